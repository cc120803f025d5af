"""Multi-GPU: one process per GPU, tile-row sharding of a frame, RCCL gather of the framebuffer.

The reference has no GPU-level parallelism at all (only process-level scene sharding,
generate_images.py:136-139); this is BASELINE.json's design: every rank holds the whole scene
(708 MB at 3 M Gaussians — 0.25 % of an MI355X's HBM), renders a contiguous band of 16-pixel tile
rows, and the bands are gathered to rank 0 over xGMI.  Tiles are independent after binning, so the
gathered frame is bit-identical to a single-GPU frame (tests: tile-row union == full frame).

Which rows a rank owns: a contiguous band of equal height by default, or (`interleave=True`) every R-th row — rank r of R
owns frame tile rows r, r+R, r+2R, ...  An indoor view puts most of its depth complexity into a few rows around the
horizon, so equal bands are uneven (the slowest of 8 ranks carries 1.5x the mean of the 256-pose sweep) while every R-th
row gives each rank the same mix whatever the camera looks at.  Interleaved rows are stored compactly
(`Renderer.render(interleave=(R, r))`), so slabs stay equal-size and the collective is unchanged; rank 0 re-interleaves
the rows when it hands out a frame (one device copy).  It is opt-in because it does not pay on the indoor sweep
(measured per rank on one MI355X, frames pipelined, ms per frame of the slowest rank, contiguous -> interleaved:
R=2 0.168 -> 0.171, R=4 0.136 -> 0.134, R=8 0.112 -> 0.101): a splat covers 2-3 tile rows, so every rank then projects,
shades and bins ~1.7x the splats a contiguous band sees, and the copy on rank 0 eats what is left.  Bands balanced by
queue length did better in the same experiment (R=8 0.087, R=4 0.116) but need a calibration pass and unequal slabs.

The one exchange step is a gather of equal-size slabs (`torch.distributed.gather`; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  Rank 0 receives straight into views of its frame buffer,
so there is no assembly copy.  A 1080p fp32 frame is 24.9 MB (3.3 MB per rank at 8 ranks): seven
peers land on seven distinct xGMI links of rank 0 concurrently, so the step is latency-, not
bandwidth-bound — one collective per frame, no ring.

For a sweep of independent frames (`ShardedRenderer.render_batch`) the bands of B frames are rendered through the
renderer's pipelined lanes and travel in one asynchronous collective, double-buffered against the next batch.

`shard_cameras` is the other natural partition (frames of a sweep are independent units): no
data-path collective at all.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TILE = 16


def row_partition(n_tile_rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous bands of ceil(rows/world) tile rows; trailing ranks may get fewer (or none).
    Uniform band height lets rank 0 gather directly into its frame buffer; the critical path
    (largest band) is the same as for the most even split."""
    if world <= 0:
        raise ValueError("world must be positive")
    per = (n_tile_rows + world - 1) // world
    return [(min(r * per, n_tile_rows), min((r + 1) * per, n_tile_rows)) for r in range(world)]


def shard_cameras(n_cameras: int, rank: int, world: int) -> range:
    """Round-robin camera ownership: rank r renders cameras r, r+world, ..."""
    return range(rank, n_cameras, world)


class FrameGather:
    """Buffers and the collective for gathering tile-row bands of H x W frames to rank `dst`.

    batch=None: one frame, slab = [slab_rows, W, C].  batch=B: B frames per collective, slab = [B, slab_rows, W, C]
    (a sweep's frames are independent, so their bands can travel together: one collective per B frames)."""

    def __init__(self, height: int, width: int, device, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, dst: int = 0, channels: int = 3, dtype=torch.float32, batch: Optional[int] = None,
                 interleave: bool = False):
        self.group = group
        self.interleave = bool(interleave)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.dst, self.h, self.w = dst, height, width
        self.batch = batch
        self.n_tile_rows = (height + TILE - 1) // TILE
        self.bands = row_partition(self.n_tile_rows, self.world)
        self.slab_rows = ((self.n_tile_rows + self.world - 1) // self.world) * TILE      # pixel rows per slab
        self.band = self.bands[self.rank]
        if self.interleave:          # rank r owns frame rows r, r+world, ...; `band` then counts OWNED rows
            self.bands = [(0, len(range(r, self.n_tile_rows, self.world))) for r in range(self.world)]
            self.band = self.bands[self.rank]
        lead = () if batch is None else (int(batch),)
        # every rank's slab has the same shape; rank dst owns the padded buffer the slabs land in
        if self.rank == dst:
            self.padded = torch.zeros((self.world,) + lead + (self.slab_rows, width, channels), dtype=dtype, device=device)
            self.slab = self.padded[self.rank]
            self._views = [self.padded[i] for i in range(self.world)]
        else:
            self.padded = None
            self.slab = torch.zeros(lead + (self.slab_rows, width, channels), dtype=dtype, device=device)
            self._views = None

    @property
    def band_pixel_rows(self) -> Tuple[int, int]:
        if self.interleave:
            raise ValueError("interleaved rows are not one range of pixel rows")
        return self.band[0] * TILE, min(self.band[1] * TILE, self.h)

    @property
    def render_rows(self) -> dict:
        """Keyword arguments for Renderer.render that select this rank's rows (out_band = a slab of this object)."""
        if self.interleave:
            return {"interleave": (self.world, self.rank)} if self.world > 1 else {"tile_rows": self.band}
        return {"tile_rows": self.band}

    def _assemble(self, padded: torch.Tensor) -> torch.Tensor:
        """[world, slab_rows, W, C] -> [H, W, C]: a view for contiguous bands, one copy for interleaved rows."""
        if self.interleave and self.world > 1:
            k = self.slab_rows // TILE
            v = padded.view(self.world, k, TILE, self.w, -1).permute(1, 0, 2, 3, 4)       # [k, world, 16, W, C]
            return v.reshape(k * self.world * TILE, self.w, -1)[: self.h]
        return padded.reshape(self.world * self.slab_rows, self.w, -1)[: self.h]

    def _collective(self, n: Optional[int], async_op: bool):
        src = self.slab if n is None else self.slab[:n]
        outs = None
        if self.rank == self.dst:
            outs = self._views if n is None else [v[:n] for v in self._views]
        if src.is_cuda and dist.get_backend(self.group) == "gloo":
            # debugging aid (several ranks on one GPU under gloo): stage through the host
            host = src.cpu()
            houts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(host, houts, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for o, h in zip(outs, houts):
                    o.copy_(h)
            return None
        return dist.gather(src, outs, dst=self.dst, group=self.group, async_op=async_op)

    def gather(self) -> Optional[torch.Tensor]:
        """Collective (single-frame buffers).  Returns the assembled [H,W,C] frame on rank dst (contiguous bands: a view
        of the receive buffer, no copy)."""
        if self.batch is not None:
            raise ValueError("gather() is for single-frame buffers; use gather_batch()")
        if self.world > 1:
            self._collective(None, False)
        if self.rank != self.dst:
            return None
        return self._assemble(self.padded)

    def gather_batch(self, n: Optional[int] = None, async_op: bool = False):
        """Collective over the first n frames of a batched buffer.  Returns the work handle (None when complete)."""
        if self.batch is None:
            raise ValueError("gather_batch() needs batch=B buffers")
        if self.world > 1:
            return self._collective(n, async_op)
        return None

    def frames(self, n: Optional[int] = None) -> Optional[torch.Tensor]:
        """Rank dst: the gathered batch as a [n, world, slab_rows, W, C] view (contiguous bands: frame b = rows of [b]
        stacked, cut at H; interleaved: use frame(b))."""
        if self.rank != self.dst:
            return None
        v = self.padded.permute(1, 0, 2, 3, 4)
        return v if n is None else v[:n]

    def frame(self, b: int) -> Optional[torch.Tensor]:
        """Rank dst: frame b of the gathered batch, assembled to [H,W,C] (one copy)."""
        if self.rank != self.dst:
            return None
        return self._assemble(self.padded[:, b])


class ShardedRenderer:
    """Tile-row-sharded rendering across the ranks of a process group: every rank renders its band of tile rows,
    the bands are gathered to rank `dst`."""

    def __init__(self, renderer, height: int, width: int, group=None, dst: int = 0, batch: int = 8, interleave: bool = False):
        self.r = renderer
        self.h, self.w, self.group, self.dst = height, width, group, dst
        self.interleave = bool(interleave)
        self.g = FrameGather(height, width, renderer.device, group=group, dst=dst, interleave=self.interleave)
        self.batch = int(batch)
        self._ring = None            # two batched buffers: one travels while the other is rendered into
        self._pending = [None, None]
        self._turn = 0
        self.last_stats = None       # statistics of the last batch this rank rendered (per-frame averages)

    def render(self, camera, scene, *, config=None):
        """One frame: every rank renders its band into its slab and joins the gather; rank dst gets the frame.
        The band is rendered synchronously: a frame that exceeds the record capacity is re-rendered after the queues
        have grown (an asynchronous frame would render nothing and the stale slab would travel)."""
        r0, r1 = self.g.band
        if r1 > r0:
            self.r.render(camera, scene, config=config, out_band=self.g.slab, sync=True, **self.g.render_rows)
        return self.g.gather()

    def render_batch(self, cameras, scene, *, config=None, timing=False):
        """Up to `batch` independent frames (a sweep): the bands are rendered through the renderer's pipelined lanes
        and travel in ONE asynchronous collective, which overlaps with the next call's rendering (two buffers
        alternate).  Returns the FrameGather holding this batch; its contents are complete after .finish() (or the
        next-but-one render_batch call)."""
        n = len(cameras)
        if n == 0 or n > self.batch:
            raise ValueError(f"1..{self.batch} cameras per batch")
        if self._ring is None:
            self._ring = [FrameGather(self.h, self.w, self.r.device, group=self.group, dst=self.dst, batch=self.batch,
                                      interleave=self.interleave) for _ in range(2)]
        k = self._turn
        self._turn ^= 1
        if self._pending[k] is not None:           # the collective that last read this buffer
            self._pending[k].wait()
            self._pending[k] = None
        g = self._ring[k]
        r0, r1 = g.band
        if r1 > r0:
            for b, cam in enumerate(cameras):
                self.r.render(cam, scene, config=config, out_band=g.slab[b], sync=False, pipelined=True, timing=timing,
                              **g.render_rows)
            self.last_stats = self.r.sync()        # bands complete (all lanes) before the collective reads them
        self._pending[k] = g.gather_batch(n, async_op=True)
        return g

    def finish(self):
        """Wait for the collectives in flight."""
        for k in (0, 1):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None
