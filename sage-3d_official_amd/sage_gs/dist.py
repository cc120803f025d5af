"""Multi-GPU: one process per GPU, tile-row sharding of a frame, RCCL gather of the framebuffer.

The reference has no GPU-level parallelism at all (only process-level scene sharding,
generate_images.py:136-139); this is BASELINE.json's design: every rank holds the whole scene
(708 MB at 3 M Gaussians — 0.25 % of an MI355X's HBM), renders a contiguous band of 16-pixel tile
rows, and the bands are gathered to rank 0 over xGMI.  Tiles are independent after binning, so the
gathered frame is bit-identical to a single-GPU frame (tests: tile-row union == full frame).

The one exchange step is a gather of equal-size slabs (`torch.distributed.gather`; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  Rank 0 receives straight into views of its frame buffer,
so there is no assembly copy.  A 1080p fp32 frame is 24.9 MB (3.3 MB per rank at 8 ranks): seven
peers land on seven distinct xGMI links of rank 0 concurrently, so the step is latency-, not
bandwidth-bound — one collective per frame, no ring.

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
    """Buffers and the collective for gathering tile-row bands of H x W frames to rank `dst`."""

    def __init__(self, height: int, width: int, device, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, dst: int = 0, channels: int = 3, dtype=torch.float32):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.dst, self.h, self.w = dst, height, width
        self.n_tile_rows = (height + TILE - 1) // TILE
        self.bands = row_partition(self.n_tile_rows, self.world)
        self.slab_rows = ((self.n_tile_rows + self.world - 1) // self.world) * TILE      # pixel rows per slab
        self.band = self.bands[self.rank]
        # every rank's slab has the same shape; rank dst owns the padded frame the slabs land in
        if self.rank == dst:
            self.padded = torch.zeros((self.world, self.slab_rows, width, channels), dtype=dtype, device=device)
            self.slab = self.padded[self.rank]
            self._views = [self.padded[i] for i in range(self.world)]
        else:
            self.padded = None
            self.slab = torch.zeros((self.slab_rows, width, channels), dtype=dtype, device=device)
            self._views = None

    @property
    def band_pixel_rows(self) -> Tuple[int, int]:
        return self.band[0] * TILE, min(self.band[1] * TILE, self.h)

    def gather(self) -> Optional[torch.Tensor]:
        """Collective.  Returns the assembled [H,W,C] frame on rank dst (a view, no copy), else None."""
        if self.world > 1:
            if self.slab.is_cuda and dist.get_backend(self.group) == "gloo":
                # debugging aid (several ranks on one GPU under gloo): stage through the host
                host = self.slab.cpu()
                outs = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
                dist.gather(host, outs, dst=self.dst, group=self.group)
                if self.rank == self.dst:
                    for i, o in enumerate(outs):
                        self._views[i].copy_(o)
            else:
                dist.gather(self.slab, self._views if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return self.padded.view(self.world * self.slab_rows, self.w, -1)[: self.h]


class ShardedRenderer:
    """Tile-row-sharded rendering of one frame across the ranks of a process group."""

    def __init__(self, renderer, height: int, width: int, group=None, dst: int = 0):
        self.r = renderer
        self.g = FrameGather(height, width, renderer.device, group=group, dst=dst)

    def render(self, camera, scene, *, config=None, sync=False):
        """Every rank renders its band into its slab and joins the gather; rank dst gets the frame."""
        r0, r1 = self.g.band
        if r1 > r0:
            self.r.render(camera, scene, config=config, out_band=self.g.slab, tile_rows=(r0, r1), sync=sync)
        return self.g.gather()
