"""Multi-GPU: one process per GPU, tile-row sharding of a frame, RCCL gather of the framebuffer.

The reference has no GPU-level parallelism at all (only process-level scene sharding,
generate_images.py:136-139); this is BASELINE.json's design: every rank holds the whole scene
(708 MB at 3 M Gaussians — 0.25 % of an MI355X's HBM), renders a contiguous band of 16-pixel tile
rows, and the bands are gathered to rank 0 over xGMI.  Tiles are independent after binning, so the
gathered frame is bit-identical to a single-GPU frame (tests: tile-row union == full frame).

What a rank pays for: the library tests every 64-Gaussian chunk of the (Z-ordered) scene against the rank's band
before it touches it (csrc: k_chunk_bounds / chunk_outside), so projection, SH, binning, sorting and blending all
scale with what the band can see — not with the scene.

Which rows a rank owns:
  * default — contiguous bands as even as they come: 68 rows over 8 ranks = 9,9,9,9,8,8,8,8 (`row_partition`);
  * `balance=True` — contiguous bands of equal COST.  An indoor view puts most of its depth complexity into a few
    rows around the horizon, so equal bands leave the slowest of 8 ranks at ~1.5x the mean.  Every rank reports the
    records its rows queued (`Renderer.row_records`, from the frames it just rendered), one tiny all-reduce makes the
    per-row profile of the whole frame known everywhere — together with the time each rank's band just took, which is
    spread over the band's rows by their records (`timed_row_cost`: records alone mispredict, a ceiling row of a few huge
    splats queues as many as a horizon row of thousands of small ones and costs a third) — and the next batch's bands are
    cut from that (`balanced_partition`: minimal largest band cost); a sweep's views change slowly, so the previous batch
    predicts the next;
  * `interleave=True` — every R-th row (rank r owns rows r, r+R, ...): balanced whatever the camera looks at, but a
    splat covers 2-3 tile rows, so every rank then projects and bins ~1.7x the splats of a contiguous band, and rank 0
    has to re-interleave the rows (one copy).  Opt-in; measured slower than balanced bands.

The one exchange step is a gatherv of exact slabs, all of it ONE group of point-to-point operations
(`torch.distributed.batch_isend_irecv`; on the "nccl" backend — RCCL on ROCm — a single ncclGroupStart/End of
ncclSend/ncclRecv, i.e. a gatherv with the seven peers on seven distinct xGMI links of rank 0 concurrently; "gloo" in the
CPU tests), in one of two shapes (`exchange=`):
  * "slab" (default for batches) — every peer keeps the bands of a batch's B frames as ONE contiguous [B, rows, W, C] slab
    (the library renders a batch at any frame stride) and sends it in ONE operation; rank 0 receives it into a staging slab
    and scatters it into the rows of its [B, H, W, C] frame buffer with one strided device copy per peer: world - 1
    operations per exchange (7 at 8 ranks) whatever B is, at the price of one more pass over the gathered bytes on rank 0;
  * "frames" — rank 0 posts one receive per (peer, frame) straight into the rows of its frame buffer, every peer one send
    per frame: no staging, no copy, (world - 1) x B operations per exchange (224 at 8 ranks x 32 frames).
Bands may be of any height either way and there is no padding.  Which shape is faster over xGMI is not known to this
repo (no multi-GPU node was available to its builder): `bench.py --gpus N` times both and says so in `collective`.  A 1080p fp32 frame is 24.9 MB (3.1 MB per rank at 8
ranks): latency-, not bandwidth-bound — so a sweep ships the bands of B frames per group, asynchronously, double-
buffered against the rendering of the next batch (`ShardedRenderer.render_batch`; 32 frames by default: the library
renders a batch in groups of eight frames per launch on two streams, and the pipeline drains at the end of every call —
measured on one GPU with every rank's band replayed, slowest of 8 ranks: 0.084 ms/frame at 8 frames per call, 0.077 at
16, 0.070 at 32).

`shard_cameras` is the other natural partition (frames of a sweep are independent units): no
data-path collective at all.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

TILE = 16


def row_partition(n_tile_rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous bands, as even as they come: the first (rows mod world) ranks get one row more
    (68 rows, 8 ranks -> 9,9,9,9,8,8,8,8).  Ranks beyond the rows get empty bands."""
    if world <= 0:
        raise ValueError("world must be positive")
    base, extra = divmod(n_tile_rows, world)
    bands, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < extra else 0)
        bands.append((a, b))
        a = b
    return bands


def balanced_partition(cost: Sequence[float], world: int, max_rows: Optional[int] = None) -> List[Tuple[int, int]]:
    """Contiguous bands over len(cost) rows that minimise the largest band cost (cost[r] >= 0 per row; a band holds at
    most `max_rows` rows — the capacity of a rank's slab).  Bisection on the bottleneck over the prefix sums with a
    greedy feasibility test, then the greedy cut at the optimum; bands left over are made by halving the costliest
    ones, so no rank that could have rows stays idle.  Deterministic: every rank computes the same bands from the
    same costs."""
    n = len(cost)
    if world <= 0:
        raise ValueError("world must be positive")
    c = np.maximum(np.nan_to_num(np.asarray(cost, np.float64), nan=0.0, posinf=0.0), 0.0)
    if max_rows is None:
        max_rows = n
    if max_rows * world < n:
        raise ValueError(f"{world} bands of at most {max_rows} rows cannot cover {n} rows")
    if n == 0:
        return [(0, 0)] * world
    pre = np.concatenate([[0.0], np.cumsum(c)])
    if not pre[-1] > 0.0:            # nothing to balance
        return row_partition(n, world)

    def cut(limit):
        """Greedy bands of cost <= limit (and <= max_rows rows), or None if they need more than `world` bands."""
        bands, a = [], 0
        while a < n:
            if len(bands) == world:
                return None
            b = int(np.searchsorted(pre, pre[a] + limit, side="right")) - 1      # furthest b with pre[b] - pre[a] <= limit
            b = min(b, a + max_rows, n)
            if b <= a:
                return None          # a single row exceeds the limit
            # the rows left must still fit the bands left
            if n - b > (world - len(bands) - 1) * max_rows:
                return None
            bands.append((a, b))
            a = b
        return bands

    lo, hi = float(c.max()), float(pre[-1]) + 1.0
    if cut(hi) is None:              # only the row cap binds: cost plays no part
        return row_partition(n, world)
    for _ in range(64):
        mid = 0.5 * (lo + hi)
        if cut(mid) is not None:
            hi = mid
        else:
            lo = mid
        if hi - lo <= 1e-9 * max(1.0, hi):
            break
    bands = cut(hi)
    while len(bands) < min(world, n):
        k = max(range(len(bands)), key=lambda i: (bands[i][1] - bands[i][0] > 1, pre[bands[i][1]] - pre[bands[i][0]]))
        a, b = bands[k]
        if b - a <= 1:
            break
        m = min(range(a + 1, b), key=lambda x: max(pre[x] - pre[a], pre[b] - pre[x]))      # the most even cut
        bands[k:k + 1] = [(a, m), (m, b)]
    bands += [(n, n)] * (world - len(bands))
    return bands


def timed_row_cost(bands: Sequence[Tuple[int, int]], band_ms: Sequence[float], row_records: Sequence[float],
                   prev_cost: Optional[Sequence[float]] = None, smooth: float = 0.5) -> np.ndarray:
    """Per-tile-row cost from what the ranks just MEASURED: rank r took band_ms[r] for its band, and that time is spread
    over the band's rows in proportion to the records each row queued (rows of an empty band, or without records, share
    evenly).  Records alone mispredict — a ceiling row of a few huge splats queues as many records as a horizon row
    of thousands of small ones and costs a third — while the measured time carries everything, including the fixed cost of
    a rank's launches.  `prev_cost` (the cost the current bands were cut from) is blended in (`smooth`) to damp the
    iteration.  Deterministic; every rank computes the same vector from the same all-reduced inputs."""
    rec = np.maximum(np.nan_to_num(np.asarray(row_records, np.float64)), 0.0)
    cost = np.zeros(len(rec))
    for (a, b), t in zip(bands, band_ms):
        if b <= a:
            continue
        t = max(float(t), 0.0)
        w = rec[a:b] + 1.0                                   # (+1: rows without records still share the band's fixed cost)
        cost[a:b] = t * w / w.sum()
    if prev_cost is not None and len(prev_cost) == len(cost) and smooth > 0.0:
        p = np.asarray(prev_cost, np.float64)
        if p.sum() > 0 and cost.sum() > 0:
            cost = (1.0 - smooth) * cost + smooth * p * (cost.sum() / p.sum())
    return cost


def shard_cameras(n_cameras: int, rank: int, world: int) -> range:
    """Round-robin camera ownership: rank r renders cameras r, r+world, ..."""
    return range(rank, n_cameras, world)


class _Works:
    """The handles of one exchange; wait() completes it (and, for the gloo-with-GPU-tensors debugging path, lands the
    host-staged slabs in the device buffers)."""

    def __init__(self, works, after=None):
        self._works, self._after = list(works or []), after

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        if self._after is not None:
            self._after()
            self._after = None


class FrameGather:
    """Buffers and the exchange for gathering the tile-row bands of H x W frames to rank `dst`.

    batch=None: one frame; batch=B: B frames per exchange (a sweep's frames are independent, so their bands travel
    together).  Rank dst owns the frames themselves ([B,] H, W, C — `frames()` / `frame(b)` are views of that buffer for
    contiguous bands) and renders its own band in place; every other rank owns a slab of `max_band_rows` tile rows per
    frame.  `render_target(b)` gives the keyword arguments for `Renderer.render` that put this rank's rows where the
    exchange expects them.  Bands can be replaced between exchanges (`set_bands`: cost-balanced sharding) as long as
    every rank installs the same ones."""

    def __init__(self, height: int, width: int, device, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, dst: int = 0, channels: int = 3, dtype=torch.float32, batch: Optional[int] = None,
                 interleave: bool = False, max_band_rows: Optional[int] = None, exchange: str = "frames"):
        if exchange not in ("frames", "slab"):
            raise ValueError("exchange must be 'frames' or 'slab'")
        self.group = group
        # "slab": one operation per peer and exchange (contiguous bands of a batch; interleaved rows and single frames keep "frames")
        self.mode = "slab" if (exchange == "slab" and not interleave and batch is not None) else "frames"
        self.interleave = bool(interleave)
        # (no process group: one rank — `bench.py --gpus 1` runs its verify step through the very same code)
        have_pg = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank(group) if have_pg else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if have_pg else 1) if world is None else world
        self.dst, self.h, self.w, self.c = dst, height, width, channels
        self.batch = batch
        self.n_tile_rows = (height + TILE - 1) // TILE
        even = -(-self.n_tile_rows // self.world)
        # a slab holds up to this many tile rows: room for cost-balanced bands (a cheap band — ceiling, floor — may be
        # several times the even height)
        self.max_band_rows = min(self.n_tile_rows, max(even, int(max_band_rows) if max_band_rows else 4 * even))
        nb = 1 if batch is None else int(batch)
        self._nb = nb
        if self.interleave:          # rank r owns frame rows r, r+world, ...; `band` then counts OWNED rows
            self.bands = [(0, len(range(r, self.n_tile_rows, self.world))) for r in range(self.world)]
            rows_cap = even
        else:
            self.bands = row_partition(self.n_tile_rows, self.world)
            rows_cap = self.max_band_rows
        self.band = self.bands[self.rank]
        self._frames = None
        self._compact = None
        self._slab = None
        self._slab_flat = None       # "slab" mode, a peer: the storage its compact [nb, rows, W, C] slab is a view of
        self._stage = None           # "slab" mode, rank dst: where the peers' slabs land before they are scattered into the frames
        self.last_ops = 0            # point-to-point operations this rank posted in its last exchange
        if self.rank == dst:
            self._frames = torch.zeros((nb, height, width, channels), dtype=dtype, device=device)
            if self.interleave and self.world > 1:
                # compact images of every rank's rows land here; frames are assembled on request (one copy)
                self._compact = torch.zeros((self.world, nb, rows_cap * TILE, width, channels), dtype=dtype, device=device)
            if self.mode == "slab" and self.world > 1:
                self._stage = torch.zeros(nb * height * width * channels, dtype=dtype, device=device)
        elif self.mode == "slab":
            self._slab_flat = torch.zeros(nb * rows_cap * TILE * width * channels, dtype=dtype, device=device)
            self._view_slab()
        else:
            self._slab = torch.zeros((nb, rows_cap * TILE, width, channels), dtype=dtype, device=device)

    def _view_slab(self):
        """"slab" mode, a peer: its slab is COMPACT for the band it currently owns — [nb, rows, W, C] contiguous, so that the first n frames
        are one contiguous message (the library renders a batch at any frame stride: sgs_render_batch_strided)."""
        y0, y1 = self._px(self.band)
        rows = max(0, y1 - y0)
        self._slab = self._slab_flat[: self._nb * rows * self.w * self.c].view(self._nb, rows, self.w, self.c)

    # -- bands ------------------------------------------------------------------------------------------------------
    def set_bands(self, bands: Sequence[Tuple[int, int]]):
        """Install a new partition (contiguous bands only); must be called with the same bands on every rank."""
        if self.interleave:
            raise ValueError("interleaved rows are fixed by (world, rank)")
        bands = [(int(a), int(b)) for a, b in bands]
        if len(bands) != self.world or bands[0][0] != 0 or bands[-1][1] != self.n_tile_rows or \
                any(b < a or b - a > self.max_band_rows for a, b in bands) or \
                any(bands[i][0] != bands[i - 1][1] for i in range(1, len(bands))):
            raise ValueError(f"not a partition of {self.n_tile_rows} tile rows into {self.world} bands of <= "
                             f"{self.max_band_rows} rows: {bands}")
        self.bands = bands
        self.band = bands[self.rank]
        if self._slab_flat is not None:
            self._view_slab()

    def _px(self, band) -> Tuple[int, int]:
        return band[0] * TILE, min(band[1] * TILE, self.h)

    @property
    def band_pixel_rows(self) -> Tuple[int, int]:
        if self.interleave:
            raise ValueError("interleaved rows are not one range of pixel rows")
        return self._px(self.band)

    @property
    def slab(self) -> torch.Tensor:
        """The tensor this rank's rows are written to ([rows, W, C], or [B, rows, W, C] with batch=B): rank dst's own
        rows of the frame buffer (contiguous bands), its compact image (interleaved), or the slab of another rank."""
        if self.rank != self.dst:
            t = self._slab
        elif self._compact is not None:
            t = self._compact[self.rank]
        elif self.interleave:                   # world == 1
            t = self._frames
        else:
            y0, y1 = self._px(self.band)
            t = self._frames[:, y0:y1]
        return t[0] if self.batch is None else t

    def render_target(self, b: int = 0) -> dict:
        """Keyword arguments for Renderer.render that select this rank's rows of frame b and where they are stored."""
        if self.interleave and self.world > 1:
            buf = self._compact[self.rank] if self.rank == self.dst else self._slab
            return {"out_band": buf[b], "interleave": (self.world, self.rank)}
        if self.rank == self.dst:
            return {"out": self._frames[b], "tile_rows": self.band}
        return {"out_band": self._slab[b], "tile_rows": self.band}

    def batch_target(self) -> dict:
        """Keyword arguments for Renderer.render_batch: this rank's rows of every frame of the batch, in place."""
        if self.interleave and self.world > 1:
            buf = self._compact[self.rank] if self.rank == self.dst else self._slab
            return {"out_bands": buf, "interleave": (self.world, self.rank)}
        if self.rank == self.dst:
            return {"out": self._frames, "tile_rows": self.band}
        return {"out_bands": self._slab, "tile_rows": self.band}

    @property
    def render_rows(self) -> dict:
        """The row selection alone (for callers that manage their own output buffers)."""
        if self.interleave and self.world > 1:
            return {"interleave": (self.world, self.rank)}
        return {"tile_rows": self.band}

    # -- the exchange ---------------------------------------------------------------------------------------------------
    def _rows_px(self, r: int) -> int:
        """Pixel rows of one frame that rank r contributes."""
        if self.interleave:
            return self.bands[r][1] * TILE
        y0, y1 = self._px(self.bands[r])
        return y1 - y0

    def _global(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def exchange(self, n: Optional[int] = None, async_op: bool = False) -> Optional[_Works]:
        """Gatherv of the first n frames' bands to rank dst, as ONE group of point-to-point operations.  Returns the
        handle (async_op) or None once complete."""
        n = self._nb if n is None else int(n)
        if not 0 < n <= self._nb:
            raise ValueError(f"1..{self._nb} frames per exchange")
        if self.world == 1:
            return None
        mine = self._frames if self.rank == self.dst else self._slab
        gloo_gpu = mine.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, staged, scatter = [], [], []
        if self.mode == "slab":
            # ONE operation per peer: its [n, rows, W, C] slab, contiguous on both sides; rank dst scatters it into the frames afterwards
            if self.rank == self.dst:
                off = 0
                for r in range(self.world):
                    rows = self._rows_px(r)
                    if r == self.dst or rows <= 0:
                        continue
                    cnt = n * rows * self.w * self.c
                    view = self._stage[off:off + cnt].view(n, rows, self.w, self.c)
                    off += cnt
                    y0 = self.bands[r][0] * TILE
                    scatter.append((self._frames[:n, y0:y0 + rows], view))
                    if gloo_gpu:
                        host = torch.empty(view.shape, dtype=view.dtype)
                        staged.append((view, host))
                        view = host
                    ops.append(dist.P2POp(dist.irecv, view, self._global(r), self.group))
            else:
                rows = self._rows_px(self.rank)
                if rows > 0:
                    src = self._slab[:n]                 # (compact: the first n frames are one contiguous range)
                    if gloo_gpu:
                        src = src.cpu()
                        staged.append((None, src))
                    ops.append(dist.P2POp(dist.isend, src, self._global(self.dst), self.group))
        elif self.rank == self.dst:
            for r in range(self.world):
                rows = self._rows_px(r)
                if r == self.dst or rows <= 0:
                    continue
                for b in range(n):
                    if self.interleave:
                        view = self._compact[r][b, :rows]
                    else:
                        y0 = self.bands[r][0] * TILE
                        view = self._frames[b, y0:y0 + rows]
                    if gloo_gpu:                 # debugging aid (several ranks on one GPU under gloo): stage through the host
                        host = torch.empty(view.shape, dtype=view.dtype)
                        staged.append((view, host))
                        view = host
                    ops.append(dist.P2POp(dist.irecv, view, self._global(r), self.group))
        else:
            rows = self._rows_px(self.rank)
            if rows > 0:
                for b in range(n):
                    src = self._slab[b, :rows]
                    if gloo_gpu:
                        src = src.cpu()
                        staged.append((None, src))       # kept alive until the send has completed
                    ops.append(dist.P2POp(dist.isend, src, self._global(self.dst), self.group))
        self.last_ops = len(ops)
        works = dist.batch_isend_irecv(ops) if ops else []

        def land():
            for view, host in staged:
                if view is not None:
                    view.copy_(host)
            for rows_of_frames, slab in scatter:         # "slab" mode: one strided device copy per peer
                rows_of_frames.copy_(slab)
        h = _Works(works, land if (staged or scatter) else None)
        if async_op:
            return h
        h.wait()
        return None

    # -- single-frame convenience ---------------------------------------------------------------------------------------
    def gather(self) -> Optional[torch.Tensor]:
        """Exchange (single-frame buffers).  Returns the [H,W,C] frame on rank dst (contiguous bands: the frame buffer
        itself, no copy)."""
        if self.batch is not None:
            raise ValueError("gather() is for single-frame buffers; use exchange()")
        self.exchange(1)
        return self.frame(0)

    def gather_batch(self, n: Optional[int] = None, async_op: bool = False):
        if self.batch is None:
            raise ValueError("gather_batch() needs batch=B buffers")
        return self.exchange(n, async_op)

    def frames(self, n: Optional[int] = None) -> Optional[torch.Tensor]:
        """Rank dst: the gathered frames, [n, H, W, C] (contiguous bands: a view of the receive buffer)."""
        if self.rank != self.dst:
            return None
        n = self._nb if n is None else n
        if self._compact is not None:
            return torch.stack([self.frame(b) for b in range(n)])
        return self._frames[:n]

    def frame(self, b: int = 0) -> Optional[torch.Tensor]:
        """Rank dst: frame b, [H, W, C] (contiguous bands: a view; interleaved rows: assembled with one copy)."""
        if self.rank != self.dst:
            return None
        if self._compact is not None:
            k = self._compact.shape[2] // TILE
            v = self._compact[:, b].view(self.world, k, TILE, self.w, self.c).permute(1, 0, 2, 3, 4)   # [k, world, 16, W, C]
            return v.reshape(k * self.world * TILE, self.w, self.c)[: self.h]
        return self._frames[b]


class ShardedRenderer:
    """Tile-row-sharded rendering across the ranks of a process group: every rank renders its band of tile rows, the
    bands are gathered to rank `dst`.  balance=True re-cuts the bands of a sweep by cost (see the module docstring)."""

    # the gather buffers of a rank (two batched FrameGather rings + the fp32 scratch of rgba8 output) may take this much
    # device memory before the constructor refuses (batch x band rows x width grows quickly at 3840x2160)
    MAX_BUFFER_BYTES = 16 << 30

    def __init__(self, renderer, height: int, width: int, group=None, dst: int = 0, batch: int = 32,
                 interleave: bool = False, balance: bool = False, output: str = "float32", exchange: str = "slab"):
        """output="float32": rank dst gets [H,W,3] float32 frames (the renderer's native output).  output="rgba8": every rank
        packs its band to uint8 RGBA (Renderer.pack_rgba8 — what `get_rgba()` hands the reference's callers) and the bands
        travel as bytes: a third of the fp32 payload over xGMI (8.3 instead of 24.9 MB per 1080p frame into rank dst);
        the gathered frame equals pack_rgba8 of the un-sharded frame bit for bit."""
        if balance and interleave:
            raise ValueError("interleaved rows are balanced by construction; balance=True is for contiguous bands")
        if output not in ("float32", "rgba8"):
            raise ValueError("output must be 'float32' or 'rgba8'")
        if output == "rgba8" and interleave:
            raise ValueError("rgba8 output is for contiguous bands")
        if exchange not in ("frames", "slab"):
            raise ValueError("exchange must be 'frames' or 'slab' (see the module docstring)")
        self.exchange = exchange
        self.r = renderer
        self.h, self.w, self.group, self.dst = height, width, group, dst
        self.interleave, self.balance = bool(interleave), bool(balance)
        self.output = output
        self._gk = {"channels": 4, "dtype": torch.uint8} if output == "rgba8" else {}
        self._f32 = None             # rgba8: the fp32 bands are rendered into this slab, then packed into the gather buffers
        self.g = FrameGather(height, width, renderer.device, group=group, dst=dst, interleave=self.interleave, **self._gk)
        self.batch = int(batch)
        self._ring = None            # two batched buffers: one travels while the other is rendered into
        self._pending = [None, None]
        self._turn = 0
        self._cost_work = None       # the all-reduce of the previous batch's per-row records and per-rank band times
        self._cost = None
        self._cost_bands = None      # the bands those measurements were taken with
        self._row_cost = None        # the per-row cost the current bands were cut from
        self.bands = list(self.g.bands)
        self.last_stats = None       # statistics of the last batch this rank rendered (per-frame averages)
        # what the two batched gather buffers (+ the fp32 scratch of rgba8 output) will take on this rank
        esz = 1 if output == "rgba8" else 4
        ch = 4 if output == "rgba8" else 3
        band_rows = min(height, self.g.max_band_rows * TILE)
        scratch = self.batch * band_rows * width * 12 if output == "rgba8" else 0
        stage = self.batch * height * width * ch * esz if (exchange == "slab" and not self.interleave) else 0     # rank dst, per ring buffer
        size = lambda rows: 2 * self.batch * rows * width * ch * esz + scratch + (2 * stage if rows == height else 0)
        self.buffer_bytes = int(size(height if self.g.rank == dst else band_rows))
        # The limit is checked against the LARGEST rank's need (rank dst holds whole frames) on EVERY rank: a check of the rank's own
        # need would raise on dst alone and leave the others blocked in the collective below until the communicator times out.
        need = size(height)
        if need > self.MAX_BUFFER_BYTES:
            raise ValueError(f"ShardedRenderer(batch={self.batch}) would hold {need / 2**30:.1f} GiB of gather buffers on rank "
                             f"{dst} at {width}x{height}; use a smaller batch (MAX_BUFFER_BYTES = {self.MAX_BUFFER_BYTES >> 30} GiB)")
        if self.world > 1:
            # create the communicator NOW, with every rank taking part: the framebuffer exchange is a group of point-to-point
            # operations in which a rank with an empty band takes no part, which is only safe on a communicator that exists
            t = torch.zeros(1, dtype=torch.float32, device="cpu" if dist.get_backend(self.group) == "gloo" else renderer.device)
            dist.all_reduce(t, group=self.group)

    @property
    def world(self) -> int:
        return self.g.world

    def render(self, camera, scene, *, config=None):
        """One frame: every rank renders its band and joins the exchange; rank dst gets the frame.
        The band is rendered synchronously: a frame that exceeds the record capacity is re-rendered after the queues
        have grown (an asynchronous frame would render nothing and the stale slab would travel)."""
        r0, r1 = self.g.band
        if r1 > r0:
            if self.output == "rgba8":
                slab = self._scratch(1)[0]
                self.r.render(camera, scene, config=config, sync=True, out_band=slab, tile_rows=self.g.band)
                self._pack(self.g, slab[None], 1)
            else:
                self.r.render(camera, scene, config=config, sync=True, **self.g.render_target(0))
        return self.g.gather()

    # -- rgba8 output -------------------------------------------------------------------------------------------------------
    def _scratch(self, n: int) -> torch.Tensor:
        if self._f32 is None or self._f32.shape[0] < n:
            self._f32 = torch.zeros((max(n, self.batch), self.g.max_band_rows * TILE, self.w, 3), dtype=torch.float32,
                                    device=self.r.device)
        return self._f32

    def _pack(self, g: "FrameGather", f32: torch.Tensor, n: int):
        """This rank's fp32 band of the first n frames -> uint8 RGBA in the gather buffers (rank dst: its own rows of the
        frames; the others: their slabs)."""
        y0, y1 = g._px(g.band)
        rows = y1 - y0
        for b in range(n):
            dst = g._frames[b, y0:y1] if g.rank == g.dst else g._slab[b, :rows]
            self.r.pack_rgba8(f32[b, :rows], out=dst)

    # -- cost-balanced bands ---------------------------------------------------------------------------------------------
    def _post_costs(self, n_frames: int, band_ms: float, bands):
        """After a batch: every rank contributes the records its rows queued and the time its band took per frame; one
        small all-reduce (rows + ranks elements), posted BEFORE the batch's exchange so that it does not queue behind
        25 MB per frame on the communicator."""
        g = self.g
        rec = self.r.row_records(g.n_tile_rows, reset=True).astype(np.float64) / max(1, n_frames)
        times = np.zeros(self.world)
        times[g.rank] = band_ms
        dev = "cpu" if dist.get_backend(self.group) == "gloo" else self.r.device
        self._cost = torch.from_numpy(np.concatenate([rec, times])).to(dev)
        self._cost_bands = list(bands)
        self._cost_work = dist.all_reduce(self._cost, group=self.group, async_op=True) if self.world > 1 else None

    def _rebalance(self):
        if self._cost is None:
            return
        if self._cost_work is not None:
            self._cost_work.wait()
            self._cost_work = None
        v = self._cost.cpu().numpy()
        self._cost = None
        n = self.g.n_tile_rows
        rec, times = v[:n], v[n:]
        # measured band times (every rank posts a positive wall-clock time), spread over the rows by their records
        self._row_cost = timed_row_cost(self._cost_bands, times, rec, self._row_cost)
        self.bands = balanced_partition(self._row_cost, self.world, self.g.max_band_rows)

    def render_batch(self, cameras, scene, *, config=None, timing=False):
        """Up to `batch` independent frames (a sweep): the bands are rendered through the renderer's pipelined lanes
        and travel in ONE asynchronous group of sends/receives, which overlaps with the next call's rendering (two
        buffers alternate).  Returns the FrameGather holding this batch; its contents are complete after .finish() (or
        the next-but-one render_batch call)."""
        n = len(cameras)
        if n == 0 or n > self.batch:
            raise ValueError(f"1..{self.batch} cameras per batch")
        if self._ring is None:
            self._ring = [FrameGather(self.h, self.w, self.r.device, group=self.group, dst=self.dst, batch=self.batch,
                                      interleave=self.interleave, exchange=self.exchange, **self._gk) for _ in range(2)]
        k = self._turn
        self._turn ^= 1
        if self._pending[k] is not None:           # the exchange that last read this buffer
            self._pending[k].wait()
            self._pending[k] = None
        g = self._ring[k]
        if self.balance:
            self._rebalance()                      # bands cut from the previous batch's per-row records (same on all ranks)
            g.set_bands(self.bands)
        import time
        t0 = time.perf_counter()
        r0, r1 = g.band
        if r1 > r0:
            rgba8 = self.output == "rgba8"
            f32 = self._scratch(n) if rgba8 else None
            if timing:                             # per-stage events: one call per frame
                for b, cam in enumerate(cameras):
                    tgt = {"out_band": f32[b], "tile_rows": g.band} if rgba8 else g.render_target(b)
                    self.r.render(cam, scene, config=config, sync=False, pipelined=True, timing=True, **tgt)
                self.last_stats = self.r.sync()    # bands complete (all lanes) before the exchange reads them
            else:                                  # ONE call into the library for the whole batch (complete on return)
                tgt = {"out_bands": f32, "tile_rows": g.band} if rgba8 else g.batch_target()
                _, st = self.r.render_batch(cameras, scene, config=config, want_stats=True, **tgt)
                self.last_stats = st[-1]
            if rgba8:
                self._pack(g, f32, n)
        if self.balance:                           # (render_batch is complete on return: the band's wall time per frame)
            self._post_costs(n, 1e3 * (time.perf_counter() - t0) / n, g.bands)
        self._pending[k] = g.exchange(n, async_op=True)
        return g

    def finish(self):
        """Wait for the exchanges in flight."""
        for k in (0, 1):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None
        if self._cost_work is not None:
            self._cost_work.wait()
            self._cost_work = None
