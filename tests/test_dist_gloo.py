"""The N>1 path on CPU: world_size-2 (and 3) `gloo` runs of the tile-row partition + framebuffer gather
(sage_gs/dist.py).  Each rank contributes the oracle's render of ITS band of tile rows; rank 0 must end up
with the full frame, bit-identical to an un-sharded render (tiles are independent after binning)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sage_gs.dist import FrameGather, row_partition, shard_cameras


def test_row_partition_covers_rows_once():
    for rows, world in ((68, 1), (68, 2), (68, 4), (68, 8), (135, 8), (3, 8), (16, 16), (1, 4)):
        bands = row_partition(rows, world)
        assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == rows
        per = -(-rows // world)
        for r, (a, b) in enumerate(bands):
            assert 0 <= b - a <= per and a == min(r * per, rows)
            if r:
                assert a == bands[r - 1][1]
    assert [len(shard_cameras(10, r, 4)) for r in range(4)] == [3, 3, 2, 2]
    assert sorted(i for r in range(4) for i in shard_cameras(10, r, 4)) == list(range(10))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _owned_pixel_rows(g):
    """Pixel rows of the frame held by this rank's slab, in slab order (interleaved FrameGather)."""
    rows = []
    for k, row in enumerate(range(g.rank, g.n_tile_rows, g.world)):
        rows += [(16 * k + i, 16 * row + i) for i in range(16) if 16 * row + i < g.h]
    return rows


def _worker(rank, world, port, h, w, n, q, interleave=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "oracle"), os.path.join(root, "sage-3d_official_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_c
    import oracle_np as onp
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene, _ = onp.config1_scene(n=n, seed=3)
        cam = onp.Camera(w, h, 0.6 * w, 0.6 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
        g = FrameGather(h, w, torch.device("cpu"), interleave=interleave)
        r0, r1 = g.band
        g.slab.fill_(-7.0)                                   # garbage that must not survive in the frame
        if interleave:
            whole, _ = oracle_c.render(*scene, cam, threads=1, want="image")
            assert g.render_rows == {"interleave": (world, rank)} and r1 == len(range(rank, g.n_tile_rows, world))
            for dst_row, src_row in _owned_pixel_rows(g):
                g.slab[dst_row] = torch.from_numpy(whole[src_row])
        elif r1 > r0:
            band, _ = oracle_c.render(*scene, cam, None, r0, r1, threads=1, want="image")
            y0, y1 = g.band_pixel_rows
            g.slab[: y1 - y0] = torch.from_numpy(band[y0:y1])
        frame = g.gather()
        if rank == 0:
            full, _ = oracle_c.render(*scene, cam, threads=1, want="image")
            q.put(bool((frame.numpy() == full).all()) and tuple(frame.shape) == (h, w, 3))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,res,interleave", [(2, (112, 160), False), (3, (100, 72), False), (2, (24, 40), False),
                                                  (2, (112, 160), True), (3, (100, 72), True), (3, (24, 40), True)])
def test_tile_row_gather_gloo(world, res, interleave):
    h, w = res
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, 1500, q, interleave)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _batch_worker(rank, world, port, h, w, q, interleave=False):
    """Batched bands: B frames per collective, asynchronous, partial last batch (FrameGather(batch=B))."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 4
        g = FrameGather(h, w, torch.device("cpu"), batch=B, interleave=interleave)
        if interleave:
            pairs = _owned_pixel_rows(g)
            dst_rows = torch.tensor([d for d, _ in pairs], dtype=torch.long)
            src_rows = torch.tensor([s_ for _, s_ in pairs], dtype=torch.float32)
        else:
            y0, y1 = g.band_pixel_rows
            dst_rows = torch.arange(0, y1 - y0, dtype=torch.long)
            src_rows = torch.arange(y0, y1, dtype=torch.float32)
        ok = True
        for n in (B, 3):                                       # a full batch, then a partial one reusing the buffer
            g.slab.fill_(-7.0)
            for b in range(n):
                # frame b, pixel row y: value 1000 b + y (+ channel/column pattern) -> any misplaced slab shows
                rows = src_rows.view(-1, 1, 1)
                g.slab[b, dst_rows] = 1000.0 * b + rows + 0.001 * torch.arange(w, dtype=torch.float32).view(1, -1, 1) \
                    + 0.25 * torch.arange(3, dtype=torch.float32).view(1, 1, -1)
            work = g.gather_batch(n, async_op=True)
            if work is not None:
                work.wait()
            if rank == 0:
                exp_rows = torch.arange(h, dtype=torch.float32).view(-1, 1, 1)
                for b in range(n):
                    exp = 1000.0 * b + exp_rows + 0.001 * torch.arange(w, dtype=torch.float32).view(1, -1, 1) \
                        + 0.25 * torch.arange(3, dtype=torch.float32).view(1, 1, -1)
                    f = g.frame(b)
                    ok = ok and tuple(f.shape) == (h, w, 3) and bool((f == exp).all())
                v = g.frames(n)
                ok = ok and tuple(v.shape) == (n, world, g.slab_rows, w, 3)
            else:
                assert g.frame(0) is None and g.frames() is None
        if rank == 0:
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,res,interleave", [(2, (112, 160), False), (3, (100, 72), False), (3, (100, 72), True)])
def test_batched_tile_row_gather_gloo(world, res, interleave):
    h, w = res
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, h, w, q, interleave)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
