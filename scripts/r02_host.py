#!/usr/bin/env python3
"""Host-side cost of issuing a frame (GPU box): enqueue time per frame for a light band, Python loop vs one C call."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes, _capi
dev = torch.device("cuda", 0)
W, H = 1920, 1080
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [(i * 77) % 256 for i in range(10, 110)]
ring = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(8)]
r = Renderer(dev, record_capacity=24 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
n = 96
for rows in ((60, 68), (27, 36), None):
    kw = {} if rows is None else {"tile_rows": rows}
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            r.render(cams[poses[i]], gs, out=ring[i % 8], sync=False, pipelined=True, **kw)
        t1 = time.perf_counter()
        r.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rows {rows}: python loop enqueue {1e3 * (t1 - t0) / n:.4f} ms/frame, total {1e3 * (t2 - t0) / n:.4f} ms/frame")
    # the same frames through ONE C call (sgs_render_batch: contiguous [B,H,W,3] output)
    out = torch.zeros((n, H, W, 3), dtype=torch.float32, device=dev)
    cl = [cams[poses[i]] for i in range(n)]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_batch(cl, gs, out=out, tile_rows=rows)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rows {rows}: render_batch (marshalling {n} cameras in Python + one C call) {1e3 * (t2 - t0) / n:.4f} ms/frame")
    arr = (_capi.SgsCamera * n)(*[r._c_camera(c, gs) for c in cl])
    cfg = r._c_config(None)
    r0, r1 = (0, -1) if rows is None else rows
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r._lib.check(r._lib.sgs_render_batch(r._ctx, gs.handle, arr, n, C.byref(cfg), r0, r1, out.data_ptr(), None, r._stream()), r._ctx)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rows {rows}: sgs_render_batch alone (cameras pre-marshalled) {1e3 * (t2 - t0) / n:.4f} ms/frame")
    del out
