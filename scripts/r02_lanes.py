#!/usr/bin/env python3
"""How many frames in flight pay for full frames and for a rank's band (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
from sage_gs.dist import row_partition
dev = torch.device("cuda", 0)
W, H = 1920, 1080
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [(i * 77) % 256 for i in range(10, 110)]
ring = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(12)]
g = scenes.to_gaussians(sc, dev)
def rate(r, gs, rows=None, n=96):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            kw = {} if rows is None else {"tile_rows": rows}
            r.render(cams[poses[i % len(poses)]], gs, out=ring[i % len(ring)], sync=False, pipelined=True, **kw)
        r.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
def rate_batch(r, gs, rows=None, n=96):
    """the same frames through ONE call into the library (sgs_render_batch_strided)"""
    cl = [cams[poses[i % len(poses)]] for i in range(n)]
    if rows is None:
        buf = torch.zeros((n, H, W, 3), dtype=torch.float32, device=dev); kw = {"out": buf}
    else:
        buf = torch.zeros((n, (rows[1] - rows[0]) * 16, W, 3), dtype=torch.float32, device=dev); kw = {"out_bands": buf, "tile_rows": rows}
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_batch(cl, gs, **kw)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
bands = row_partition(68, 8)
os.environ["SGS_LANES"] = "3"
r = Renderer(dev, record_capacity=24 << 20); gs = r.upload(g)
per = [rate(r, gs, b) for b in bands]
print(f"one frame per launch set, 3 lanes (python loop): full {rate(r, gs):.4f}  bands max {max(per):.4f} mean {np.mean(per):.4f} {np.round(per, 3).tolist()}", flush=True)
gs.free(); r.close()
for group, gl in ((1, 3), (2, 2), (2, 4), (4, 1), (4, 2), (8, 1)):
    os.environ["SGS_GROUP"], os.environ["SGS_GROUP_LANES"] = str(group), str(gl)
    r = Renderer(dev, record_capacity=24 << 20); gs = r.upload(g)
    per = [rate_batch(r, gs, b) for b in bands]
    print(f"group {group} x {gl} streams: full {rate_batch(r, gs):.4f}  bands max {max(per):.4f} mean {np.mean(per):.4f} {np.round(per, 3).tolist()}", flush=True)
    gs.free(); r.close()
