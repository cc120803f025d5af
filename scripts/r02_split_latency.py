#!/usr/bin/env python3
"""Latency of ONE frame when it is issued as K bands of tile rows on the pipelined lanes at once (bands are independent
after binning and their union is the frame bit for bit) against the plain synchronous frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda", 0)
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [(i * 77) % 256 for i in range(10, 110)]
r = Renderer(dev, record_capacity=96 << 20); gs = r.upload(scenes.to_gaussians(sc, dev))
out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
gy = (H + 15) // 16
def lat(fn):
    ts = []
    for p in poses:
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(cams[p]); ts.append(1e3 * (time.perf_counter() - t0))
    return np.percentile(ts, [10, 50, 90]).round(4).tolist(), round(float(np.mean(ts)), 4)
def whole(c):
    r.render(c, gs, out=out)
def split(k, cuts=None):
    cuts = cuts or [round(i * gy / k) for i in range(k + 1)]
    def f(c):
        for a, b in zip(cuts[:-1], cuts[1:]):
            r.render(c, gs, out=out, tile_rows=(a, b), sync=False, pipelined=True)
        r.sync()
    return f
ref = None
for name, fn in (("whole", whole), ("2 bands", split(2)), ("3 bands", split(3)), ("2 bands 40/60", split(2, [0, int(gy * 0.45), gy])), ("whole", whole)):
    fn(cams[poses[0]]); img = out.clone()
    if ref is None: ref = img
    assert (img == ref).all()
    print(f"{W}x{H} {name}: p10/p50/p90 {lat(fn)}", flush=True)
