#!/usr/bin/env python3
"""One rank's band of a tile-row-sharded sweep, as ShardedRenderer.render_batch issues it (frame groups), for a kernel trace:
    rocprofv3 --kernel-trace --stats -d out -- python scripts/r03_band_prof.py <row0> <row1> [per_call]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
r0, r1 = int(sys.argv[1]), int(sys.argv[2]); per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 32
W, H = 1920, 1080
dev = torch.device("cuda", 0)
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [(i * 77) % 256 for i in range(10, 110)]
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
n = 64
cl = [cams[poses[i % len(poses)]] for i in range(n)]
buf = torch.zeros((per_call, (r1 - r0) * 16, W, 3), dtype=torch.float32, device=dev)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for c0 in range(0, n, per_call):
        r.render_batch(cl[c0:c0 + per_call], gs, out_bands=buf, tile_rows=(r0, r1))
    torch.cuda.synchronize()
    print(f"rows [{r0},{r1}) per_call {per_call}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/frame", flush=True)
