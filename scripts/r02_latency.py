#!/usr/bin/env python3
"""One-frame latency (host-timed call -> frame complete) at the reference's own resolutions (GPU box):
640x480 (simple_env.py:52), 1024x768 (generate_images.py:43), 1920x1080, 3840x2160; 3 M-Gaussian scene and the 500 k room."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
dev = torch.device("cuda", 0)
r = Renderer(dev, record_capacity=128 << 20)
for n, seed in ((3_000_000, 2), (500_000, 1)):
    sc = scenes.cached_room(n, seed=seed)
    gs = r.upload(scenes.to_gaussians(sc, dev))
    for (w, h) in ((640, 480), (1024, 768), (1920, 1080), (3840, 2160)):
        cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=seed)
        out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        poses = [(i * 77) % len(cams) for i in range(60)]
        for p in poses[:10]:
            r.render(cams[p], gs, out=out)
        lat, gpu = [], []
        for p in poses[10:]:
            t0 = time.perf_counter(); r.render(cams[p], gs, out=out); lat.append(1e3 * (time.perf_counter() - t0))
        for p in poses[10:]:
            r.render(cams[p], gs, out=out, timing=True); gpu.append(r.last_stats["ms"])
        st = {k: round(1e3 * float(np.mean([g[k] for g in gpu])), 1) for k in gpu[0]}
        print(f"N={n} {w}x{h}: latency ms p10 {np.percentile(lat,10):.3f} p50 {np.percentile(lat,50):.3f} p90 {np.percentile(lat,90):.3f}  stages us {st}", flush=True)
    gs.free()
