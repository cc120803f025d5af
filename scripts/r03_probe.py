#!/usr/bin/env python3
"""Round-3 probe (GPU box): per-stage times alone + host-timed one-frame latency for scene kinds and resolutions.
    python scripts/r03_probe.py [room|trained] [WxH ...]        SAGE_GS_LIB=<variant .so> for A/B"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes

kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1][0].isdigit() else "room"
res = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if a[0].isdigit()] or [(1920, 1080)]
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 3_000_000))
sc = scenes.make_trained_like(N, seed=2) if kind == "trained" else scenes.cached_room(N, seed=2)
r = Renderer(dev, record_capacity=192 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
STAGES = ("preprocess", "count", "emit", "render")
tag = os.path.basename(os.environ.get("SAGE_GS_LIB", "default"))
for (W, H) in res:
    cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
    poses = [(i * 77) % 256 for i in range(5, 5 + int(os.environ.get("NPOSES", 24)))]
    out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    for p in poses[:4]:
        r.render(cams[p], gs, out=out)
    acc = {s: 0.0 for s in STAGES}; d = dv = 0
    for p in poses:
        r.render(cams[p], gs, out=out, timing=True)
        st = r.last_stats
        for s in STAGES: acc[s] += st["ms"][s]
        d += st["d_total"]; dv += st["n_visible"]
    lat = []
    for p in poses:
        t0 = time.perf_counter(); r.render(cams[p], gs, out=out); lat.append(1e3 * (time.perf_counter() - t0))
    n = len(poses)
    print(f"{tag} {kind} {W}x{H} ({n} poses): alone us { {s: round(1e3 * acc[s] / n, 1) for s in STAGES} } N_v={dv // n} D={d // n} | latency ms p50 {np.percentile(lat, 50):.3f} p90 {np.percentile(lat, 90):.3f}", flush=True)
