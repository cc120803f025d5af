#!/usr/bin/env python3
"""Fine tiles A/B (GPU box): stage times ALONE, one-frame latency and the pipelined rate at the reference's resolutions with
sgs_tuning.fine_tile_pixels = 0 (16x16-pixel tiles), = the resolution's pixel count (8x8-pixel tiles) and four times that (4x4), one process, one upload;
the two renderings of every probed pose are compared (max |d|).
    python scripts/fine_probe.py [trained] [n=24] [res=320x240,640x480,1024x768]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sage_gs import Renderer, scenes

args = sys.argv[1:]
N = next((int(a[2:]) for a in args if a.startswith("n=")), 24)
RES = [tuple(int(v) for v in r.split("x")) for r in next((a[4:] for a in args if a.startswith("res=")), "320x240,640x480,1024x768").split(",")]
dev = torch.device("cuda", 0)
sc = scenes.make_trained_like(3_000_000, seed=2) if "trained" in args else scenes.cached_room(3_000_000, seed=2)
STAGES = ("preprocess", "count", "emit", "render")
poses = [(i * 77) % 256 for i in range(5, 105)]
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
for (w, h) in RES:
    cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=2)
    buf = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    ring = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in range(4)]
    frames = {}
    for tag, fp in (("coarse", 0), ("fine", w * h), ("fine2", 4 * w * h), ("coarse", 0), ("fine", w * h), ("fine2", 4 * w * h)):
        r.set_tuning(fine_tile_pixels=fp)
        acc = {s: [] for s in STAGES}; tot = []; nv = d = mx = 0
        for p in poses[:4]:
            r.render(cams[p], gs, out=buf)
        for p in poses[:N]:
            r.render(cams[p], gs, out=buf, timing=True)
            st = r.last_stats
            for s in STAGES:
                acc[s].append(st["ms"][s])
            tot.append(st["ms_total"]); nv += st["n_visible"]; d += st["d_total"]; mx = max(mx, st["max_tile_len"])
        frames[tag] = [r.render(cams[p], gs).clone() for p in poses[:8]]
        lat = []
        for p in poses[:32]:
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render(cams[p], gs, out=buf); lat.append(1e3 * (time.perf_counter() - t0))
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(100):
                r.render(cams[poses[i % len(poses)]], gs, out=ring[i % 4], sync=False, pipelined=True)
            r.sync(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        worst = {s: round(1e3 * float(np.max(acc[s])), 1) for s in STAGES}
        print(f"[{tag}] {w}x{h}: alone us { {s: round(1e3 * float(np.mean(acc[s])), 1) for s in STAGES} } total {1e3 * float(np.mean(tot)):.1f} "
              f"(slowest render {worst['render']})  N_v={nv // N} D={d // N} max_tile_len={mx} n_tiles={st['n_tiles']} | latency p50 {np.percentile(lat, 50):.3f} "
              f"p90 {np.percentile(lat, 90):.3f} ms | pipelined {best:.4f} ms/frame", flush=True)
    for other in ("fine", "fine2"):
        dd = max(float((a - b).abs().max()) for a, b in zip(frames["coarse"], frames[other]))
        nd = sum(int(((a - b).abs() > 1e-5).any(dim=-1).sum()) for a, b in zip(frames["coarse"], frames[other]))
        print(f"    {w}x{h}: {other} vs coarse over 8 poses: max |d| {dd:.3e}, pixels differing by more than 1e-5: {nd}", flush=True)
gs.free(); r.close()
