#!/usr/bin/env python3
"""The deep-tile path of k_tile_render on the scenes and resolutions where it runs (3 M room scene and the trained-like scene at the reference's
resolutions and at 1080p): the production frame — which culls windows of long-lived tiles against their live pixels — must equal, bit for bit, the
frame of the D_f-counting instantiation (never takes that path) and the frame of reference binning + extent-only quadrant tests (LOOSE_CULL).
    python scripts/gpu_deep_identity.py [poses per resolution = 12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
r = Renderer(dev, record_capacity=96 << 20)
bad = 0
for kind in ("room", "trained"):
    sc = scenes.cached_room(3_000_000, seed=2) if kind == "room" else scenes.make_trained_like(1_000_000, seed=2)
    gs = r.upload(scenes.to_gaussians(sc, dev))
    for (w, h) in ((320, 240), (640, 480), (1024, 768), (1920, 1080)):
        cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=2)
        deep = mism = 0
        for i in range(n_poses):
            c = cams[(i * 77 + 129) % len(cams)]
            a = r.render(c, gs).clone(); deep += r.last_stats["n_deep_windows"]
            b = r.render(c, gs, stats=True).clone()
            assert r.last_stats["n_deep_windows"] == 0
            l = r.render(c, gs, loose_cull=True)
            if not (bool((a == b).all()) and bool((a == l).all())):
                mism += 1
        bad += mism
        print(f"{kind} {w}x{h}: {n_poses} poses, {deep} windows culled first, {mism} frames differ from the D_f-counting / reference-binning frames", flush=True)
    gs.free()
print("deep-tile identity:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
