#!/usr/bin/env python3
"""3 M room scene at the reference's resolutions: production frame == reference-binning frame (LOOSE_CULL + FULL_SORT) == lazily sorted
reference binning == D_f-counting instantiation, bit for bit, over a set of poses (the long-queue paths: refinement, windows, full batches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
sc = scenes.cached_room(3_000_000, seed=2) if kind == "room" else scenes.make_trained_like(1_000_000, seed=2)
r = Renderer("cuda:0", record_capacity=128 << 20); gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
bad = 0
for (W, H) in ((640, 480), (1024, 768), (320, 240)):
    cams = scenes.room_cameras(sc, W, H, 4, 64, seed=2)
    for i in range(16):
        c = cams[(i * 77 + 5) % 256]
        img = r.render(c, gs).clone()
        st = None
        for name, kw in (("stats", dict(stats=True)), ("loose", dict(loose_cull=True)), ("loose_full", dict(loose_cull=True, full_sort=True)), ("full", dict(full_sort=True))):
            im2 = r.render(c, gs, **kw)
            if name == "stats": st = dict(r.last_stats)
            if not bool((im2 == img).all()):
                bad += 1; d = (im2 - img).abs().amax(-1); print(f"MISMATCH {W}x{H} pose {i} {name}: {int((d > 0).sum())} px, max {float(d.max()):.3e}", flush=True)
    print(f"{kind} {W}x{H}: 16 poses checked, last: D={st['d_total']} D_f={st['d_fetched']} max_tile_len={st['max_tile_len']} spill={st['n_spill_tiles']}", flush=True)
print(f"{bad} mismatches"); sys.exit(1 if bad else 0)
