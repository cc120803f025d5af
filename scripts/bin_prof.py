#!/usr/bin/env python3
"""Per-workgroup phase cycles of k_bin_count (profiling build build/lib/libsage_gs_prof.so, debug buffer 101).  Run on the GPU box.
    python scripts/bin_prof.py [room|trained] [W H]        (POSES=5,129 selects poses)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
os.environ["SAGE_GS_LIB"] = os.path.join(ROOT, "build", "lib", "libsage_gs_prof.so")
import numpy as np, torch
from sage_gs import Renderer, scenes
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sc = scenes.cached_room(3_000_000, seed=2) if kind == "room" else scenes.make_trained_like(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
for ci in [int(v) for v in os.environ.get("POSES", "5,129,206").split(",")]:
    for _ in range(3):
        r.render(cams[ci], gs, timing=True)
    st = r.last_stats
    p = r.debug_buffer(101, np.uint64).reshape(-1, 8).astype(np.float64)
    live, find, walk, flush, nlist, nvis, tot, big = p.T
    m = tot > 0
    print(f"[{kind} {W}x{H}] cam {ci}: count stage {st['ms']['count']*1e3:.0f} us emit {st['ms']['emit']*1e3:.0f} us  N_v={st['n_visible']} D_s={st['d_super']} D={st['d_total']} | k_bin_count per workgroup mean (max) cycles: "
          f"find {find[m].mean():.0f} ({find[m].max():.0f})  walk {walk[m].mean():.0f} ({walk[m].max():.0f})  big {big[m].mean():.0f} ({big[m].max():.0f})  flush {flush[m].mean():.0f} ({flush[m].max():.0f})  "
          f"total {tot[m].mean():.0f} ({tot[m].max():.0f}); touched super-tiles {nlist[m].mean():.0f}, live chunks (last pass) {live[m].mean():.0f}, splats {nvis[m].mean():.0f}", flush=True)
