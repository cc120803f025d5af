#!/usr/bin/env python3
"""A scene several times the benchmark's (default 12 M Gaussians): upload, render, and the size-independent properties —
production frame == reference-binning frame == union of tile-row bands, D consistent.  python scripts/gpu_big_scene.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
t0 = time.time(); sc = scenes.make_room(n, seed=3); print(f"scene {n} in {time.time() - t0:.1f} s", flush=True)
cams = scenes.room_cameras(sc, 1920, 1080, n_positions=2, n_yaw=4, seed=3)
r = Renderer("cuda:0", record_capacity=256 << 20)
t0 = time.time(); gs = r.upload(scenes.to_gaussians(sc, "cuda:0")); torch.cuda.synchronize(); print(f"upload {time.time() - t0:.1f} s", flush=True)
bad = 0
for i, c in enumerate(cams):
    img = r.render(c, gs, timing=True, stats=True).clone(); st = r.last_stats
    plain = r.render(c, gs).clone(); deep = r.last_stats["n_deep_windows"]            # the instantiation a sweep runs (no D_f): deep-tile path included
    ref = r.render(c, gs, full_sort=True, loose_cull=True).clone(); st_ref = r.last_stats
    union = torch.zeros_like(img); d = 0
    for a, b in ((0, 20), (20, 31), (31, 33), (33, 50), (50, 68)):
        r.render(c, gs, out=union, tile_rows=(a, b)); d += r.last_stats["d_total"]
    ok = bool((img == ref).all()) and bool((plain == img).all()) and bool((union == img).all()) and d == st["d_total"] and st["d_total"] <= st_ref["d_total"] and st["n_visible"] == st_ref["n_visible"]
    print(f"cam {i}: N_v={st['n_visible']} D={st['d_total']} (reference binning {st_ref['d_total']}) D_f={st['d_fetched']} deep windows {deep} ms={ {k: round(v, 3) for k, v in st['ms'].items()} } -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
# ... and the same frames as ONE batch (groups of eight: the group's work list of (chunk, frame) pairs, k_preprocess_shared) equal the frames alone
alone = [r.render(c, gs).clone() for c in cams]
batch = r.render_batch(cams, gs)
nb = sum(0 if bool((batch[i] == alone[i]).all()) else 1 for i in range(len(cams)))
print(f"batch of {len(cams)}: {nb} frames differ from the frames rendered alone")
bad += nb
print(f"{bad} mismatches"); sys.exit(1 if bad else 0)
