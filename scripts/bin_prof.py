#!/usr/bin/env python3
"""Per-workgroup phase breakdown of k_bin_count (profiling build).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
os.environ.setdefault("SAGE_GS_LIB", os.path.join(ROOT, "build", "lib", "libsage_gs_prof.so"))
import numpy as np, torch
from sage_gs import Renderer, scenes
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, 1920, 1080, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
slab = torch.zeros((9 * 16, 1920, 3), device="cuda:0")
for ci, rows in ((5, None), (70, None), (140, None), (20, (27, 36)), (140, (27, 36))):
    for _ in range(3):
        if rows is None:
            r.render(cams[ci], gs, timing=True)
        else:
            r.render(cams[ci], gs, out_band=slab, tile_rows=rows, timing=True)
    st = r.last_stats
    p = r.debug_buffer(101, np.uint64).reshape(-1, 8).astype(np.float64)
    nlive, find, walk, flush, nlist, nvis, tot, big = p.T
    order = np.argsort(-tot)[:5]
    print(f"cam {ci} rows {rows}: count stage {st['ms']['count']*1e3:.0f} us N_v={st['n_visible']} D={st['d_total']} | per workgroup mean (max): "
          f"find {find.mean():.0f} ({find.max():.0f})  walk {walk.mean():.0f} ({walk.max():.0f})  big-rect walk {big.mean():.0f} ({big.max():.0f})  "
          f"flush {flush.mean():.0f} ({flush.max():.0f})  total {tot.mean():.0f} ({tot.max():.0f}) cycles; touched tiles {nlist.mean():.0f}, live chunks {nlive.mean():.0f}")
    print("   slowest five:", [(int(b), int(find[b]), int(walk[b]), int(big[b]), int(flush[b]), int(nlive[b]), int(nvis[b])) for b in order], "(wg, find, walk, big, flush, live chunks of the last pass, visible)")
