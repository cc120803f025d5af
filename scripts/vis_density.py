"""Development probe: how dense is visibility inside the 64-Gaussian chunks k_preprocess works on?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes, _capi
sc = scenes.make_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, 1920, 1080, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
means = np.asarray(sc.means, np.float64)
for ci in (5, 70, 140, 200):
    r.render(cams[ci], gs)
    ids = r.debug_buffer(_capi.BUF_SLOT_IDS, np.uint32)
    live = (ids != 0xFFFFFFFF)[: 3_000_000 // 64 * 64].reshape(-1, 64)
    per = live.sum(1)
    V = np.asarray(cams[ci].view, np.float64) @ np.asarray(sc.model_to_world, np.float64)
    tz = means @ V[2, :3] + V[2, 3]
    front = (tz > 0.2)[: 3_000_000 // 64 * 64].reshape(-1, 64)
    fper = front.sum(1)
    print(f"cam {ci}: visible {live.sum()}  chunks with a visible lane {np.count_nonzero(per)} of {len(per)} (mean fill {per[per>0].mean():.1f}/64); "
          f"front {front.sum()} chunks with a front lane {np.count_nonzero(fper)} (mean fill {fper[fper>0].mean():.1f}/64); "
          f"128-B SH lines touched/needed: {np.count_nonzero(live.reshape(-1, 8, 8).any(2)) * 128 * 12 / 1e6:.0f} MB vs {live.sum() * 192 / 1e6:.0f} MB")
