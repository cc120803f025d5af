#!/usr/bin/env python3
"""Frames of library variants (build/variants/*.so) against the product library's, bit for bit.  python scripts/variant_identity.py a.so b.so ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import torch
from sage_gs import Renderer, scenes, _capi
sc = scenes.cached_room(3_000_000, seed=2)
g = scenes.to_gaussians(sc, "cuda:0")
ref = {}
for name in [None] + sys.argv[1:]:
    lib = None if name is None else _capi.Lib(os.path.join(ROOT, "build", "variants", name))
    r = Renderer("cuda:0", record_capacity=96 << 20, lib=lib)
    gs = r.upload(g)
    bad = 0
    for (w, h) in ((1920, 1080), (640, 480), (320, 240)):
        cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=2)
        for p in (0, 77, 129, 206):
            f = r.render(cams[p], gs).clone()
            if name is None:
                ref[(w, p)] = f
            elif not torch.equal(f, ref[(w, p)]):
                bad += 1
    print(f"[{name or 'product'}] frames differing from the product library's: {bad} of 12", flush=True)
    gs.free(); r.close()
