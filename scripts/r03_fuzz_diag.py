#!/usr/bin/env python3
"""Diagnose one fuzz seed: which of the test-hook frames differs from the production frame, where, by how much."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import conftest, parity_cases as pc
import oracle_np as onp
from test_gpu_parity import GpuDriver
seed = int(sys.argv[1]); max_n = int(sys.argv[2]); max_res = (2400, 1400); wild = True
drv = GpuDriver()
caught = {}
orig = pc.check_against_oracle
def probe(drv_, scene, cam, cfg=None, rows=(0, -1), what="", queues=True):
    drv_.upload(*scene)
    img, st = drv_.render(cam, cfg, rows)
    print(what, "N_v", st["n_visible"], "D", st["d_total"], "D_f", st["d_fetched"], "max_tile_len", st["max_tile_len"], "spill", st["n_spill_tiles"])
    for name, kw in (("plain", dict(stats=False)), ("no_chunk_cull", dict(chunk_cull=False)), ("full_sort", dict(full_sort=True)),
                     ("loose_lazy", dict(loose_cull=True)), ("loose_full", dict(full_sort=True, loose_cull=True))):
        im2, st2 = drv_.render(cam, cfg, rows, **kw)
        d = np.abs(im2.astype(np.float64) - img.astype(np.float64)).max(axis=-1)
        ys, xs = np.nonzero(d > 0)
        print(f"  {name}: {len(ys)} pixels differ, max {d.max():.3e}; D {st2['d_total']} D_f {st2['d_fetched']}", end="")
        if len(ys):
            tiles = sorted(set((int(y) // 16, int(x) // 16) for y, x in zip(ys, xs)))
            print(f"  tiles {tiles[:8]} first px {(int(ys[0]), int(xs[0]))} prod {img[ys[0], xs[0]]} other {im2[ys[0], xs[0]]}", end="")
        print()
    ref, aux = __import__("oracle_c").render(*scene, cam, cfg) if cfg is not None else __import__("oracle_c").render(*scene, cam)
    d = np.abs(img.astype(np.float64) - ref).max(axis=-1)
    print("  vs oracle: max", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
    im2, _ = drv_.render(cam, cfg, rows, loose_cull=True)
    dd = np.abs(im2.astype(np.float64) - img.astype(np.float64)).max(axis=-1)
    for y, x in zip(*np.nonzero(dd > 0)):
        print(f"  px ({y},{x}): prod {img[y, x]} loose {im2[y, x]} oracle {ref[y, x]} margin {aux['margin'][y, x]:.3e}  |prod-or| {np.abs(img[y,x]-ref[y,x]).max():.2e} |loose-or| {np.abs(im2[y,x]-ref[y,x]).max():.2e}")
    print("  cfg", cfg)
pc.check_against_oracle = probe
pc.case_fuzz(drv, [seed], max_n, max_res, wild)
