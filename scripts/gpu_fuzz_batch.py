#!/usr/bin/env python3
"""sgs_render_batch* against the same frames rendered alone, bit for bit, on random indoor scenes (scenes.make_room / make_trained_like) with
random camera sets — runs of neighbouring headings (the groups the library projects with ONE grid over the scene's chunks,
k_preprocess_shared), views that share nothing (every frame its own live list), and mixes of both — at random resolutions (some small
enough for fine tiles), whole frames and bands of tile rows:   python scripts/gpu_fuzz_batch.py FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from sage_gs import Renderer, scenes

dev = torch.device("cuda", 0)
r = Renderer(dev)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad, frames = [], 0
for seed in range(a, b):
    rng = np.random.default_rng(90_000 + seed)
    n = int(rng.integers(20_000, 500_000))
    w, h = [(320, 240), (640, 480), (1024, 768), (1920, 1080), (int(rng.integers(200, 1500)), int(rng.integers(150, 900)))][int(rng.integers(5))]
    sc = (scenes.make_trained_like if rng.random() < 0.3 else scenes.make_room)(n, seed=int(rng.integers(1 << 30)))
    cams = scenes.room_cameras(sc, w, h, n_positions=2, n_yaw=32, seed=int(rng.integers(1 << 30)))
    nb = int(rng.integers(2, 14))
    kind = int(rng.integers(3))
    if kind == 0:      # a path: neighbouring headings from one position
        p0 = int(rng.integers(64)); sel = [(p0 // 32) * 32 + (p0 + k) % 32 for k in range(nb)]
    elif kind == 1:    # anything
        sel = [int(v) for v in rng.integers(0, 64, nb)]
    else:              # runs of neighbours between unrelated views
        sel = []
        while len(sel) < nb:
            p0 = int(rng.integers(64)); run = int(rng.integers(1, 5))
            sel += [(p0 // 32) * 32 + (p0 + k) % 32 for k in range(run)]
        sel = sel[:nb]
    cl = [cams[p] for p in sel]
    gy = (h + 15) // 16
    rows = None if rng.random() < 0.6 else tuple(sorted(int(v) for v in rng.choice(gy + 1, 2, replace=False)))
    fine = bool(rng.random() < 0.7)
    try:
        gs = r.upload(scenes.to_gaussians(sc, dev))
        alone = [r.render(c, gs, tile_rows=rows, fine_tiles=fine).clone() for c in cl]
        out = torch.full((nb, h, w, 3), -1.0, dtype=torch.float32, device=dev)
        if rows is None:
            r.render_batch(cl, gs, out=out, fine_tiles=fine)
            y0, y1 = 0, h
        else:
            r.render_batch(cl, gs, out=out, tile_rows=rows, fine_tiles=fine)
            y0, y1 = 16 * rows[0], min(h, 16 * rows[1])
        for i, fr in enumerate(alone):
            if not torch.equal(out[i, y0:y1], fr[y0:y1]):
                raise AssertionError(f"frame {i} of {nb} (pose {sel[i]}) differs: {int((out[i, y0:y1] != fr[y0:y1]).any(dim=-1).sum())} pixels")
        if y1 > y0 and float(torch.stack([f[y0:y1].max() for f in alone]).max()) <= 0.0:
            print(f"  (seed {seed}: empty frames)")
        frames += nb
        gs.free()
    except Exception as e:                                   # noqa: BLE001
        bad.append(seed); print("FAIL", seed, f"n={n} {w}x{h} kind={kind} rows={rows} fine={fine}", repr(e)[:400], flush=True)
print(f"batch seeds [{a},{b}): {frames} frames, {len(bad)} failures {bad}")
r.close()
