#!/usr/bin/env python3
"""Random scenes with trained-3DGS statistics (scenes.make_trained_like, SH degree 3, the asset's -90 degree model transform) at random sizes, resolutions
and poses through the full oracle comparison:  python scripts/gpu_fuzz_rooms.py FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import conftest, parity_cases as pc, oracle_np as onp
from test_gpu_parity import GpuDriver
from sage_gs import scenes
drv = GpuDriver()
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.integers(10_000, 250_000))
    w, h = int(rng.integers(160, 1400)), int(rng.integers(120, 900))
    sc = scenes.make_trained_like(n, seed=int(rng.integers(1 << 30)))
    cams = scenes.room_cameras(sc, w, h, n_positions=2, n_yaw=8, seed=int(rng.integers(1 << 30)))
    c = cams[int(rng.integers(len(cams)))]
    view = (np.asarray(c.view, np.float64) @ np.asarray(scenes.MODEL_TO_WORLD, np.float64)).astype(np.float32)
    cam = onp.Camera(c.width, c.height, c.fx, c.fy, c.cx, c.cy, view)
    gy = (h + 15) // 16
    rows = (0, -1) if rng.random() < 0.6 else tuple(sorted(int(v) for v in rng.choice(gy + 1, 2, replace=False)))
    try:
        pc.check_against_oracle(drv, sc.as_tuple(), cam, None, rows, what=f"trained-like seed {seed} (n={n} {w}x{h} rows {rows})")
    except Exception as e:                                   # noqa: BLE001
        bad.append(seed); print("FAIL", seed, repr(e)[:500], flush=True)
print(f"trained-like seeds [{a},{b}): {len(bad)} failures {bad}")
