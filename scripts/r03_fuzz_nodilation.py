#!/usr/bin/env python3
"""Seeded frames with dilation 0 (sub-pixel specks and needles keep their true, nearly singular conics: the exact-trip path of the
composite for conics that round to an indefinite form) and extreme thresholds, through the full comparison with the oracle."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import conftest, parity_cases as pc
import oracle_np as onp
from test_gpu_parity import GpuDriver
drv = GpuDriver()
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(77_000 + seed)
    n = int(rng.integers(1, 1500)); w, h = int(rng.integers(17, 500)), int(rng.integers(17, 400)); deg = int(rng.integers(0, 4))
    lo = float(10 ** rng.uniform(-5.0, -1.5)); hi = lo * float(10 ** rng.uniform(0.3, 4.5))
    scene = pc.random_scene(n, 88_000 + seed, deg, box=((-3, 3), (-2, 2), (-1, 9)), scale=(lo, min(hi, 20.0)), opac_mu=float(rng.uniform(-3.0, 3.0)))
    eye = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(-3, 6)]); target = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(3, 8)])
    if np.linalg.norm(target - eye) < 0.5: target = eye + np.array([0.1, 0.0, 1.0])
    view = pc.look_at_view(eye, target, np.array([rng.normal(0, 0.3), 1.0, rng.normal(0, 0.3)]))
    f = float(w * rng.uniform(0.12, 1.6))
    cam = onp.Camera(w, h, f, f * float(rng.uniform(0.9, 1.1)), w / 2.0 + float(rng.uniform(-3, 3)), h / 2.0 + float(rng.uniform(-3, 3)), view)
    cfg = onp.Config(near=float(rng.choice([0.2, 0.05, 0.5])), dilation=float(rng.choice([0.0, 0.0, 1e-3, 0.02])),
                     alpha_min=float(rng.choice([1.0 / 255.0, 0.05, 0.0005])), alpha_max=float(rng.choice([0.99, 0.5, 0.999])),
                     t_min=float(rng.choice([1.0e-4, 1.0e-2, 1.0e-6])), background=tuple(float(v) for v in rng.uniform(0, 1, 3)), sh_degree=-1)
    try:
        pc.check_against_oracle(drv, scene, cam, cfg, (0, -1), what=f"no-dilation seed {seed} (n={n} {w}x{h} deg {deg} dil {cfg.dilation} amin {cfg.alpha_min} amax {cfg.alpha_max} tmin {cfg.t_min})")
    except Exception as e:                                   # noqa: BLE001
        bad.append(seed); print("FAIL", seed, repr(e)[:500], flush=True)
print(f"no-dilation seeds [{a},{b}): {len(bad)} failures {bad}")
