#!/usr/bin/env python3
"""Hundreds of thousands of splats in a handful of tiles: the long-queue paths of k_tile_render (refinement, windows, rank sort, HBM radix
and its tie repair) through the full comparison with the oracle (queues bit-exact, frames to tolerance, production == reference binning)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import conftest, parity_cases as pc
import oracle_np as onp
from test_gpu_parity import GpuDriver
drv = GpuDriver()
f32 = lambda a: np.ascontiguousarray(a, np.float32)
bad = 0
for name, n, zmode, opac in (("distinct depths, faint", 200_000, "spread", 0.006), ("distinct depths, mixed opacity", 150_000, "spread", None),
                             ("three depth sheets (ties)", 120_000, "sheets", 0.006), ("one depth (all ties)", 60_000, "one", 0.005),
                             ("narrow slab", 250_000, "slab", 0.008)):
    rng = np.random.default_rng(len(name) + n)
    W, H = 96, 80
    cam = onp.Camera(W, H, 80.0, 80.0, W / 2.0, H / 2.0, np.eye(4, dtype=np.float32))
    if zmode == "spread": z = rng.uniform(2.0, 9.0, n)
    elif zmode == "sheets": z = rng.choice([3.0, 3.0000002, 5.5], n)
    elif zmode == "one": z = np.full(n, 4.0)
    else: z = rng.uniform(4.0, 4.02, n)
    # screen position within ~2 tiles around the centre
    u = rng.normal(0.0, 9.0, n); v = rng.normal(0.0, 7.0, n)
    means = np.stack([u * z / 80.0, v * z / 80.0, z], 1)
    scales = np.exp(rng.uniform(math.log(0.002), math.log(0.03), (n, 3)))
    quats = rng.normal(size=(n, 4))
    o = np.full(n, opac) if opac is not None else 1.0 / (1.0 + np.exp(-rng.normal(-4.0, 1.5, n)))
    sh = 0.5 * rng.normal(size=(n, 1, 3))
    scene = (f32(means), f32(scales), f32(quats), f32(np.clip(o, 1e-5, 0.999)), f32(sh), 0)
    try:
        pc.check_against_oracle(drv, scene, cam, None, (0, -1), what=name)
        st = drv.r.last_stats if hasattr(drv, "r") else {}
        print(f"{name}: n={n} ok", flush=True)
    except Exception as e:                               # noqa: BLE001
        bad += 1; print(f"FAIL {name}: {repr(e)[:600]}", flush=True)
print(f"{bad} failures"); sys.exit(1 if bad else 0)
