#!/usr/bin/env python3
"""One seed of gpu_fuzz_rooms.py / gpu_fuzz_trained.py once more, with the worst splat-attribute deviations printed (mean2D, conic) beside the
Gaussians they belong to:   python scripts/fuzz_seed_detail.py rooms|trained SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import conftest, parity_cases as pc, oracle_np as onp, oracle_c
from test_gpu_parity import GpuDriver
from sage_gs import scenes
kind, seed = sys.argv[1], int(sys.argv[2])
drv = GpuDriver()
rng = np.random.default_rng(50_000 + seed)
if kind == "rooms":
    n = int(rng.integers(20_000, 600_000)); w, h = int(rng.integers(320, 2000)), int(rng.integers(240, 1200))
    sc = scenes.make_room(n, seed=int(rng.integers(1 << 30)))
else:
    n = int(rng.integers(10_000, 250_000)); w, h = int(rng.integers(160, 1400)), int(rng.integers(120, 900))
    sc = scenes.make_trained_like(n, seed=int(rng.integers(1 << 30)))
cams = scenes.room_cameras(sc, w, h, n_positions=2, n_yaw=8, seed=int(rng.integers(1 << 30)))
c = cams[int(rng.integers(len(cams)))]
view = (np.asarray(c.view, np.float64) @ np.asarray(scenes.MODEL_TO_WORLD, np.float64)).astype(np.float32)
cam = onp.Camera(c.width, c.height, c.fx, c.fy, c.cx, c.cy, view)
scene = sc.as_tuple()
drv.upload(*scene)
drv.render(cam, None, (0, -1), full_sort=True, loose_cull=True)
ref, aux = oracle_c.render(*scene, cam, None, 0, -1)
off, ids, slot_ids, splats = drv.intermediates()
vis = np.nonzero(aux["tiles"] > 0)[0]
order = np.argsort(slot_ids); sp = splats[order]; f = sp.view(np.float32)
dxy = np.abs(f[:, 0:2] - aux["xy"][vis])
i = int(np.argmax(dxy.max(axis=1))); g = vis[i]
print(f"n={n} {w}x{h}; worst mean2D deviation {dxy.max():.3e} at Gaussian {g}: device {f[i, 0:2]} oracle {aux['xy'][g]} (fp32 ulp there {np.spacing(np.float32(np.abs(aux['xy'][g]).max())):.3e}); "
      f"mean {scene[0][g]} scale {scene[1][g]} depth {aux['depth'][g] if 'depth' in aux else '?'} tiles {aux['tiles'][g]} rect {aux['rect'][g]}")
rel = np.abs(np.stack([f[:, 2], f[:, 3], f[:, 4]], 1) - aux["conic"][vis]) / (np.abs(aux["conic"][vis]) + 1e-12)
j = int(np.argmax(rel.max(axis=1))); g = vis[j]
print(f"worst conic deviation {rel.max():.3e} at Gaussian {g}: device {f[j, 2:5]} oracle {aux['conic'][g]}; scale {scene[1][g]} quat {scene[2][g]} mean {scene[0][g]} tiles {aux['tiles'][g]} rect {aux['rect'][g]} xy {aux['xy'][g]}")
print("pixels beyond tolerance:", int((np.abs(drv.render(cam)[0] - ref).max(axis=-1) > 1e-3).sum()))
