#!/usr/bin/env python3
"""How many instructions does k_tile_render spend on a tile BEFORE it blends anything?  A scene of k small splats per 16x16 tile
(k = argv[1]), 1920x1080; run under  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES  and divide by 4 waves x 8160 tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, Camera, Gaussians
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, f = 1920, 1080, 733.0
gx, gy = W // 16, (H + 15) // 16
rng = np.random.default_rng(0)
tx, ty = np.meshgrid(np.arange(gx), np.arange(gy), indexing="xy")
px = (tx.reshape(-1, 1) * 16 + rng.uniform(3, 12, (gx * gy, k))).reshape(-1)
py = (ty.reshape(-1, 1) * 16 + rng.uniform(3, 12, (gx * gy, k))).reshape(-1)
py = np.minimum(py, H - 2)
z = rng.uniform(3.0, 6.0, px.shape)
means = np.stack([(px - W / 2) / f * z, (py - H / 2) / f * z, z], 1).astype(np.float32)
n = means.shape[0]
dev = "cuda:0"
g = Gaussians(torch.tensor(means, device=dev), torch.full((n, 3), 0.004, dtype=torch.float32, device=dev),
              torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(n, 1), torch.full((n,), 0.5, device=dev), torch.rand(n, 1, 3, device=dev), 0)
r = Renderer("cuda:0")
gs = r.upload(g)
cam = Camera(W, H, f, f, W / 2, H / 2, np.eye(4, dtype=np.float32))
for _ in range(5):
    r.render(cam, gs)
st = r.last_stats
print(f"k={k}: N={n} N_v={st['n_visible']} D={st['d_total']} max_tile_len={st['max_tile_len']}")
