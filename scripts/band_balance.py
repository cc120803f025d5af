"""Per-band GPU time of tile-row-sharded frames on ONE GPU: how uneven are equal-height bands?   (GPU box)
usage: python scripts/band_balance.py [world ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
from sage_gs import Renderer, scenes
from sage_gs.dist import row_partition

dev = torch.device("cuda", 0)
scene = scenes.make_room(3_000_000, seed=2)
cams = scenes.room_cameras(scene, 1920, 1080, n_positions=4, n_yaw=64, seed=2)
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(scene, dev))
rows = (1080 + 15) // 16
poses = list(range(10, 110, 6))
worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
# per-row cost profile (one row at a time) for the balanced variant
prof = np.zeros((len(poses), rows))
slab = torch.zeros((rows * 16, 1920, 3), dtype=torch.float32, device=dev)
for w in worlds:
    bands = row_partition(rows, w)
    t = np.zeros((len(poses), w)); d = np.zeros((len(poses), w))
    for pi, p in enumerate(poses):
        for k, (r0, r1) in enumerate(bands):
            if r1 <= r0:
                continue
            for rep in range(2):
                r.render(cams[p], gs, out_band=slab[: (r1 - r0) * 16], tile_rows=(r0, r1), timing=True)
            st = r.last_stats
            t[pi, k] = st["ms_total"]; d[pi, k] = st["d_total"]
    print(f"world {w}: band ms mean over poses {np.round(t.mean(0), 3)}")
    print(f"   per-pose max/mean: median {np.median(t.max(1) / t.mean(1)):.2f}  worst {np.max(t.max(1) / t.mean(1)):.2f};"
          f"  sweep-level (mean band times) max/mean {t.mean(0).max() / t.mean(0).mean():.2f}")
    print(f"   D share per band {np.round(d.mean(0) / d.mean(0).sum(), 3)}")
