"""Equal-height vs queue-length-balanced tile-row bands, band times on ONE GPU.   (GPU box)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
from sage_gs import Renderer, scenes, _capi
from sage_gs.dist import row_partition

dev = torch.device("cuda", 0)
scene = scenes.make_room(3_000_000, seed=2)
cams = scenes.room_cameras(scene, 1920, 1080, n_positions=4, n_yaw=64, seed=2)
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(scene, dev))
rows, gx = 68, 120
frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev)
slab = torch.zeros((rows * 16, 1920, 3), dtype=torch.float32, device=dev)

def balanced(w, world):
    """contiguous bands minimising the largest band weight (every band >= 1 row)"""
    n = len(w); pre = np.concatenate([[0.0], np.cumsum(w)])
    best = np.full((world + 1, n + 1), np.inf); arg = np.zeros((world + 1, n + 1), int)
    best[0, 0] = 0
    for k in range(1, world + 1):
        for j in range(k, n + 1):
            for i in range(k - 1, j):
                v = max(best[k - 1, i], pre[j] - pre[i])
                if v < best[k, j]:
                    best[k, j] = v; arg[k, j] = i
    cuts = [n]
    for k in range(world, 0, -1):
        cuts.append(arg[k, cuts[-1]])
    cuts = cuts[::-1]
    return [(cuts[i], cuts[i + 1]) for i in range(world)]

calib = list(range(10, 110, 12))
drow = np.zeros(rows)
for p in calib:
    r.render(cams[p], gs, out=frame)
    off = r.debug_buffer(_capi.BUF_TILE_OFFSETS, np.uint32).astype(np.int64)
    d = np.diff(off)[: rows * gx].reshape(rows, gx).sum(1)
    drow += d
drow /= len(calib)
print("D per row (k):", np.round(drow / 1e3).astype(int).tolist())
poses = list(range(13, 110, 6))

def band_times(bands):
    t = np.zeros((len(poses), len(bands)))
    for pi, p in enumerate(poses):
        for k, (r0, r1) in enumerate(bands):
            for rep in range(2):
                r.render(cams[p], gs, out_band=slab[: (r1 - r0) * 16], tile_rows=(r0, r1), timing=True)
            t[pi, k] = r.last_stats["ms_total"]
    return t

for world in ():
    t = band_times(row_partition(rows, world))
    print(f"world {world} equal   : sweep max {t.mean(0).max():.3f} mean {t.mean(0).mean():.3f}  per-pose-max mean {t.max(1).mean():.3f}")
    for kappa in (0.0, 10e3, 25e3, 50e3):
        bands = balanced(drow + kappa, world)
        t = band_times(bands)
        print(f"world {world} kappa {kappa/1e3:4.0f}k: sweep max {t.mean(0).max():.3f} mean {t.mean(0).mean():.3f}  per-pose-max mean {t.max(1).mean():.3f}  rows {[b - a for a, b in bands]}")

import time
def band_rate(bands, n=96):
    """ms per frame of each band when the sweep's frames go through the pipelined lanes (what a rank does in rows mode)"""
    out = []
    ring = [torch.zeros((rows * 16, 1920, 3), dtype=torch.float32, device=dev) for _ in range(4)]
    for (r0, r1) in bands:
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                r.render(cams[(10 + i) % len(cams)], gs, out_band=ring[i % 4][: (r1 - r0) * 16], tile_rows=(r0, r1), sync=False, pipelined=True)
            r.sync(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        out.append(dt)
    return np.array(out)

def interleaved_rate(world, n=96):
    out = []
    ring = [torch.zeros((rows * 16, 1920, 3), dtype=torch.float32, device=dev) for _ in range(4)]
    for phase in range(world):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                r.render(cams[(10 + i) % len(cams)], gs, out_band=ring[i % 4], interleave=(world, phase), sync=False, pipelined=True)
            r.sync(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        out.append(dt)
    return np.array(out)

print("---- pipelined band rates (ms per frame per rank)")
for world in (2, 4, 8):
    e = band_rate(row_partition(rows, world))
    print(f"world {world} contiguous : max {e.max():.3f} mean {e.mean():.3f}  {np.round(e, 3).tolist()}")
    b = interleaved_rate(world)
    print(f"world {world} interleaved: max {b.max():.3f} mean {b.mean():.3f}  {np.round(b, 3).tolist()}")
