#!/usr/bin/env python3
"""Round-4 probe (GPU box): a sweep's frame time under library variants chosen by environment, ONE process, scene loaded once.
    python scripts/r04_sweep.py "SGS_FUSE=0" "SGS_FUSE=1" "SGS_FUSE=1 SGS_FUSE_LDS_PAD=8192" ...
Each variant: a fresh Renderer (the library reads its knobs in sgs_create), then per resolution the driver-shaped sweep
(20 poses as ONE render_batch call, 5 warm-up) and the 100-pose sweep of pipelined single frames; best and median of REPS runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from sage_gs import Renderer, scenes

variants = [a for a in sys.argv[1:] if "=" in a or a == "-"] or ["-"]
RES = [tuple(int(v) for v in a.split("x")) for a in os.environ.get("RES", "1920x1080").split(",")]
REPS = int(os.environ.get("REPS", 5))
KIND = os.environ.get("KIND", "room")
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 3_000_000))
sc = scenes.make_trained_like(N, seed=2) if KIND == "trained" else scenes.cached_room(N, seed=2)
g = scenes.to_gaussians(sc, dev)
KNOBS = ("SGS_FUSE", "SGS_FUSE_LDS_PAD", "SGS_LANES", "SGS_GROUP", "SGS_GROUP_LANES", "SAGE_GS_LIB", "SGS_FUSE_DEPTH")
ref = {}
for v in variants:
    for k in KNOBS:
        os.environ.pop(k, None)
    if v != "-":
        for kv in v.split():
            k, val = kv.split("=", 1); os.environ[k] = val
    r = Renderer(dev, record_capacity=96 << 20)
    gs = r.upload(g)
    for (W, H) in RES:
        cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
        pose = lambda i: (i * 77) % len(cams)
        batch = torch.zeros((20, H, W, 3), dtype=torch.float32, device=dev)
        ring = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(4)]
        r.render_batch([cams[pose(i)] for i in range(8)], gs, out=batch)          # warm-up: every lane's buffers
        for i in range(12):
            r.render(cams[pose(i)], gs, out=ring[i % 4], sync=False, pipelined=True)
        r.sync()
        t20, t100 = [], []
        for rep in range(REPS):
            r.render_batch([cams[pose(i)] for i in range(5)], gs, out=batch)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            r.render_batch([cams[pose(5 + i)] for i in range(20)], gs, out=batch)
            torch.cuda.synchronize(dev); t20.append((time.perf_counter() - t0) / 20)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for i in range(100):
                r.render(cams[pose(10 + i)], gs, out=ring[i % 4], sync=False, pipelined=True)
            r.sync(); torch.cuda.synchronize(dev); t100.append((time.perf_counter() - t0) / 100)
        # identity of the frames against the first variant (bit-exact: every variant runs the same arithmetic)
        chk = r.render_batch([cams[pose(5 + i)] for i in range(4)], gs).cpu()
        key = (W, H)
        same = "ref" if key not in ref else ("identical" if bool((ref[key] == chk).all()) else f"DIFFERENT max|d|={float((ref[key]-chk).abs().max()):.3g}")
        ref.setdefault(key, chk)
        print(f"[{v}] {KIND} {W}x{H}: batch20 ms/frame best {1e3*min(t20):.4f} med {1e3*np.median(t20):.4f} ({1/np.median(t20):.0f} fps) | "
              f"pipelined100 best {1e3*min(t100):.4f} med {1e3*np.median(t100):.4f} ({1/np.median(t100):.0f} fps) | frames {same}", flush=True)
    del gs; r.close(); del r
