"""Development probe: host time to ENQUEUE pipelined frames vs time until they complete."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
import torch
from sage_gs import Renderer, scenes
sc = scenes.make_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, 1920, 1080, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
outs = [torch.zeros((1080, 1920, 3), device="cuda:0") for _ in range(4)]
for timing in (False, True):
    for i in range(10):
        r.render(cams[i], gs, out=outs[i % 4], sync=False, pipelined=True)
    r.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        r.render(cams[(10 + i) % 256], gs, out=outs[i % 4], sync=False, pipelined=True, timing=timing)
    t1 = time.perf_counter()
    r.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"timing={timing}: enqueue {1e6 * (t1 - t0) / 200:.0f} us/frame, complete {1e6 * (t2 - t0) / 200:.0f} us/frame -> {200 / (t2 - t0):.0f} frames/s")
