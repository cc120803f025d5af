import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sage_gs import Renderer, scenes
dev = torch.device("cuda", 0)
sc = scenes.make_room(2000, seed=1)
r = Renderer(dev)
gs = r.upload(scenes.to_gaussians(sc, dev))
cams = scenes.room_cameras(sc, 64, 64, n_positions=1, n_yaw=4, seed=1)
buf = torch.zeros((64, 64, 3), dtype=torch.float32, device=dev)
for _ in range(20): r.render(cams[0], gs, out=buf)
lat = []
for i in range(400):
    t0 = time.perf_counter(); r.render(cams[i % 4], gs, out=buf); lat.append(1e6 * (time.perf_counter() - t0))
print("tiny frame, host-timed call -> complete: p10 %.1f p50 %.1f p90 %.1f us" % tuple(np.percentile(lat, [10, 50, 90])))
r.render(cams[0], gs, out=buf, timing=True); print("GPU stages (events) us:", {k: round(1e3 * v, 1) for k, v in r.last_stats["ms"].items()}, "total", round(1e3 * r.last_stats["ms_total"], 1))
# launch-only cost: async issue without waiting
t0 = time.perf_counter()
for i in range(200): r.render(cams[i % 4], gs, out=buf, sync=False, pipelined=True)
t1 = time.perf_counter(); r.sync(); torch.cuda.synchronize()
print("issue cost per frame (async): %.1f us" % (1e6 * (t1 - t0) / 200))
