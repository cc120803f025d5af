import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd")); sys.path.insert(0, ROOT)
import torch
from sage_gs import Renderer, scenes
dev = torch.device("cuda", 0)
sc = scenes.cached_room(3_000_000, seed=2)
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
cams = scenes.room_cameras(sc, 1920, 1080, n_positions=4, n_yaw=64, seed=2)
cl = [cams[(i * 77) % 256] for i in range(10, 110)]
out = torch.zeros((100, 1080, 1920, 3), dtype=torch.float32, device=dev)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_batch(cl, gs, out=out); torch.cuda.synchronize()
    print("batch", rep, (time.perf_counter() - t0) / 100 * 1e3, "ms/frame", flush=True)
