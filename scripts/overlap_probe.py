"""Development probe: how much do independent frames overlap when issued on K streams?
K contexts (one renderer each, scene uploaded K times) render frames round-robin, each on its own torch stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
import torch
from sage_gs import Renderer, scenes

N = int(os.environ.get("PROBE_N", 3_000_000)); W, H = 1920, 1080
scene = scenes.make_room(N, seed=2)
cams = scenes.room_cameras(scene, W, H, n_positions=4, n_yaw=64, seed=2)
g = scenes.to_gaussians(scene, "cuda:0")
for K in (1, 2, 3, 4):
    rs = [Renderer("cuda:0") for _ in range(K)]
    scs = [r.upload(g) for r in rs]
    streams = [torch.cuda.Stream() for _ in range(K)]
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(K)]
    def run(frames):
        for i in frames:
            k = i % K
            with torch.cuda.stream(streams[k]):
                rs[k].render(cams[i % len(cams)], scs[k], out=outs[k], sync=False)
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                rs[k].sync()
    run(range(10)); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(range(100)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={K}: {100 / dt:.1f} frames/s  ({dt * 10:.3f} ms/frame)", flush=True)
    del rs, scs
