"""Development probe: cost of the depth/coverage output (sgs_render_rgbd) against plain RGB."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sage-3d_official_amd"))
import torch
from sage_gs import Renderer, scenes
sc = scenes.make_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, 1920, 1080, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
out = torch.zeros((1080, 1920, 3), device="cuda:0"); aux = torch.zeros((1080, 1920, 2), device="cuda:0")
for use_aux in (False, True):
    tot = {}
    for i in range(10, 60):
        if use_aux:
            r.render(cams[i], gs, out=out, out_aux=aux, timing=True)
        else:
            r.render(cams[i], gs, out=out, timing=True)
        for k, v in r.last_stats["ms"].items():
            tot[k] = tot.get(k, 0.0) + v
    print("aux" if use_aux else "rgb", {k: round(1e3 * v / 50) for k, v in tot.items()})
