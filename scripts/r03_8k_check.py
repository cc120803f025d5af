import os, sys
sys.path.insert(0, "/root/repo/sage-3d_official_amd")
import numpy as np, torch
from sage_gs import Renderer, scenes
sc = scenes.cached_room(3_000_000, seed=2)
r = Renderer("cuda:0", record_capacity=512 << 20); gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
for (W, H) in ((7680, 4320), (8176, 4608)):
    cams = scenes.room_cameras(sc, W, H, 4, 64, seed=2)
    for i in (5, 82):
        c = cams[i]
        try:
            img = r.render(c, gs, timing=True, stats=True).clone(); st = dict(r.last_stats)
            ref = r.render(c, gs, loose_cull=True, full_sort=True)
            union = torch.zeros_like(img)
            gy = (H + 15) // 16
            for a, b in ((0, gy // 3), (gy // 3, gy // 2 + 1), (gy // 2 + 1, gy)):
                r.render(c, gs, out=union, tile_rows=(a, b))
            print(f"{W}x{H} pose {i}: N_v={st['n_visible']} D={st['d_total']} D_f={st['d_fetched']} ms={ {k: round(v,3) for k,v in st['ms'].items()} } identical to reference binning: {bool((img==ref).all())}, to the union of bands: {bool((union==img).all())}", flush=True)
        except Exception as e:
            print(f"{W}x{H} pose {i}: {type(e).__name__}: {str(e)[:300]}", flush=True)
