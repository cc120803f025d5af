"""Every way of issuing frames must give the same bits: synchronous, pipelined on the lanes, render_batch (frame groups),
bands into slabs — over random scenes, resolutions (buffers re-grown in between) and camera lists, with ordinary frames
mixed into the pipelined ones.  Used by tests/test_gpu_parity.py (a few iterations) and scripts/gpu_stress_paths.py (many)."""
import numpy as np


def stress_issue_paths(r, iters, seed, verbose=False):
    import torch
    from sage_gs import scenes
    rng = np.random.default_rng(seed)
    dev = r.device
    bad = 0
    for it in range(iters):
        n = int(rng.integers(5_000, 250_000))
        w, h = int(rng.integers(64, 1400)), int(rng.integers(48, 900))
        sc = scenes.make_room(n, seed=int(rng.integers(1 << 30)))
        cams = scenes.room_cameras(sc, w, h, n_positions=2, n_yaw=int(rng.integers(3, 9)), seed=int(rng.integers(1 << 30)))
        gs = r.upload(scenes.to_gaussians(sc, dev))
        ref = [r.render(c, gs).clone() for c in cams]
        # pipelined frames, an ordinary frame in the middle
        outs = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in cams]
        for i, c in enumerate(cams):
            r.render(c, gs, out=outs[i], sync=False, pipelined=True)
            if i == len(cams) // 2:
                mid = r.render(cams[0], gs).clone()
        r.sync()
        ok = all((a == b).all().item() for a, b in zip(outs, ref)) and (mid == ref[0]).all().item()
        # one call for the batch
        batch = r.render_batch(cams, gs)
        ok = ok and all((batch[i] == ref[i]).all().item() for i in range(len(cams)))
        # a band of tile rows into slabs
        gy = (h + 15) // 16
        r0 = int(rng.integers(0, gy)); r1 = int(rng.integers(r0 + 1, gy + 1))
        slabs = torch.zeros((len(cams), (r1 - r0) * 16, w, 3), dtype=torch.float32, device=dev)
        r.render_batch(cams, gs, out_bands=slabs, tile_rows=(r0, r1))
        y0, y1 = r0 * 16, min(r1 * 16, h)
        ok = ok and all((slabs[i, : y1 - y0] == ref[i][y0:y1]).all().item() for i in range(len(cams)))
        # the same band, one frame at a time and pipelined
        for i, c in enumerate(cams):
            slabs[i].zero_()
            r.render(c, gs, out_band=slabs[i], tile_rows=(r0, r1), sync=False, pipelined=True)
        r.sync()
        ok = ok and all((slabs[i, : y1 - y0] == ref[i][y0:y1]).all().item() for i in range(len(cams)))
        if verbose:
            print(f"iteration {it}: n={n} {w}x{h} {len(cams)} cameras rows [{r0},{r1}) -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
        gs.free()
    return bad
