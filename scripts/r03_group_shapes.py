#!/usr/bin/env python3
"""CPU estimate (oracle arithmetic) of what the composite's pixel-group shape costs: for sampled tiles of a 3 M-Gaussian
1080p frame, the number of (group, splat) evaluations a wave spends under different groupings of its 64 lanes —
one 8x8 quadrant with one splat list (today), two 8x4 halves / four 4x4 quarters with a list each (a trip serves all
sub-groups at once: trips = the longest of them), and lists that are re-cut against the still-live pixels every batch.
    python scripts/r03_group_shapes.py [pose ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_np as onp
from sage_gs import scenes

N = int(os.environ.get("N", 3_000_000)); W, H = 1920, 1080
sc = scenes.cached_room(N, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [int(a) for a in sys.argv[1:]] or [77 * 10 % 256]
cfg = onp.Config().f32()
rng = np.random.default_rng(0)
GROUP = 192

def run(pose, n_tiles=300):
    cam = cams[pose]
    view = (np.asarray(cam.view, np.float64) @ np.asarray(sc.model_to_world, np.float64)).astype(np.float32)
    ocam = onp.Camera(W, H, cam.fx, cam.fy, cam.cx, cam.cy, view)
    t0 = time.time()
    pre = onp.preprocess(sc.means, sc.scales, sc.quats, sc.opacities, sc.sh, sc.sh_degree, ocam, onp.Config())
    off, ids = onp.bin_and_sort(pre, ocam)
    print(f"pose {pose}: N_v={int(pre['visible'].sum())} D={len(ids)}  ({time.time()-t0:.0f} s)", flush=True)
    gx, gy = ocam.grid
    xy = pre["xy"].astype(np.float64); con = pre["conic"].astype(np.float64); op = pre["opacity"].astype(np.float64)
    tiles = rng.choice(gx * (gy - 1), n_tiles, replace=False)
    tot = dict(q8x8=0, h8x4=0, h8x4_sum=0, q4x4=0, q4x4_sum=0, s16x4=0, q8x8_live=0, h8x4_live=0, useful=0, consumed=0, staged=0)
    for t in tiles:
        ty, tx = divmod(int(t), gx)
        PX, PY = np.meshgrid(np.arange(tx * 16, tx * 16 + 16, dtype=np.float64), np.arange(ty * 16, ty * 16 + 16, dtype=np.float64))
        q = ids[off[t]:off[t + 1]]
        T = np.ones((16, 16)); done = np.zeros((16, 16), bool)
        valid_l = []; live_l = []
        for k, g in enumerate(q):
            if done.all(): break
            dx = xy[g, 0] - PX; dy = xy[g, 1] - PY
            power = -0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) - con[g, 1] * dx * dy
            alpha = np.minimum(cfg.alpha_max, op[g] * np.exp(np.minimum(power, 0.0)))
            valid = (power <= 0) & (alpha >= cfg.alpha_min)
            valid_l.append(valid); live_l.append(~done)
            hit = valid & ~done
            testT = T * (1 - alpha)
            stop = hit & (testT < cfg.t_min)
            T = np.where(hit & ~stop, testT, T); done |= stop
        if not valid_l: continue
        Vd = np.stack(valid_l); L = np.stack(live_l)            # [K,16,16]
        K = Vd.shape[0]
        tot["consumed"] += K; tot["staged"] += min(len(q), -(-K // GROUP) * GROUP)
        tot["useful"] += int((Vd & L).sum())

        def lists(h, w):
            """per group of h x w pixels: (splat touches the group) [K, G], group still has a live pixel before splat k [K, G]"""
            v = Vd.reshape(K, 16 // h, h, 16 // w, w).any(axis=(2, 4)).reshape(K, -1)
            l = L.reshape(K, 16 // h, h, 16 // w, w).any(axis=(2, 4)).reshape(K, -1)
            return v, l
        def cost(h, w, per_wave, live_cut=False):
            """wave-trips (in splats): groups are packed per_wave to a wave in row-major order of an 8x8 quadrant"""
            v, l = lists(h, w)
            if live_cut:     # lists re-cut per batch of GROUP against the pixels live at the batch start
                vl = np.zeros_like(v)
                for b0 in range(0, K, GROUP):
                    lv = L[b0]                                    # live at batch start
                    vb = (Vd[b0:b0 + GROUP] & lv[None]).reshape(-1, 16 // h, h, 16 // w, w).any(axis=(2, 4)).reshape(-1, v.shape[1])
                    vl[b0:b0 + GROUP] = vb
                v = vl
            act = v & l                                          # evaluated: in the list and the group not yet finished
            # map groups to waves: quadrant of the group's origin
            gh, gw = 16 // h, 16 // w
            gy_, gx_ = np.divmod(np.arange(gh * gw), gw)
            quad = (gy_ * h // 8) * 2 + (gx_ * w // 8)
            c = 0; s = 0
            for qd in range(4):
                m = act[:, quad == qd]
                if m.shape[1] == 0: continue
                # each sub-group walks ITS list; a trip serves all of the wave's sub-groups: per batch, trips = the longest list
                for b0 in range(0, K, GROUP):
                    cnts = m[b0:b0 + GROUP].sum(axis=0)
                    c += int(cnts.max()); s += int(cnts.sum())
            return c, s
        tot["q8x8"] += cost(8, 8, 1)[0]
        c, s = cost(4, 8, 2); tot["h8x4"] += c; tot["h8x4_sum"] += s
        c, s = cost(4, 4, 4); tot["q4x4"] += c; tot["q4x4_sum"] += s
        tot["q8x8_live"] += cost(8, 8, 1, True)[0]
        tot["h8x4_live"] += cost(4, 8, 2, True)[0]
        # 16x4 strips as the wave's group (one list per wave)
        v, l = lists(4, 16); act = v & l
        tot["s16x4"] += int(sum(act[b0:b0 + GROUP].sum() for b0 in range(0, K, GROUP)))
    base = tot["q8x8"]
    print(f"  {n_tiles} tiles: consumed/tile {tot['consumed']/n_tiles:.0f} staged/tile {tot['staged']/n_tiles:.0f}; wave evaluations per tile (8x8 quadrants): {base/n_tiles:.0f}; "
          f"useful lanes {tot['useful']/(base*64):.3f}")
    for k in ("h8x4", "q4x4", "s16x4", "q8x8_live", "h8x4_live"):
        print(f"    {k:10s}: {tot[k]/base:.3f} of the 8x8 wave-trips" + (f"   (perfectly balanced sub-lists: {tot[k+'_sum']/base/ (2 if k=='h8x4' else 4):.3f})" if k + "_sum" in tot else ""))

for p in poses:
    run(p)
