#!/usr/bin/env python3
"""Round-5 probe (GPU box): stage times ALONE and the pipelined rate of the bench's sweep, for the library SAGE_GS_LIB names
(A/B of variants in one box visit), on the fp32-uploaded scene and/or the scene uploaded from the compressed payload.
    python scripts/r05_probe.py [fp32] [packed] [lowres] [n=24] [libs=a.so,b.so,...]     (libs: several variants in ONE process, names under build/variants/)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sage_gs import Renderer, scenes, _capi

args = sys.argv[1:]
modes = [a for a in args if a in ("fp32", "packed")] or ["fp32"]
N = next((int(a[2:]) for a in args if a.startswith("n=")), 24)
dev = torch.device("cuda", 0)
sc = scenes.make_trained_like(3_000_000, seed=2) if "trained" in args else scenes.cached_room(3_000_000, seed=2)
STAGES = ("preprocess", "count", "emit", "render")
poses = [(i * 77) % 256 for i in range(5, 105)]
TRAJ = next((int(a[5:] or 1) for a in args if a.startswith("traj")), 0)
if TRAJ:                   # headings TRAJ x 5.6 degrees apart from one position: what a trajectory's frames look like (traj, traj=4, ...)
    poses = [(i * TRAJ) % 64 for i in range(5, 105)]
tag = os.path.basename(os.environ.get("SAGE_GS_LIB", "default"))
SCENE_TAG = ("trained" if "trained" in args else "room") + (f" traj={TRAJ}" if TRAJ else "")
LIBS = next((a[5:].split(",") for a in args if a.startswith("libs=")), [None])


def alone(r, gs, cams, buf, n):
    acc = {s: [] for s in STAGES}; tot = []; nv = d = df = 0
    for p in poses[:4]:
        r.render(cams[p], gs, out=buf)
    for p in poses[:n]:
        r.render(cams[p], gs, out=buf, timing=True)
        st = r.last_stats
        for s in STAGES:
            acc[s].append(st["ms"][s])
        tot.append(st["ms_total"]); nv += st["n_visible"]; d += st["d_total"]
    for p in poses[:n]:
        r.render(cams[p], gs, out=buf, stats=True)
        df += r.last_stats["d_fetched"]
    return {s: round(1e3 * float(np.mean(acc[s])), 1) for s in STAGES}, round(1e3 * float(np.mean(tot)), 1), nv // n, d // n, df // n


def rate(r, gs, cams, ring, n=100):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            r.render(cams[poses[i % len(poses)]], gs, out=ring[i % 4], sync=False, pipelined=True)
        r.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


def rate_batch(r, gs, cams, n=20):
    """ms per frame of ONE render_batch call of n frames (the bench's headline path: frame groups of four on two streams)"""
    cl = [cams[poses[i % len(poses)]] for i in range(n)]
    out = torch.zeros((n, cams[0].height, cams[0].width, 3), dtype=torch.float32, device=dev)
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_batch(cl, gs, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


def latency(r, gs, cams, buf, n=32):
    lat = []
    for p in poses[:4]:
        r.render(cams[p], gs, out=buf)
    for p in poses[:n]:
        t0 = time.perf_counter(); r.render(cams[p], gs, out=buf); lat.append(1e3 * (time.perf_counter() - t0))
    return np.percentile(lat, 50), np.percentile(lat, 90)


def group_overlap(cams, n=20):
    """the library's group_overlap (sgs_api.hip) on 8192 Gaussians of the scene: sum over the frames of a group of four of the Gaussians in
    view / those in view of any — averaged over the groups of the first n poses"""
    rng = np.random.default_rng(0)
    m = sc.means[rng.choice(sc.means.shape[0], 8192, replace=False)].astype(np.float64)
    m = np.c_[m, np.ones(len(m))] @ (np.asarray(sc.model_to_world, np.float64).T if sc.model_to_world is not None else np.eye(4))
    r = []
    for g0 in range(0, n, 4):
        seen = []
        for p in poses[g0:g0 + 4]:
            c = cams[p]; t = m @ np.asarray(c.view, np.float64).T
            px, py = c.fx * t[:, 0] / t[:, 2] + c.cx, c.fy * t[:, 1] / t[:, 2] + c.cy
            seen.append((t[:, 2] > 0.2) & (px >= -0.1 * c.width) & (px < 1.1 * c.width) & (py >= -0.1 * c.height) & (py < 1.1 * c.height))
        r.append(sum(s_.sum() for s_ in seen) / max(1, np.logical_or.reduce(seen).sum()))
    return float(np.mean(r))


g_dev = scenes.to_gaussians(sc, dev)
for libname in LIBS:
    if libname is None:
        r = Renderer(dev, record_capacity=96 << 20)
    else:
        tag = libname
        r = Renderer(dev, record_capacity=96 << 20, lib=_capi.Lib(os.path.join(ROOT, "build", "variants", libname if libname.endswith(".so") else libname + ".so")))
    GROUP = next((int(a[6:]) for a in args if a.startswith("group=")), 0); GL = next((int(a[3:]) for a in args if a.startswith("gl=")), 2)
    if GROUP:
        r.set_tuning(group=GROUP, group_lanes=GL)
        tag = f"{tag} group={GROUP}"
    for mode in modes:
        if mode == "packed":
            from bench import quantise_on_gpu
            dv = quantise_on_gpu(g_dev)
            gs = r.upload_compressed(dv[0], dv[1], dv[2], sc.sh_degree, model_to_world=sc.model_to_world, sh_decode="bin_centre")
            del dv
        else:
            gs = r.upload(g_dev)
        for (w, h) in ([(1920, 1080)] + ([(640, 480), (320, 240)] if "lowres" in args else [])):
            cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=2)
            ring = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in range(4)]
            a = alone(r, gs, cams, ring[0], N)
            line = f"[{tag}] {SCENE_TAG} {mode} {w}x{h}: alone us {a[0]} total {a[1]}  N_v={a[2]} D={a[3]} D_f={a[4]}"
            if (w, h) == (1920, 1080):
                line += f" | pipelined {rate(r, gs, cams, ring):.4f} ms/frame | render_batch x20 {rate_batch(r, gs, cams, 20):.4f} x100 {rate_batch(r, gs, cams, 100):.4f} (group overlap {group_overlap(cams):.2f})"
            else:
                p50, p90 = latency(r, gs, cams, ring[0])
                line += f" | latency p50 {p50:.3f} p90 {p90:.3f} ms"
            print(line, flush=True)
        gs.free()
    r.close()
