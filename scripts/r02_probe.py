#!/usr/bin/env python3
"""Round-2 probe (GPU box): what the per-chunk bounds + Z-order buy on one GPU, and what a rank of a tile-row-sharded
frame pays — per-stage times alone, the pipelined rate of the sweep, and the 8-rank replay (every rank's band rendered
on this one GPU, pipelined, as a rank does in rows mode) for even and cost-balanced bands.
    python scripts/r02_probe.py [full|bands|all]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
import numpy as np, torch
from sage_gs import Renderer, scenes
from sage_gs.dist import row_partition, balanced_partition, timed_row_cost, ShardedRenderer

what = sys.argv[1] if len(sys.argv) > 1 else "all"
W, H = (3840, 2160) if "4k" in sys.argv else (1920, 1080)
dev = torch.device("cuda", 0)
sc = scenes.cached_room(3_000_000, seed=2)
cams = scenes.room_cameras(sc, W, H, n_positions=4, n_yaw=64, seed=2)
poses = [(i * 77) % 256 for i in range(10, 110)]
STAGES = ("preprocess", "count", "emit", "render")
ring = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(4)]


def make(morton):
    ev = lambda k: int(os.environ[k]) if k in os.environ else None       # (the probe's own knobs; the library reads no environment)
    r = Renderer(dev, record_capacity=(192 << 20) if W > 1920 else (96 << 20), morton=morton, lanes=ev("SGS_LANES"), group=ev("SGS_GROUP"),
                 group_lanes=ev("SGS_GROUP_LANES"))
    return r, r.upload(scenes.to_gaussians(sc, dev))


def alone(r, gs, chunk_cull=True, rows=None, n=24):
    acc = {s: 0.0 for s in STAGES}; tot = 0.0; nv = d = df = 0
    for p in poses[:n]:
        kw = {} if rows is None else {"tile_rows": rows}
        r.render(cams[p], gs, out=ring[0], timing=True, chunk_cull=chunk_cull, **kw)
        st = r.last_stats
        for s in STAGES:
            acc[s] += st["ms"][s]
        tot += st["ms_total"]; nv += st["n_visible"]; d += st["d_total"]
        r.render(cams[p], gs, out=ring[0], chunk_cull=chunk_cull, stats=True, **kw)        # (D_f is counted on request only)
        df += r.last_stats["d_fetched"]
    return {s: round(1e3 * acc[s] / n, 1) for s in STAGES}, round(1e3 * tot / n, 1), nv // n, d // n, df // n


def rate(r, gs, chunk_cull=True, rows=None, n=100):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            kw = {} if rows is None else {"tile_rows": rows}
            r.render(cams[poses[i % len(poses)]], gs, out=ring[i % 4], sync=False, pipelined=True, chunk_cull=chunk_cull, **kw)
        r.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


if what == "quick":
    r, gs = make(True)
    a = alone(r, gs, True)
    print(f"{os.environ.get('SAGE_GS_LIB', 'default lib')}: alone us {a[0]} total {a[1]}  N_v={a[2]} D={a[3]} D_f={a[4]} | pipelined {rate(r, gs, True):.4f} ms/frame", flush=True)
    a = alone(r, gs, True)
    print(f"   again: alone us {a[0]} total {a[1]} | pipelined {rate(r, gs, True):.4f} ms/frame", flush=True)
    gs.free(); r.close()

if what in ("full", "all"):
    for morton in (True, False):
        r, gs = make(morton)
        for cull in (True, False):
            a = alone(r, gs, cull)
            print(f"morton={int(morton)} chunk_cull={int(cull)}: alone us {a[0]} total {a[1]}  N_v={a[2]} D={a[3]} D_f={a[4]} | pipelined {rate(r, gs, cull):.4f} ms/frame", flush=True)
        if morton:
            r.render(cams[poses[0]], gs, out=ring[0])
            sk = r.debug_buffer(4, np.uint8)
            print(f"   chunks skipped by bounds at pose {poses[0]}: {sk.mean():.3f}")
        gs.free(); r.close()

def rate_batch(r, gs, rows, per_call=8, n=64):
    """ms per frame of a band when the sweep goes through Renderer.render_batch (frame groups), `per_call` frames per call
    — what a rank of ShardedRenderer.render_batch does"""
    cl = [cams[poses[i % len(poses)]] for i in range(n)]
    buf = torch.zeros((per_call, (rows[1] - rows[0]) * 16, W, 3), dtype=torch.float32, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c0 in range(0, n, per_call):
            r.render_batch(cl[c0:c0 + per_call], gs, out_bands=buf, tile_rows=rows)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


if what in ("bands", "all"):
    r, gs = make(True)
    gy = (H + 15) // 16
    gx = (W + 15) // 16
    for world in (8, 4, 2):
        even = row_partition(gy, world)
        # per-row records over the sweep sample -> balanced bands (what ShardedRenderer(balance=True) converges to)
        r.row_records(gy, reset=True)
        for p in poses[:24]:
            r.render(cams[p], gs, out=ring[0])
        rec = r.row_records(gy, reset=True) / 24.0
        for name, bands in (("even", even),) + tuple((f"balanced(tile_cost={tc})", balanced_partition(rec + tc * gx, world, 4 * -(-gy // world))) for tc in (48.0,)):
            per = np.array([rate(r, gs, True, b, n=60) for b in bands])
            print(f"world {world} {name}: one frame per launch set: slowest {per.max():.4f} mean {per.mean():.4f} ms/frame/rank  rows {[b - a for a, b in bands]}  {np.round(per, 3).tolist()}", flush=True)
            for pc in (8, 16, 32):
                per = np.array([rate_batch(r, gs, b, pc) for b in bands])
                print(f"world {world} {name}: render_batch x{pc}: slowest {per.max():.4f} mean {per.mean():.4f}  {np.round(per, 3).tolist()}", flush=True)
        if world == 8:
            for k, (r0, r1) in enumerate(even):
                a = alone(r, gs, True, (r0, r1), n=12)
                print(f"   rank {k} rows {(r0, r1)} alone us {a[0]} total {a[1]} N_v={a[2]} D={a[3]}")
    full = rate(r, gs)
    print(f"full frame pipelined: {full:.4f} ms/frame")

if what in ("iterate", "all"):
    # what ShardedRenderer(balance=True) does over the batches of a sweep: bands re-cut from the MEASURED band times
    r, gs = make(True)
    gy = (H + 15) // 16
    r.row_records(gy, reset=True)
    for p in poses[:24]:
        r.render(cams[p], gs, out=ring[0])
    rec = r.row_records(gy, reset=True) / 24.0
    full = rate(r, gs)
    for world in (8, 4, 2):
        bands, cost = row_partition(gy, world), None
        for it in range(6):
            per = np.array([rate_batch(r, gs, b, 32) if b[1] > b[0] else 0.0 for b in bands])
            print(f"world {world} iteration {it}: slowest {per.max():.4f} mean {per.mean():.4f} (x{full / per.max():.2f} of {full:.4f})  rows {[b - a for a, b in bands]}  {np.round(per, 3).tolist()}", flush=True)
            cost = timed_row_cost(bands, per, rec, cost)
            bands = balanced_partition(cost, world, 4 * -(-gy // world))

if what == "batchfull":
    # full frames through Renderer.render_batch (frame groups: SGS_GROUP frames per launch set on SGS_GROUP_LANES streams)
    # against the per-frame pipelined path
    r, gs = make(True)
    gy = (H + 15) // 16
    print(f"SGS_GROUP={os.environ.get('SGS_GROUP')} SGS_GROUP_LANES={os.environ.get('SGS_GROUP_LANES')}: per-frame pipelined {rate(r, gs):.4f} | "
          f"render_batch x32 {rate_batch(r, gs, (0, gy), 32, 96):.4f} x20 {rate_batch(r, gs, (0, gy), 20, 100):.4f} ms/frame", flush=True)

if what == "bands8":
    # the converged 8-rank bands of the sweep, every rank's band through render_batch x32 (env: SGS_GROUP, SGS_GROUP_LANES)
    r, gs = make(True)
    rows = [18, 8, 4, 4, 4, 4, 11, 15]
    bands, a = [], 0
    for n in rows:
        bands.append((a, a + n)); a += n
    per = np.array([rate_batch(r, gs, b, 32) for b in bands])
    print(f"SGS_GROUP={os.environ.get('SGS_GROUP')} SGS_GROUP_LANES={os.environ.get('SGS_GROUP_LANES')} SGS_LANES={os.environ.get('SGS_LANES')}: slowest {per.max():.4f} mean {per.mean():.4f}  {np.round(per, 3).tolist()}", flush=True)
