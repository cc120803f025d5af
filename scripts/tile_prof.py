#!/usr/bin/env python3
"""Per-tile phase breakdown of k_tile_render (profiling build build/lib/libsage_gs_prof.so).  Run on the GPU box."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
os.environ["SAGE_GS_LIB"] = os.path.join(ROOT, "build", "lib", "libsage_gs_prof.so")
import numpy as np, torch
from sage_gs import Renderer, scenes
sc = scenes.make_trained_like(3_000_000, seed=2) if os.environ.get("SCENE") == "trained" else scenes.cached_room(int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, seed=2)
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
cams = scenes.room_cameras(sc, W, H, 4, 64, seed=2)
r = Renderer("cuda:0", record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, "cuda:0"))
out = {}
TOT = {'ev': 0.0, 'val': 0.0, 'use': 0.0, 'stg': 0.0, 'hit': 0.0}
for ci in [int(v) for v in os.environ.get('POSES', '5,20,70,140,200').split(',')]:
    r.render(cams[ci], gs, stats=True); d_f = r.last_stats["d_fetched"]      # (D_f is counted on request only)
    for _ in range(3):
        r.render(cams[ci], gs, timing=True, deep_cull=os.environ.get('DEEP', '1') != '0')      # the kernels a sweep runs
    st = dict(r.last_stats); st["d_fetched"] = d_f
    p = r.debug_buffer(100, np.uint64).reshape(-1, 32).astype(np.float64)
    n, part, sort, blend, ng, nb, tot, t0, ev, emp, val, use, stg, hit, rt0, rt1, prank, pbar1, pstage, pjob, prec = p.T[:21]
    praw = r.debug_buffer(100, np.uint64).reshape(-1, 32)
    nref, nfill = (praw[:, 21] & np.uint64(0xffff)).astype(np.float64), (praw[:, 22] & np.uint64(0xffff)).astype(np.float64)
    dtry, dok = ((praw[:, 21] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64), ((praw[:, 21] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64)
    dsurv = (praw[:, 22] >> np.uint64(16)).astype(np.float64)
    pu = r.debug_buffer(100, np.uint64).reshape(-1, 32)[:, 23]
    dead, few = (pu & np.uint64(0xffffffff)).astype(np.float64), (pu >> np.uint64(32)).astype(np.float64)
    clk = 1e-3 * tot.sum() / max(1e-9, 1.0)   # cycles
    for k_, v_ in (('ev', ev), ('val', val), ('use', use), ('stg', stg), ('hit', hit)): TOT[k_] += float(v_.sum())
    print(f"cam {ci}: render {st['ms']['render']*1e3:.0f} us  D={st['d_total']} D_f={st['d_fetched']} | tile-cycles sum: part {part.sum()/1e6:.1f}M sort {sort.sum()/1e6:.1f}M blend {blend.sum()/1e6:.1f}M total {tot.sum()/1e6:.1f}M | "
          f"groups/tile {ng.mean():.2f} batches/tile {nb.mean():.2f} | max tile total {tot.max()/1e3:.0f}k cyc (n={int(n[tot.argmax()])}) | span {(t0+tot).max()-t0.min():.0f} cyc")
    print(f"     blend evaluations (wave x splat): {ev.sum()/1e6:.2f}M = {ev.sum()/max(1,st['d_fetched']):.2f} per consumed record; no pixel inside the cut-off: "
          f"{100*emp.sum()/max(1,ev.sum()):.1f} %; lanes inside the cut-off {100*val.sum()/max(1,64*ev.sum()):.1f} %, of them on live pixels {100*use.sum()/max(1,val.sum()):.1f} %")
    print(f"     evaluations with NO live pixel inside the cut-off: {100*dead.sum()/max(1,ev.sum()):.1f} %, with one or two: {100*few.sum()/max(1,ev.sum()):.1f} %")
    print(f"     staged splats (single-batch groups): {stg.sum()/1e6:.2f}M, reaching at least one quadrant: {100*hit.sum()/max(1,stg.sum()):.1f} %  (D = {st['d_total']/1e6:.2f}M records, D_f = {st['d_fetched']/1e6:.2f}M)")
    # occupancy over the kernel's span: how many tiles (workgroups) are in flight
    evs = np.concatenate([np.stack([rt0, np.ones_like(rt0)], 1), np.stack([rt1, -np.ones_like(rt0)], 1)])
    evs = evs[np.argsort(evs[:, 0])]
    act = np.cumsum(evs[:, 1]); dt = np.diff(evs[:, 0], append=evs[-1, 0]); span = evs[-1, 0] - evs[0, 0]
    peak = act.max()
    print(f"     workgroups in flight: peak {int(peak)}, time-average {float((act * dt).sum() / span):.0f}; share of the span with < 50 % of the peak: "
          f"{100 * dt[act < 0.5 * peak].sum() / span:.1f} %, < 90 %: {100 * dt[act < 0.9 * peak].sum() / span:.1f} %  (span {span / 100:.0f} us)")
    q = np.quantile(rt1 - evs[0, 0], [0.5, 0.9, 0.99, 1.0]) / 100
    print(f"     tiles finished by: 50 % {q[0]:.0f} us, 90 % {q[1]:.0f} us, 99 % {q[2]:.0f} us, all {q[3]:.0f} us; tiles started after 50 % of the span: {100 * (rt0 - evs[0, 0] > 0.5 * span).mean():.1f} %")
    print(f"     single-batch path, cycle sums: rank (records resident -> ranked) {prank.sum()/1e6:.0f}M, barrier 1 {pbar1.sum()/1e6:.0f}M, stage (gather wait + extents + quadrant test) {pstage.sum()/1e6:.0f}M, barrier 2 {sort.sum()/1e6:.0f}M; partition {part.sum()/1e6:.0f}M; blend {blend.sum()/1e6:.0f}M; total {tot.sum()/1e6:.0f}M")
    print(f"     start of a tile, mean cycles: entry -> job arrived {pjob.mean():.0f}, -> records arrived {prec.mean():.0f}, -> partitioned {part.mean():.0f}  (p90: {np.quantile(pjob,0.9):.0f}, {np.quantile(prec,0.9):.0f}, {np.quantile(part,0.9):.0f})")
    print(f"     deep-tile culls: attempted {int(dtry.sum())} (tiles {int((dtry > 0).sum())}), taken {int(dok.sum())}, survivors per taken window {dsurv.sum() / max(1.0, dok.sum()):.0f}; cull cycles {pbar1.sum()/1e6:.1f}M")
    h = praw[:, 24:29].astype(np.float64); lcyc = praw[:, 31].astype(np.float64)
    print(f"     listed splats (wave x batch lists) by live pixels of the wave: 1-4 {h[:,0].sum()/1e3:.0f}k  5-8 {h[:,1].sum()/1e3:.0f}k  9-16 {h[:,2].sum()/1e3:.0f}k  17-32 {h[:,3].sum()/1e3:.0f}k  33-64 {h[:,4].sum()/1e3:.0f}k | "
          f"wave-cycles in the trip loops {lcyc.sum()/1e6:.1f}M")
    order = np.argsort(-tot)[:8]
    for o in order:
        print(f"     tile {o}: n={int(n[o])} part {part[o]/1e3:.0f}k sort {sort[o]/1e3:.0f}k blend {blend[o]/1e3:.0f}k groups {int(ng[o])} batches {int(nb[o])} | rank {prank[o]/1e3:.0f}k stage {pstage[o]/1e3:.0f}k total {tot[o]/1e3:.0f}k evals {int(ev[o])} (no live pixel {int(dead[o])}, 1-2 {int(few[o])}) refinements {int(nref[o])} window fills {int(nfill[o])} deep {int(dok[o])}/{int(dtry[o])} surv {int(dsurv[o])} cull {pbar1[o]/1e3:.0f}k | listed by live px {[int(v) for v in h[o]]} trips {lcyc[o]/1e3:.0f}k cyc")
    # by size class
    for lo, hi in ((0, 256), (256, 1024), (1024, 4096), (4096, 10**9)):
        m = (n > lo) & (n <= hi)
        if m.any():
            print(f"     n in ({lo},{hi}]: {m.sum()} tiles, mean cyc job {pjob[m].mean():.0f} rec {prec[m].mean():.0f} part {part[m].mean():.0f} rank {prank[m].mean():.0f} stage {pstage[m].mean():.0f} sort {sort[m].mean():.0f} blend {blend[m].mean():.0f} total {tot[m].mean():.0f}")

# over all the poses: lanes of the blend's (wave, splat) evaluations inside the cut-off, and of those on a live pixel
print("TOTAL " + json.dumps({"poses": os.environ.get('POSES', '5,20,70,140,200'), "evaluations": TOT['ev'],
                             "lanes_inside_cutoff": TOT['val'] / max(1.0, 64 * TOT['ev']), "of_those_on_live_pixels": TOT['use'] / max(1.0, TOT['val']),
                             "useful_lane_frac": TOT['use'] / max(1.0, 64 * TOT['ev']), "staged_reaching_a_quadrant": TOT['hit'] / max(1.0, TOT['stg'])}))
