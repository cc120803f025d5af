#!/usr/bin/env python3
"""get_rgba() breakdown (GPU box): one frame at a time at the reference's resolutions — render alone, render + pack + the pinned host buffer (view / numpy copy),
the pack kernel writing the pinned buffer directly, and GsCamera.set_world_pose + get_rgba()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
from sage_gs import Renderer, scenes
from sage_gs.adapter import GsCamera
from sage_gs import camera as cam_conv
dev = torch.device("cuda", 0)
sc = scenes.cached_room(3_000_000, seed=2)
r = Renderer(dev, record_capacity=96 << 20)
gs = r.upload(scenes.to_gaussians(sc, dev))
poses = [(i * 77) % 256 for i in range(5, 105)]
def pct(a): return f"p50 {np.percentile(a,50):.3f} p90 {np.percentile(a,90):.3f}"
for (w, h) in ((320, 240), (640, 480), (1024, 768)):
    cams = scenes.room_cameras(sc, w, h, n_positions=4, n_yaw=64, seed=2)
    buf = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    sel = poses[:48]
    for p in sel[:4]: r.render(cams[p], gs, out=buf)
    a = []
    for p in sel:
        t0 = time.perf_counter(); r.render(cams[p], gs, out=buf); a.append(1e3 * (time.perf_counter() - t0))
    print(f"{w}x{h} render sync            {pct(a)}")
    a = []
    for p in sel:
        t0 = time.perf_counter(); r.render_rgba8_host(cams[p], gs); a.append(1e3 * (time.perf_counter() - t0))
    print(f"{w}x{h} render_rgba8_host view {pct(a)}")
    a = []
    for p in sel:
        t0 = time.perf_counter(); x = r.render_rgba8_host(cams[p], gs).copy(); a.append(1e3 * (time.perf_counter() - t0))
    print(f"{w}x{h} ... + numpy copy       {pct(a)}")
    # pack straight into pinned host memory (the kernel writes over PCIe), one stream, one wait
    host = torch.empty((h, w, 4), dtype=torch.uint8, pin_memory=True); hn = host.numpy()
    a = []; ok = True
    for p in sel:
        t0 = time.perf_counter()
        rgb = r.render(cams[p], gs, out=buf, sync=False)
        r._lib.check(r._lib.sgs_pack_rgba8(r._ctx, rgb.data_ptr(), host.data_ptr(), w, h, r._stream()), r._ctx)
        r.sync(); torch.cuda.current_stream(dev).synchronize()
        a.append(1e3 * (time.perf_counter() - t0))
    ref = r.render_rgba8_host(cams[sel[-1]], gs)
    print(f"{w}x{h} pack -> pinned direct  {pct(a)}  equal={bool((hn == ref).all())}")
    a = []
    for p in sel:
        t0 = time.perf_counter()
        rgb = r.render(cams[p], gs, out=buf, sync=False)
        r._lib.check(r._lib.sgs_pack_rgba8(r._ctx, rgb.data_ptr(), host.data_ptr(), w, h, r._stream()), r._ctx)
        r.sync(); torch.cuda.current_stream(dev).synchronize(); x = hn.copy()
        a.append(1e3 * (time.perf_counter() - t0))
    print(f"{w}x{h} ... + numpy copy       {pct(a)}")
    t0 = time.perf_counter()
    for _ in range(50): x = hn.copy()
    print(f"{w}x{h} numpy copy alone {(time.perf_counter()-t0)/50*1e3:.3f} ms")
    gcam = GsCamera(r, gs, resolution=(w, h)); gcam.initialize()
    ip = [cam_conv.isaac_pose_from_view(cams[p].view) for p in sel]
    for pos_, q_ in ip[:4]:
        gcam.set_world_pose(pos_, q_); gcam.get_rgba()
    for cp in (True, False):
        a = []
        for pos_, q_ in ip:
            t0 = time.perf_counter(); gcam.set_world_pose(pos_, q_); x = gcam.get_rgba(copy=cp); a.append(1e3 * (time.perf_counter() - t0))
        print(f"{w}x{h} GsCamera.get_rgba(copy={cp}) {pct(a)}")
