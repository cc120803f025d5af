"""Synthetic scenes and camera paths of BASELINE.md §2 (InteriorGS itself is not available offline).

`config1` is the 10k-Gaussian plumbing case; `make_room` builds indoor scenes matched to the
"~500k room" / "~3M scene" configurations on Gaussian count, SH degree and indoor statistics
(70 % flat surfels on floor/ceiling/walls, 30 % furniture blobs, mostly opaque).  Scenes are built
in the reference's WORLD frame (Z up, metres — Data/template.usda:105-108) and stored in MODEL space
so that `model_to_world` = Rx(-90 deg) (template.usda:120) brings them back, exactly as a SAGE-3D
asset is laid out.  Everything is NumPy on the host; outputs are float32 arrays.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SH_C0 = 0.28209479177387814

# the reference's lens: focalLength 8.0 on the USD default 20.955 mm aperture (simple_env.py:905)
REF_FOCAL_OVER_APERTURE = 8.0 / 20.955
EYE_HEIGHT = 1.2                      # generate_images.py:45 / simple_env.py:1204


def rx(deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


MODEL_TO_WORLD = rx(-90.0)            # Data/template.usda:120  rotateXYZ = (-90, 0, 0)


@dataclass
class SceneArrays:
    means: np.ndarray
    scales: np.ndarray
    quats: np.ndarray
    opacities: np.ndarray
    sh: np.ndarray
    sh_degree: int
    model_to_world: np.ndarray
    extent: tuple = (0.0, 0.0, 0.0)   # world-space size (x, y, z)
    rooms: tuple = ()                 # (x0, y0, x1, y1) of every room, world frame

    def as_tuple(self):
        return self.means, self.scales, self.quats, self.opacities, self.sh, self.sh_degree


def config1(n=10_000, seed=0):
    """10k random Gaussians in a camera-space box, SH degree 0."""
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(2, 8, n)], 1)
    scales = np.exp(rng.uniform(math.log(0.02), math.log(0.2), (n, 3)))
    quats = rng.normal(size=(n, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, n)))
    sh = 0.5 * rng.normal(size=(n, 1, 3))
    f32 = lambda a: np.asarray(a, np.float32)
    return SceneArrays(f32(means), f32(scales), f32(quats), f32(opac), f32(sh), 0, np.eye(4))


def _mat_to_quat_wxyz(R):
    """Batched rotation matrix -> (w,x,y,z), numerically safe branch per row."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    tr = m00 + m11 + m22
    c0 = tr > 0
    c1 = ~c0 & (m00 >= m11) & (m00 >= m22)
    c2 = ~c0 & ~c1 & (m11 >= m22)
    c3 = ~c0 & ~c1 & ~c2
    for cond, idx in ((c0, 0), (c1, 1), (c2, 2), (c3, 3)):
        if not cond.any():
            continue
        r = R[cond]
        if idx == 0:
            s = np.sqrt(r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2] + 1.0) * 2
            q[cond] = np.stack([0.25 * s, (r[:, 2, 1] - r[:, 1, 2]) / s, (r[:, 0, 2] - r[:, 2, 0]) / s,
                                (r[:, 1, 0] - r[:, 0, 1]) / s], 1)
        elif idx == 1:
            s = np.sqrt(1.0 + r[:, 0, 0] - r[:, 1, 1] - r[:, 2, 2]) * 2
            q[cond] = np.stack([(r[:, 2, 1] - r[:, 1, 2]) / s, 0.25 * s, (r[:, 0, 1] + r[:, 1, 0]) / s,
                                (r[:, 0, 2] + r[:, 2, 0]) / s], 1)
        elif idx == 2:
            s = np.sqrt(1.0 + r[:, 1, 1] - r[:, 0, 0] - r[:, 2, 2]) * 2
            q[cond] = np.stack([(r[:, 0, 2] - r[:, 2, 0]) / s, (r[:, 0, 1] + r[:, 1, 0]) / s, 0.25 * s,
                                (r[:, 1, 2] + r[:, 2, 1]) / s], 1)
        else:
            s = np.sqrt(1.0 + r[:, 2, 2] - r[:, 0, 0] - r[:, 1, 1]) * 2
            q[cond] = np.stack([(r[:, 1, 0] - r[:, 0, 1]) / s, (r[:, 0, 2] + r[:, 2, 0]) / s,
                                (r[:, 1, 2] + r[:, 2, 1]) / s, 0.25 * s], 1)
    return q


def _room_grid(size_x, size_y, target_rooms):
    """Split the floor plan into a grid of rooms of roughly 30-40 m^2."""
    if target_rooms <= 1:
        return 1, 1
    nx = max(1, round(math.sqrt(target_rooms * size_x / size_y)))
    ny = max(1, round(target_rooms / nx))
    return nx, ny


def make_room(n, seed=1, size=None, height=2.8, sh_degree=3, n_rooms=None):
    """Indoor scene of `n` Gaussians.  Defaults follow BASELINE.md: n<=1M -> one 6x5x2.8 m room,
    larger -> a 20x15x2.8 m floor of ~8 rooms with interior walls and door openings."""
    rng = np.random.default_rng(seed)
    if size is None:
        size = (6.0, 5.0) if n <= 1_000_000 else (20.0, 15.0)
    if n_rooms is None:
        n_rooms = 1 if n <= 1_000_000 else 8
    sx, sy = size
    gx, gy = _room_grid(sx, sy, n_rooms)
    rw, rd = sx / gx, sy / gy
    rooms = tuple((i * rw, j * rd, (i + 1) * rw, (j + 1) * rd) for j in range(gy) for i in range(gx))

    # ---- planar surfaces: (origin, edge u, edge v, normal) rectangles in the world frame ----------
    rects = []
    def add(o, u, v, nrm):
        rects.append((np.array(o, float), np.array(u, float), np.array(v, float), np.array(nrm, float)))
    add((0, 0, 0), (sx, 0, 0), (0, sy, 0), (0, 0, 1))                     # floor
    add((0, 0, height), (sx, 0, 0), (0, sy, 0), (0, 0, -1))                # ceiling
    add((0, 0, 0), (sx, 0, 0), (0, 0, height), (0, 1, 0))                  # outer walls
    add((0, sy, 0), (sx, 0, 0), (0, 0, height), (0, -1, 0))
    add((0, 0, 0), (0, sy, 0), (0, 0, height), (1, 0, 0))
    add((sx, 0, 0), (0, sy, 0), (0, 0, height), (-1, 0, 0))
    door = 1.0
    for i in range(1, gx):                                                 # interior walls along y
        for j in range(gy):
            y0, y1 = j * rd, (j + 1) * rd
            c = rng.uniform(y0 + 0.8, y1 - 0.8 - door)
            add((i * rw, y0, 0), (0, c - y0, 0), (0, 0, height), (1, 0, 0))
            add((i * rw, c + door, 0), (0, y1 - c - door, 0), (0, 0, height), (1, 0, 0))
            add((i * rw, c, 2.0), (0, door, 0), (0, 0, height - 2.0), (1, 0, 0))   # lintel
    for j in range(1, gy):                                                 # interior walls along x
        for i in range(gx):
            x0, x1 = i * rw, (i + 1) * rw
            c = rng.uniform(x0 + 0.8, x1 - 0.8 - door)
            add((x0, j * rd, 0), (c - x0, 0, 0), (0, 0, height), (0, 1, 0))
            add((c + door, j * rd, 0), (x1 - c - door, 0, 0), (0, 0, height), (0, 1, 0))
            add((c, j * rd, 2.0), (door, 0, 0), (0, 0, height - 2.0), (0, 1, 0))
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v, _ in rects])

    n_surf = int(round(0.7 * n))
    n_furn = n - n_surf

    # ---- 70 %: flat surfels on the rectangles -----------------------------------------------------
    which = rng.choice(len(rects), size=n_surf, p=areas / areas.sum())
    O = np.stack([r[0] for r in rects])[which]; U = np.stack([r[1] for r in rects])[which]
    V = np.stack([r[2] for r in rects])[which]; Nn = np.stack([r[3] for r in rects])[which]
    a, b = rng.random(n_surf), rng.random(n_surf)
    pos_s = O + a[:, None] * U + b[:, None] * V + rng.normal(0.0, 0.003, n_surf)[:, None] * Nn
    s_in = 0.025 * np.exp(rng.normal(0.0, 0.5, (n_surf, 2)))
    scl_s = np.concatenate([s_in, 0.1 * s_in[:, :1]], 1)
    t1 = U / np.linalg.norm(U, axis=1, keepdims=True)
    t2 = np.cross(Nn, t1)
    ang = rng.uniform(0, 2 * math.pi, n_surf)
    e1 = np.cos(ang)[:, None] * t1 + np.sin(ang)[:, None] * t2
    e2 = np.cross(Nn, e1)
    R_s = np.stack([e1, e2, Nn], axis=2)                                   # columns = principal axes
    alb_rect = rng.uniform(0.15, 0.9, (len(rects), 3))
    alb_s = alb_rect[which]

    # ---- 30 %: furniture blobs (axis-aligned boxes standing on the floor) --------------------------
    boxes = []
    for (x0, y0, x1, y1) in rooms:
        for _ in range(6):
            w, d, h = rng.uniform(0.4, 2.0), rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.9)
            bx = rng.uniform(x0 + 0.2, max(x0 + 0.21, x1 - 0.2 - w)); by = rng.uniform(y0 + 0.2, max(y0 + 0.21, y1 - 0.2 - d))
            boxes.append((bx, by, 0.0, w, d, h))
    boxes = np.array(boxes)
    vol = boxes[:, 3] * boxes[:, 4] * boxes[:, 5]
    wb = rng.choice(len(boxes), size=n_furn, p=vol / vol.sum())
    B = boxes[wb]
    u3 = rng.random((n_furn, 3))
    # half of the blobs hug a face of the box, half fill its volume
    face = rng.integers(0, 3, n_furn); side = rng.integers(0, 2, n_furn); hug = rng.random(n_furn) < 0.5
    u3[np.arange(n_furn), face] = np.where(hug, side + rng.normal(0, 0.01, n_furn), u3[np.arange(n_furn), face])
    pos_f = B[:, :3] + u3 * B[:, 3:6]
    scl_f = np.repeat(0.015 * np.exp(rng.normal(0.0, 0.5, (n_furn, 1))), 3, 1)
    qf = rng.normal(size=(n_furn, 4)); qf /= np.linalg.norm(qf, axis=1, keepdims=True)
    alb_box = rng.uniform(0.1, 0.95, (len(boxes), 3))
    alb_f = alb_box[wb]

    # ---- world -> model (inverse of MODEL_TO_WORLD), quaternions, colours --------------------------
    Minv = MODEL_TO_WORLD[:3, :3].T
    means = np.concatenate([pos_s, pos_f]) @ Minv.T
    q_s = _mat_to_quat_wxyz(np.einsum("ij,njk->nik", Minv, R_s))
    # furniture orientations are random; rotating a uniformly random rotation leaves it uniform
    quats = np.concatenate([q_s, qf])
    scales = np.concatenate([scl_s, scl_f])
    opac = 1.0 / (1.0 + np.exp(-rng.normal(2.0, 2.0, n)))
    k = (sh_degree + 1) ** 2
    sh = np.empty((n, k, 3), np.float32)
    alb = np.concatenate([alb_s, alb_f])
    sh[:, 0, :] = ((alb - 0.5) / SH_C0 + rng.normal(0.0, 0.1, (n, 3))).astype(np.float32)
    if k > 1:
        sh[:, 1:, :] = 0.05 * rng.standard_normal((n, k - 1, 3), dtype=np.float32)
    # interleave surfels and furniture the way a trained scene is stored: spatially coherent runs
    order = np.argsort((np.floor(means[:, 0] / 0.5) * 4096 + np.floor(means[:, 2] / 0.5)).astype(np.int64),
                       kind="stable")
    f32 = lambda a: np.ascontiguousarray(a[order], np.float32)
    return SceneArrays(f32(means), f32(scales), f32(quats), f32(opac), f32(sh), sh_degree,
                       MODEL_TO_WORLD.copy(), (sx, sy, height), rooms)


def make_trained_like(n, seed=1, **kw):
    """The statistics of a TRAINED 3DGS scene laid over `make_room`'s geometry (what InteriorGS assets look like, as far
    as published 3DGS checkpoints go — none is available offline): log-normal scales with a heavy tail (sigma 0.9 instead
    of 0.5, 1 % of the splats ten times larger still), strong anisotropy (one tangent axis stretched by exp|N(0, 0.8)|),
    40 % of the splats nearly transparent (opacity sigmoid(N(-3.5, 1)): < 0.1), 8 % floaters — large, faint blobs anywhere in
    the volume — and NO spatial order in the arrays.  Against `make_room` this multiplies the records queued per tile (D)
    and the records a pixel consumes before it saturates (D_f): the scene to tune the binning and the tail tiles on."""
    sc = make_room(n, seed=seed, **kw)
    rng = np.random.default_rng(seed + 7919)
    scales = sc.scales.astype(np.float64) * np.exp(rng.normal(0.0, 0.75, (n, 1)))           # 0.5 (+) 0.75 -> sigma 0.9
    stretch = np.exp(np.abs(rng.normal(0.0, 0.8, n)))
    scales[:, 0] *= stretch                                                               # one in-plane axis (surfels: a tangent)
    giant = rng.random(n) < 0.01
    scales[giant] *= 10.0
    opac = np.where(rng.random(n) < 0.4, 1.0 / (1.0 + np.exp(-rng.normal(-3.5, 1.0, n))), sc.opacities.astype(np.float64))
    means = sc.means.astype(np.float64).copy()
    fl = rng.random(n) < 0.08                                                             # floaters: anywhere in the volume
    nf = int(fl.sum())
    world = np.stack([rng.uniform(0, sc.extent[0], nf), rng.uniform(0, sc.extent[1], nf), rng.uniform(0, sc.extent[2], nf)], 1)
    means[fl] = world @ MODEL_TO_WORLD[:3, :3]                                            # world -> model (M^-1 = M^T, row vectors)
    scales[fl] = 0.06 * np.exp(rng.normal(0.0, 0.7, (nf, 3)))
    opac[fl] = 1.0 / (1.0 + np.exp(-rng.normal(-2.5, 1.0, nf)))
    order = rng.permutation(n)
    f32 = lambda a: np.ascontiguousarray(a[order], np.float32)
    return SceneArrays(f32(means), f32(scales), f32(sc.quats), f32(opac), f32(sc.sh), sc.sh_degree,
                       sc.model_to_world, sc.extent, sc.rooms)


def scene_from_arrays(arrays, model_to_world=None):
    """SceneArrays for a loaded scene (sage_gs.ply: means, scales, quats, opacities, sh, degree): the world-space bounds
    (2nd-98th percentile of the means: trained scenes carry floaters far outside) stand in for the room list that
    `room_cameras` / `sweep_cameras` place their eye points in."""
    means, scales, quats, opac, sh, deg = arrays
    m2w = MODEL_TO_WORLD.copy() if model_to_world is None else np.asarray(model_to_world, float)
    w = np.asarray(means, np.float64) @ m2w[:3, :3].T + m2w[:3, 3]
    lo, hi = np.percentile(w, 2, axis=0), np.percentile(w, 98, axis=0)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return SceneArrays(f32(means), f32(scales), f32(quats), f32(opac), f32(sh), int(deg), m2w,
                       tuple(float(v) for v in (hi - lo)), ((float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1])),))


def cached_room(n, seed=1, cache_dir=None, **kw):
    """`make_room`, kept on disk between processes (the generator takes ~25 s for 3 M Gaussians; benchmarks and profiling
    runs that start many processes on one box load the arrays instead).  cache_dir=None -> $SGS_SCENE_CACHE or
    /tmp/sage_gs_scenes; any failure to read or write the cache falls back to generating."""
    import os
    if kw:
        return make_room(n, seed=seed, **kw)
    d = cache_dir or os.environ.get("SGS_SCENE_CACHE") or "/tmp/sage_gs_scenes"
    path = os.path.join(d, f"room_{int(n)}_{int(seed)}_v1.npz")
    try:
        z = np.load(path)
        return SceneArrays(z["means"], z["scales"], z["quats"], z["opacities"], z["sh"], int(z["sh_degree"]),
                           z["model_to_world"], tuple(float(v) for v in z["extent"]), tuple(tuple(float(v) for v in r) for r in z["rooms"]))
    except Exception:
        pass
    sc = make_room(n, seed=seed)
    try:
        os.makedirs(d, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, means=sc.means, scales=sc.scales, quats=sc.quats, opacities=sc.opacities, sh=sc.sh, sh_degree=sc.sh_degree,
                 model_to_world=sc.model_to_world, extent=np.asarray(sc.extent), rooms=np.asarray(sc.rooms))
        os.replace(tmp, path)
    except Exception:
        pass
    return sc


def view_from_yaw(position, yaw, pitch=0.0):
    """world->camera matrix of a camera at `position` (world, Z up) looking along heading `yaw`
    (radians about +Z, 0 = +X) with `pitch` up; camera axes +X right, +Y down, +Z forward."""
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    fwd = np.array([cy * cp, sy * cp, sp])
    right = np.array([sy, -cy, 0.0])
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])
    V = np.eye(4)
    V[:3, :3] = R
    V[:3, 3] = -R @ np.asarray(position, float)
    return V


def reference_intrinsics(width, height):
    """(fx, fy, cx, cy) of the reference's camera: 8 mm lens on a 20.955 mm aperture, square pixels,
    principal point at the image centre (simple_env.py:840-905)."""
    f = width * REF_FOCAL_OVER_APERTURE
    return f, f, width / 2.0, height / 2.0


def room_cameras(scene: SceneArrays, width=1920, height=1080, n_positions=4, n_yaw=64, seed=0):
    """BASELINE.md pose list: n_yaw headings at each of n_positions eye-height points (seeded)."""
    from .renderer import Camera
    rng = np.random.default_rng(seed + 1000)
    fx, fy, cx, cy = reference_intrinsics(width, height)
    rooms = scene.rooms or ((0.0, 0.0, scene.extent[0], scene.extent[1]),)
    cams = []
    for p in range(n_positions):
        x0, y0, x1, y1 = rooms[(p * max(1, len(rooms) // n_positions)) % len(rooms)]
        mx, my = min(1.0, 0.25 * (x1 - x0)), min(1.0, 0.25 * (y1 - y0))      # a metre from the walls (less in a small scene)
        pos = (rng.uniform(x0 + mx, x1 - mx), rng.uniform(y0 + my, y1 - my), EYE_HEIGHT)
        for k in range(n_yaw):
            cams.append(Camera(width, height, fx, fy, cx, cy, view_from_yaw(pos, 2 * math.pi * k / n_yaw)))
    return cams


def sweep_cameras(scene: SceneArrays, width=3840, height=2160, n=360, position=None, seed=0):
    """BASELINE.md config 5 / SURVEY.md §8d: a 360-degree camera sweep — `n` cameras at ONE eye-height position, yaw
    steps of 360/n degrees (1 degree at n = 360), the reference's lens (fx = 1466 px at 3840).  This is the shape of the
    reference's batch caller (generate_images.py:408-436: one pose per waypoint, same scene, same camera).  `position`
    defaults to a seeded point inside the scene's first room."""
    from .renderer import Camera
    fx, fy, cx, cy = reference_intrinsics(width, height)
    if position is None:
        rng = np.random.default_rng(seed + 2000)
        x0, y0, x1, y1 = (scene.rooms or ((0.0, 0.0, scene.extent[0], scene.extent[1]),))[0]
        mx, my = min(1.0, 0.25 * (x1 - x0)), min(1.0, 0.25 * (y1 - y0))
        position = (rng.uniform(x0 + mx, x1 - mx), rng.uniform(y0 + my, y1 - my), EYE_HEIGHT)
    position = (float(position[0]), float(position[1]), float(position[2]) if len(position) > 2 else EYE_HEIGHT)
    return [Camera(width, height, fx, fy, cx, cy, view_from_yaw(position, 2 * math.pi * k / n)) for k in range(n)]


def to_gaussians(scene: SceneArrays, device):
    """SceneArrays -> renderer.Gaussians with tensors on `device`."""
    import torch
    from .renderer import Gaussians
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return Gaussians(t(scene.means), t(scene.scales), t(scene.quats), t(scene.opacities), t(scene.sh),
                     scene.sh_degree, scene.model_to_world)
