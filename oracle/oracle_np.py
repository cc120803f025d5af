"""fp64 NumPy oracle for the 3DGS scene-render hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference (Galery23/SAGE-3D_Official) contains no rasterizer: its frames come
from NVIDIA Isaac Sim (`Code/benchmark/environment_evaluation/simple_env.py:1368-1380`,
`Code/data_pipeline/training_data_construction/generate_images.py:425-428`), a closed binary that
is absent here, and it ships no golden frames or tests.  This file therefore
restates the six stages that `BASELINE.json:north_star` names, with the constants tabulated in
SURVEY.md §8(a) rows S1-S6 (the public EWA / 3D-Gaussian-splatting equations, written from the
maths, not from any source file).  It is pinned only by closed-form known-answer tests
(`tests/test_oracle_known_answers.py`) and by agreement with the independent C restatement
(`oracle/sgs_oracle.c`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product (`sage-3d_official_amd/`) never does.

Precision contract (shared with the HIP path so integer results can be compared bit-exactly):
  * inputs are fp32 arrays;
  * all per-Gaussian geometry (S2, S3) is evaluated in fp64 from those fp32 values;
  * the per-Gaussian results handed to the composite are ROUNDED TO fp32 (xy, conic, opacity, rgb,
    depth) — that is the storage format of the path — and the composite itself runs in fp64 here
    (fp32 on the GPU; tolerance |d| < 1e-3 per pixel, BASELINE.json);
  * ordering is by (tile, fp32 depth bits, Gaussian index) — ties on depth break on the index.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

TILE = 16

# S1 — real spherical-harmonic basis constants (bands 0..3), the standard 3DGS normalisation.
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


@dataclass
class Config:
    """SURVEY.md §5 'Config / flags' row: the constants of S2-S6."""
    near: float = 0.2            # S2: cull if tz <= near
    far: float = 1.0e30          # optional far cull (reference camera clip (0.1, 50), simple_env.py:899)
    dilation: float = 0.3        # S2: + 0.3 px^2 on the 2-D covariance diagonal
    clamp: float = 1.3           # S2: clamp t.xy/t.z to +-1.3 tan(fov/2)
    alpha_min: float = 1.0 / 255.0   # S6
    alpha_max: float = 0.99          # S6
    t_min: float = 1.0e-4            # S6
    background: tuple = (0.0, 0.0, 0.0)   # A7: black, linear, un-tonemapped
    sh_degree: int = -1          # -1: use the scene's degree

    def f32(self):
        """The ABI carries these constants as fp32 (sgs_config); evaluate with the rounded values."""
        r = lambda v: float(np.float32(v))
        return Config(r(self.near), r(self.far), r(self.dilation), r(self.clamp), r(self.alpha_min),
                      r(self.alpha_max), r(self.t_min), tuple(r(b) for b in self.background), self.sh_degree)


@dataclass
class Camera:
    """Pinhole camera, +Z forward, +X right, +Y down."""
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    view: np.ndarray = field(default_factory=lambda: np.eye(4))   # model -> camera, row-major 4x4

    @property
    def grid(self):
        return ((self.width + TILE - 1) // TILE, (self.height + TILE - 1) // TILE)


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def camera_position(view):
    """Camera centre in model space: -R^T t."""
    v = np.asarray(view, dtype=np.float64)
    return -v[:3, :3].T @ v[:3, 3]


def eval_sh(sh, dirs, degree):
    """S1. sh: [N,K,3] (fp32 values, evaluated in fp64), dirs: [N,3] unit, -> rgb [N,3] (pre-clamp +0.5)."""
    sh = np.asarray(sh, dtype=np.float64)
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if degree >= 1:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
               + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if degree >= 3:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
               + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
               + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
               + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res + 0.5


def preprocess(means, scales, quats, opacities, sh, sh_degree, cam: Camera, cfg: Config,
               tile_row_begin=0, tile_row_end=None):
    """S1-S3 for every Gaussian.  Returns a dict of per-Gaussian arrays (length N).

    `visible[i]` is False when the Gaussian is culled (tz <= near, tz > far, det <= 0, or its tile
    rect — clipped to tile rows [tile_row_begin, tile_row_end) — is empty).
    """
    cfg = cfg.f32()
    means = np.asarray(means, np.float64)
    scales = np.asarray(scales, np.float64)
    quats = np.asarray(quats, np.float64)
    N = means.shape[0]
    gx, gy = cam.grid
    if tile_row_end is None:
        tile_row_end = gy
    V = np.asarray(cam.view, np.float32).astype(np.float64)      # the ABI carries the view as fp32
    R, tvec = V[:3, :3], V[:3, 3]

    # world(model) -> view
    t = means @ R.T + tvec
    tz = t[:, 2]
    depth32 = tz.astype(np.float32)
    in_front = (tz > cfg.near) & (tz <= cfg.far)
    tzs = np.where(in_front, tz, 1.0)       # keep the maths finite for culled ones

    # S2: 3-D covariance from (linear) scale + normalised quaternion (w,x,y,z)
    qn = quats / np.linalg.norm(quats, axis=1, keepdims=True)
    w, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    Rq = np.empty((N, 3, 3))
    Rq[:, 0, 0] = 1 - 2 * (y * y + z * z); Rq[:, 0, 1] = 2 * (x * y - w * z); Rq[:, 0, 2] = 2 * (x * z + w * y)
    Rq[:, 1, 0] = 2 * (x * y + w * z); Rq[:, 1, 1] = 1 - 2 * (x * x + z * z); Rq[:, 1, 2] = 2 * (y * z - w * x)
    Rq[:, 2, 0] = 2 * (x * z - w * y); Rq[:, 2, 1] = 2 * (y * z + w * x); Rq[:, 2, 2] = 1 - 2 * (x * x + y * y)
    M = Rq * scales[:, None, :]                 # R S
    Sigma = M @ np.transpose(M, (0, 2, 1))      # R S S^T R^T

    # S2: EWA projection
    fx, fy, cx, cy = (float(np.float32(v)) for v in (cam.fx, cam.fy, cam.cx, cam.cy))
    tan_x = 0.5 * cam.width / fx
    tan_y = 0.5 * cam.height / fy
    limx, limy = cfg.clamp * tan_x, cfg.clamp * tan_y
    txc = np.clip(t[:, 0] / tzs, -limx, limx) * tzs
    tyc = np.clip(t[:, 1] / tzs, -limy, limy) * tzs
    J = np.zeros((N, 2, 3))
    J[:, 0, 0] = fx / tzs
    J[:, 0, 2] = -fx * txc / (tzs * tzs)
    J[:, 1, 1] = fy / tzs
    J[:, 1, 2] = -fy * tyc / (tzs * tzs)
    T = J @ R                                   # 2x3
    cov = T @ Sigma @ np.transpose(T, (0, 2, 1))
    a = cov[:, 0, 0] + cfg.dilation
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + cfg.dilation
    det = a * c - b * b
    ok = in_front & (det > 0.0)
    dets = np.where(ok, det, 1.0)
    conic = np.stack([c / dets, -b / dets, a / dets], axis=1)

    # S3: 3-sigma radius and tile rect.  Pixel i has centre coordinate i (mean2D = f*x/z + c - 0.5).
    mid = 0.5 * (a + c)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(lam))
    px = fx * t[:, 0] / tzs + cx - 0.5
    py = fy * t[:, 1] / tzs + cy - 0.5

    def _tile(v, lo, hi):
        return np.clip(np.floor(v / TILE), lo, hi).astype(np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        x0 = _tile(px - radius, 0, gx); x1 = _tile(px + radius + (TILE - 1), 0, gx)
        y0 = _tile(py - radius, tile_row_begin, tile_row_end)
        y1 = _tile(py + radius + (TILE - 1), tile_row_begin, tile_row_end)
    tiles = np.where(ok, (x1 - x0) * (y1 - y0), 0)
    visible = ok & (tiles > 0)
    tiles = np.where(visible, tiles, 0)

    # S1: colour
    deg = sh_degree if cfg.sh_degree < 0 else min(cfg.sh_degree, sh_degree)
    cam_pos = camera_position(V)
    d = means - cam_pos
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    rgb = np.maximum(eval_sh(sh, d, deg), 0.0)

    return dict(
        visible=visible, depth=depth32, tiles=tiles.astype(np.int64),
        rect=np.stack([x0, y0, x1, y1], axis=1).astype(np.int64),
        radius=radius,
        xy=_f32(np.stack([px, py], axis=1)), conic=_f32(conic),
        opacity=_f32(opacities), rgb=_f32(rgb),
        xy64=np.stack([px, py], axis=1), conic64=conic, rgb64=rgb,
    )


def bin_and_sort(pre, cam: Camera, tile_row_begin=0, tile_row_end=None):
    """S4+S5. Returns (tile_offsets[T+1], ids[D]) — ids of each tile's queue in (depth bits, index) order.

    Tiles are numbered row-major over the FULL grid; tiles outside the row range have empty queues.
    """
    gx, gy = cam.grid
    vis = np.nonzero(pre["visible"])[0]
    rect = pre["rect"][vis]
    cnt = pre["tiles"][vis]
    D = int(cnt.sum())
    gid = np.repeat(vis, cnt)
    # local index of each record inside its Gaussian's rect
    starts = np.cumsum(cnt) - cnt
    local = np.arange(D) - np.repeat(starts, cnt)
    w = np.repeat(rect[:, 2] - rect[:, 0], cnt)
    tx = np.repeat(rect[:, 0], cnt) + local % np.maximum(w, 1)
    ty = np.repeat(rect[:, 1], cnt) + local // np.maximum(w, 1)
    tile = ty * gx + tx
    key = pre["depth"][gid].view(np.uint32).astype(np.uint64)
    order = np.lexsort((gid, key, tile))        # tile major, then depth bits, then index
    ids = gid[order]
    counts = np.bincount(tile, minlength=gx * gy)
    offsets = np.zeros(gx * gy + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets, ids


def composite(pre, offsets, ids, cam: Camera, cfg: Config, tile_row_begin=0, tile_row_end=None):
    """S6 front-to-back alpha composite, fp64, tile by tile.

    Returns dict(image[H,W,3], final_T[H,W], depth_image[H,W] (sum T alpha z),
    n_contrib[H,W] (index+1 of the last blended record),
    consumed[T] (records any pixel of the tile examined = the D_f term), margin[H,W]).
    `margin` is the smallest relative distance of any examined (pixel, Gaussian) pair to one of S6's two
    thresholds — |alpha/alpha_min - 1| (the cut-off: the blend jumps by alpha_min*T*c there) and, for pairs past the
    cut-off, |T(1-alpha)/t_min - 1| (the stop: the pair's alpha*T*c is blended or not, up to ~1e-2*c when alpha is
    near alpha_max).  fp32 and fp64 evaluations may legitimately decide differently inside that margin;
    `pixel_variants` enumerates what each admissible decision gives.
    """
    cfg = cfg.f32()
    gx, gy = cam.grid
    if tile_row_end is None:
        tile_row_end = gy
    H, W = cam.height, cam.width
    img = np.zeros((H, W, 3)); finT = np.ones((H, W)); ncon = np.zeros((H, W), np.int64)
    zimg = np.zeros((H, W)); zs = pre["depth"].astype(np.float64)
    margin = np.full((H, W), np.inf)
    consumed = np.zeros(gx * gy, np.int64)
    xy = pre["xy"].astype(np.float64); con = pre["conic"].astype(np.float64)
    op = pre["opacity"].astype(np.float64); rgb = pre["rgb"].astype(np.float64)
    bg = np.asarray(cfg.background, np.float64)
    for ty in range(tile_row_begin, tile_row_end):
        for tx in range(gx):
            t = ty * gx + tx
            ys = np.arange(ty * TILE, min((ty + 1) * TILE, H))
            xs = np.arange(tx * TILE, min((tx + 1) * TILE, W))
            PX, PY = np.meshgrid(xs.astype(np.float64), ys.astype(np.float64))
            T = np.ones_like(PX); C = np.zeros(PX.shape + (3,)); Z = np.zeros_like(PX)
            done = np.zeros(PX.shape, bool); nc = np.zeros(PX.shape, np.int64)
            mg = np.full(PX.shape, np.inf)
            q = ids[offsets[t]:offsets[t + 1]]
            used = 0
            for k, g in enumerate(q):
                if done.all():
                    break
                used = k + 1
                dx = xy[g, 0] - PX; dy = xy[g, 1] - PY
                power = -0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) - con[g, 1] * dx * dy
                alpha = np.minimum(cfg.alpha_max, op[g] * np.exp(np.minimum(power, 0.0)))
                live = ~done & (power <= 0.0)
                mg = np.where(live, np.minimum(mg, np.abs(alpha / cfg.alpha_min - 1.0)), mg)
                hit = live & (alpha >= cfg.alpha_min)
                testT = T * (1.0 - alpha)
                mg = np.where(hit, np.minimum(mg, np.abs(testT / cfg.t_min - 1.0)), mg)
                stop = hit & (testT < cfg.t_min)
                blend = hit & ~stop
                C = np.where(blend[..., None], C + (alpha * T)[..., None] * rgb[g], C)
                Z = np.where(blend, Z + alpha * T * zs[g], Z)
                T = np.where(blend, testT, T)
                nc = np.where(blend, k + 1, nc)
                done |= stop
            consumed[t] = used
            sl = (slice(ys[0], ys[-1] + 1), slice(xs[0], xs[-1] + 1))
            img[sl] = C + T[..., None] * bg
            finT[sl] = T; ncon[sl] = nc; margin[sl] = mg; zimg[sl] = Z
    return dict(image=img, final_T=finT, n_contrib=ncon, consumed=consumed, margin=margin, depth_image=zimg)


def pixel_variants(pre, offsets, ids, cam: Camera, cfg: Config, px: int, py: int, rel_margin=1.0e-4, cap=12):
    """Two-sided evaluation of one pixel: every colour an evaluation may produce that decides the pairs lying within
    `rel_margin` (relative) of the alpha cut-off or of the stop threshold either way (depth-first over the branch
    points, at most `cap` on a path).  Returns float32 [L,3]; L == 1 when no decision of the pixel is that close.
    The checker accepts a pixel iff it matches ONE of these within the parity tolerance."""
    cfg = cfg.f32()
    gx, _ = cam.grid
    t = (py // TILE) * gx + px // TILE
    q = ids[offsets[t]:offsets[t + 1]]
    xy = pre["xy"].astype(np.float64); con = pre["conic"].astype(np.float64)
    op = pre["opacity"].astype(np.float64); rgb = pre["rgb"].astype(np.float64)
    bg = np.asarray(cfg.background, np.float64)
    leaves = []

    def walk(k, T, C, depth):
        C = C.copy()
        while k < len(q):
            g = q[k]; k += 1
            dx = xy[g, 0] - px; dy = xy[g, 1] - py
            power = -0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) - con[g, 1] * dx * dy
            if power > 0.0:
                continue
            alpha = min(cfg.alpha_max, op[g] * math.exp(power))
            hit = alpha >= cfg.alpha_min
            if abs(alpha / cfg.alpha_min - 1.0) < rel_margin and depth < cap:
                depth += 1
                walk(k, T, C, depth)                     # this pair skipped ...
                hit = True                               # ... or taken
            if not hit:
                continue
            testT = T * (1.0 - alpha)
            stop = testT < cfg.t_min
            if abs(testT / cfg.t_min - 1.0) < rel_margin and depth < cap:
                depth += 1
                leaves.append(C + T * bg)                # the pixel ends here ...
                stop = False                             # ... or this pair is blended
            if stop:
                break
            C = C + alpha * T * rgb[g]
            T = testT
        leaves.append(C + T * bg)

    walk(0, 1.0, np.zeros(3), 0)
    return np.asarray(leaves, np.float32).reshape(-1, 3)


def render(means, scales, quats, opacities, sh, sh_degree, cam: Camera, cfg: Config = None,
           tile_row_begin=0, tile_row_end=None):
    """All six stages; returns (image fp64 [H,W,3], aux dict with every intermediate)."""
    cfg = cfg or Config()
    pre = preprocess(means, scales, quats, opacities, sh, sh_degree, cam, cfg,
                     tile_row_begin, tile_row_end)
    offsets, ids = bin_and_sort(pre, cam, tile_row_begin, tile_row_end)
    out = composite(pre, offsets, ids, cam, cfg, tile_row_begin, tile_row_end)
    aux = dict(pre=pre, offsets=offsets, ids=ids, **out,
               n_visible=int(pre["visible"].sum()), D=int(offsets[-1]),
               D_f=int(out["consumed"].sum()))
    return out["image"], aux


# ---------------------------------------------------------------------------------------------
# Synthetic inputs of BASELINE.md §2 (the generators the tests and the bench share with the product's
# own `sage_gs.scenes`; kept here too so the oracle stays self-contained for fixture generation).
def config1_scene(n=10_000, seed=0):
    """10k random Gaussians, SH deg 0, camera-space box, 256x256."""
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(2, 8, n)], 1)
    scales = np.exp(rng.uniform(math.log(0.02), math.log(0.2), (n, 3)))
    quats = rng.normal(size=(n, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, n)))
    sh = (0.5 * rng.normal(size=(n, 1, 3)))
    cam = Camera(256, 256, 128.0, 128.0, 128.0, 128.0, np.eye(4))
    return (_f32(means), _f32(scales), _f32(quats), _f32(opac), _f32(sh), 0), cam
