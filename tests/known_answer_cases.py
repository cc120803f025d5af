"""Closed-form known answers (SURVEY.md §8c iii) run against the KERNELS — no oracle in between.

The same analytic cases that pin the oracle (tests/test_oracle_known_answers.py), restated for a driver of the
product: `upload(*scene)`, `render(cam, cfg, rows, full_sort=, loose_cull=) -> (image, stats)`,
`render_aux(cam) -> (image, aux[H,W,2])`, `intermediates()`.  tests/test_emu_parity.py runs them on the CPU (the
product's kernel source under the wave64 emulator), tests/test_gpu_parity.py on the MI355X through the C ABI.
The composite is fp32 on the device, so pixel tolerances are a few fp32 ulps of O(1) values.
"""
import math

import numpy as np

import oracle_np as onp          # Camera / Config containers and the SH constants only — nothing is rendered with it

C0, C1 = onp.SH_C0, onp.SH_C1
TOL = 2e-5


def _scene(means, scales, opac, dc, quats=None, sh=None, deg=0):
    n = len(means)
    means = np.asarray(means, np.float32).reshape(n, 3)
    scales = np.asarray(scales, np.float32).reshape(n, 3)
    quats = np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)) if quats is None else np.asarray(quats, np.float32)
    opac = np.asarray(opac, np.float32).reshape(n)
    if sh is None:
        sh = np.asarray(dc, np.float32).reshape(n, 1, 3)
    return means, scales, quats, opac, np.asarray(sh, np.float32), deg


def _cam(w=64, h=64, f=64.0, cx=None, cy=None, view=None):
    # cx = 32.5 puts the optical axis exactly on pixel 32 (pixel centres are integers)
    return onp.Camera(w, h, f, f, 32.5 if cx is None else cx, 32.5 if cy is None else cy,
                      np.eye(4, dtype=np.float32) if view is None else np.asarray(view, np.float32))


def _render(drv, scene, cam, cfg=None, rows=(0, -1), **kw):
    drv.upload(*scene)
    img, st = drv.render(cam, cfg, rows, **kw)
    return img.astype(np.float64), st


def case_isotropic_profile(drv):
    s, z, o, dc, f = 0.1, 4.0, 0.8, 0.6, 64.0
    scene = _scene([[0, 0, z]], [[s, s, s]], [o], [[dc, dc / 2, -dc]])
    img, st = _render(drv, scene, _cam(f=f))
    var = (f * s / z) ** 2 + 0.3
    col = np.maximum(0.0, 0.5 + C0 * np.array([dc, dc / 2, -dc]))
    yy, xx = np.mgrid[0:64, 0:64]
    r2 = (xx - 32.0) ** 2 + (yy - 32.0) ** 2
    raw = o * np.exp(-r2 / (2 * var))
    alpha = np.where(raw >= 1 / 255.0, np.minimum(0.99, raw), 0.0)
    rad = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    x0, x1 = (32 - rad) // 16, (32 + rad + 15) // 16
    inrect = (xx // 16 >= x0) & (xx // 16 < x1) & (yy // 16 >= x0) & (yy // 16 < x1)
    expect = np.where(inrect, alpha, 0.0)[..., None] * col
    on_cut = np.abs(raw * 255.0 - 1.0) < 1e-3                    # an fp32 evaluation may take either branch there
    assert np.abs(img - expect)[~on_cut].max() < TOL
    assert st["n_visible"] == 1
    _, st_ref = _render(drv, scene, _cam(f=f), loose_cull=True)   # reference binning: D = the 3-sigma rect's area
    _, st_16 = _render(drv, scene, _cam(f=f), fine=False)          # (production binning on the reference's 16x16-pixel tiles; `img` above: the
    assert st_ref["d_total"] == (x1 - x0) ** 2 and st_16["d_total"] <= st_ref["d_total"]      #  default — fine tiles for a frame this small)


def case_front_to_back(drv):
    sc = _scene([[0, 0, 5.0], [0, 0, 3.0]], [[0.2] * 3, [0.15] * 3], [0.6, 0.5], [[1.0, 0, 0], [0, 1.0, 0]])
    img, _ = _render(drv, sc, _cam())
    v1, v2 = (64 * 0.15 / 3.0) ** 2 + 0.3, (64 * 0.2 / 5.0) ** 2 + 0.3     # nearer one first
    c1 = np.maximum(0, 0.5 + C0 * np.array([0, 1.0, 0])); c2 = np.maximum(0, 0.5 + C0 * np.array([1.0, 0, 0]))
    assert np.allclose(img[32, 32], 0.5 * c1 + 0.5 * 0.6 * c2, atol=TOL)
    b1, b2 = 0.5 * math.exp(-9.0 / (2 * v1)), 0.6 * math.exp(-9.0 / (2 * v2))
    assert np.allclose(img[32, 35], b1 * c1 + (1 - b1) * b2 * c2, atol=TOL)


def case_alpha_clamp_and_cutoff(drv):
    img, _ = _render(drv, _scene([[0, 0, 2.0]], [[0.05] * 3], [1.0], [[2.0, 2.0, 2.0]]), _cam())
    col = 0.5 + C0 * 2.0
    assert abs(img[32, 32, 0] - 0.99 * col) < TOL                            # alpha clamped to 0.99
    var = (64 * 0.05 / 2.0) ** 2 + 0.3
    yy, xx = np.mgrid[0:64, 0:64]
    alpha = np.exp(-((xx - 32.0) ** 2 + (yy - 32.0) ** 2) / (2 * var))
    assert (img[alpha < 0.999 / 255.0] == 0).all()                           # below 1/255: contributes nothing
    assert (img[(alpha > 1.001 / 255.0) & (alpha < 0.5)][:, 0] > 0).all()


def case_transmittance_termination(drv):
    # five coincident splats of opacity 0.95: T = .05, .0025, 1.25e-4, then 6.25e-6 < 1e-4 stops the pixel
    n = 5
    means = [[0, 0, 2.0 + 0.1 * i] for i in range(n)]
    dc = [[1.0, 1.0, 1.0]] * 3 + [[-1.0, 3.0, 0.0]] * 2                      # the last two must stay unseen
    scene = _scene(means, [[0.5] * 3] * n, [0.95] * n, dc)
    img, st = _render(drv, scene, _cam())
    col = 0.5 + C0 * 1.0
    assert np.allclose(img[32, 32], col * (0.95 + 0.05 * 0.95 + 0.0025 * 0.95), atol=TOL)
    drv.upload(*scene)
    _, aux = drv.render_aux(_cam())
    assert abs((1.0 - float(aux[32, 32, 1])) - 1.25e-4) < 2e-7               # coverage = 1 - T at the stop
    z = np.array([2.0, 2.1, 2.2])
    w = np.array([0.95, 0.05 * 0.95, 0.0025 * 0.95])
    assert abs(float(aux[32, 32, 0]) - float((w * z).sum())) < 1e-5          # expected depth over the three blended


def case_sh_signs(drv):
    sh = np.zeros((1, 4, 3), np.float32)
    sh[0, 0] = [0.3, 0.3, 0.3]; sh[0, 1] = [0.2, 0, 0]; sh[0, 2] = [0, 0.2, 0]; sh[0, 3] = [0, 0, 0.2]

    def centre(view, mean, cfg=None):
        img, _ = _render(drv, _scene([mean], [[0.2] * 3], [0.5], None, sh=sh, deg=1), _cam(view=view), cfg)
        return img[32, 32] / 0.5
    base = 0.5 + C0 * 0.3
    assert np.allclose(centre(np.eye(4), [0, 0, 3.0]), [base, base + C1 * 0.2, base], atol=2 * TOL)
    vx = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float32)
    assert np.allclose(centre(vx, [3.0, 0, 0]), [base, base, base - C1 * 0.2], atol=2 * TOL)
    vy = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    assert np.allclose(centre(vy, [0, 3.0, 0]), [base - C1 * 0.2, base, base], atol=2 * TOL)
    vz = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)
    assert np.allclose(centre(vz, [0, 0, -3.0]), [base, base - C1 * 0.2, base], atol=2 * TOL)
    assert np.allclose(centre(np.eye(4), [0, 0, 3.0], onp.Config(sh_degree=0)), [base] * 3, atol=2 * TOL)


def case_culling(drv):
    def nvis(mean, cfg=None):
        _, st = _render(drv, _scene([mean], [[0.05] * 3], [0.9], [[1, 1, 1]]), _cam(), cfg, loose_cull=True)
        return st["n_visible"], st["d_total"]
    assert nvis([0, 0, 0.2]) == (0, 0)                    # tz <= 0.2 is culled
    assert nvis([0, 0, 0.2001])[0] == 1
    assert nvis([0, 0, -1.0]) == (0, 0)                   # behind the camera
    assert nvis([50.0, 0, 2.0]) == (0, 0)                 # in front but far off-screen: empty tile rect
    assert nvis([0, 0, 10.5], onp.Config(far=10.0))[0] == 0


def case_rect_and_duplicates(drv):
    s, z, f = 0.3, 3.0, 64.0
    cam = _cam(w=128, h=96, f=f, cx=70.5, cy=40.5)        # mean2D = (70, 40)
    _, st = _render(drv, _scene([[0, 0, z]], [[s] * 3], [0.9], [[1, 1, 1]]), cam, full_sort=True, loose_cull=True)
    var = (f * s / z) ** 2 + 0.3
    rad = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    x0, x1 = max(0, (70 - rad) // 16), min(8, (70 + rad + 15) // 16)
    y0, y1 = max(0, (40 - rad) // 16), min(6, (40 + rad + 15) // 16)
    off, ids, slot_ids, splats = drv.intermediates()
    assert (splats[0, 10] & 0xffff, splats[0, 10] >> 16, splats[0, 11] & 0xffff, splats[0, 11] >> 16) == (x0, y0, x1, y1)
    assert st["d_total"] == (x1 - x0) * (y1 - y0)
    expect = np.zeros((6, 8), int); expect[y0:y1, x0:x1] = 1
    assert (np.diff(off).reshape(6, 8) == expect).all() and (ids == 0).all()


def case_depth_ties_on_index(drv):
    a = 0.5
    c_red = np.maximum(0, 0.5 + C0 * np.array([2.0, -2.0, -2.0])); c_blue = np.maximum(0, 0.5 + C0 * np.array([-2.0, -2.0, 2.0]))
    img, _ = _render(drv, _scene([[0, 0, 3.0], [0, 0, 3.0]], [[0.3] * 3] * 2, [a, a], [[2.0, -2, -2], [-2.0, -2, 2.0]]), _cam())
    assert np.allclose(img[32, 32], a * c_red + (1 - a) * a * c_blue, atol=TOL)       # index 0 in front
    img, _ = _render(drv, _scene([[0, 0, 3.0], [0, 0, 3.0]], [[0.3] * 3] * 2, [a, a], [[-2.0, -2, 2.0], [2.0, -2, -2]]), _cam())
    assert np.allclose(img[32, 32], a * c_blue + (1 - a) * a * c_red, atol=TOL)


def case_background(drv):
    cfg = onp.Config(background=(0.2, 0.4, 0.6))
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32),
             np.zeros((0,), np.float32), np.zeros((0, 1, 3), np.float32), 0)
    img, st = _render(drv, empty, _cam(), cfg)
    assert np.allclose(img, [0.2, 0.4, 0.6], atol=1e-7) and st["d_total"] == 0
    img, _ = _render(drv, _scene([[0, 0, 3.0]], [[0.3] * 3], [0.5], [[0, 0, 0]]), _cam(), cfg)
    assert np.allclose(img[32, 32], 0.5 * 0.5 + 0.5 * np.array([0.2, 0.4, 0.6]), atol=TOL)


ALL = (case_isotropic_profile, case_front_to_back, case_alpha_clamp_and_cutoff, case_transmittance_termination,
       case_sh_signs, case_culling, case_rect_and_duplicates, case_depth_ties_on_index, case_background)
