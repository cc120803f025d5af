"""Parity cases, written once and run twice: under the wave64 emulator on CPU (test_emu_parity.py,
kernel LOGIC) and on a real MI355X through the C ABI (test_gpu_parity.py, the parity tests proper).

A `drv` offers: upload(means, scales, quats, opac, sh, deg); render(cam, cfg=None, rows=(0,-1))
-> (image ndarray, stats dict); intermediates() -> (tile_offsets, sorted ids, slot ids, splat words).
Cameras/configs are oracle_np.Camera / oracle_np.Config objects.
"""
import math

import numpy as np

import oracle_c
import oracle_np as onp
from conftest import assert_frame_close


def random_scene(n, seed, deg=0, box=((-2, 2), (-2, 2), (2, 8)), scale=(0.02, 0.2), opac_mu=0.0):
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(*box[0], n), rng.uniform(*box[1], n), rng.uniform(*box[2], n)], 1)
    scales = np.exp(rng.uniform(math.log(scale[0]), math.log(scale[1]), (n, 3)))
    quats = rng.normal(size=(n, 4)) + 1e-3
    opac = 1.0 / (1.0 + np.exp(-rng.normal(opac_mu, 1.5, n)))
    k = (deg + 1) ** 2
    sh = 0.5 * rng.normal(size=(n, k, 3))
    if k > 1:
        sh[:, 1:] *= 0.3
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return f(means), f(scales), f(quats), f(opac), f(sh), deg


def look_at_view(eye, target, down=(0, 1, 0)):
    """A rigid model->camera matrix (+Z forward, +Y roughly `down`, +X right)."""
    eye, target, down = (np.asarray(v, float) for v in (eye, target, down))
    z = target - eye; z /= np.linalg.norm(z)
    x = np.cross(down, z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    V = np.eye(4); V[:3, :3] = R; V[:3, 3] = -R @ eye
    return V.astype(np.float32)


class forced_fine:
    """with forced_fine(drv): frames small enough for fine tiles (sgs_tuning.fine_tile_pixels) ARE rendered through them, whatever the
    growth of their record count says (sgs_tuning.fine_tile_growth >= 16: the pixel rule alone) — the tests' way to the fine-tile path on
    scenes of large splats, which the library itself would render through 16x16-pixel tiles."""

    def __init__(self, drv):
        self.drv = drv

    def __enter__(self):
        self.keep = self.drv.tuning()["fine_tile_growth"]
        self.drv.set_tuning(fine_tile_growth=1.0e9)
        return self

    def __exit__(self, *exc):
        self.drv.set_tuning(fine_tile_growth=self.keep)
        return False


def check_against_oracle(drv, scene, cam, cfg=None, rows=(0, -1), what="", queues=True, upload=True, thorough=True):
    """Full comparison of one frame: counts and queues bit-exact, splat attributes to fp32 rounding,
    image within the parity tolerance.  upload=False: `scene` is already the driver's uploaded scene (several poses
    of a multi-million-Gaussian scene are checked against one upload).  thorough=False (a few long cases of the CPU emulator, whose point
    lies elsewhere): the renders that only hold one switch against another — SGS_FLAG_NO_DEEP, no chunk culling, reference binning with
    the lazy sort, and the same under fine tiles — are left to the other cases; everything held against the ORACLE stays."""
    if upload:
        drv.upload(*scene)
    # production path: tight bin rects, queues sorted lazily and only as far as the composite reads them — on 16x16-pixel tiles, the tiling
    # of the reference's integer structures and of the test hooks (a small frame's default, fine tiles, is checked at the end)
    img, st = drv.render(cam, cfg, rows, fine=False)
    # ... which counts D_f only on request (the drivers ask for it): the instantiation without the bookkeeping — the one
    # a sweep runs — must produce the same frame from the same queues
    img_plain, st_plain = drv.render(cam, cfg, rows, stats=False, fine=False)
    assert (img_plain == img).all() and st_plain["d_total"] == st["d_total"] and st_plain["n_visible"] == st["n_visible"] \
        and st_plain["d_fetched"] == 0, f"{what}: the frame depends on whether D_f is counted"
    # (the instantiation without D_f is also the only one that takes the deep-tile path: windows of a long-lived tile culled against its
    #  live pixels before anything is ranked — the comparison above holds the two paths against each other, bit for bit; and so does the
    #  runtime switch, SGS_FLAG_NO_DEEP, inside the one instantiation)
    if thorough:
        img_nodeep, st_nodeep = drv.render(cam, cfg, rows, stats=False, deep=False, fine=False)
        assert (img_nodeep == img).all() and st_nodeep["n_deep_windows"] == 0, f"{what}: the deep-tile cull changed a pixel"
    assert st["n_deep_windows"] == 0
    st["n_deep_windows_plain"] = st_plain["n_deep_windows"]
    # test hook: no chunk culling (every chunk of the scene projected).  The per-chunk bounds may only have skipped
    # chunks none of whose Gaussians is visible: same N_v, same queues, same frame.
    if thorough:
        drv.row_records(0, reset=True)
        img_all, st_all = drv.render(cam, cfg, rows, chunk_cull=False, fine=False)
        assert (img_all == img).all() and st_all["n_visible"] == st["n_visible"] and st_all["d_total"] == st["d_total"] \
            and st_all["d_fetched"] == st["d_fetched"], f"{what}: chunk culling changed the frame"
        # the per-row record counters (what cost-balanced bands are cut from) add up to D
        gy_ = (cam.height + 15) // 16
        rr = drv.row_records(gy_, reset=True)
        assert int(rr.sum()) == st_all["d_total"] and (rr[:rows[0]] == 0).all() and (rows[1] < 0 or (rr[rows[1]:] == 0).all())
    # test hook: order every queue completely (production binning)
    img_full, st_full = drv.render(cam, cfg, rows, full_sort=True)
    assert (img_full == img).all(), f"{what}: lazy and full sort must blend the same records in the same order"
    assert st_full["d_fetched"] == st["d_fetched"] and st_full["d_total"] == st["d_total"]
    # test hook: REFERENCE binning (S3's rect, what the oracle defines) + extent-only quadrant test.  The production
    # path may only have dropped records / (wave, splat) pairs that no pixel could use: frames bit-identical.
    img_loose, st_loose = drv.render(cam, cfg, rows, full_sort=True, loose_cull=True)
    assert (img_loose == img).all(), f"{what}: culling changed the frame"
    if thorough:
        img_ref_lazy, st_ref_lazy = drv.render(cam, cfg, rows, loose_cull=True)
        assert (img_ref_lazy == img).all(), f"{what}: culling changed the frame (lazy sort)"
        assert st_ref_lazy["d_fetched"] == st_loose["d_fetched"] and st_ref_lazy["d_total"] == st_loose["d_total"]
    assert st["d_total"] <= st_loose["d_total"] and st["d_fetched"] <= st_loose["d_fetched"]
    ref, aux = oracle_c.render(*scene, cam, cfg, rows[0], rows[1])
    assert st["n_visible"] == aux["n_visible"] == st_loose["n_visible"], (what, st["n_visible"], aux["n_visible"])
    assert st_loose["d_total"] == aux["D"], (what, st_loose["d_total"], aux["D"])
    off, ids, slot_ids, splats = drv.intermediates()
    assert (off == aux["offsets"]).all(), f"{what}: tile offsets differ"
    if queues:
        assert (ids == aux["ids"]).all(), f"{what}: per-tile queue order differs from (depth bits, index) order"
    # splat table, keyed by Gaussian index
    vis = np.nonzero(aux["tiles"] > 0)[0]
    assert (np.sort(slot_ids) == vis).all(), f"{what}: set of visible Gaussians differs"
    sp = splats[np.argsort(slot_ids)]
    f = sp.view(np.float32)
    assert (sp[:, 9] == aux["depth_bits"][vis]).all(), f"{what}: depth bits differ"
    r = aux["rect"][vis]
    assert ((sp[:, 10] & 0xffff) == r[:, 0]).all() and ((sp[:, 10] >> 16) == r[:, 1]).all() and \
           ((sp[:, 11] & 0xffff) == r[:, 2]).all() and ((sp[:, 11] >> 16) == r[:, 3]).all(), f"{what}: tile rects differ"
    # mean2D: the oracle's fp64 value rounded to fp32 — bit for bit, but for ONE fp32 ulp where the two fp64 evaluations (a reciprocal and a
    # product here, a division there) fall on either side of a rounding boundary (gpu_fuzz_trained seed 415: one of 155 k splats, r06zh)
    dxy = np.abs(f[:, 0:2] - aux["xy"][vis])
    assert (dxy <= np.maximum(1e-3 * 2 ** -10, np.spacing(np.abs(aux["xy"][vis])))).all(), f"{what}: mean2D {dxy.max():.2e}"
    # conic: 1e-5 relative, entry by entry — an entry below a millionth of the conic's largest (a numerical zero: the off-diagonal of an
    # isotropic Gaussian comes out as 1e-17 on both sides, gpu_fuzz_rooms seed 434) is held against that scale instead of against itself
    cref = aux["conic"][vis]
    rel = np.abs(np.stack([f[:, 2], f[:, 3], f[:, 4]], 1) - cref) / np.maximum(np.abs(cref), 1e-6 * np.abs(cref).max(axis=1, keepdims=True, initial=0) + 1e-30)
    assert rel.max(initial=0) < 1e-5, f"{what}: conic {rel.max():.2e}"
    assert (f[:, 5] == aux["opacity"][vis]).all(), f"{what}: opacity"
    rgb = np.stack([f[:, 6], f[:, 7], f[:, 8]], 1)
    assert np.abs(rgb - aux["rgb"][vis]).max(initial=0) < 2e-5, f"{what}: SH colour"
    # the band's pixel rows only (rows outside it are not rendered by either side)
    gy = (cam.height + 15) // 16
    ya, yb = 16 * rows[0], min(cam.height, 16 * (gy if rows[1] < 0 else rows[1]))
    worst = assert_frame_close(img[ya:yb], ref[ya:yb], aux["margin"][ya:yb], aux["recheck"], what=what, y0=ya)
    if "d_fetched" in st_loose and st_loose["d_fetched"]:
        # D_f (reference binning) only differs from the oracle's where a pixel sat on the termination threshold
        assert abs(st_loose["d_fetched"] - aux["D_f"]) <= max(8, 2e-3 * aux["D_f"]), (what, st_loose["d_fetched"], aux["D_f"])
    # FINE TILES — the frame through tiles of 8x8 or 4x4 pixels (sgs_tuning.fine_tile_pixels; sgs_common.h), forced here for every frame small
    # enough (the library itself also asks that the split does not multiply the records by more than fine_tile_growth).  The same splats reach
    # every pixel (S3's rect stays a rect of 16x16-pixel tiles: N_v is the oracle's), the blend's coordinates are relative to another origin:
    # the oracle's frame within the same tolerance, every pixel, threshold-sensitive ones two-sidedly — and the instantiations / switches
    # that must not change a frame do not change this one either.
    with forced_fine(drv):
        img_f, st_f = drv.render(cam, cfg, rows)
        st["n_tiles_fine"] = st_f["n_tiles"]
        if st_f["n_tiles"] != st["n_tiles"]:
            assert st_f["n_visible"] == aux["n_visible"], (what, st_f["n_visible"], aux["n_visible"])
            img_fp, st_fp = drv.render(cam, cfg, rows, stats=False)
            assert (img_fp == img_f).all() and st_fp["d_total"] == st_f["d_total"], f"{what}: fine tiles: the frame depends on whether D_f is counted"
            if thorough:
                img_fn, _ = drv.render(cam, cfg, rows, stats=False, deep=False)
                img_fa, st_fa = drv.render(cam, cfg, rows, chunk_cull=False)
                assert (img_fn == img_f).all() and (img_fa == img_f).all() and st_fa["d_total"] == st_f["d_total"], f"{what}: fine tiles: a switch changed the frame"
            worst = max(worst, assert_frame_close(img_f[ya:yb], ref[ya:yb], aux["margin"][ya:yb], aux["recheck"], what=what + " [fine tiles]", y0=ya))
        else:
            assert (img_f == img).all(), f"{what}: SGS_FLAG_NO_FINE_TILES changed a frame that is not rendered through fine tiles"
    # ... and what the library does with this frame when nobody tells it: one of the tilings above (or the one between them), its own decision
    # (fine_shift_of: the pixel rule and the growth of the record count, estimated from the scene's probe) — the same frame, bit for bit
    img_d, st_d = drv.render(cam, cfg, rows)
    st["n_tiles_default"] = st_d["n_tiles"]
    if st_d["n_tiles"] == st["n_tiles"]:
        assert (img_d == img).all(), f"{what}: the default tiling is 16x16 and the frame is not the 16x16 frame"
    elif st_d["n_tiles"] == st_f["n_tiles"]:
        assert (img_d == img_f).all() and st_d["d_total"] == st_f["d_total"], f"{what}: the default tiling is the forced one and the frame is not"
    else:
        assert st["n_tiles"] < st_d["n_tiles"] < st_f["n_tiles"] and st_d["n_visible"] == aux["n_visible"]
        worst = max(worst, assert_frame_close(img_d[ya:yb], ref[ya:yb], aux["margin"][ya:yb], aux["recheck"], what=what + " [8x8-pixel tiles]", y0=ya))
    return img, st, aux, worst


# ------------------------------------------------------------------------------------------------
def case_config1(drv, n=10_000):
    scene, cam = onp.config1_scene(n=n, seed=0)
    return check_against_oracle(drv, scene, cam, what=f"config1 n={n}")


def case_sh_degrees(drv, n=1500, thorough=True):
    for deg in (0, 1, 2, 3):
        scene = random_scene(n, 20 + deg, deg, box=((-3, 3), (-2, 2), (-3, 3)))
        view = look_at_view((4.0, 0.5, 5.0), (0.0, 0.0, 0.0))
        cam = onp.Camera(160, 128, 110.0, 105.0, 81.2, 60.7, view)
        check_against_oracle(drv, scene, cam, what=f"sh degree {deg}", thorough=thorough or deg == 3)
        if deg == 3:     # evaluate a degree-3 scene at lower degrees
            for d in (0, 2):
                check_against_oracle(drv, scene, cam, onp.Config(sh_degree=d), what=f"deg3 scene at degree {d}", thorough=thorough)


def case_ragged(drv):
    for n, (w, h) in ((1, (16, 16)), (63, (40, 24)), (65, (100, 70)), (129, (33, 17)), (1000, (250, 130))):
        scene = random_scene(n, 100 + n, 1, scale=(0.05, 0.4))
        cam = onp.Camera(w, h, 0.7 * w, 0.7 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
        check_against_oracle(drv, scene, cam, what=f"ragged n={n} {w}x{h}")


def case_fuzz(drv, seeds, max_n=700, max_res=(260, 160), wild=False, thorough_every=1):
    """Seeded random small frames through the full comparison: scene size, resolution (ragged tiles), SH degree, splat
    sizes from sub-pixel to screen-filling, opacities down to the cut-off, cameras inside and outside the cloud (Gaussians
    behind the camera, across the near plane, off screen), non-default thresholds / dilation / background, and a band of
    tile rows now and then."""
    for seed in seeds:
        rng = np.random.default_rng(10_000 + seed)
        n = int(rng.integers(1, max_n))
        w, h = int(rng.integers(17, max_res[0])), int(rng.integers(17, max_res[1]))
        deg = int(rng.integers(0, 4))
        lo = float(10 ** rng.uniform(-2.7, -1.0)); hi = lo * float(10 ** rng.uniform(0.3, 2.0))
        if wild:                     # needles and pancakes up to the size of the scene, specks far below a pixel
            lo = float(10 ** rng.uniform(-4.0, -1.0)); hi = lo * float(10 ** rng.uniform(0.3, 4.0))
        scene = random_scene(n, 20_000 + seed, deg, box=((-3, 3), (-2, 2), (-1, 9)), scale=(lo, min(hi, 20.0 if wild else 3.0)),
                             opac_mu=float(rng.uniform(-3.0, 2.0)))
        if wild and seed % 3 == 0:   # opacities at both ends of (0, 1); a tenth of the cloud pushed far away (last depth buckets)
            scene[3][:] = np.clip(np.where(rng.random(n) < 0.5, 1.0 - 10 ** rng.uniform(-7, -2, n), 10 ** rng.uniform(-4, -1, n)), 1e-6, 1.0 - 1e-7).astype(np.float32)
            far = rng.random(n) < 0.1
            scene[0][far, 2] = (scene[0][far, 2] * float(10 ** rng.uniform(1, 3.5))).astype(np.float32)
            scene[1][far] *= np.float32(30.0)
        eye = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(-3, 6)])
        target = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(3, 8)])
        if np.linalg.norm(target - eye) < 0.5:
            target = eye + np.array([0.1, 0.0, 1.0])
        down = np.array([rng.normal(0, 0.3), 1.0, rng.normal(0, 0.3)])
        view = look_at_view(eye, target, down)
        f = float(w * rng.uniform(0.12 if wild else 0.35, 1.6))
        cam = onp.Camera(w, h, f, f * float(rng.uniform(0.9, 1.1)), w / 2.0 + float(rng.uniform(-3, 3)),
                         h / 2.0 + float(rng.uniform(-3, 3)), view)
        cfg = onp.Config(near=float(rng.choice([0.2, 0.05, 0.5])), dilation=float(rng.choice([0.3, 0.1, 0.6])),
                         alpha_min=float(rng.choice([1.0 / 255.0, 0.01, 0.002])), alpha_max=float(rng.choice([0.99, 0.9])),
                         t_min=float(rng.choice([1.0e-4, 1.0e-3])), background=tuple(float(v) for v in rng.uniform(0, 1, 3)),
                         sh_degree=int(rng.integers(0, deg + 1)) if rng.random() < 0.3 else -1)
        gy = (h + 15) // 16
        rows = (0, -1)
        if gy >= 3 and rng.random() < 0.3:
            r0 = int(rng.integers(0, gy - 1)); rows = (r0, int(rng.integers(r0 + 1, gy + 1)))
        if seed % 4 == 0:            # default constants and the whole frame: also the depth / coverage outputs (f-4)
            cfg, rows = None, (0, -1)
        check_against_oracle(drv, scene, cam, cfg, rows, what=f"fuzz seed {seed} (n={n} {w}x{h} deg {deg} rows {rows})",
                             thorough=seed % thorough_every == 0)
        if seed % 5 == 1 and rows == (0, -1):      # interleaved tile-row shards reproduce the frame's rows bit for bit
            full, st_full = drv.render(cam, cfg)
            stride = int(rng.integers(2, 5))
            d_sum = 0
            for phase in range(min(stride, gy)):         # (a shard beyond the frame's rows owns nothing and is never rendered)
                part, st_p = drv.render(cam, cfg, interleave=(stride, phase))
                for k, row in enumerate(range(phase, gy, stride)):
                    y0, y1 = 16 * row, min(16 * row + 16, h)
                    assert (part[16 * k: 16 * k + (y1 - y0)] == full[y0:y1]).all(), f"fuzz seed {seed}: stride {stride} phase {phase} row {row}"
                d_sum += st_p["d_total"]
            assert d_sum == st_full["d_total"], f"fuzz seed {seed}: interleaved shards queue {d_sum} records, the frame {st_full['d_total']}"
        if seed % 4 == 0:
            rgb0, _ = drv.render(cam, stats=False)          # (the production instantiation)
            rgb, aux = drv.render_aux(cam)
            assert (rgb == rgb0).all(), f"fuzz seed {seed}: the aux instantiation changed the colours"
            ref, o = oracle_c.render(*scene, cam)
            safe = o["margin"] >= 1e-4
            zmax = max(1.0, float(np.max(o["depth_image"])))
            if safe.any():
                assert np.abs(aux[..., 0] - o["depth_image"])[safe].max() < 1e-3 * zmax, f"fuzz seed {seed}: depth"
                assert np.abs(aux[..., 1] - (1.0 - o["final_T"]))[safe].max() < 1e-3, f"fuzz seed {seed}: coverage"


def case_non_finite_gaussians(drv, n=3000, res=(160, 120), seed=3):
    """NaN / +-inf in a Gaussian's mean, scale, rotation or opacity (or an all-zero quaternion): that Gaussian is never
    visible and the frame is, bit for bit, the frame of the scene without it — its chunk neighbours, the Z-order and the
    per-chunk bounds are unaffected."""
    rng = np.random.default_rng(seed)
    means, scales, quats, opac, sh, deg = random_scene(n, 900 + seed, 2, scale=(0.03, 0.4))
    hit = rng.random(n) < 0.05
    bm, bs, bq, bo = means.copy(), scales.copy(), quats.copy(), opac.copy()
    vals = [np.nan, np.inf, -np.inf]
    for i in np.nonzero(hit)[0]:
        f = int(rng.integers(0, 6)); v = vals[int(rng.integers(0, 3))]
        if f == 0: bm[i, int(rng.integers(0, 3))] = v
        elif f == 1: bs[i, int(rng.integers(0, 3))] = v
        elif f == 2: bq[i, int(rng.integers(0, 4))] = v
        elif f == 3: bq[i] = 0.0
        elif f == 4: bo[i] = np.nan
        else: bm[i] = v
    w, h = res
    cam = onp.Camera(w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
    keep = ~hit
    drv.upload(means[keep], scales[keep], quats[keep], opac[keep], sh[keep], deg)
    ref, st_ref = drv.render(cam)
    drv.upload(bm, bs, bq, bo, sh, deg)
    img, st = drv.render(cam)
    assert np.isfinite(img).all() and (img == ref).all(), "a non-finite Gaussian changed the frame"
    assert st["d_total"] == st_ref["d_total"]


def case_padding_lanes(drv):
    """N not a multiple of 64 with a camera looking down -z: the padding lanes of the last chunk must stay
    culled (their placeholder mean projects to a huge POSITIVE depth for such a view)."""
    scene = random_scene(70, 77, 0, box=((-1, 1), (-1, 1), (-6, -2)), scale=(0.05, 0.3))
    view = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)        # rotate pi about y: looks down -z
    cam = onp.Camera(64, 64, 50.0, 50.0, 32.0, 32.0, view)
    img, st, aux, _ = check_against_oracle(drv, scene, cam, what="padding lanes")
    assert st["n_visible"] == aux["n_visible"] <= 70


def case_empty(drv):
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32),
             np.zeros((0,), np.float32), np.zeros((0, 1, 3), np.float32), 0)
    cam = onp.Camera(48, 32, 40.0, 40.0, 24.0, 16.0, np.eye(4, dtype=np.float32))
    drv.upload(*empty)
    img, st = drv.render(cam, onp.Config(background=(0.25, 0.5, 0.75)))
    assert st["n_visible"] == 0 and st["d_total"] == 0
    assert np.allclose(img, [0.25, 0.5, 0.75], atol=1e-7)
    # everything culled (all behind the camera)
    scene = random_scene(300, 1, 0, box=((-1, 1), (-1, 1), (-8, -2)))
    drv.upload(*scene)
    img, st = drv.render(cam)
    assert st["n_visible"] == 0 and st["d_total"] == 0 and (img == 0).all()


def case_interleaved_rows(drv, stride, n=2500, res=(208, 150)):
    """Interleaved tile-row shards (rank p of `stride` owns frame rows p, p+stride, ...): every shard's compact image
    holds exactly those rows of the full frame, bit for bit, and the shards' queues add up to the frame's — on 16x16-pixel tiles and on
    the fine tiles a frame this small gets by default (a 16-pixel row is then 2 or 4 rows of tiles, owned or not as a whole)."""
    scene = random_scene(n, 7, 2, scale=(0.03, 0.3))
    w, h = res
    cam = onp.Camera(w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
    drv.upload(*scene)
    gy = (h + 15) // 16
    with forced_fine(drv):              # (switches nothing for the fine=False renders and the hooks: those are 16x16 by flag)
        for fine in (False, True):
            full, st_full = drv.render(cam, fine=fine)
            _, st_full_ref = drv.render(cam, loose_cull=True)
            assert (st_full["n_tiles"] != gy * ((w + 15) // 16)) == fine, "this frame is meant to be small enough for fine tiles"
            cp = next(c for c in (16, 8, 4) if -(-w // c) * -(-h // c) == st_full["n_tiles"])          # the tile's side in pixels
            d_sum = d_ref_sum = pix = 0
            seen = np.zeros(gy, bool)
            for phase in range(stride):
                img, st = drv.render(cam, interleave=(stride, phase), fine=fine)
                _, st_ref = drv.render(cam, interleave=(stride, phase), loose_cull=True)
                owned = list(range(phase, gy, stride))
                rows_of_tiles = sum(-(-(min(16 * row + 16, h) - 16 * row) // cp) for row in owned)
                assert img.shape[0] == 16 * len(owned) and st["n_tiles"] == rows_of_tiles * (-(-w // cp)), (st["n_tiles"], rows_of_tiles, cp)
                for k, row in enumerate(owned):
                    y0, y1 = 16 * row, min(16 * row + 16, h)
                    assert (img[16 * k: 16 * k + (y1 - y0)] == full[y0:y1]).all(), f"stride {stride} phase {phase}: frame row {row} (fine {fine})"
                    assert (img[16 * k + (y1 - y0): 16 * k + 16] == -1).all(), "pixels below the frame must stay untouched"
                    seen[row] = True
                    pix += (y1 - y0) * w
                d_sum += st["d_total"]; d_ref_sum += st_ref["d_total"]
                # a sub-range of the owned rows: only those are written
                if len(owned) >= 2:
                    part, st_p = drv.render(cam, rows=(1, 2), interleave=(stride, phase), fine=fine)
                    y0, y1 = 16 * owned[1], min(16 * owned[1] + 16, h)
                    assert (part[16: 16 + (y1 - y0)] == full[y0:y1]).all() and (part[:16] == -1).all() and (part[32:] == -1).all()
            assert seen.all() and pix == w * h
            assert d_sum == st_full["d_total"] and d_ref_sum == st_full_ref["d_total"]


def case_chunk_bounds(drv, n=6000, res=(208, 150)):
    """Per-chunk bounds (k_chunk_bounds / chunk_outside): a scene much wider than the view, so most 64-Gaussian chunks
    (Z-ordered at upload) lie outside the image or behind the camera.  The skipped chunks must hold no visible Gaussian
    — N_v, D, queues and frames equal the oracle's and the no-culling run's, for the full frame and for bands of tile
    rows (where chunks above and below the band are skipped too) — and a good share of the chunks must actually be
    skipped, or the test would prove nothing."""
    # a wall 24 m x 12 m facing the camera at ~6 m (the view covers a tenth of it) and a cloud behind the camera
    means, scales, quats, opac, sh, deg = random_scene(n, 17, 1, box=((-12, 12), (-6, 6), (5.9, 6.1)), scale=(0.02, 0.12))
    rng = np.random.default_rng(18)
    back = rng.random(n) < 0.3
    means[back, 2] = rng.uniform(-10.0, -2.0, int(back.sum())).astype(np.float32)
    scene = (means, scales, quats, opac, sh, deg)
    w, h = res
    view = look_at_view((0.3, -0.2, -1.0), (0.5, 0.4, 6.0))
    cam = onp.Camera(w, h, 0.9 * w, 0.9 * w, w / 2.0 + 3.3, h / 2.0 - 2.1, view)
    gy = (h + 15) // 16
    n_chunks = (n + 63) // 64
    skipped = {}
    for rows in ((0, -1), (0, 2), (gy // 2, gy // 2 + 1), (gy - 2, gy)):
        check_against_oracle(drv, scene, cam, rows=rows, what=f"chunk bounds rows {rows}")
        drv.render(cam, None, rows)
        sk = drv.chunk_skipped()
        assert len(sk) == n_chunks
        skipped[rows] = int(sk.sum())
        drv.render(cam, None, rows, chunk_cull=False)
        assert int(drv.chunk_skipped().sum()) == 0
    assert skipped[(0, -1)] > 0.4 * n_chunks, skipped              # most of the scene is outside the view
    for rows in list(skipped)[1:]:
        assert skipped[rows] > skipped[(0, -1)], skipped            # a band skips what is above / below it as well


def case_tile_rows(drv, n=2500, res=(208, 150)):
    """Tile-row shards: each band equals the oracle's band; their union equals the full frame bit-exactly (16x16-pixel tiles and the fine
    tiles a frame this small gets by default)."""
    scene = random_scene(n, 7, 2, scale=(0.03, 0.3))
    w, h = res
    cam = onp.Camera(w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
    drv.upload(*scene)
    gy = (h + 15) // 16
    cuts = [0, gy // 3, gy // 3 + 1, gy]
    with forced_fine(drv):              # (switches nothing for the fine=False renders and the hooks: those are 16x16 by flag)
        for fine in (False, True):
            full, st_full = drv.render(cam, fine=fine)
            union = np.full_like(full, -1.0)
            d_sum = 0
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                out = np.full_like(full, -1.0)
                img, st = drv.render(cam, None, (r0, r1), out=out, fine=fine)
                y0, y1 = r0 * 16, min(r1 * 16, h)
                assert (img[:y0] == -1).all() and (img[y1:] == -1).all(), "rows outside the band must be untouched"
                ref, aux = oracle_c.render(*scene, cam, None, r0, r1)
                _, st_ref = drv.render(cam, None, (r0, r1), loose_cull=True)      # reference binning: the oracle's D
                assert st_ref["d_total"] == aux["D"] and st["n_visible"] == aux["n_visible"] and (fine or st["d_total"] <= aux["D"])
                assert_frame_close(img[y0:y1], ref[y0:y1], aux["margin"][y0:y1], aux["recheck"], what=f"tile rows [{r0}, {r1}) fine={fine}", y0=y0)
                union[y0:y1] = img[y0:y1]
                d_sum += st["d_total"]
            assert (union == full).all(), f"union of tile-row bands != full frame (fine {fine})"
            assert d_sum == st_full["d_total"]


def case_depth_ties(drv):
    """Equal fp32 depths inside a tile must order by Gaussian index: short runs (in-place fix-up) and
    a run longer than SGS_TIE_RUN_MAX (the (index, depth) resort)."""
    rng = np.random.default_rng(5)
    n = 240
    means = np.zeros((n, 3), np.float32)
    means[:, 0] = rng.uniform(-0.15, 0.15, n); means[:, 1] = rng.uniform(-0.15, 0.15, n)
    depth = np.full(n, 3.0, np.float32)
    depth[:60] = 2.5                                   # one run of 60 (> 32) ...
    depth[60:120] = np.repeat(np.linspace(2.6, 2.9, 20, dtype=np.float32), 3)    # ... twenty runs of 3
    depth[120:] = np.linspace(3.0, 4.0, 120, dtype=np.float32)                   # ... and distinct keys
    means[:, 2] = depth
    perm = rng.permutation(n)
    means = means[perm]
    scales = np.full((n, 3), 0.05, np.float32)
    quats = np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1))
    opac = np.full(n, 0.05, np.float32)
    sh = rng.normal(size=(n, 1, 3)).astype(np.float32)
    cam = onp.Camera(64, 64, 64.0, 64.0, 32.0, 32.0, np.eye(4, dtype=np.float32))
    check_against_oracle(drv, (means, scales, quats, opac, sh, 0), cam, what="depth ties")


def case_sort_classes(drv, sizes=(700, 2500, 6000, 9500)):
    """Queue lengths that land in each sort class: S (<=1024), M (<=4096), L (<=9216, all in LDS)
    and X (spill: ping-pong in HBM)."""
    for n in sizes:
        rng = np.random.default_rng(n)
        means = np.stack([rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), rng.uniform(1.0, 30.0, n)], 1).astype(np.float32)
        scales = np.full((n, 3), 2.0, np.float32) * means[:, 2:3]          # every splat covers all four tiles
        quats = rng.normal(size=(n, 4)).astype(np.float32)
        opac = rng.uniform(0.005, 0.012, n).astype(np.float32)            # faint, but clear of the 1/255 cut-off
        sh = rng.normal(size=(n, 1, 3)).astype(np.float32)
        cam = onp.Camera(32, 32, 32.0, 32.0, 16.0, 16.0, np.eye(4, dtype=np.float32))
        img, st, aux, _ = check_against_oracle(drv, (means, scales, quats, opac, sh, 0), cam, what=f"sort class n={n}")
        assert st["max_tile_len"] == n


def case_sparse_lists(drv, n=9000, res=(48, 40)):
    """Thousands of faint splats a pixel or two wide over a small image: no pixel saturates, every tile reads its whole queue in many batches,
    and every listed splat reaches a handful of a quadrant's 64 pixels — the shape of a low-resolution frame's slowest tiles (and the case the
    per-pixel-mask walk of round 6 was built for, profiles/r06n).  A second scene mixes in splats that cover whole quadrants and a few opaque
    ones (pixels that end in the middle of a list)."""
    w, h = res
    f = 40.0
    for seed, big in ((33, 0.0), (34, 0.05)):
        rng = np.random.default_rng(seed)
        z = rng.uniform(2.0, 30.0, n)
        means = np.stack([rng.uniform(-0.5 * w, 0.5 * w, n) * z / f, rng.uniform(-0.5 * h, 0.5 * h, n) * z / f, z], 1).astype(np.float32)
        s_px = np.where(rng.random(n) < big, rng.uniform(3.0, 12.0, n), rng.uniform(0.15, 0.7, n))          # sigma in pixels
        scales = (np.repeat((s_px * z / f)[:, None], 3, 1) * rng.uniform(0.6, 1.0, (n, 3))).astype(np.float32)
        quats = rng.normal(size=(n, 4)).astype(np.float32)
        opac = rng.uniform(0.02, 0.25, n).astype(np.float32)
        if big:
            opac[(rng.random(n) < 0.03) & (s_px < 1.0)] = 0.995                  # (a pixel here and there ends inside a chunk)
            opac[s_px >= 3.0] *= 0.2
        sh = rng.normal(size=(n, 1, 3)).astype(np.float32)
        cam = onp.Camera(w, h, f, f, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
        img, st, aux, _ = check_against_oracle(drv, (means, scales, quats, opac, sh, 0), cam, what=f"sparse lists (seed {seed})")
        assert st["max_tile_len"] > 512, st["max_tile_len"]
        assert float((aux["final_T"] > 0.05).mean()) > 0.5 or big


def case_deep_tile(drv, n_back=6000):
    """Tiles that keep consuming batches for a few pixels: three opaque layers saturate every pixel of a 32x32 image except a 5x5 hole
    per tile, and thousands of splats behind them are visible only through the holes.  After SGS_DEEP_AFTER batches such a tile culls every
    resident window of its queue against the live pixels BEFORE ranking / staging (k_tile_render, deep tiles) and blends the survivors as one
    batch: the frame must equal the oracle's AND the frame of the ordinary path (the D_f-counting instantiation) bit for bit — both
    checked by check_against_oracle — and the path must actually have run."""
    rng = np.random.default_rng(21)
    ys, xs = np.mgrid[0:32, 0:32]
    hole = ((xs % 16 >= 8) & (xs % 16 <= 12) & (ys % 16 >= 3) & (ys % 16 <= 7))
    px, py = xs[~hole].astype(np.float64), ys[~hole].astype(np.float64)
    f, c = 32.0, 16.0
    front = []
    for z in (1.0, 1.05, 1.1):
        front.append(np.stack([(px - c + 0.5) * z / f, (py - c + 0.5) * z / f, np.full(px.shape, z)], 1))
    nb = n_back
    zb = rng.uniform(2.0, 40.0, nb)
    back = np.stack([rng.uniform(-15.5, 15.5, nb) * zb / f, rng.uniform(-15.5, 15.5, nb) * zb / f, zb], 1)
    means = np.concatenate(front + [back]).astype(np.float32)
    nf = means.shape[0] - nb
    scales = np.concatenate([np.repeat((0.55 * means[:nf, 2] / f)[:, None], 3, 1),                # ~0.55 px: covers its own pixel
                             np.repeat((rng.uniform(0.6, 2.5, nb) * zb / f)[:, None], 3, 1)]).astype(np.float32)
    quats = rng.normal(size=(means.shape[0], 4)).astype(np.float32)
    opac = np.concatenate([np.full(nf, 0.999), rng.uniform(0.05, 0.6, nb)]).astype(np.float32)
    sh = rng.normal(size=(means.shape[0], 1, 3)).astype(np.float32)
    perm = rng.permutation(means.shape[0])
    scene = (means[perm], scales[perm], quats[perm], opac[perm], sh[perm], 0)
    cam = onp.Camera(32, 32, f, f, c, c, np.eye(4, dtype=np.float32))
    img, st, aux, _ = check_against_oracle(drv, scene, cam, what="deep tile")
    assert st["max_tile_len"] > 1024 and st["n_deep_windows_plain"] >= 4, (st["max_tile_len"], st["n_deep_windows_plain"])
    # most pixels stopped inside the front layers, the holes read on deep into the queue
    stopped = float((aux["final_T"] < 1e-3).mean())
    assert stopped > 0.5 and st["d_fetched"] > 0.1 * st["d_total"], (stopped, st["d_fetched"], st["d_total"], st["n_deep_windows_plain"])
    print(f"[deep tile] {st['n_deep_windows_plain']} windows culled first; {100 * stopped:.0f} % of the pixels stopped; D_f / D = {st['d_fetched'] / st['d_total']:.2f}")


def case_big_depth_bucket(drv, n_slab=3000):
    """Thousands of splats of one tile inside a 2-cm slab of a queue that spans 1-12 m (a wall facing the camera), four fifths
    of them at EXACTLY the same depth.  The composite cuts 256 depth buckets from the queue's own range (~3 cm each here), finds
    the slab's bucket longer than one batch and partitions the queue again over that bucket's key range (refinement) — which
    separates everything but the equal depths: that bucket exceeds the LDS capacity, cannot be refined and is sorted through
    HBM, its run of equal keys ordered by index."""
    rng = np.random.default_rng(8)
    n = n_slab + 500
    means = np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n), rng.uniform(1.0, 12.0, n)], 1).astype(np.float32)
    means[:n_slab, 2] = rng.uniform(3.0, 3.02, n_slab).astype(np.float32)
    means[100:100 + (4 * n_slab) // 5, 2] = 3.01               # four fifths of the slab at one depth (the tight bin rects drop some per tile)
    perm = rng.permutation(n); means = means[perm]
    scales = np.full((n, 3), 2.0, np.float32) * means[:, 2:3]      # broad: alpha stays clear of the 1/255 cut-off
    quats = rng.normal(size=(n, 4)).astype(np.float32)
    opac = rng.uniform(0.005, 0.012, n).astype(np.float32)
    sh = rng.normal(size=(n, 1, 3)).astype(np.float32)
    cam = onp.Camera(32, 32, 32.0, 32.0, 16.0, 16.0, np.eye(4, dtype=np.float32))
    img, st, aux, _ = check_against_oracle(drv, (means, scales, quats, opac, sh, 0), cam, what="big depth bucket")
    assert st["n_spill_tiles"] >= 1


def case_full_grid_splat(drv, res=(1920, 1080), thorough=True):
    """One huge splat covering every tile of a 1080p grid (120x68 = 8160 records from one lane) plus
    small ones: exercises the balanced duplication's row-major expansion over a wide rect."""
    w, h = res
    rng = np.random.default_rng(3)
    n = 40
    means = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.5, 0.5, n), rng.uniform(2, 4, n)], 1).astype(np.float32)
    scales = np.full((n, 3), 0.01, np.float32)
    means[17] = [0.0, 0.0, 1.0]; scales[17] = [4.0, 4.0, 0.5]
    quats = np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1))
    opac = np.full(n, 0.3, np.float32); opac[17] = 0.003             # below 1/255 everywhere: binned but never blended
    sh = rng.normal(size=(n, 1, 3)).astype(np.float32)
    cam = onp.Camera(w, h, 0.38 * w, 0.38 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
    img, st, aux, _ = check_against_oracle(drv, (means, scales, quats, opac, sh, 0), cam, what="full-grid splat", thorough=thorough)
    assert aux["tiles"][17] == ((w + 15) // 16) * ((h + 15) // 16)


def case_overflow_retry(drv):
    scene = random_scene(1200, 9, 0, scale=(0.1, 0.5))
    cam = onp.Camera(96, 96, 80.0, 80.0, 48.0, 48.0, np.eye(4, dtype=np.float32))
    drv.set_record_capacity(1024)
    img, st, aux, _ = check_against_oracle(drv, scene, cam, what="overflow -> grow -> retry")
    assert st["d_total"] > 1024 and st["retries"] >= 1
    # and once grown, no more retries
    img2, st2 = drv.render(cam, fine=False)
    assert st2["retries"] == 0 and (img2 == img).all()


def case_depth_aux(drv, n=1500, res=(96, 80)):
    """f-4: expected view depth sum(T alpha z) and coverage 1 - T_final, against the oracle; the RGB of the
    aux kernel instantiation must equal the RGB-only kernel's bit for bit."""
    scene = random_scene(n, 55, 1, scale=(0.03, 0.3))
    w, h = res
    cam = onp.Camera(w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
    drv.upload(*scene)
    rgb0, _ = drv.render(cam, stats=False)
    rgb, aux = drv.render_aux(cam)
    assert (rgb == rgb0).all()
    ref, o = oracle_c.render(*scene, cam)
    safe = o["margin"] >= 1e-4
    zmax = float(np.max(scene[0][:, 2]))
    assert np.abs(aux[..., 0] - o["depth_image"])[safe].max() < 1e-3 * zmax
    assert np.abs(aux[..., 1] - (1.0 - o["final_T"]))[safe].max() < 1e-3
    assert np.abs(aux[..., 0] - o["depth_image"]).max() < zmax / 255.0 + 1e-3 * zmax


def case_determinism(drv, n=4000):
    scene = random_scene(n, 31, 1, scale=(0.03, 0.3))
    cam = onp.Camera(176, 144, 120.0, 120.0, 88.0, 72.0, np.eye(4, dtype=np.float32))
    drv.upload(*scene)
    a, st = drv.render(cam, full_sort=True)
    ids_a = drv.intermediates()[1].copy()
    b, _ = drv.render(cam, full_sort=True)
    ids_b = drv.intermediates()[1]
    c, _ = drv.render(cam, fine=False)
    assert (c == a).all()
    p1, _ = drv.render(cam, stats=False, fine=False); p2, _ = drv.render(cam, stats=False, fine=False)      # the production instantiation twice
    assert (p1 == p2).all() and (p1 == a).all(), "two production renders of the same frame differ"
    assert (a == b).all() and (ids_a == ids_b).all(), "two renders of the same frame differ"
    with forced_fine(drv):
        f1, st1 = drv.render(cam, stats=False); f2, st2 = drv.render(cam, stats=False)  # ... and through fine tiles
    assert st1["n_tiles"] > st["n_tiles"] and (f1 == f2).all() and st1 == st2, "two fine-tile renders of the same frame differ"
    assert float(np.mean(np.abs(f1 - a).max(axis=-1) > 1e-5)) < 2e-3 and float(np.median(np.abs(f1 - a))) < 1e-6, "fine tiles moved the frame by more than rounding"
    d1, sd1 = drv.render(cam, stats=False); d2, sd2 = drv.render(cam, stats=False)      # ... and as the library decides by itself
    assert (d1 == d2).all() and sd1 == sd2 and ((d1 == f1).all() or (d1 == a).all() or sd1["n_tiles"] not in (st["n_tiles"], st1["n_tiles"]))


def case_batch_shares_scene_reads(drv, n=4000, res=(160, 112), quick=False):
    """sgs_render_batch*: the full frames of a group are projected by ONE launch whose waves take the group's work list — (chunk, frame)
    pairs, chunk-major (k_chunk_cull_group / k_preprocess_shared: the frames that want a chunk run side by side, the group reads the scene
    once).  That must not show: every frame of a batch equals the frame rendered alone, bit for bit — for nearly identical views (every
    chunk wanted by all frames), for views turned a quarter circle apart (hardly a chunk wanted twice), for a batch that mixes them
    (groups of four: the kinds alternate), with fine tiles and without."""
    scene = random_scene(n, 77, 2, box=((-4, 4), (-2, 2), (-4, 4)), scale=(0.02, 0.2))
    drv.upload(*scene)
    w, h = res

    def cam_at(yaw_deg, dx=0.0):
        a = np.deg2rad(yaw_deg)
        eye = (0.3 + dx, 0.1, 0.2)
        return onp.Camera(w, h, 0.7 * w, 0.7 * w, w / 2.0, h / 2.0, look_at_view(eye, (eye[0] + np.sin(a), eye[1], eye[2] + np.cos(a))))
    near = [cam_at(10.0 + 2.0 * k, 0.02 * k) for k in range(6)]                  # consecutive frames of a path
    far = [cam_at(90.0 * k) for k in range(4)] + [cam_at(45.0), cam_at(225.0)]   # views that share next to nothing
    mixed = near[:4] + far[:4] + near[4:] + far[4:]
    for cams in ((mixed,) if quick else (near, far, mixed)):       # (quick — the CPU emulator: the mixed batch holds both kinds of group)
        for fine in (False, True):
            with forced_fine(drv):
                alone = [drv.render(c, stats=False, fine=fine)[0] for c in cams]
                batch = drv.render_batch(cams, fine=fine)
            for i, a in enumerate(alone):
                assert (batch[i] == a).all(), f"frame {i} of a batch of {len(cams)} differs from the frame rendered alone (fine {fine})"
    assert any(a.max() > 0.05 for a in alone)


def case_fine_tile_decision(drv, n=500):
    """fine_shift_of (sgs_api.hip): a frame small enough for fine tiles is split while halving the tiles multiplies its (Gaussian, tile)
    records by no more than sgs_tuning.fine_tile_growth — a ratio the library ESTIMATES on the host from the scene's probe (here the
    whole scene: fewer Gaussians than a probe holds).  Scenes of splats from far below a tile to several tiles wide, at two resolutions:
    the decision must agree with the frame's REAL record counts wherever those are clear of the threshold, the frame must be the frame of
    the tiling it chose (bit for bit), a batch and a band must make the same choice as the single frame."""
    g = drv.tuning()["fine_tile_growth"]
    assert g == 2.2
    seen = set()
    for seed, scale in ((1, (0.004, 0.02)), (2, (0.01, 0.05)), (3, (0.03, 0.15)), (4, (0.1, 0.5)), (5, (0.3, 1.2))):
        scene = random_scene(n, 300 + seed, 1, scale=scale, opac_mu=1.0)
        drv.upload(*scene)
        for (w, h) in ((96, 64), (200, 152)):
            cam = onp.Camera(w, h, 0.9 * w, 0.9 * w, w / 2.0, h / 2.0, np.eye(4, dtype=np.float32))
            tiles = lambda c: -(-w // c) * -(-h // c)
            # the three tilings, each forced (the pixel rule alone: fine_tile_pixels = 0 / W H / 4 W H under a growth limit that never binds)
            keep = drv.tuning()
            D, img = {}, {}
            for side, fp in ((16, 0), (8, w * h), (4, 4 * w * h)):
                drv.set_tuning(fine_tile_pixels=fp, fine_tile_growth=1.0e9)
                img[side], st = drv.render(cam, stats=False)
                assert st["n_tiles"] == tiles(side)
                D[side] = st["d_total"]
            drv.set_tuning(fine_tile_pixels=4 * w * h, fine_tile_growth=keep["fine_tile_growth"])     # (both splits allowed by the pixel rule)
            got, st = drv.render(cam, stats=False)
            side = next(c for c in (16, 8, 4) if tiles(c) == st["n_tiles"])
            assert (got == img[side]).all() and st["d_total"] == D[side], f"seed {seed} {w}x{h}: the frame is not the frame of the tiling it chose"
            g1, g2 = D[8] / max(1, D[16]), D[4] / max(1, D[8])
            want = 16 if g1 > g else 8 if g2 > g else 4
            clear = abs(g1 - g) > 0.12 * g and (g1 > g or abs(g2 - g) > 0.12 * g)        # (the estimate bins extents, not the exact rects: a few per cent)
            assert side == want or not clear, f"seed {seed} {w}x{h}: records grow by {g1:.2f}, {g2:.2f} per split, the library chose {side}x{side}"
            seen.add(side)
            gy = (h + 15) // 16
            band, st_b = drv.render(cam, rows=(1, gy - 1), stats=False, out=np.full((h, w, 3), -1.0, np.float32))
            assert (band[16:16 * (gy - 1)] == got[16:16 * (gy - 1)]).all(), f"seed {seed} {w}x{h}: a band chose another tiling than its frame"
            drv.set_tuning(fine_tile_pixels=keep["fine_tile_pixels"], fine_tile_growth=keep["fine_tile_growth"])
    assert seen == {16, 8, 4}, seen          # (the five scenes span the three outcomes)
