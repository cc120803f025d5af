"""Kernel LOGIC under the wave64 emulator (CPU; see tests/hipemu/hip/hip_runtime.h): the product's
own kernel source, compiled for the host, must reproduce the oracle — queues bit-exactly, images within
the parity tolerance.  The same cases run on the real GPU in test_gpu_parity.py."""
import numpy as np
import pytest

import emu_harness
import oracle_np as onp
import parity_cases as pc


@pytest.fixture(scope="module")
def drv():
    d = emu_harness.EmuRenderer(record_capacity=1 << 21)
    yield d
    d.close()


def test_config1(drv):
    pc.case_config1(drv, n=10_000)


def test_sh_degrees(drv):
    pc.case_sh_degrees(drv, n=600)


def test_ragged_sizes(drv):
    pc.case_ragged(drv)


def test_padding_lanes_stay_culled(drv):
    pc.case_padding_lanes(drv)


def test_empty_and_all_culled(drv):
    pc.case_empty(drv)


def test_tile_row_bands(drv):
    pc.case_tile_rows(drv, n=1200, res=(112, 100))


def test_depth_ties(drv):
    pc.case_depth_ties(drv)


def test_big_depth_bucket(drv):
    pc.case_big_depth_bucket(drv, n_slab=1500)


def test_sort_classes(drv):
    pc.case_sort_classes(drv, sizes=(700, 2500, 9500))


def test_full_grid_splat(drv):
    pc.case_full_grid_splat(drv, res=(640, 368))


def test_two_binning_windows(drv):
    # 129 x 65 = 8385 tiles > SGS_WT (8192): the binning kernels walk two LDS windows
    pc.case_full_grid_splat(drv, res=(2064, 1040))


def test_depth_and_coverage_outputs(drv):
    pc.case_depth_aux(drv, n=1500, res=(96, 80))


def test_determinism(drv):
    pc.case_determinism(drv, n=1500)


def test_overflow_retry():
    d = emu_harness.EmuRenderer(record_capacity=1 << 16)
    try:
        pc.case_overflow_retry(d)
    finally:
        d.close()


def test_float_reciprocal_division_is_exact():
    """wave_expand() replaces idx / w by (unsigned)((idx + 0.5f) * (1.0f / w)); exact over every
    rect a 16384x16384 frame can produce (w <= 1024 tiles, idx < 2^20)."""
    for w in range(1, 1025):
        idx = np.arange(0, min(1 << 20, w * 1024), dtype=np.uint32)
        rw = np.float32(1.0) / np.float32(w)
        q = ((idx.astype(np.float32) + np.float32(0.5)) * rw).astype(np.uint32)
        assert (q == idx // w).all(), w


def test_against_committed_golden_fixture(drv):
    """The kernels (under the emulator) against tests/golden/config1_golden.npz — no oracle run involved."""
    import os
    from conftest import assert_frame_close
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_golden.npz"))
    scene, _ = onp.config1_scene(n=int(g["n"]), seed=int(g["seed"]))
    cam = onp.Camera(int(g["width"]), int(g["height"]), float(g["f"]), float(g["f"]), 64.0, 64.0, np.eye(4, dtype=np.float32))
    drv.upload(*scene)
    prod, st_prod = drv.render(cam)                                   # production: tight bin rects, lazy sort
    img, st = drv.render(cam, full_sort=True, loose_cull=True)          # reference binning: the fixture's integer structures
    assert (prod == img).all() and st_prod["d_total"] <= st["d_total"]
    off, ids, _, _ = drv.intermediates()
    assert st["d_total"] == int(g["D"]) and st["n_visible"] == int(g["n_visible"]) and st["d_fetched"] == int(g["D_f"])
    assert (off == g["offsets"]).all() and (ids == g["ids"]).all()
    assert_frame_close(img, g["image"], g["margin"], cmax=2.5, what="golden config1")
