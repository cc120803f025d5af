import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sage-3d_official_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must fail loudly — never silently skip — when selected on a box without a GPU
    or without the built HIP library; on the CPU box they are simply deselected by -m "not gpu"."""
    return


# ---------------------------------------------------------------------------------------------
# Parity check shared by the emulator tests (CPU) and the GPU tests.
TOL = 1.0e-3            # BASELINE.json: per-pixel |d| < 1e-3 (fp32)
MARGIN = 1.0e-4         # relative distance to a discontinuity below which fp32/fp64 may decide differently


def assert_frame_close(img, ref, margin, cmax=1.0, tol=TOL, what="frame"):
    """|img - ref| < tol on every pixel whose oracle evaluation stayed clear of the path's own
    discontinuities (alpha == 1/255 cut-off, T == 1e-4 stop); on the few pixels that sit within
    MARGIN of one, a differently-rounded but correct evaluation may take the other branch, which
    moves the pixel by at most alpha_min * T * c <= cmax/255 — that looser bound is asserted there."""
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    err = np.abs(img - ref).max(axis=-1)
    safe = np.asarray(margin) >= MARGIN
    assert safe.mean() > 0.98, f"{what}: too many threshold-sensitive pixels ({1 - safe.mean():.3%})"
    worst = err[safe].max() if safe.any() else 0.0
    assert worst < tol, f"{what}: max |d| = {worst:.3e} on discontinuity-free pixels (tol {tol})"
    if (~safe).any():
        loose = cmax / 255.0 + tol
        assert err[~safe].max() < loose, f"{what}: max |d| = {err[~safe].max():.3e} on sensitive pixels (bound {loose:.3e})"
    return float(worst)


@pytest.fixture(scope="session")
def oracle_c_mod():
    import oracle_c
    oracle_c.build()
    return oracle_c
