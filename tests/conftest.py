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
MARGIN = 1.0e-4         # relative distance to a threshold below which fp32/fp64 may decide differently (oracle_c.REL_MARGIN)
PARITY_LOG = []         # one line per checked frame: (what, pixels, flagged, worst off-threshold, worst two-sided)


def stored_variants(yx, ptr, rgb):
    """A `recheck` callable over variants stored in a fixture: pixel i (yx[i]) has the leaf colours rgb[ptr[i]:ptr[i+1]]."""
    index = {(int(y), int(x)): i for i, (y, x) in enumerate(np.asarray(yx))}

    def recheck(ys, xs, got):
        got = np.asarray(got, np.float64).reshape(-1, 3)
        best = np.full(len(got), np.inf); leaves = np.zeros(len(got), np.int64)
        for k, (y, x) in enumerate(zip(ys, xs)):
            i = index.get((int(y), int(x)))
            if i is None:
                continue                                    # not in the fixture: stays inf -> the check fails loudly
            v = np.asarray(rgb[ptr[i]:ptr[i + 1]], np.float64)
            best[k] = np.abs(v - got[k]).max(axis=1).min(); leaves[k] = len(v)
        return best, leaves, np.zeros(len(got), bool)
    return recheck


def assert_frame_close(img, ref, margin, recheck, tol=TOL, what="frame", y0=0):
    """|img - ref| < tol on EVERY pixel, with one refinement and no exemption: S6 has two discontinuities (the
    alpha >= 1/255 cut-off and the T(1-alpha) < 1e-4 stop), and a pixel holding a (pixel, Gaussian) pair within MARGIN
    (relative) of one of them may be decided either way by a correctly rounded fp32 evaluation.  For those pixels the
    oracle evaluates every admissible set of decisions (`recheck`: oracle_c.Recheck / oracle_np.pixel_variants) and the
    pixel must match ONE of the resulting colours within the same tol.  `margin`/`img`/`ref` may be a band of the frame
    starting at pixel row y0 (recheck takes frame coordinates).  Returns the worst error; logs the flagged fraction."""
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    assert np.isfinite(img).all(), f"{what}: {int((~np.isfinite(img)).sum())} non-finite output value(s)"    # (a NaN compares False both ways below)
    err = np.abs(img - ref).max(axis=-1)
    safe = np.asarray(margin) >= MARGIN
    worst = float(err[safe].max()) if safe.any() else 0.0
    assert worst < tol, f"{what}: max |d| = {worst:.3e} on pixels with no decision near a threshold (tol {tol})"
    n_flag = int((~safe).sum()); worst2 = 0.0
    if n_flag:
        # pixels inside the margin that agree with the nominal evaluation anyway need no second look
        ys, xs = np.nonzero(~safe & ~(err < tol))
        if len(ys):
            assert recheck is not None, f"{what}: {len(ys)} threshold-sensitive pixels differ and no two-sided oracle was supplied"
            best, leaves, capped = recheck(ys + y0, xs, img[ys, xs])
            bad = ~(best < tol)
            assert not bad.any(), (f"{what}: {int(bad.sum())} threshold-sensitive pixel(s) match NO admissible evaluation: "
                                   f"worst {np.nanmax(best[bad]):.3e} at (y,x)=({ys[bad][0] + y0},{xs[bad][0]}), "
                                   f"{int(leaves[bad][0])} variants, capped={bool(capped[bad][0])}")
            worst2 = float(best.max())
        worst2 = max(worst2, float(err[~safe & (err < tol)].max(initial=0.0)))
    PARITY_LOG.append((what, int(safe.size), n_flag, worst, worst2))
    print(f"[parity] {what}: {safe.size} px, {n_flag} ({n_flag / max(1, safe.size):.3%}) within {MARGIN:g} of a threshold, "
          f"max|d| {worst:.2e} off-threshold, {worst2:.2e} two-sided")
    return max(worst, worst2)


def pytest_terminal_summary(terminalreporter):
    """The parity ledger: how many pixels of each oracle-checked frame sat near a threshold and how far off the worst was."""
    if not PARITY_LOG:
        return
    px = sum(r[1] for r in PARITY_LOG); fl = sum(r[2] for r in PARITY_LOG)
    terminalreporter.write_line(f"[parity] {len(PARITY_LOG)} oracle-checked frames, {px} pixels, {fl} ({fl / max(1, px):.4%}) "
                                f"within {MARGIN:g} of a threshold (checked two-sidedly, same tol {TOL:g}); "
                                f"worst |d| off-threshold {max(r[3] for r in PARITY_LOG):.2e}, two-sided {max(r[4] for r in PARITY_LOG):.2e}")


@pytest.fixture(scope="session")
def oracle_c_mod():
    import oracle_c
    oracle_c.build()
    return oracle_c
