#!/usr/bin/env python3
"""Seeded random frames, HIP path against the oracle (tests/parity_cases.py::case_fuzz), for many more seeds than the
suite runs:  python scripts/gpu_fuzz.py FIRST LAST [max_n] [wild] [big]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "sage-3d_official_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import conftest, parity_cases as pc
from test_gpu_parity import GpuDriver
drv = GpuDriver()
a, b = int(sys.argv[1]), int(sys.argv[2]); max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 700
wild = "wild" in sys.argv; max_res = (3900, 2200) if "uhd" in sys.argv else (2400, 1400) if "huge" in sys.argv else (900, 600) if "big" in sys.argv else (260, 160)
bad = []
for seed in range(a, b):
    try:
        pc.case_fuzz(drv, [seed], max_n, max_res, wild)
    except Exception as e:                                   # noqa: BLE001
        bad.append(seed); print("FAIL", seed, repr(e)[:500], flush=True)
print(f"seeds [{a},{b}) max_n {max_n}: {len(bad)} failures {bad}")
