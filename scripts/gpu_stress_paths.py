#!/usr/bin/env python3
"""Many iterations of tests/gpu_stress.py (every way of issuing frames must give the same bits):
    python scripts/gpu_stress_paths.py [iterations] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sage_gs import Renderer
from gpu_stress import stress_issue_paths
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = stress_issue_paths(Renderer("cuda:0"), iters, int(sys.argv[2]) if len(sys.argv) > 2 else 0, verbose=True)
print(f"{iters} iterations, {bad} mismatches")
sys.exit(1 if bad else 0)
