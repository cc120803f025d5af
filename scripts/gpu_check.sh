#!/bin/bash
# One GPU-box visit: smoke -> GPU parity tests -> short bench.  Logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -15 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-40} --warmup 5 2>&1 | tail -5 | tee gpurun_out/bench.log
