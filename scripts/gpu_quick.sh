#!/bin/bash
# quick GPU visit:  scripts/gpu_quick.sh <tag> [probe args...]   -> gpurun_out/<tag>/
export TMPDIR=/tmp
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
(time timeout 900 python -m pytest tests -m gpu -x -q --durations=5) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 900 python scripts/r02_probe.py "$@" > $OUT/probe.log 2>&1; cat $OUT/probe.log
