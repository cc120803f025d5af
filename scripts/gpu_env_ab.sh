#!/bin/bash
# A/B of environment settings in ONE GPU-box visit:  scripts/gpu_env_ab.sh <tag> "ENV=.. ENV2=.." "..." ...
export TMPDIR=/tmp
TAG=${1:-e}; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
for e in "$@"; do
  echo "== $e" | tee -a $OUT/env_ab.log
  env $e timeout 600 python scripts/r02_probe.py quick 2>&1 | grep -v amdgpu.ids | tee -a $OUT/env_ab.log
done
