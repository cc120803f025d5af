#!/bin/bash
# A/B of library variants in ONE GPU-box visit:  scripts/gpu_variants.sh <tag> <variant>...   (names under build/variants/)
export TMPDIR=/tmp
TAG=${1:-v}; shift
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = TESTS ]; then
    (time timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
  else
    SAGE_GS_LIB=$PWD/build/variants/$v.so timeout 600 python scripts/r02_probe.py quick 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.log
  fi
done
