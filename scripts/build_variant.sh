#!/bin/bash
# scripts/build_variant.sh <git-rev|WORK> <name> [extra hipcc flags...]
#   compiles the library from the sources of a git revision (or of the working tree) into
#   build/variants/<name>.so, for A/B runs in ONE GPU-box visit (box-to-box variance is +-1.5 %):
#   SAGE_GS_LIB=build/variants/<name>.so python scripts/r02_probe.py quick
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants; mkdir -p $OUT
TMP=$(mktemp -d)
mkdir -p $TMP/pkg/csrc $TMP/include
if [ "$REV" = WORK ]; then
  cp $ROOT/sage-3d_official_amd/csrc/* $TMP/pkg/csrc/; cp $ROOT/include/sage_gs.h $TMP/include/
else
  for f in sgs_api.hip sgs_kernels.h sgs_common.h; do git -C $ROOT show $REV:sage-3d_official_amd/csrc/$f > $TMP/pkg/csrc/$f; done
  git -C $ROOT show $REV:include/sage_gs.h > $TMP/include/sage_gs.h
fi
(cd $TMP/pkg && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize "$@" -shared csrc/sgs_api.hip -o $OUT/$NAME.so)
rm -rf $TMP; ls -la $OUT/$NAME.so
