#!/bin/bash
# round 4, visit b: the software pipeline's knobs (binning priority, depth) + k_fused alone (serial trace)
export TMPDIR=/tmp
TAG=${1:-r04b}
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
REPS=4 timeout 900 python scripts/r04_sweep.py "SGS_FUSE=0" "SGS_FUSE=1" "SGS_FUSE=1 SGS_FUSE_B_PRIO=1" "SGS_FUSE=1 SGS_FUSE_DEPTH=3" "SGS_FUSE=1 SGS_FUSE_DEPTH=3 SGS_FUSE_B_PRIO=1" "SGS_FUSE=1 SGS_FUSE_SERIAL=1" "SGS_FUSE=0" 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
cd /tmp
trace() { # name env... -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-lowres "$@" > $OUT/$name.log 2>&1
  local db=$(find $OUT/raw_$name -name "*.db" | head -1)
  python $ROOT/scripts/rocpd_timeline.py $db 0.03 0.26 > $OUT/timeline_$name.txt 2>/dev/null
  rm -rf $OUT/raw_$name
  echo "== $name"; cat $OUT/timeline_$name.txt
}
trace serial SGS_FUSE=1 SGS_FUSE_SERIAL=1 -- --steps 100 --warmup 10
trace d3prio SGS_FUSE=1 SGS_FUSE_DEPTH=3 SGS_FUSE_B_PRIO=1 -- --steps 100 --warmup 10
trace unfused SGS_FUSE=0 -- --steps 100 --warmup 10
