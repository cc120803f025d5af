#!/bin/bash
# per-kernel durations, one frame at a time:  scripts/gpu_r03_trace.sh <tag> [bench args...]
TAG=${1:-r03t}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/raw -o trace -- python $ROOT/bench.py --no-pipeline --no-cpu-baseline --no-lowres --no-events --steps 20 --warmup 5 "$@" > $OUT/bench.log 2>&1
db=$(find $OUT/raw -name "*.db" | head -1)
python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_alone.csv
python $ROOT/scripts/rocpd_timeline.py $db > $OUT/timeline_alone.txt 2>/dev/null
rm -rf $OUT/raw
cut -c1-110 $OUT/kernel_stats_alone.csv | head -16
