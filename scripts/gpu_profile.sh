#!/bin/bash
# rocprofv3 kernel trace of the bench command; summary copied to gpurun_out/prof_<tag>/
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps ${BENCH_STEPS:-40} --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
tail -2 $OUT/bench_under_rocprof.log
find $OUT/raw -name "*kernel_stats*" | head
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -30 $OUT/kernel_stats.csv
# keep the merged-back payload small
find $OUT/raw -name "*kernel_trace.csv" -size +20M -delete
