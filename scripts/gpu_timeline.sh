#!/bin/bash
# rocprofv3 kernel trace of a short bench run + concurrency summary
TAG=${1:-tl}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/raw -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps ${BENCH_STEPS:-60} --warmup 10 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
db=$(find $OUT/raw -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $db | tee $OUT/timeline.txt
python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats.csv
rm -rf $OUT/raw
