#!/bin/bash
# kernel durations of one frame at a time (rocprofv3 kernel trace):  scripts/gpu_scan_time.sh [bench args]
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/scan_time; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-pipeline "$@" > $OUT/log.txt 2>&1
db=$(find $OUT/raw -name "*.db" | head -1)
python $ROOT/scripts/rocpd_stats.py $db | python -c "
import csv, sys
for i, r in enumerate(csv.reader(sys.stdin)):
    if i and i < 8: print('%-28s calls %4s avg %8.1f us' % (r[0].split('(')[0][-28:], r[1], float(r[3]) / 1e3))"
rm -rf $OUT/raw
