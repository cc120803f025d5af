#!/bin/bash
# rocprofv3 kernel trace of a bench command -> per-kernel stats + concurrency timeline of the timed region
#   scripts/gpu_trace.sh <tag> [bench args...]
export TMPDIR=/tmp
TAG=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw -o trace -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
db=$(find $OUT/raw -name "*.db" | head -1)
python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats.csv
python $ROOT/scripts/rocpd_timeline.py $db 0.04 0.34 > $OUT/timeline.txt
rm -rf $OUT/raw
grep '^{' $OUT/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), {k:round(v['ms_alone']*1e3,1) for k,v in d['roofline']['stages'].items()})"
cat $OUT/timeline.txt
