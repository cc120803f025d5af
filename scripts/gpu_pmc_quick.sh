#!/bin/bash
# quick PMC look at the frame's kernels (one frame at a time, 20 poses):  scripts/gpu_pmc_quick.sh <tag> COUNTER [COUNTER...]
#   one rocprofv3 --pmc pass per counter argument (quote several names to put them into one pass)
export TMPDIR=/tmp
TAG=${1:-pq}; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT/csv
cd /tmp
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/raw_$i -o p -- python $ROOT/bench.py --no-cpu-baseline --no-events --no-pipeline --steps 20 --warmup 2 > $OUT/pass_$i.log 2>&1
  find $OUT/raw_$i -name "*counter_collection.csv" -exec cp {} $OUT/csv/p$i.csv \;
  rm -rf $OUT/raw_$i
done
python $ROOT/scripts/pmc_summary.py $OUT/csv 4 > $OUT/pmc_summary.json
python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, v in d.items():
    if "sgs::" in k and "layout" not in k and "bounds" not in k:
        print(k, {c: round(x) for c, x in v.items()})
PY
