#!/bin/bash
# PMC pass: instruction-cache behaviour of the kernels (one frame at a time)
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT/pmc
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL" | head -20 > $OUT/avail.txt
cat $OUT/avail.txt | cut -c1-200
pmc() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/raw_$name -o p -- python $ROOT/bench.py --no-cpu-baseline --no-events --no-pipeline --steps 30 > $OUT/$name.log 2>&1
  find $OUT/raw_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc/$name.csv \;
  rm -rf $OUT/raw_$name; tail -2 $OUT/$name.log | cut -c1-200; }
pmc ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES
python $ROOT/scripts/pmc_summary.py $OUT/pmc 10 > $OUT/summary.json 2>/dev/null
python - <<PY
import json
try:
    d = json.load(open("$OUT/summary.json"))
    for k, v in d.items():
        if "sgs::" in k:
            print(k.split("(")[0], {a: round(b / 1e6, 3) for a, b in v.items() if a != "_launches"})
except Exception as e:
    print("no summary:", e)
PY
