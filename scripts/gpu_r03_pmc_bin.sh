#!/bin/bash
# round-3 PMC look at the binning kernels on the trained-like scene (one frame at a time)
export TMPDIR=/tmp
TAG=${1:-r03pmcb}; KIND=${2:-trained}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT/csv
cd /tmp
i=0
for c in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/raw_$i -o p -- python $ROOT/bench.py --scene-kind $KIND --no-cpu-baseline --no-events --no-pipeline --no-lowres --steps 12 --warmup 2 > $OUT/pass_$i.log 2>&1
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
