#!/bin/bash
# two PMC passes (instruction mix + busy cycles) over the default frames, one frame at a time
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT/pmc
cd /tmp
pmc() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/raw_$name -o p -- python $ROOT/bench.py --no-cpu-baseline --no-events --no-pipeline --steps ${STEPS:-40} > $OUT/$name.log 2>&1
  find $OUT/raw_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc/$name.csv \;
  rm -rf $OUT/raw_$name; }
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc grbm GRBM_GUI_ACTIVE
python $ROOT/scripts/pmc_summary.py $OUT/pmc 10 > $OUT/summary.json
python - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in d.items():
    if "sgs::" in k and "layout" not in k:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        print(f"{k.split('(')[0]:34s} {cyc/2.4e3:7.1f} us  VALU insts {v['SQ_INSTS_VALU']/1e6:6.1f}M  VALU busy {100*v['SQ_ACTIVE_INST_VALU']*4/1024/cyc:5.1f} %  "
              f"SALU {v['SQ_INSTS_SALU']/1e6:5.1f}M  LDS {v['SQ_INSTS_LDS']/1e6:5.1f}M  waves/SIMD {v['SQ_WAVE_CYCLES']*4/1024/cyc:4.1f}  VALU-active cycles/SIMD {v['SQ_ACTIVE_INST_VALU']*4/1024/1e3:6.0f}k")
PY
