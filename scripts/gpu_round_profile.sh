#!/bin/bash
# Everything profiles/ holds for one round, from ONE GPU-box visit:  scripts/gpu_round_profile.sh <tag>
#   bench JSON (default command), rocprofv3 kernel trace of the same command (+ one frame at a time), PMC passes
#   (counters in their own runs, --kernel-trace only) over the SAME frames, traffic.json.
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
echo "== bench"; timeout 600 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
trace() { # name args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o trace -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1
  local db=$(find $OUT/raw_$name -name "*.db" | head -1)
  python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_$name.csv
  python $ROOT/scripts/rocpd_timeline.py $db 0.05 0.52 > $OUT/timeline_$name.txt
  rm -rf $OUT/raw_$name
}
echo "== kernel trace (default command: frames in flight)"; trace pipelined
echo "== kernel trace (one frame at a time)"; trace alone --no-pipeline
pmc() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python $ROOT/bench.py --no-cpu-baseline --no-events --no-pipeline > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
  rm -rf $OUT/pmc_$name
}
echo "== PMC passes"
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc sq2 SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc sq3 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
mkdir -p $OUT/pmc; mv $OUT/pmc_*.csv $OUT/pmc/
python $ROOT/scripts/pmc_summary.py $OUT/pmc 10 > $OUT/pmc_summary.json
python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
stage = {"preprocess": ["sgs::k_preprocess"], "count": ["sgs::k_bin_count", "sgs::k_tile_scan"], "emit": ["sgs::k_bin_emit"],
         "render": ["sgs::k_tile_render<false>"]}
out = {}
for s, ks in stage.items():
    out[s] = sum((2.0 * d[k]["FETCH_SIZE"] + d[k]["WRITE_SIZE"]) * 1024.0 for k in ks if k in d)
# share of the kernel's cycles in which a SIMD's VALU was executing an instruction (SQ_ACTIVE_INST_VALU counts
# quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
out["_valu_busy"] = {s: max((d[k]["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (d[k]["GRBM_GUI_ACTIVE"] / 8.0) for k in ks if k in d)
                     for s, ks in stage.items()}
out["_lds_bank_conflict_share"] = {s: max(d[k]["SQ_LDS_BANK_CONFLICT"] / max(1.0, d[k]["SQ_LDS_IDX_ACTIVE"]) for k in ks if k in d)
                                   for s, ks in stage.items()}
out["_valu_insts"] = {s: sum(d[k]["SQ_INSTS_VALU"] for k in ks if k in d) for s, ks in stage.items()}   # wave instructions per launch
out["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc (separate passes, --kernel-trace only), "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM and this repo's own calibration (profiles/r01_hbm_calib_*.csv: "
                "2 GiB streamed reads report 1 GiB at 16 and at 4 B/lane; streamed writes are exact); averaged over the launches of "
                "the default bench command's frames (warm-up launches skipped), one frame at a time")
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
ls $OUT
