#!/bin/bash
# Everything profiles/ holds for one round, from ONE GPU-box visit:  scripts/gpu_round_profile.sh <tag>
#   bench JSON (default command and the driver's --steps 20 --warmup 5), rocprofv3 kernel traces of the same command
#   (frames in flight, and one frame at a time), PMC passes (counters in their own runs, --kernel-trace only) over the SAME
#   pose sets, traffic.json keyed by pose set (bench.py quotes PMC figures only for the pose set they were taken on),
#   the 4K sweep of BASELINE config 5.
TAG=${1:-r05}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
echo "== bench"; timeout 600 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json
timeout 600 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; tail -c 300 $OUT/bench_k20.json
echo "== N = 2 on this one GPU under gloo (the N > 1 code path incl. verify; not a scaling number)"; SGS_BENCH_SHARE_GPU=1 timeout 400 python $ROOT/bench.py --gpus 2 --steps 20 --warmup 5 --no-secondary > $OUT/bench_n2_shared_gpu.json 2> $OUT/bench_n2_shared_gpu.err; grep verify $OUT/bench_n2_shared_gpu.err
echo "== N = 8 on this one GPU under gloo: configs[3] and configs[4] (the N > 1 code path at the driver's world size incl. verify and both gatherv shapes; not a scaling number)"
SGS_BENCH_SHARE_GPU=1 timeout 900 python $ROOT/bench.py --gpus 8 --steps 20 --warmup 5 --no-secondary > $OUT/bench_n8_shared_gpu.json 2> $OUT/bench_n8_shared_gpu.err; grep verify $OUT/bench_n8_shared_gpu.err
SGS_BENCH_SHARE_GPU=1 timeout 900 python $ROOT/bench.py --gpus 8 --config 5 --steps 16 --warmup 4 --no-secondary > $OUT/bench_n8_shared_gpu_config5.json 2> $OUT/bench_n8_shared_gpu_config5.err; grep verify $OUT/bench_n8_shared_gpu_config5.err
echo "== fp32 / compressed scene, the reference's resolutions (stage times alone)"; (cd $ROOT && timeout 400 python scripts/r05_probe.py fp32 packed lowres n=20 2>&1 | grep -v amdgpu.ids > $OUT/probe_fp32_packed_lowres.txt; cat $OUT/probe_fp32_packed_lowres.txt)
echo "== config 5 (3840x2160, 360-camera sweep)"; timeout 600 python $ROOT/bench.py --config 5 --no-cpu-baseline --no-lowres --no-trained --no-verify --no-upload-probe > $OUT/bench_config5.json 2> $OUT/bench_config5.err; tail -c 300 $OUT/bench_config5.json
trace() { # name args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-lowres --no-trained --no-verify --no-upload-probe --preheat-ms 0 "$@" > $OUT/$name.log 2>&1
  local db=$(find $OUT/raw_$name -name "*.db" | head -1)
  python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_$name.csv
  python $ROOT/scripts/rocpd_timeline.py $db ${WIN:-0.04 0.34} > $OUT/timeline_$name.txt 2>/dev/null
  rm -rf $OUT/raw_$name
}
echo "== kernel trace (default command: frames in flight)"; WIN="0.03 0.26" trace pipelined   # (warm-up 10 + 100 timed frames of ~410 frames launched)
echo "== kernel trace (one frame at a time)"; trace alone --no-pipeline
echo "== kernel trace (driver's command, one frame at a time)"; trace alone_k20 --no-pipeline --steps 20 --warmup 5
echo "== kernel trace (config 5, one frame at a time)"; trace config5_alone --config 5 --no-pipeline --steps 40
pmc() { # set name flags... -- counters...
  local set=$1 name=$2; shift 2
  local flags=()
  while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${set}_$name -o p -- python $ROOT/bench.py --no-cpu-baseline --no-lowres --no-trained --no-verify --no-events --no-pipeline --no-upload-probe --preheat-ms 0 "${flags[@]}" > $OUT/pmc_${set}_$name.log 2>&1
  mkdir -p $OUT/pmc_$set
  find $OUT/pmc_${set}_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$set/$name.csv \;
  rm -rf $OUT/pmc_${set}_$name
}
passes() { # set skip flags...
  local set=$1 skip=$2; shift 2
  pmc $set sq1 "$@" -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  pmc $set sq2 "$@" -- SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  pmc $set fetch "$@" -- FETCH_SIZE
  pmc $set write "$@" -- WRITE_SIZE
  pmc $set grbm "$@" -- GRBM_GUI_ACTIVE
  python $ROOT/scripts/pmc_summary.py $OUT/pmc_$set $skip > $OUT/pmc_summary_$set.json
}
echo "== what a kernel waits for when it wants to start workgroups (VERDICT r3 item 1: SPI resource-allocation stalls, SQ wait cycles) — the default command, frames in flight"
stall() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/stall_$name -o p -- python $ROOT/bench.py --no-cpu-baseline --no-lowres --no-trained --no-verify --no-events --no-upload-probe --preheat-ms 0 > $OUT/stall_$name.log 2>&1
  mkdir -p $OUT/stalls
  find $OUT/stall_$name -name "*counter_collection.csv" -exec cp {} $OUT/stalls/$name.csv \;
  rm -rf $OUT/stall_$name
}
stall sq SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
stall spi1 SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN
stall spi2 SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN
python $ROOT/scripts/pmc_summary.py $OUT/stalls 10 > $OUT/pipeline_stalls.json 2>/dev/null; head -c 1500 $OUT/pipeline_stalls.json
echo "== lane use of the composite (profiling build, the driver's first poses)"
(cd $ROOT && make -s -C sage-3d_official_amd prof >/dev/null 2>&1; POSES=$(python -c "print(','.join(str((i*77)%256) for i in range(5,25)))") timeout 300 python scripts/tile_prof.py > $OUT/tile_prof_k20.txt 2>/dev/null; grep TOTAL $OUT/tile_prof_k20.txt | cut -c1-300)
echo "== trained-like scene, bench line"; timeout 600 python $ROOT/bench.py --scene-kind trained --steps 20 --warmup 5 --no-cpu-baseline --no-upload-probe > $OUT/bench_trained_k20.json 2> $OUT/bench_trained.err; tail -c 200 $OUT/bench_trained_k20.json
echo "== PMC passes (default pose set)"; passes default 10
echo "== PMC passes (driver's pose set)"; passes k20 5 --steps 20 --warmup 5
python - <<PY
import json, os
out = {}
for set_, log in (("default", "pmc_default_sq1.log"), ("k20", "pmc_k20_sq1.log")):
    d = json.load(open("$OUT/pmc_summary_%s.json" % set_))
    # the pose set the passes ran on, as bench.py itself names it (the JSON line of the profiled run)
    line = [l for l in open("$OUT/" + log) if l.startswith("{")][-1]
    tag = json.loads(line)["config"]["pose_set"]
    # (two-level binning: stage "count" = level 1, splats -> super-tile queues; stage "emit" = level 2, super-tile queues -> tile queues)
    stage = {"preprocess": ["sgs::k_chunk_cull", "sgs::k_preprocess<false, false>", "sgs::k_preprocess<true, false>", "sgs::k_preprocess<false, true>", "sgs::k_preprocess<true, true>"], "count": ["sgs::k_bin_count", "sgs::k_stile_scan", "sgs::k_bin_emit"],
             "emit": ["sgs::k_expand<false>", "sgs::k_tile_scan", "sgs::k_expand<true>"],
             # the instantiation a sweep runs (no aux output, no D_f bookkeeping), however the profiler spells it
             "render": ["sgs::k_tile_render<false, false, false>"]}
    t = {}
    for s, ks in stage.items():
        t[s] = sum((2.0 * d[k]["FETCH_SIZE"] + d[k]["WRITE_SIZE"]) * 1024.0 for k in ks if k in d)
    # share of the kernel's cycles in which a SIMD's VALU was executing an instruction (SQ_ACTIVE_INST_VALU counts
    # quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    t["_valu_busy"] = {s: max((d[k]["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (d[k]["GRBM_GUI_ACTIVE"] / 8.0) for k in ks if k in d) for s, ks in stage.items()}
    t["_lds_bank_conflict_share"] = {s: max(d[k]["SQ_LDS_BANK_CONFLICT"] / max(1.0, d[k]["SQ_LDS_IDX_ACTIVE"]) for k in ks if k in d) for s, ks in stage.items()}
    t["_valu_insts"] = {s: sum(d[k]["SQ_INSTS_VALU"] for k in ks if k in d) for s, ks in stage.items()}   # wave instructions per launch
    t["_launches_averaged"] = {k: d[k].get("_launches") for ks in stage.values() for k in ks if k in d}
    if set_ == "k20":
        try:
            tl = [l for l in open("$OUT/tile_prof_k20.txt") if l.startswith("TOTAL ")][-1]
            t["_lane_use"] = json.loads(tl[6:])
        except Exception:
            pass
    out[tag] = t
out["_note"] = ("per pose set (bench.py config.pose_set): HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc (separate passes, "
                "--kernel-trace only), FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM and this repo's own calibration (profiles/r01_hbm_calib_*.csv: "
                "2 GiB streamed reads report 1 GiB at 16 and at 4 B/lane; streamed writes are exact); averaged over the launches of the bench "
                "command's frames (warm-up launches skipped), one frame at a time")
import subprocess, sys
sys.path.insert(0, "$ROOT")
import bench
out["_kernel_sha"] = bench.kernel_sha()           # bench.py quotes these figures only while csrc/ hashes to this
try:
    out["_commit"] = subprocess.check_output(["git", "-C", "$ROOT", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["_commit"] = os.environ.get("SGS_COMMIT")  # (the GPU box has no .git: the caller passes the commit)
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps({k: (v if k.startswith("_") else {s: v[s] for s in ("preprocess", "count", "emit", "render")}) for k, v in out.items() if k != "_note"}, indent=1))
PY
ls $OUT
