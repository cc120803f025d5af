#!/bin/bash
# round 4, first GPU visit: parity suite on the software pipeline, then A/B of the sweep under SGS_FUSE / LDS padding
export TMPDIR=/tmp
TAG=${1:-r04a}
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
(time timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 900 python scripts/r04_sweep.py "SGS_FUSE=0" "SGS_FUSE=1" "SGS_FUSE=1 SGS_FUSE_LDS_PAD=8192" "SGS_FUSE=1 SGS_FUSE_LDS_PAD=16384" "SGS_FUSE=0" "SGS_FUSE=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
RES=640x480,1024x768 REPS=3 timeout 600 python scripts/r04_sweep.py "SGS_FUSE=0" "SGS_FUSE=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_lowres.log
cd /tmp
(rocprofv3 -L 2>/dev/null | grep -o -E "\bSPI_[A-Z0-9_]+" | sort -u | tr '\n' ' ') > $OUT/spi_counters.txt
echo "== kernel trace of the fused sweep"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_fused -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-lowres --steps 100 --warmup 10 > $OUT/trace_fused.log 2>&1
db=$(find $OUT/raw_fused -name "*.db" | head -1)
python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_fused.csv
python $ROOT/scripts/rocpd_timeline.py $db 0.03 0.26 > $OUT/timeline_fused.txt 2>/dev/null
rm -rf $OUT/raw_fused
tail -c 600 $OUT/trace_fused.log
echo "== bench (driver command)"
timeout 600 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; tail -c 1500 $OUT/bench_k20.json
