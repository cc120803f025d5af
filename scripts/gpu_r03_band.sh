#!/bin/bash
# kernel stats + timeline of one rank's band (frame groups): scripts/gpu_r03_band.sh <tag> <row0> <row1>
export TMPDIR=/tmp
TAG=${1:-r03band}; OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
python scripts/r03_band_prof.py $2 $3 32 2>&1 | grep -v amdgpu.ids | tee $OUT/plain.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/raw -o trace -- python $GRAFT_REPO_ROOT/scripts/r03_band_prof.py $2 $3 32 > $OUT/traced.txt 2>&1
db=$(find $OUT/raw -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $db | tee $OUT/timeline_band.txt
python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_band.csv
rm -rf $OUT/raw
