#!/bin/bash
# round 4: A/B of sweep variants (args: tag, then variant strings) + optional traces (TRACE="name|ENV ENV|bench args;...")
export TMPDIR=/tmp
TAG=${1:-r04c}; shift
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
[ -n "$TESTS" ] && { (time timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log; }
[ $# -gt 0 ] && timeout 1200 python scripts/r04_sweep.py "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
cd /tmp
IFS=';' read -ra TR <<< "$TRACE"
for t in "${TR[@]}"; do
  IFS='|' read -r name envs args <<< "$t"
  env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-lowres $args > $OUT/$name.log 2>&1
  db=$(find $OUT/raw_$name -name "*.db" | head -1)
  python $ROOT/scripts/rocpd_timeline.py $db ${WIN:-0.03 0.26} > $OUT/timeline_$name.txt 2>/dev/null
  python $ROOT/scripts/rocpd_stats.py $db > $OUT/kernel_stats_$name.csv 2>/dev/null
  rm -rf $OUT/raw_$name
  echo "== $name"; cat $OUT/timeline_$name.txt
done
