#!/bin/bash
# Frame-group size x group streams of sgs_render_batch on the driver's command (separate processes: stream creation pollutes HW queues)
export TMPDIR=/tmp
OUT=gpurun_out/r04g; mkdir -p $OUT
for cfg in "4 2" "2 4" "3 2" "2 3" "2 2" "1 3" "1 4" "1 6" "8 1" "5 1" "4 2"; do
  set -- $cfg
  for rep in 1 2; do
    SGS_GROUP=$1 SGS_GROUP_LANES=$2 timeout 100 python bench.py --steps 20 --warmup 5 --no-upload-probe --no-lowres --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group $1 x streams $2:', round(d['value'],1), round(d['ms_per_step'],4), 'v100', round(d['value_100']['value'],1))" | tee -a $OUT/groups.txt
  done
done
