#!/bin/bash
# bench.py one-shot behaviour: the driver's command with and without the pre-heat, twice each
export TMPDIR=/tmp
TAG=${1:-r04f}; OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  for ph in 0 60 200; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lowres --preheat-ms $ph > $OUT/k20_ph${ph}_$i.json 2>$OUT/err.txt
    python - <<PY
import json; d=json.load(open("$OUT/k20_ph${ph}_$i.json")); print("preheat $ph run $i: value %.0f ms_per_step %.4f value_100 %.0f preheat_steps %d" % (d["value"], d["ms_per_step"], d["value_100"]["value"], d["preheat_steps"]))
PY
  done
done
