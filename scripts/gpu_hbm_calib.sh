#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (separate --pmc passes, like scripts/gpu_pmc.sh)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/hbm_calib
rm -rf $OUT; mkdir -p $OUT
BIN=$PWD/build/hbm_calib
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- $BIN > $OUT/$c.log 2>&1
  find $OUT/$c -name "*counter_collection.csv" -exec cp {} $OUT/$c.csv \;
  rm -rf $OUT/$c
done
tail -1 $OUT/FETCH_SIZE.log
python3 - <<PY
import csv
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for row in csv.DictReader(open("$OUT/%s.csv" % c)):
        print(c, row["Kernel_Name"].split("(")[0], float(row["Counter_Value"]))
PY
