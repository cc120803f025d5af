#!/bin/bash
# scripts/gpu_r04_ab.sh <variant>...   fixed path (k=1 scene, VALU/SALU per wave of k_tile_render) and the driver's bench line per library variant, separate processes
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r04ab; mkdir -p $OUT
for v in "$@"; do
  cd /tmp
  SAGE_GS_LIB=$ROOT/build/variants/$v.so timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/x -o p -- python $ROOT/scripts/r04_fixed_path.py 1 > /dev/null 2>&1
  f=$(find $OUT/x -name "*counter_collection.csv" | head -1)
  python - $f $v <<PY | tee -a $OUT/ab.txt
import csv,sys,collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_tile_render" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w=sum(acc["SQ_WAVES"][-3:])/3
print(sys.argv[2], "fixed path per wave:", {k: round(sum(v[-3:])/3/w,1) for k,v in acc.items() if k!="SQ_WAVES"})
PY
  rm -rf $OUT/x; cd $ROOT
  for rep in 1 2; do SAGE_GS_LIB=$ROOT/build/variants/$v.so timeout 120 python bench.py --steps 20 --warmup 5 --no-upload-probe --no-lowres --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4), 'render alone us', round(1e3*d['roofline']['avg_launch_ms'],1))" | tee -a $OUT/ab.txt; done
done
