#!/bin/bash
# k_tile_render's fixed path, cut by cut (build/variants/cutN.so: -DSGS_CUT=N returns after stage N): SQ_INSTS_VALU / SALU per wave
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r04cut; mkdir -p $OUT; cd /tmp
for k in 1 64; do for v in cut1 cut2 cut3 cut4 pf0; do
  SAGE_GS_LIB=$ROOT/build/variants/$v.so timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/x -o p -- python $ROOT/scripts/r04_fixed_path.py $k > /dev/null 2>&1
  f=$(find $OUT/x -name "*counter_collection.csv" | head -1)
  python - $f $v $k <<PY | tee -a $OUT/cuts.txt
import csv,sys,collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_tile_render" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w=sum(acc["SQ_WAVES"][-3:])/3
print(sys.argv[2], "k="+sys.argv[3], {k: round(sum(v[-3:])/3/w,1) for k,v in acc.items() if k!="SQ_WAVES"}, "waves", round(w))
PY
  rm -rf $OUT/x
done; done
