#!/bin/bash
# usage (GPU box, repo root): bash tools/fit_valu_classes.sh <tag>   (after tools/fit_valu.sh <tag>: same workloads)
# One more PMC pass per workload: vector instructions by class (f32 add / mul / fma, transcendental, conversions, integer).
tag=$1
out=gpurun_out/fit_$tag
mkdir -p $out
export TMPDIR=/tmp
: > $out/pmc_classes.txt
while read -r mesh refr refl; do
  [ -z "$mesh" ] && continue
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d $out/raw -- python3 tools/prof_target.py $mesh $refr 16 4 1920 1080 $refl > $out/run.log 2>&1
  f=$(find $out/raw -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$mesh $refr/$refl" >> $out/pmc_classes.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if "k_render" not in row.get("Kernel_Name", ""): continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(sys.argv[2], "|", " ".join("%s=%.0f" % kv for kv in sorted(acc.items())))
PY
  rm -rf $out/raw
done <<'LIST'
monkey.obj 8 2
monkey.obj 0 0
monkey.obj 2 2
sphere.obj 4 2
shell.obj 5 2
cube.obj 8 2
sphere.obj 0 0
ott.obj 8 2
LIST
cat $out/pmc_classes.txt
