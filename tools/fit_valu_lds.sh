#!/bin/bash
# usage (GPU box, repo root): bash tools/fit_valu_lds.sh <tag>
# tools/fit_valu.sh + tools/fit_valu_classes.sh for k_render_lds (RR_DEBUG_KERNEL=lds) on the meshes whose nodes fit LDS:
# trip counters, vector instructions in total and by class.  python3 tools/fit_valu.py gpurun_out/fit_<tag> fits them.
tag=$1
out=gpurun_out/fit_$tag
mkdir -p $out
export TMPDIR=/tmp RR_DEBUG_KERNEL=lds
: > $out/pmc.txt; : > $out/stats.jsonl; : > $out/pmc_classes.txt
while read -r mesh refr refl; do
  [ -z "$mesh" ] && continue
  PROF_STATS=1 python3 tools/prof_target.py $mesh $refr 16 4 1920 1080 $refl >> $out/stats.jsonl 2>> $out/err.log
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/raw -- python3 tools/prof_target.py $mesh $refr 16 4 1920 1080 $refl > $out/run.log 2>&1
  f=$(find $out/raw -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$mesh $refr/$refl" >> $out/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); kern = set()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if "k_render" not in k: continue
        kern.add(k.split("(")[0][:60]); acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(sys.argv[2], "|", " ".join("%s=%.0f" % kv for kv in sorted(acc.items())), "|", ";".join(sorted(kern)))
PY
  rm -rf $out/raw
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
  echo "done $mesh $refr/$refl" >> $out/progress.txt
done <<'LIST'
monkey.obj 8 2
monkey.obj 0 0
monkey.obj 2 2
sphere.obj 4 2
shell.obj 5 2
cube.obj 8 2
sphere.obj 0 0
sphere.obj 8 2
LIST
cat $out/stats.jsonl $out/pmc.txt
