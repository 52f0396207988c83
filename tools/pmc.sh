#!/bin/bash
# usage: bash tools/pmc.sh <tag> "<counters>" <kernel-substring> <python script + args...>
# one PMC pass; prints per-launch averages for kernels whose name contains <kernel-substring>
tag=$1; shift
ctrs=$1; shift
kern=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw -- python3 "$@" > $out/run.log 2>&1
echo "rc=$?" >> $out/run.log
f=$(find $out/raw -name '*counter_collection.csv' | head -1)
python3 - "$f" "$kern" > $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if sys.argv[2] not in k: continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-32s avg_per_launch %16.1f  launches %d" % (k, s / max(n, 1), n))
PY
rm -rf $out/raw
cat $out/summary.txt
tail -3 $out/run.log
