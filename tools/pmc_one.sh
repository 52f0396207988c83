#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_one.sh <tag> "<counters>" <script + args...>
# one rocprofv3 --pmc pass (with --kernel-trace only); prints per-kernel-name averages of every counter
tag=$1; shift; ctrs=$1; shift
out=gpurun_out/pmc1_$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw -- python3 "$@" > $out/run.log 2>&1
f=$(find $out/raw -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee $out/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:64]
    a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("    %-34s avg/launch %16.1f  launches %d" % (c, s / n, n))
PY
rm -rf $out/raw
