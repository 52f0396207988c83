#!/bin/bash
# usage (GPU box, repo root): bash tools/prof.sh <tag> [bench args]
# rocprofv3 kernel trace + stats of the bench command, then two PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command.
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-subdiv "$@" > $out/bench.log 2>&1
echo "rc=$?" >> $out/bench.log
find $out/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE VALUBusy VALUUtilization"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -- python3 bench.py --no-cpu-baseline --no-subdiv "$@" > $out/pmc_$n.log 2>&1
  f=$(find $out/pmc_$n -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $out/pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if "k_render_fused<" not in k or "true, false>" in k.replace("false, false", ""): pass
        if "k_render_fused" not in k: continue
        key = (k.split("(")[0][-60:], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-62s %-28s avg_per_launch %18.1f  launches %d" % (k, c, s / max(n, 1), n))
PY
  rm -rf $out/pmc_$n
done
tail -2 $out/bench.log | head -1
head -8 $out/kernel_stats.csv
cat $out/pmc_summary.txt
