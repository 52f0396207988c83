#!/bin/bash
# usage (GPU box, repo root): bash tools/prof.sh <tag> [bench args]
# rocprofv3 kernel trace + stats of the bench command, then two PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command.
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-subdiv --no-configs "$@" > $out/bench.log 2>&1
echo "rc=$?" >> $out/bench.log
find $out/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE VALUBusy VALUUtilization"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -- python3 bench.py --no-cpu-baseline --no-subdiv --no-configs --no-depth1 "$@" > $out/pmc_$n.log 2>&1
  f=$(find $out/pmc_$n -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $out/pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if "k_render_" not in k: continue
        key = (k.split("(")[0][-60:], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-62s %-28s avg_per_launch %18.1f  launches %d" % (k, c, s / max(n, 1), n))
PY
  rm -rf $out/pmc_$n
done
# HBM traffic of the timed loop's kernel per launch (MI355X_MICROARCH.md, HBM section: separate passes; FETCH_SIZE counts half of
# the bytes of 16 B/lane loads on gfx950 and is doubled; the counters report KB)
python3 - $out/pmc_summary.txt "$tag" > $out/hbm_traffic.json <<'PY'
import json, re, sys
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg_per_launch\s+([0-9.]+)\s+launches\s+(\d+)", l)
    if m and ", false, false" in m.group(1): rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
best = max(rows.items(), key=lambda kv: kv[1].get("FETCH_SIZE", (0, 0))[1]) if rows else None
if best:
    k, v = best
    f, w = v.get("FETCH_SIZE", (0, 0))[0], v.get("WRITE_SIZE", (0, 0))[0]
    print(json.dumps({"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate pass) --kernel-trace -- python3 bench.py --no-cpu-baseline --no-subdiv --no-configs --no-depth1 (tools/prof.sh %s)" % sys.argv[2],
                      "kernel": k, "frames_per_launch": 64, "launches": v.get("FETCH_SIZE", (0, 0))[1],
                      "FETCH_SIZE_KB_avg_per_launch": f, "WRITE_SIZE_KB_avg_per_launch": w,
                      "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of the bytes of 16 B/lane loads on gfx950 -> doubled; WRITE_SIZE taken as is",
                      "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024)}, indent=1))
PY
cat $out/hbm_traffic.json
tail -2 $out/bench.log | head -1
head -8 $out/kernel_stats.csv
cat $out/pmc_summary.txt
