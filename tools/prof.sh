#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args]
# kernel trace + stats of the bench command; summaries land in gpurun_out/prof_<tag>/
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
echo "rc=$?" >> $out/bench.log
find $out/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} sh -c 'head -1 {} > '$out'/kernel_trace_head.csv; grep k_render_fused {} | head -400 >> '$out'/kernel_trace_head.csv'
rm -rf $out/trace
tail -2 $out/bench.log
head -12 $out/kernel_stats.csv
