#!/bin/bash
# usage (GPU box): bash tools/exp_variants.sh "<defines A>" "<defines B>" ...   -- rebuilds rr_render with each set of
# defines and times monkey/sphere/ott at Depth 16 and 64 (tools/exp_batch.py).  "" = the product build.
for d in "$@"; do
  rm -f refraction_raytracing_dxr_amd/build/rr_render.hip.o
  RR_EXTRA_DEFINES="$d" python refraction_raytracing_dxr_amd/_build.py > /dev/null 2>&1 || { echo "build failed: $d"; continue; }
  echo "=== variant: '$d'"
  timeout -k 10 200 python tools/exp_batch.py monkey.obj sphere.obj ott.obj 2>&1 | grep -v amdgpu.ids
done
rm -f refraction_raytracing_dxr_amd/build/rr_render.hip.o
