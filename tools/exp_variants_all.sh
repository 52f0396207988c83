#!/bin/bash
# usage (GPU box): bash tools/exp_variants_all.sh "<defines A>" "<defines B>" ...  -- rebuilds the library with each set of defines
# (RR_EXTRA_DEFINES; "" = the product build) and times monkey / sphere / ott at Depth 16 and 64, the 16k-triangle monkey, and the
# two TLAS configurations on the lock-step kernel
for d in "$@"; do
  export RR_EXTRA_DEFINES="$d"
  python refraction_raytracing_dxr_amd/_build.py > /dev/null 2>&1 || { echo "build failed: $d"; continue; }
  echo "=== variant: '$d'"
  timeout -k 10 200 python tools/exp_batch.py monkey.obj sphere.obj ott.obj 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python tools/exp_16k.py 2>&1 | grep PLOC
  RR_DEBUG_KERNEL=fused timeout -k 10 200 python tools/exp_tlas.py both 16 2>&1 | grep "^C" | cut -c1-260
done
