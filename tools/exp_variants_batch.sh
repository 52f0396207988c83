#!/bin/bash
# usage (GPU box): bash tools/exp_variants_batch.sh "<defines A>" ...: frames-per-launch sweep (tools/exp_batch.py) per build variant
for d in "$@"; do
  export RR_EXTRA_DEFINES="$d"
  python refraction_raytracing_dxr_amd/_build.py > /dev/null 2>&1 || { echo "build failed: $d"; continue; }
  echo "=== variant: '$d'"
  timeout -k 10 200 python tools/exp_batch_clean.py monkey.obj sphere.obj ott.obj shell.obj 2>&1 | grep -v amdgpu.ids
done
