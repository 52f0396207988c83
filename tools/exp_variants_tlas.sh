#!/bin/bash
# like exp_variants.sh, but times the two TLAS configurations (C4, C5) of tools/exp_configs.py
for d in "$@"; do
  rm -f refraction_raytracing_dxr_amd/build/rr_render.hip.o
  RR_EXTRA_DEFINES="$d" python refraction_raytracing_dxr_amd/_build.py > /dev/null 2>&1 || { echo "build failed: $d"; continue; }
  echo "=== variant: '$d'"
  timeout -k 10 300 python tools/exp_configs.py 2>&1 | grep "^C4\|^C5"
done
rm -f refraction_raytracing_dxr_amd/build/rr_render.hip.o
