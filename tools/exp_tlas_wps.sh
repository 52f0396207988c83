#!/bin/bash
# the lock-step two-level kernel by the waves per SIMD its two builds are compiled for (run on the GPU box: rebuilds per variant)
for d in "-DRR_TLAS30_WPS=7 -DRR_TLAS39_WPS=5" "-DRR_TLAS30_WPS=6 -DRR_TLAS39_WPS=6" "-DRR_TLAS30_WPS=8 -DRR_TLAS39_WPS=4" "-DRR_TLAS30_WPS=5 -DRR_TLAS39_WPS=7"; do
  export RR_EXTRA_DEFINES="$d"
  echo "=== $d"
  RR_DEBUG_KERNEL=fused timeout -k 10 500 python tools/exp_tlas.py both 16 2>&1 | tail -2 | cut -c1-150
done
