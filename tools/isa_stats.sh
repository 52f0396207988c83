#!/bin/bash
# usage: bash tools/isa_stats.sh <file.hip> [extra hipcc flags]: per kernel of the TU -- registers, spills, scratch, and the vector
# instruction count with the moves (v_mov / v_readlane / v_writelane / scratch_) that only shuffle state around
f=$1; shift
out=$(mktemp -d)
hipcc -x hip refraction_raytracing_dxr_amd/csrc/$f --offload-arch=gfx950 -fno-gpu-rdc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S -o $out/k.s "$@" 2>/dev/null
python3 - $out/k.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.end_amdhsa_kernel", t, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    v = [i for i in ins if i.startswith("v_")]
    mov = sum(1 for i in v if i.startswith("v_mov"))
    lane = sum(1 for i in v if i.startswith(("v_readlane", "v_writelane")))
    scr = sum(1 for i in ins if i.startswith("scratch_"))
    g = lambda k: (re.search(r"\.%s\s+(\d+)" % k, body) or [0, "?"])[1]
    print("%-90s VGPR %s SGPR %s scratch %s B | %5d instr, %5d vector (%d v_mov, %d lane moves), %d scratch ops" % (
        name[:90], g("amdhsa_next_free_vgpr"), g("amdhsa_next_free_sgpr"), g("amdhsa_private_segment_fixed_size"), len(ins), len(v), mov, lane, scr))
PY
rm -rf $out
