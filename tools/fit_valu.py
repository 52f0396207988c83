"""Least-squares fit of the vector instructions the render kernel issues (PMC SQ_INSTS_VALU) against the wave-level trip
counters of the same frames: INSTS_VALU ~ a * node_trips + b * leaf_trips + c * passes + d * waves + e * background waves,
where a background wave (a block outside the scene's screen rectangle: RayGen + one Miss on a branch of its own) is taken out
of the pass and wave columns.
    python3 tools/fit_valu.py gpurun_out/fit_<tag>      (the directory tools/fit_valu.sh wrote)"""
import json, sys
import numpy as np
d = sys.argv[1]
stats = [json.loads(l) for l in open(d + "/stats.jsonl") if l.strip().startswith("{")]
pmc = []
for l in open(d + "/pmc.txt"):
    if "|" not in l: continue
    kv = dict(x.split("=") for x in l.split("|")[1].split())
    pmc.append({k: float(v) for k, v in kv.items()})
A = np.array([[s["node_trips"], s["leaf_trips"], s["shade_passes"] - s["background_waves"], s["waves"] - s["background_waves"],
               s["background_waves"]] for s in stats], float)
y = np.array([p["SQ_INSTS_VALU"] for p in pmc])
x, res, rank, sv = np.linalg.lstsq(A, y, rcond=None)
print("VALU instructions per internal-node trip %.1f, per leaf trip %.1f, per shading pass %.1f, per wave %.1f, per background wave %.1f" % tuple(x))
for s, p, row in zip(stats, pmc, A):
    model = float(row @ x)
    cyc = p["GRBM_GUI_ACTIVE"] / 8.0
    print("%-34s measured %.4g  model %.4g (%+.1f %%) | cycles per VALU instruction per SIMD %.2f | SQ_ACTIVE_INST_VALU*4 / SIMD cycles = %.3f | "
          "lane utilisation node %.1f %% leaf %.1f %% pass %.1f %%" % (
              s["workload"], p["SQ_INSTS_VALU"], model, 100 * (model / p["SQ_INSTS_VALU"] - 1), cyc * 1024 / p["SQ_INSTS_VALU"],
              p["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 100 * s["node_visits"] / (64 * s["node_trips"]),
              100 * s["tri_tests"] / (64 * max(s["leaf_trips"], 1)), 100 * s["rays"] / (64 * s["shade_passes"])))

# ---- the same fit per instruction class (tools/fit_valu_classes.sh), if its file is there
import os
pc = d + "/pmc_classes.txt"
if os.path.exists(pc):
    cls = []
    for l in open(pc):
        if "|" not in l: continue
        cls.append({k: float(v) for k, v in (x.split("=") for x in l.split("|")[1].split())})
    names = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"]
    print("\nper-trip counts by class (node trip, leaf trip, shading pass, wave, background wave):")
    per = {}
    for n in names:
        yy = np.array([c[n] for c in cls])
        xx = np.linalg.lstsq(A, yy, rcond=None)[0]
        per[n] = xx
        print("  %-26s %7.1f %7.1f %7.1f %7.1f %7.1f   (worst residual %.1f %%)" % (n, *xx, 100 * np.abs(A @ xx / yy - 1).max()))
    # SIMD cycles per trip: f32 add/mul at the full rate; FMA_F32 counts v_fma_f32 / v_fmac_f32 (full rate) AND v_fma_mix_f32
    # (quarter rate: the 12 slab distances of a node trip), which the node-trip row shows
    full = per["SQ_INSTS_VALU_ADD_F32"] + per["SQ_INSTS_VALU_MUL_F32"] + per["SQ_INSTS_VALU_FMA_F32"]
    full[0] -= per["SQ_INSTS_VALU_FMA_F32"][0]            # the node trip's "FMA_F32" are its v_fma_mix_f32: quarter rate
    trans = per["SQ_INSTS_VALU_TRANS_F32"]
    quarter = x - full - trans
    C_FULL, C_QUARTER, C_TRANS = 2.2, 4.2, 8.0
    cyc = full * C_FULL + quarter * C_QUARTER + trans * C_TRANS
    print("instructions per trip: total %s\n   full rate %s\n   quarter %s\n   transcendental %s" % (np.round(x, 1), np.round(full, 1), np.round(quarter, 1), np.round(trans, 1)))
    print("SIMD cycles per trip (%.1f / %.1f / %.1f cycles per class): node %.1f leaf %.1f pass %.1f wave %.1f background wave %.1f" % (C_FULL, C_QUARTER, C_TRANS, *cyc))
    for s, p, row in zip(stats, pmc, A):
        print("%-34s vector issue busy %.3f" % (s["workload"], float(row @ cyc) / (1024 * p["GRBM_GUI_ACTIVE"] / 8.0)))
