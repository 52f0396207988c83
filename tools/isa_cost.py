"""Static vector-issue cost of the render kernel's loops, from its gfx950 assembly.

    python3 tools/isa_cost.py [kernel-name-substring]     (default: the bench kernel, k_render_fused<19, 2, false, false, ...>)

Compiles csrc/rr_render.hip to assembly with the product flags, takes the kernel whose mangled name contains the
substring, and sorts every basic block by the loop it sits in (the compiler annotates each block with its innermost
loop):  the innermost loop that holds the slab test (v_fma_mix_f32) is the INTERNAL-NODE trip; the rest of its parent loop
is the LEAF trip (triangle test, pop); the rest of that one's parent is the per-ray PASS (box-test set-up, hit attributes,
ClosestHit / Miss); everything outside is per-BLOCK work (RayGen, store).  Vector instructions are counted in three classes
with the SIMD time tools/ubench_valu.hip measured for a wave64 instruction once several waves share a SIMD:
    full rate   v_fma_f32 v_fmac_f32 v_mul_f32 v_add_f32 v_sub_f32 v_subrev_f32 v_mac_f32        2 cycles (measured 2.2-2.9)
    quarter     every other v_* (selects, compares, min/max, v_fma_mix_f32, conversions, packed, integer)   4 cycles (4.2)
    transcend.  v_rcp* v_rsq* v_sqrt* v_exp* v_log* v_sin* v_cos*                                8 cycles (MI355X_MICROARCH.md)
A block inside a loop counts once per trip of that loop although branches may skip it, so the figures are upper bounds
per trip; bench.py scales them by the fraction of the static count the PMC counter SQ_INSTS_VALU confirms.
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "refraction_raytracing_dxr_amd", "csrc", "rr_render.hip")
FULL = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32"}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def klass(op):
    base = op.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0]
    if base in FULL:
        return "full"
    if base.startswith(TRANS):
        return "trans"
    return "quarter"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_render_fusedILi19ELi2ELb0ELb0ELb0EjLb0EE"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "render.s")
        subprocess.run(["hipcc", "-x", "hip", SRC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math", "--cuda-device-only", "-S", "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l))
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
    name = text[start].split(":")[0]
    # blocks: (label, innermost loop header or None, depth, parents[list of (header, depth)], instructions)
    blocks, cur = [], {"label": "entry", "loop": None, "depth": 0, "parents": [], "ins": []}
    i = start + 1
    while i < end:
        t = text[i].strip()
        m = re.match(r"^(?:(\.LBB\d+_\d+):|; %bb\.(\d+):)\s*(?:;\s*(.*))?$", t)
        if m:
            blocks.append(cur)
            label = m.group(1) or ("bb." + m.group(2))
            cur = {"label": label, "loop": None, "depth": 0, "parents": [], "ins": []}
            info = [m.group(3) or ""]
            j = i + 1
            while j < end and text[j].strip().startswith(";") and "Loop" in text[j]:
                info.append(text[j].strip().lstrip("; ").strip()); j += 1
            for s in info:
                mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", s)
                if mm: cur["loop"], cur["depth"] = mm.group(1), int(mm.group(2))
                mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", s)
                if mm: cur["parents"].append((mm.group(1), int(mm.group(2))))
                mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", s)
                if mm and m.group(1): cur["loop"], cur["depth"] = m.group(1)[2:], int(mm.group(1))
        elif t and not t.startswith(";") and not t.startswith("."):
            cur["ins"].append(t.split()[0])
        i += 1
    blocks.append(cur)
    # loop tree: header -> parent header
    parent = {}
    for b in blocks:
        if b["parents"] and b["label"].startswith(".L") and b["loop"] == b["label"][2:]:
            ps = sorted(b["parents"], key=lambda x: x[1])
            parent[b["loop"]] = ps[-1][0]
    node_loop = next(b["loop"] for b in blocks if any(op.startswith("v_fma_mix_f32") for op in b["ins"]) and b["loop"])
    leaf_loop = parent.get(node_loop)
    ray_loop = parent.get(leaf_loop)
    def phase(b):
        l = b["loop"]
        if l == node_loop: return "node trip"
        if l == leaf_loop: return "leaf trip"
        if l == ray_loop: return "ray pass"
        # deeper loops nested elsewhere (none expected) fall to their ancestors
        while l in parent:
            l = parent[l]
            if l == node_loop: return "node trip"
            if l == leaf_loop: return "leaf trip"
            if l == ray_loop: return "ray pass"
        return "per block"
    tab = {}
    for b in blocks:
        ph = phase(b)
        t = tab.setdefault(ph, {"full": 0, "quarter": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0})
        for op in b["ins"]:
            if op.startswith("v_") and not op.startswith("v_nop"):
                t[klass(op)] += 1
            elif op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt", "s_endpgm")):
                t["salu"] += 1
            elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
                t["vmem"] += 1
            elif op.startswith("ds_"):
                t["lds"] += 1
    print("kernel", name)
    print("%-10s %6s %8s %6s | %6s %11s | %5s %5s %4s" % ("phase", "full", "quarter", "trans", "VALU", "SIMD cycles", "SALU", "VMEM", "LDS"))
    for ph in ("node trip", "leaf trip", "ray pass", "per block"):
        t = tab.get(ph, {"full": 0, "quarter": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0})
        n = t["full"] + t["quarter"] + t["trans"]
        print("%-10s %6d %8d %6d | %6d %11d | %5d %5d %4d" % (ph, t["full"], t["quarter"], t["trans"], n,
                                                              2 * t["full"] + 4 * t["quarter"] + 8 * t["trans"], t["salu"], t["vmem"], t["lds"]))


if __name__ == "__main__":
    main()
