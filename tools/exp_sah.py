"""SAH cost of the built hierarchies (internal-node term: sum of child-box areas over the root area) for the two GPU builders,
against a host top-down full-sweep SAH build of the same triangles -- how much hierarchy quality is left on the table.
    python tools/exp_sah.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset

def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])

def cost_gpu(nodes):
    lo = np.stack([nodes["lox"], nodes["loy"], nodes["loz"]], -1)      # [n, 2, 3]
    hi = np.stack([nodes["hix"], nodes["hiy"], nodes["hiz"]], -1)
    a = area(lo, hi)                                                  # [n, 2]
    root = area(lo[0].min(0), hi[0].max(0))
    internal = nodes["c"] >= 0
    return float(a[internal].sum() / root + 1.0), float(a[~internal].sum() / root)      # (node-visit term incl. root, leaf term)

def sah_build(lo, hi):
    """full-sweep SAH, one triangle per leaf: returns (internal term incl. root, leaf term) relative to the root area"""
    cen = 0.5 * (lo + hi)
    root = area(lo.min(0), hi.max(0))
    tot_i, tot_l = 0.0, 0.0
    stack = [np.arange(len(lo))]
    while stack:
        idx = stack.pop()
        a_node = area(lo[idx].min(0), hi[idx].max(0))
        if len(idx) == 1:
            tot_l += a_node; continue
        tot_i += a_node
        best = (np.inf, None, None)
        for ax in range(3):
            o = idx[np.argsort(cen[idx, ax], kind="stable")]
            l_lo = np.minimum.accumulate(lo[o], 0); l_hi = np.maximum.accumulate(hi[o], 0)
            r_lo = np.minimum.accumulate(lo[o][::-1], 0)[::-1]; r_hi = np.maximum.accumulate(hi[o][::-1], 0)[::-1]
            n = len(o)
            k = np.arange(1, n)
            c = area(l_lo[:-1], l_hi[:-1]) * k + area(r_lo[1:], r_hi[1:]) * (n - k)
            j = int(np.argmin(c))
            if c[j] < best[0]: best = (c[j], o, j + 1)
        _, o, split = best
        stack.append(o[:split]); stack.append(o[split:])
    return tot_i / root, tot_l / root

r = rr.Renderer(0)
for name in ("sphere.obj", "monkey.obj", "shell.obj", "ott.obj"):
    m = rr.Mesh(); m.load(asset(name))
    mid = r.upload_mesh(m.verts, m.indices)
    out = []
    for fb in (True, False):
        r.build_blas(mid, fast_build=fb)
        nodes, tris = r.download_blas(mid)
        ci, cl = cost_gpu(nodes)
        out.append("%s internal %.2f leaf %.2f" % ("LBVH" if fb else "PLOC", ci, cl))
    if len(m.indices) // 3 <= 2000:
        p = m.verts["position"][m.indices].reshape(-1, 3, 3)
        si, sl = sah_build(p.min(1).astype(np.float64), p.max(1).astype(np.float64))
        out.append("host SAH internal %.2f leaf %.2f" % (si, sl))
    print("%-11s %s" % (name, " | ".join(out)), flush=True)
