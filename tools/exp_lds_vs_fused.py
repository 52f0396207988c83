"""k_render_fused against k_render_lds launch by launch round the orbit (Depth 64): how far one measurement can be from the orbit's
mean.  RR_DEBUG_KERNEL is read at rr_create, so one context per kernel.  python tools/exp_lds_vs_fused.py [mesh] [refract]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
refr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = rr.Mesh(); m.load(asset(name))
env = procedural_env(2048, 1024, seed=0)
res = {}
for k in ("fused", "lds"):
    os.environ["RR_DEBUG_KERNEL"] = k
    r = rr.Renderer(0)
    r.load_scene(m.verts, m.indices, env)
    p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
    out = []
    for b in range(10):
        a0 = 0.01 + 0.64 * b
        best = 1e9
        for rep in range(3):
            r.render_orbit(1920, 1080, 64, angle=a0, params=p, frames_per_dispatch=64)
            ms, n = r.kernel_time()
            best = min(best, ms / n)
        out.append(best)
    res[k] = out
    r.close()
print(name, "launch ms fused:", " ".join("%.2f" % x for x in res["fused"]))
print(name, "launch ms lds  :", " ".join("%.2f" % x for x in res["lds"]))
print(name, "lds / fused    :", " ".join("%.3f" % (b / a) for a, b in zip(res["fused"], res["lds"])), "| mean %.3f" % (sum(res["lds"]) / sum(res["fused"])))
