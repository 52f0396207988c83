"""Depth-64 kernel time around the orbit, L1-fed kernel against the LDS one.  python3 tools/exp_orbit.py [mesh] [refract]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
refr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
env = procedural_env(2048, 1024, seed=0)
res = {}
for kern in ("fused", "lds"):
    os.environ["RR_DEBUG_KERNEL"] = kern
    r = rr.Renderer(0)
    m = rr.Mesh(); assert m.load(asset(name))
    r.load_scene(m.verts, m.indices, env)
    p = rr.default_params(max_refract=refr, max_reflect=2, flags=rr.DISPATCH_TIME_KERNEL)
    for a0 in (0.01, 0.8, 1.6, 2.4, 3.2, 4.0, 4.8, 5.6):
        r.render_orbit(1920, 1080, 64, angle=a0, params=p, frames_per_dispatch=64); r.kernel_time()
        r.render_orbit(1920, 1080, 128, angle=a0, params=p, frames_per_dispatch=64)
        ms, n = r.kernel_time()
        res.setdefault(a0, {})[kern] = ms / n * 1e3 / 64
    r.close()
for a0, v in res.items():
    print("%s angle %.2f: fused %6.1f us/frame, lds %6.1f (%+.1f %%)" % (name, a0, v["fused"], v["lds"], 100 * (v["fused"] / v["lds"] - 1)))
