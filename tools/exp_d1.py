"""Depth-1 launches (the reference's DispatchRays(W,H,1)): kernel time by bounce limits, L1-fed kernel against the LDS one.
python3 tools/exp_d1.py [mesh]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
env = procedural_env(2048, 1024, seed=0)
for kern, envv in (("fused", {"RR_DEBUG_KERNEL": "fused"}), ("paths", {"RR_DEBUG_KERNEL": "paths"}), ("lds", {"RR_DEBUG_KERNEL": "lds"})):
    for k in ("RR_DEBUG_KERNEL", "RR_DEBUG_TICKET", "RR_DEBUG_SHAPE"): os.environ.pop(k, None)
    os.environ.update(envv)
    r = rr.Renderer(0)
    m = rr.Mesh(); assert m.load(asset(name))
    r.load_scene(m.verts, m.indices, env)
    out = []
    for refr, refl in ((0, 0), (1, 0), (1, 2), (2, 2), (8, 2)):
        p = rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_TIME_KERNEL)
        r.render_orbit(1920, 1080, 4, angle=0.01, params=p, frames_per_dispatch=1); r.kernel_time()
        r.render_orbit(1920, 1080, 16, angle=0.01, params=p, frames_per_dispatch=1)
        ms, n = r.kernel_time()
        out.append("%d/%d %6.1f us" % (refr, refl, ms / n * 1e3))
    print("%-11s %s | %s" % (kern, name, " | ".join(out)), flush=True)
    r.close()
