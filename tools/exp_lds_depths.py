"""k_render_lds against k_render_fused by launch depth, kernel time alone (HIP events round each launch), same frames:
what a launch of the persistent kernel costs beyond its frames.  python tools/exp_lds_depths.py [mesh] [refract]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
refr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = rr.Mesh(); m.load(asset(name))
env = procedural_env(2048, 1024, seed=0)
for k in ("fused", "lds"):
    os.environ["RR_DEBUG_KERNEL"] = k
    r = rr.Renderer(0)
    r.load_scene(m.verts, m.indices, env)
    p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
    out = []
    for depth in (3, 4, 8, 16, 20, 32, 64, 128):
        best = 1e9
        for rep in range(4):
            r.render_orbit(1920, 1080, depth, angle=0.01, params=p, frames_per_dispatch=depth)
            ms, n = r.kernel_time()
            best = min(best, ms / n)
        out.append("D%d %.0f us (%.1f/frame)" % (depth, best * 1e3, best * 1e3 / depth))
    print("%-6s %s" % (k, " | ".join(out)), flush=True)
    r.close()
