"""Depth-1 launches of the product path (k_render_paths where the dispatcher picks it): python tools/exp_d1_paths.py mesh..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(2048, 1024, seed=0)
r = rr.Renderer(0)
for name in (sys.argv[1:] or ["monkey.obj"]):
    m = rr.Mesh(); assert m.load(asset(name))
    r.load_scene(m.verts, m.indices, env)
    out = []
    for refr, refl in ((0, 0), (2, 2), (8, 2)):
        p = rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_TIME_KERNEL)
        r.render_orbit(1920, 1080, 4, angle=0.01, params=p, frames_per_dispatch=1); r.kernel_time()
        r.render_orbit(1920, 1080, 32, angle=0.01, params=p, frames_per_dispatch=1)
        ms, n = r.kernel_time()
        out.append("%d/%d %6.1f us" % (refr, refl, ms / n * 1e3))
    print("%-11s %s" % (name, " | ".join(out)), flush=True)
r.close()
