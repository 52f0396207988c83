import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
r = rr.Renderer(0)
m = rr.Mesh(); m.load(O.asset("monkey.obj"))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
for label, kw, (W, H) in [("all rays cull at root (tmax 1e-3)", dict(max_refract=0, tmax_primary=1e-3), (1920, 1080)),
                  ("primary only", dict(max_refract=0), (1920, 1080)),
                  ("root cull 960x540", dict(max_refract=0, tmax_primary=1e-3), (960, 540)),
                  ("root cull 3840x2160", dict(max_refract=0, tmax_primary=1e-3), (3840, 2160))]:
    p = rr.default_params(flags=rr.DISPATCH_TIME_KERNEL, **kw)
    r.render_orbit(W, H, 5, params=p); r.kernel_time()
    r.render_orbit(W, H, 30, params=p)
    ms, n = r.kernel_time()
    print("%-40s %8.1f us/frame (%d px)" % (label, ms / n * 1e3, W * H), flush=True)
r.close()
