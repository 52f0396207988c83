import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
W, H = 1920, 1080
for name in ("monkey.obj", "ott.obj", "sphere.obj"):
    m = rr.Mesh(); m.load(asset(name))
    r = rr.Renderer(0)
    r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
    p = rr.default_params(max_refract=8 if name != "sphere.obj" else 4, max_reflect=2)
    for F in (1, 2):
        n = 512
        for fl in (1, 2, 3, 4):
            r.set_frames_in_flight(fl)
            for rep in range(3):
                r.wait(); t0 = time.perf_counter()
                r.render_orbit(W, H, n, angle=0.01, params=p, frames_per_dispatch=F)
                r.wait(); dt = time.perf_counter() - t0
            print("%s F %d in-flight %d: %.1f us/frame | %s" % (name, F, fl, dt / n * 1e6, r.stats().render_kernel_name.decode()), flush=True)
    r.close()
