"""Kernel-time experiments on the GPU box: python tools/exp_time.py  (prints a small table)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env

W, H = 1920, 1080
r = rr.Renderer(0)
env = procedural_env(2048, 1024, seed=0)
for name in (sys.argv[1:] or ["monkey.obj"]):
    m = rr.Mesh(); m.load(O.asset(name))
    r.load_scene(m.verts, m.indices, env)
    for (refr, refl) in [(0, 0), (1, 0), (1, 2), (2, 2), (5, 2), (8, 2), (8, 0)]:
        p = rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_TIME_KERNEL)
        r.render_orbit(W, H, 5, params=p); r.kernel_time()
        r.render_orbit(W, H, 30, params=p)
        ms, n = r.kernel_time()
        st = r.stats()
        print("%-11s refract %d reflect %d : %8.1f us/frame  %6.2f Mrays/frame  %7.2f Grays/s" %
              (name, refr, refl, ms / n * 1e3, st.rays / 30 / 1e6, st.rays / 30 / (ms / n * 1e-3) / 1e9), flush=True)
r.close()
