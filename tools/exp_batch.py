"""frames per dispatch sweep: python tools/exp_batch.py [mesh]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
r = rr.Renderer(0)
for name in (sys.argv[1:] or ["monkey.obj"]):
    m = rr.Mesh(); m.load(O.asset(name))
    r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
    p = rr.default_params(max_refract=8)
    for F in (1, 16, 64):
        n = max(64, F * 2)
        r.render_orbit(1920, 1080, F, params=p, frames_per_dispatch=F); r.wait()
        t0 = time.perf_counter()
        r.timing_begin()
        r.render_orbit(1920, 1080, n, params=p, frames_per_dispatch=F)
        ms = r.timing_end()
        dt = time.perf_counter() - t0
        st = r.stats()
        print("%-11s F=%2d: %7.1f us/frame (events) %7.1f us/frame (wall)  %6.2f Grays/s" % (name, F, ms / n * 1e3, dt / n * 1e6, st.rays / (ms * 1e-3) / 1e9), flush=True)
r.close()
