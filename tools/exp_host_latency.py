import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
W, H = 1920, 1080
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8, max_reflect=2)
r.render_orbit(W, H, 512, angle=0.01, params=p, frames_per_dispatch=64); r.wait()
for K in (20, 20, 20, 1, 1, 64):
    r.wait()
    t0 = time.perf_counter()
    r.timing_begin()
    t1 = time.perf_counter()
    r.render_orbit(W, H, K, angle=0.01, params=p, frames_per_dispatch=K)
    t2 = time.perf_counter()
    ms = r.timing_end()
    t3 = time.perf_counter()
    print("K %2d: timing_begin %.1f us, render_orbit returns after %.1f us, timing_end returns after %.1f us; device region %.1f us; wall %.1f us" % (
        K, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, ms * 1e3, (t3 - t0) * 1e6), flush=True)
