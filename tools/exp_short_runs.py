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
for K in (20, 64):
  for F in (K, (K + 1) // 2, (K + 2) // 3, (K + 3) // 4, 5):
    for fl in (1, 2, 3):
        r.set_frames_in_flight(fl)
        best = 1e9
        for rep in range(5):
            r.render_orbit(W, H, 5, angle=0.01, params=p, frames_per_dispatch=F); r.wait()
            r.timing_begin(); r.render_orbit(W, H, K, angle=0.01, params=p, frames_per_dispatch=F); ms = r.timing_end()
            best = min(best, ms)
        print("K %d F %2d in-flight %d: %.1f us/frame" % (K, F, fl, best / K * 1e3), flush=True)
