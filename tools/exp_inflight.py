"""rr_render_orbit with 1..3 launches in flight, frames_per_dispatch sweep (monkey 1080p, 8/2 bounces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
m = rr.Mesh(); m.load(asset(name))
r = rr.Renderer(0)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8, max_reflect=2)
for F in (1, 2, 4, 16, 64):
    n = max(256, 4 * F)
    for fl in (1, 2, 3, 4):
        r.set_frames_in_flight(fl)
        for rep in range(2):
            r.wait(); t0 = time.perf_counter()
            r.render_orbit(W, H, n, angle=0.01, params=p, frames_per_dispatch=F)
            r.wait(); dt = time.perf_counter() - t0
        rays = r.stats().rays
        print("%s F %3d in-flight %d: %.1f us/frame  %.2f Grays/s" % (name, F, fl, dt / n * 1e6, rays / dt / 1e9), flush=True)
# the frames are the same ones
r.set_frames_in_flight(1); r.render_orbit(W, H, 6, angle=0.01, params=p, frames_per_dispatch=2); a = r.read_frame(slice=1).copy()
r.set_frames_in_flight(2); r.render_orbit(W, H, 6, angle=0.01, params=p, frames_per_dispatch=2); b = r.read_frame(slice=1).copy()
print("same last frame:", np.array_equal(a, b))
