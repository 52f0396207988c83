"""Profiling target: a few launches of ONE workload, nothing else (no stats launches, no second mesh), so every
k_render_fused row in a rocprofv3 trace / PMC pass is a launch of that workload.
  python3 tools/prof_target.py [mesh] [max_refract] [depth] [launches] [W] [H]
Defaults: the bench workload -- monkey.obj, 8 bounces, Depth 64, 6 launches, 1920x1080."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

a = sys.argv[1:]
mesh = a[0] if len(a) > 0 else "monkey.obj"
refr = int(a[1]) if len(a) > 1 else 8
depth = int(a[2]) if len(a) > 2 else 64
launches = int(a[3]) if len(a) > 3 else 6
W = int(a[4]) if len(a) > 4 else 1920
H = int(a[5]) if len(a) > 5 else 1080
r = rr.Renderer(0)
m = rr.Mesh(); assert m.load(asset(mesh))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=refr, max_reflect=2)
r.render_orbit(W, H, depth * launches, angle=0.01, params=p, frames_per_dispatch=depth)
r.wait()
print("rays", r.stats().rays, flush=True)
r.close()
