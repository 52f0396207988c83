import os, sys
sys.path.insert(0, "/root/repo")
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
r = rr.Renderer(0)
m = rr.Mesh(); m.load(asset(sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8)
r.render_orbit(1920, 1080, 64, params=p, frames_per_dispatch=16); r.wait()
