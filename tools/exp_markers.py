import sys; sys.path.insert(0, "/root/repo")
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
r = rr.Renderer(0)
m = rr.Mesh(); m.load(asset("monkey.obj"))
r.load_scene(m.verts, m.indices, procedural_env(256, 128, seed=0))
r.render_orbit(640, 360, 8, frames_per_dispatch=4); r.wait()
r.set_camera(rr.camera_orbit(0.3)); r.dispatch_rays(320, 200); r.read_frame()
