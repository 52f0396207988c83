"""Depth-1 launches (the reference's DispatchRays(W,H,1)) of monkey / sphere / ott at 1080p on whatever RR_DEBUG_* selects:
kernel us per frame over 32 frames of the orbit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(2048, 1024, seed=0)
r = rr.Renderer(0)
out = []
for name, refr in (("monkey.obj", 8), ("sphere.obj", 4), ("ott.obj", 8), ("shell.obj", 5)):
    m = rr.Mesh(); assert m.load(asset(name))
    r.load_scene(m.verts, m.indices, env)
    p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
    r.render_orbit(1920, 1080, 16, angle=0.01, params=p, frames_per_dispatch=1); r.kernel_time()
    r.render_orbit(1920, 1080, 32, angle=0.01, params=p, frames_per_dispatch=1)
    ms, k = r.kernel_time()
    out.append("%s %6.1f (%s)" % (name.split(".")[0], ms * 1e3 / k, r.stats().render_kernel_name.decode().split("<")[0][9:]))
print("D1 us/frame: " + " | ".join(out), flush=True)
r.close()
