import os, sys
sys.path.insert(0, "/root/repo")
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(2048, 1024, seed=0)
r = rr.Renderer(0)
for name, refr in (("monkey.obj", 8), ("sphere.obj", 4), ("ott.obj", 8)):
    m = rr.Mesh(); assert m.load(asset(name))
    r.load_scene(m.verts, m.indices, env)
    out = []
    for depth in (1, 4, 16, 20, 64):
        p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
        n = max(depth * 2, 16)
        r.render_orbit(1920, 1080, n, angle=0.01, params=p, frames_per_dispatch=depth); r.kernel_time()
        r.render_orbit(1920, 1080, n * 2, angle=0.01, params=p, frames_per_dispatch=depth)
        ms, k = r.kernel_time()
        out.append("D%-2d %6.1f" % (depth, ms * 1e3 / (n * 2)))
    print("%-11s us/frame: %s" % (name, " | ".join(out)), flush=True)
r.close()
