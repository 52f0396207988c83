import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(256, 128, seed=9)
def render(kernel, group, name, kw, W=323, H=181):
    os.environ["RR_DEBUG_KERNEL"] = kernel; os.environ["RR_DEBUG_GROUP_TRACE"] = str(group)
    r = rr.Renderer(0)
    m = rr.Mesh(); m.load(asset(name)); r.load_scene(m.verts, m.indices, env)
    r.set_camera(rr.camera_orbit(0.3))
    r.dispatch_rays(W, H, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS, **kw))
    rgba, f32 = r.read_frame(want_float=True); st = r.stats()
    r.close()
    return f32.copy(), (st.rays, st.hits, st.misses, st.terminal_hits, st.tir, st.render_kernel)
for name, kw in (("monkey.obj", dict(max_refract=8)), ("sphere.obj", dict(max_refract=4, max_reflect=1)), ("cube.obj", dict())):
    a, ca = render("fused", 0, name, kw)
    for g in (0, 1):
        b, cb = render("paths", g, name, kw)
        d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
        print(name, "group", g, "differing pixels", int(d.sum()), "counters", ca, cb)
        if d.sum():
            ys, xs = np.nonzero(d)
            for k in range(min(5, len(ys))):
                print("   ", xs[k], ys[k], a[ys[k], xs[k]], b[ys[k], xs[k]])
