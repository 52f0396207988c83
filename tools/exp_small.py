"""Fixed cost of a launch: tiny frames, primary rays only.  python3 tools/exp_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(256, 128, seed=0)
for kern, envv in (("fused", {"RR_DEBUG_KERNEL": "fused"}), ("lds", {"RR_DEBUG_KERNEL": "lds"}), ("lds 16x1", {"RR_DEBUG_KERNEL": "lds", "RR_DEBUG_SHAPE": "2"})):
    for k in ("RR_DEBUG_KERNEL", "RR_DEBUG_TICKET", "RR_DEBUG_SHAPE"): os.environ.pop(k, None)
    os.environ.update(envv)
    r = rr.Renderer(0)
    m = rr.Mesh(); assert m.load(asset("monkey.obj"))
    r.load_scene(m.verts, m.indices, env)
    out = []
    for W, H in ((8, 8), (256, 256), (1024, 768), (1920, 1080)):
        p = rr.default_params(max_refract=0, max_reflect=0, flags=rr.DISPATCH_TIME_KERNEL)
        r.render_orbit(W, H, 4, angle=0.01, params=p, frames_per_dispatch=1); r.kernel_time()
        r.render_orbit(W, H, 16, angle=0.01, params=p, frames_per_dispatch=1)
        ms, n = r.kernel_time()
        out.append("%dx%d %6.1f us" % (W, H, ms / n * 1e3))
    print("%-9s %s" % (kern, " | ".join(out)), flush=True)
    r.close()
