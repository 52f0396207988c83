"""k_render_lds (BLAS nodes in LDS, persistent workgroups) against k_render_fused (nodes through L1) on the same frames:
HIP-event kernel time per frame at several dispatch depths.  python3 tools/exp_lds.py [mesh:refract ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

W, H = 1920, 1080
env = procedural_env(2048, 1024, seed=0)
cases = [c.split(":") for c in (sys.argv[1:] or ["monkey.obj:8", "sphere.obj:4", "shell.obj:5"])]
for kern, envv in (("fused", {"RR_DEBUG_KERNEL": "fused"}), ("lds q8", {"RR_DEBUG_KERNEL": "lds"}), ("lds q64", {"RR_DEBUG_KERNEL": "lds", "RR_DEBUG_TICKET": "64"}),
                   ("lds q512", {"RR_DEBUG_KERNEL": "lds", "RR_DEBUG_TICKET": "128"})):
    for k in ("RR_DEBUG_KERNEL", "RR_DEBUG_TICKET", "RR_DEBUG_SHAPE"): os.environ.pop(k, None)
    os.environ.update(envv)
    r = rr.Renderer(0)
    for name, refr in cases:
        m = rr.Mesh(); assert m.load(asset(name))
        r.load_scene(m.verts, m.indices, env)
        out = []
        for depth, frames in ((64, 256), (16, 128), (1, 32)):
            p = rr.default_params(max_refract=int(refr), max_reflect=2, flags=rr.DISPATCH_TIME_KERNEL)
            r.render_orbit(W, H, depth, angle=0.01, params=p, frames_per_dispatch=depth); r.kernel_time()
            r.render_orbit(W, H, frames, angle=0.01, params=p, frames_per_dispatch=depth)
            ms, n = r.kernel_time()
            st = r.stats()
            us = ms * 1e3 / frames
            out.append("D%-2d %7.1f us %6.2f Gr/s" % (depth, us, st.rays / frames / us / 1e3))
        print("%-22s %-11s refract %s | %s" % (kern or "lds", name, refr, " | ".join(out)), flush=True)
    r.close()
