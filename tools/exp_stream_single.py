"""one mesh under a TLAS whose single instance is NOT the identity (a 1e-3 shift), so the two-level renderers take it:
the stream renderer against the lock-step kernel on a single-BLAS workload.  python tools/exp_stream_single.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env, subdivide
W, H, F = 1920, 1080, 16
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
env = procedural_env(2048, 1024, seed=0)
xf = np.eye(4, dtype=np.float32)[:3].copy(); xf[1, 3] = 1e-3
for levels in (0, 2, 3):
    v, i = subdivide(m.verts, levels) if levels else (m.verts, m.indices)
    mid = r.upload_mesh(v, i); r.build_blas(mid)
    for ident in (True, False):
        r.build_tlas(rr.make_instances(meshes=[mid]) if ident else rr.make_instances(transforms=[xf], meshes=[mid])); r.upload_envmap(env)
        cams = [rr.camera_orbit(0.01 * (k + 1)) for k in range(F)]
        p = rr.default_params(max_refract=8, flags=rr.DISPATCH_TIME_KERNEL)
        for rep in range(5):
            r.dispatch_rays_batch(W, H, cams, p)
            if rep == 2: r.kernel_time()
        ms, n = r.kernel_time()
        st = r.stats()
        print("%6d tri %s: %7.1f us/frame %6.2f Grays/s | %s" % (len(i) // 3, "identity" if ident else "shifted ", ms / n * 1e3 / F,
              st.rays / F / (ms / n * 1e3 / F) / 1e3, st.render_kernel_name.decode()), flush=True)
