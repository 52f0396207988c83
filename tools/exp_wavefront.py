"""fused kernel vs the experimental queue-per-bounce kernels (RR_DEBUG_KERNEL=wavefront): same frames? how fast?
usage: python tools/exp_wavefront.py            (runs itself twice, once per kernel)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import refraction_raytracing_dxr_amd as rr
    from refraction_raytracing_dxr_amd.synth import asset, procedural_env
    r = rr.Renderer(0)
    out = {}
    for name in ("monkey.obj", "sphere.obj", "ott.obj"):
        m = rr.Mesh(); m.load(asset(name))
        r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
        p = rr.default_params(max_refract=8)
        r.render_orbit(640, 360, 3, params=p, frames_per_dispatch=3)
        out[name] = np.stack([r.read_frame(slice=k) for k in range(3)])
        for F in (1, 16, 64):
            n = max(64, F * 2)
            r.render_orbit(1920, 1080, F, params=p, frames_per_dispatch=F); r.wait()
            r.timing_begin()
            r.render_orbit(1920, 1080, n, params=p, frames_per_dispatch=F)
            ms = r.timing_end()
            st = r.stats()
            print("%-10s %-9s F=%2d: %7.1f us/frame  %6.2f Grays/s  (overflow %d)" % (sys.argv[1], name, F, ms / n * 1e3, st.rays / (ms * 1e-3) / 1e9, st.traversal_overflow), flush=True)
    np.savez(sys.argv[2], **out)
else:
    import numpy as np
    for k in ("fused", "wavefront"):
        env = dict(os.environ); env["RR_DEBUG_KERNEL"] = k
        subprocess.run([sys.executable, __file__, k, "/tmp/wf_%s.npz" % k], env=env, check=True)
    a, b = np.load("/tmp/wf_fused.npz"), np.load("/tmp/wf_wavefront.npz")
    for name in a.files:
        print(name, "frames identical:", np.array_equal(a[name], b[name]), "differing pixels:", int((a[name] != b[name]).any(-1).sum()))
