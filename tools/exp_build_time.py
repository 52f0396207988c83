"""BLAS build times by mesh size and builder: python tools/exp_build_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, subdivide
r = rr.Renderer(0)
m = rr.Mesh(); m.load(asset("monkey.obj"))
o = rr.Mesh(); o.load(asset("ott.obj"))
cases = [("monkey.obj", m.verts, m.indices), ("ott.obj", o.verts, o.indices)]
for k in (2, 3, 4):
    v, i = subdivide(m.verts, k); cases.append(("monkey subdivided x%d" % k, v, i))
for name, v, i in cases:
    mid = r.upload_mesh(v, i)
    out = []
    for fast_build in (False, True):
        ts = []
        for rep in range(3):
            r.wait(); t0 = time.perf_counter()
            r.build_blas(mid, fast_build=fast_build)
            r.wait(); ts.append(time.perf_counter() - t0)
        out.append("%s %.2f ms" % ("fast_build (LBVH)" if fast_build else "fast_trace", min(ts) * 1e3))
    print("%-24s %7d tris | %s" % (name, len(i) // 3, " | ".join(out)), flush=True)
