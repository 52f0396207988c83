"""Throughput with several frames in flight (one context = one HIP stream each): python tools/exp_overlap.py [mesh]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
m = rr.Mesh(); m.load(O.asset(name))
env = procedural_env(2048, 1024, seed=0)
ctxs = []
for i in range(8):
    r = rr.Renderer(0); r.load_scene(m.verts, m.indices, env); ctxs.append(r)
p = rr.default_params(max_refract=8)
F = 64
for n in (1, 2, 3, 4, 6, 8):
    use = ctxs[:n]
    for r in use: r.render_orbit(1920, 1080, 4, params=p)
    for r in use: r.wait()
    t0 = time.perf_counter()
    per = F // n
    # interleave submissions so all streams have work queued
    for k in range(per):
        for i, r in enumerate(use):
            r.render_orbit(1920, 1080, 1, angle=0.01 * (1 + k * n + i), params=p)
    for r in use: r.wait()
    dt = time.perf_counter() - t0
    print("%s: %d streams: %7.1f us per frame (%d frames)" % (name, n, dt / (per * n) * 1e6, per * n), flush=True)
