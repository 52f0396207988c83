import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
W, H = 1920, 1080
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8, max_reflect=2)
def timed(K, label):
    r.wait(); t0 = time.perf_counter(); r.timing_begin()
    r.render_orbit(W, H, K, angle=0.01, params=p, frames_per_dispatch=64)
    ms = r.timing_end(); t1 = time.perf_counter()
    print("%-40s device %.1f us wall %.1f us | %s" % (label, ms * 1e3, (t1 - t0) * 1e6, r.stats().render_kernel_name.decode()[:16]), flush=True)
r.render_orbit(W, H, 512, angle=0.01, params=p, frames_per_dispatch=64); r.wait()
r.render_orbit(W, H, 5, angle=0.01, params=p, frames_per_dispatch=64)
timed(20, "after prewarm 512 + warmup 5 (bench)")
timed(20, "again")
timed(20, "again")
r.render_orbit(W, H, 512, angle=0.01, params=p, frames_per_dispatch=64); r.wait()
timed(20, "after 512 frames, no warm-up")
r.render_orbit(W, H, 512, angle=0.01, params=p, frames_per_dispatch=64); r.wait()
r.render_orbit(W, H, 20, angle=3.0, params=p, frames_per_dispatch=64)
timed(20, "after 512 + a 20-frame launch elsewhere")
time.sleep(0.05)
timed(20, "after 50 ms of idle")
