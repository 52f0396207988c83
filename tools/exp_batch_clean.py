"""frames per launch sweep, best of several passes (the scene's measured kernel choice is made before anything is timed):
INFLIGHT=<n> python tools/exp_batch_clean.py [mesh...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
r = rr.Renderer(0)
if os.environ.get("INFLIGHT"):
    r.set_frames_in_flight(int(os.environ["INFLIGHT"]))             # launches of one rr_render_orbit that may overlap (default: the library's, 2)
for name in (sys.argv[1:] or ["monkey.obj"]):
    m = rr.Mesh(); m.load(asset(name))
    r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
    p = rr.default_params(max_refract=4 if name == "sphere.obj" else 8)
    out = []
    for F in (4, 8, 16, 20, 32, 64):
        n = max(128, 4 * F)
        best = 1e9
        for rep in range(4):
            r.render_orbit(1920, 1080, n, params=p, frames_per_dispatch=F); r.wait()
            r.timing_begin(); r.render_orbit(1920, 1080, n, params=p, frames_per_dispatch=F); ms = r.timing_end()
            best = min(best, ms / n * 1e3)
        out.append("F=%d %.1f" % (F, best))
    print("%-11s us/frame: %s | %s" % (name, "  ".join(out), r.stats().render_kernel_name.decode()), flush=True)
r.close()
