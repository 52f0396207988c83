"""Profiling target: a few launches of ONE workload, nothing else (no stats launches, no second mesh), so every
render-kernel row in a rocprofv3 trace / PMC pass is a launch of that workload.
  python3 tools/prof_target.py [mesh] [max_refract] [depth] [launches] [W] [H] [max_reflect]
Defaults: the bench workload -- monkey.obj, 8 bounces, Depth 64, 6 launches, 1920x1080, 2 reflections.
PROF_STATS=1: the same frames through the RR_DISPATCH_COLLECT_STATS kernels instead; prints the exact counters as JSON
(run it unprofiled: the stats builds are different kernels)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

a = sys.argv[1:]
mesh = a[0] if len(a) > 0 else "monkey.obj"
refr = int(a[1]) if len(a) > 1 else 8
depth = int(a[2]) if len(a) > 2 else 64
launches = int(a[3]) if len(a) > 3 else 6
W = int(a[4]) if len(a) > 4 else 1920
H = int(a[5]) if len(a) > 5 else 1080
refl = int(a[6]) if len(a) > 6 else 2
stats = os.environ.get("PROF_STATS") == "1"
r = rr.Renderer(0)
m = rr.Mesh(); assert m.load(asset("monkey.obj" if mesh == "monkey16k" else mesh))
if mesh == "monkey16k":                # BASELINE's "~16k tri Suzanne": monkey.obj midpoint-subdivided twice (15 472 triangles)
    from refraction_raytracing_dxr_amd.synth import subdivide
    v16, i16 = subdivide(m.verts, 2)
    r.load_scene(v16, i16, procedural_env(2048, 1024, seed=0))
else:
    r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_COLLECT_STATS if stats else 0)
r.render_orbit(W, H, depth * launches, angle=0.01, params=p, frames_per_dispatch=depth)
r.wait()
st = r.stats()
if stats:
    print(json.dumps({"workload": "%s %d/%d D%d x%d %dx%d" % (mesh, refr, refl, depth, launches, W, H), "launches": launches,
                      **{k: int(getattr(st, k)) for k in ("rays", "hits", "misses", "node_visits", "tri_tests", "node_trips", "leaf_trips",
                                                         "shade_passes", "waves", "background_waves", "pixels")}}), flush=True)
else:
    print("rays", st.rays, flush=True)
r.close()
