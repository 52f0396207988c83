"""the 15 472-triangle monkey (BASELINE's '~16k tri Suzanne'): hierarchy stats and Depth-64 time, both builders"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env, subdivide
W, H, F = 1920, 1080, 64
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
env = procedural_env(2048, 1024, seed=0)
for levels in (2, 3):
    v, i = subdivide(m.verts, levels)
    for fast_build in (False, True):
        mid = r.upload_mesh(v, i); r.build_blas(mid, fast_build=fast_build); r.build_tlas(rr.make_instances(meshes=[mid])); r.upload_envmap(env)
        r.render_orbit(W, H, F, params=rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS), frames_per_dispatch=F)
        st = r.stats()
        p = rr.default_params(max_refract=8)
        r.render_orbit(W, H, F, params=p, frames_per_dispatch=F); r.wait()
        r.timing_begin(); r.render_orbit(W, H, 2 * F, params=p, frames_per_dispatch=F); ms = r.timing_end()
        print("%6d tri %s: depth %2d  %.2f node visits + %.2f tri tests per ray  %.1f us/frame  %.2f Grays/s" % (
            len(i) // 3, "LBVH" if fast_build else "PLOC", st.bvh_depth, st.node_visits / st.rays, st.tri_tests / st.rays,
            ms / (2 * F) * 1e3, st.rays / F / (ms / (2 * F) * 1e-3) / 1e9), flush=True)
