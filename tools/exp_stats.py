"""python tools/exp_stats.py <mesh> -- node visits / tri tests per ray and kernel time (C3 params)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
r = rr.Renderer(0)
for name in sys.argv[1:]:
    m = rr.Mesh(); m.load(O.asset(name))
    r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
    r.render_orbit(1920, 1080, 10, params=rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS))
    st = r.stats()
    r.render_orbit(1920, 1080, 30, params=rr.default_params(max_refract=8, flags=rr.DISPATCH_TIME_KERNEL))
    ms, n = r.kernel_time()
    print("%-11s depth %2d  nodes/ray %6.2f  tris/ray %5.2f  rays/frame %.2fM  kernel %7.1f us  %6.2f Grays/s" % (
        name, st.bvh_depth, st.node_visits / st.rays, st.tri_tests / st.rays, st.rays / 10 / 1e6, ms / n * 1e3,
        st.rays / 10 / (ms / n * 1e-3) / 1e9), flush=True)
r.close()
