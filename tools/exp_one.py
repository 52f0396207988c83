"""python tools/exp_one.py <mesh> <max_refract> <max_reflect> [frames]  -- renders frames of one config (for profiling)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
name, refr, refl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 20
r = rr.Renderer(0)
m = rr.Mesh(); m.load(O.asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_TIME_KERNEL)
r.render_orbit(1920, 1080, frames, params=p)
ms, n = r.kernel_time()
print("%s %d/%d: %.1f us/frame, %.2f Mrays/frame" % (name, refr, refl, ms / n * 1e3, r.stats().rays / frames / 1e6))
r.close()
