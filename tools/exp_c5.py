import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
def xf(tx, ty, tz, s=1.0):
    m = np.eye(4, dtype=np.float32)[:3] * np.float32(s); m[:, 3] = (tx, ty, tz); return m
m = rr.Mesh(); m.load(O.asset("monkey.obj"))
r = rr.Renderer(0)
mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid)
r.build_tlas(rr.make_instances(transforms=[xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)], meshes=[mid] * 1024))
r.upload_envmap(procedural_env(2048, 1024, seed=0))
sc = rr.camera_orbit(0.01); sc.camera_loc[0] *= 14; sc.camera_loc[2] *= 14; sc.camera_loc[1] = 11.2
r.set_camera(sc)
W, H = 3840, 2160
r.dispatch_rays(W, H, rr.default_params(max_refract=16, flags=rr.DISPATCH_COLLECT_STATS))
st = r.stats()
print("rays %.2fM  rays/px %.2f  nodes/ray %.1f  tris/ray %.2f  hits %.2fM  depth %d" % (st.rays / 1e6, st.rays / (W * H), st.node_visits / st.rays, st.tri_tests / st.rays, st.hits / 1e6, st.bvh_depth))
print("wave trips: node %.2fM leaf %.2fM passes %.3fM waves %.3fM | lane utilisation node %.1f %% leaf(tri) %.1f %% pass %.1f %%" % (
    st.node_trips / 1e6, st.leaf_trips / 1e6, st.shade_passes / 1e6, st.waves / 1e6, 100 * st.node_visits / (64.0 * st.node_trips),
    100 * st.tri_tests / (64.0 * st.leaf_trips), 100 * st.rays / (64.0 * st.shade_passes)))
p = rr.default_params(max_refract=16, flags=rr.DISPATCH_TIME_KERNEL)
for _ in range(3): r.dispatch_rays(W, H, p)
ms, n = r.kernel_time()
print("%.2f ms/frame  %.2f Grays/s" % (ms / n, st.rays / (ms / n * 1e-3) / 1e9))
