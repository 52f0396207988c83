"""C4 / C5 (two-level scenes) on whichever renderer RR_DEBUG_KERNEL selects: python tools/exp_tlas.py [C4|C5|both] [depth]
HIP-event kernel time per frame, exact trip counters of the STATS build (PROF=1: no stats launch, for counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

def xf(tx, ty, tz, s=1.0):
    m = np.eye(4, dtype=np.float32)[:3] * np.float32(s); m[:, 3] = (tx, ty, tz); return m
def load(n):
    m = rr.Mesh(); assert m.load(asset(n)); return m

which = sys.argv[1] if len(sys.argv) > 1 else "both"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
env = procedural_env(2048, 1024, seed=0)
r = rr.Renderer(0)
cases = [
    ("C4", ["shell.obj", "cube.obj", "ott.obj"], ([xf(0, 0, 0), xf(0, 0, -4.0), xf(0, 0, 4.0)], [0, 1, 2]), 3840, 2160, 8, 1.6),
    ("C5", ["monkey.obj"], ([xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)], [0] * 1024), 3840, 2160, 16, 14.0),
]
for label, names, inst, W, H, refr, radius in cases:
    if which not in ("both", label): continue
    ids = []
    for m in [load(n) for n in names]:
        mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid); ids.append(mid)
    r.build_tlas(rr.make_instances(transforms=inst[0], meshes=[ids[k] for k in inst[1]]))
    r.upload_envmap(env)
    cams = []
    for k in range(depth):
        sc = rr.camera_orbit(0.01 * (k + 1))
        sc.camera_loc[0] *= radius; sc.camera_loc[2] *= radius; sc.camera_loc[1] = 0.8 * radius
        cams.append(sc)
    p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
    for rep in range(5):        # (the kernel choice is measured on the second and third launch of a shape)
        r.dispatch_rays_batch(W, H, cams, p)
        if rep == 1: r.kernel_time()
    ms, n = r.kernel_time()
    st = r.stats()
    us = ms / n * 1e3 / depth
    name = st.render_kernel_name.decode()
    if os.environ.get("PROF") == "1":          # under a counter pass: only the timed launches (the stats build is another kernel)
        print("%s depth %d: %8.1f us/frame | %s" % (label, depth, us, name), flush=True); continue
    r.dispatch_rays_batch(W, H, cams, rr.default_params(max_refract=refr, flags=rr.DISPATCH_COLLECT_STATS))
    ss = r.stats()
    print("%s depth %d: %8.1f us/frame %6.2f Grays/s  %.2f Mrays/frame | %s | node trips %.1f M/frame (lanes %.3f) leaf trips %.1f M (%.3f) passes %.2f M (%.3f) overflow %d" % (
        label, depth, us, st.rays / depth / us / 1e3, st.rays / depth / 1e6, name, ss.node_trips / depth / 1e6, ss.node_visits / 64 / max(ss.node_trips, 1),
        ss.leaf_trips / depth / 1e6, ss.tri_tests / 64 / max(ss.leaf_trips, 1), ss.shade_passes / depth / 1e6, ss.rays / 64 / max(ss.shade_passes, 1), ss.traversal_overflow), flush=True)
r.close()
