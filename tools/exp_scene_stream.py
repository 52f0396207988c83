"""k_render_scene_stream (RR_DEBUG_KERNEL=scene-stream, RR_EXPERIMENTAL=1 builds) against the product's lock-step two-level kernel:
frames and counters of a small instanced scene, then kernel time on C5 (monkey x 1024, 2160p, 16 bounces).
    python tools/exp_scene_stream.py [leaf,shade thresholds ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env

def xf(tx, ty, tz, s=1.0, rot=0.0):
    c, sn = np.cos(rot), np.sin(rot)
    m = np.array([[c * s, 0, sn * s, tx], [0, s, 0, ty], [-sn * s, 0, c * s, tz]], np.float32)
    return m

def make(kernel, thr=None, waves=None):
    for k in ("RR_DEBUG_KERNEL", "RR_DEBUG_ASYNC", "RR_DEBUG_STREAM_WAVES"): os.environ.pop(k, None)
    if kernel: os.environ["RR_DEBUG_KERNEL"] = kernel
    if thr: os.environ["RR_DEBUG_ASYNC"] = thr
    if waves: os.environ["RR_DEBUG_STREAM_WAVES"] = str(waves)
    return rr.Renderer(0)

m = rr.Mesh(); assert m.load(asset("monkey.obj"))
env = procedural_env(512, 256, seed=3)
# ---- parity: 3x3 instances, some rotated and scaled, odd frame size, several slices
res = {}
for tag, kern in (("lockstep", None), ("stream", "scene-stream")):
    r = make(kern)
    mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid)
    T = [xf(3.0 * (i - 1), 0.2 * j, 3.0 * (j - 1), 1.0 + 0.1 * i, 0.3 * (i + j)) for i in range(3) for j in range(3)]
    r.build_tlas(rr.make_instances(transforms=T, meshes=[mid] * 9))
    r.upload_envmap(env)
    cams = []
    for k in range(5):
        sc = rr.camera_orbit(0.3 + 0.05 * k); sc.camera_loc[0] *= 2.5; sc.camera_loc[2] *= 2.5; sc.camera_loc[1] = 2.0
        cams.append(sc)
    r.dispatch_rays_batch(333, 201, cams, rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS | rr.DISPATCH_FLOAT_OUTPUT))
    frames = [r.read_frame(want_float=True, slice=k) for k in range(5)]
    st = r.stats()
    res[tag] = (frames, (st.rays, st.hits, st.misses, st.terminal_hits, st.tir, st.node_visits, st.tri_tests, st.pixels), st.render_kernel)
    r.close()
ok = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) for a, b in zip(res["lockstep"][0], res["stream"][0]))
print("parity: frames identical", ok, "| counters identical", res["lockstep"][1] == res["stream"][1], "| kernels", res["lockstep"][2], res["stream"][2], flush=True)
if res["lockstep"][1] != res["stream"][1]: print(res["lockstep"][1], res["stream"][1])

# ---- C5 timing
env2 = procedural_env(2048, 1024, seed=0)
def c5(kern, thr=None, waves=None):
    r = make(kern, thr, waves)
    mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid)
    r.build_tlas(rr.make_instances(transforms=[xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)], meshes=[mid] * 1024))
    r.upload_envmap(env2)
    cams = []
    for k in range(16):
        sc = rr.camera_orbit(0.01 * (k + 1)); sc.camera_loc[0] *= 14; sc.camera_loc[2] *= 14; sc.camera_loc[1] = 11.2
        cams.append(sc)
    W, H = 3840, 2160
    out = []
    r.set_camera(cams[0])
    r.dispatch_rays(W, H, rr.default_params(max_refract=16, flags=rr.DISPATCH_COLLECT_STATS))
    st = r.stats()
    trips = "node %.1fM leaf %.1fM pass %.2fM" % (st.node_trips / 1e6, st.leaf_trips / 1e6, st.shade_passes / 1e6)
    for depth in (16, 1):
        p = rr.default_params(max_refract=16, flags=rr.DISPATCH_TIME_KERNEL)
        for rep in range(2):
            if depth == 1:
                for c in cams[:3]:
                    r.set_camera(c); r.dispatch_rays(W, H, p)
            else:
                r.dispatch_rays_batch(W, H, cams, p)
            if rep == 0: r.kernel_time()
        ms, n = r.kernel_time()
        rays = r.stats().rays / (1 if depth == 1 else 16)
        us = ms / n * 1e3 / (1 if depth == 1 else 16)
        out.append("D%d %7.1f us %5.2f Grays/s" % (depth, us, rays / us / 1e3))
    print("%-12s thr %-5s waves %s | %s | %s" % (kern or "lockstep", thr or "-", waves or "-", " | ".join(out), trips), flush=True)
    r.close()
c5(None)
for thr in (sys.argv[1:] or ["2,2", "4,4", "3,5"]):
    c5("scene-stream", thr, 7)
c5("scene-stream", "4,4", 6)
c5("scene-stream", "4,4", 5)
