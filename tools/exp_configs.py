"""BASELINE.json configs on one MI355X: python tools/exp_configs.py  (kernel time via HIP events, Depth 16 and 1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env

def xf(tx, ty, tz, s=1.0):
    m = np.eye(4, dtype=np.float32)[:3] * np.float32(s); m[:, 3] = (tx, ty, tz); return m

def load(n):
    m = rr.Mesh(); assert m.load(O.asset(n)); return m

env = procedural_env(2048, 1024, seed=0)
r = rr.Renderer(0)
cases = [
    ("C1 sphere 256x256 refract 1", ["sphere.obj"], None, 256, 256, 1, 1.0),
    ("C2 sphere 1920x1080 refract 4", ["sphere.obj"], None, 1920, 1080, 4, 1.0),
    ("C3 monkey 1920x1080 refract 8", ["monkey.obj"], None, 1920, 1080, 8, 1.0),
    ("   shell 1024x768 refract 5 (the demo)", ["shell.obj"], None, 1024, 768, 5, 1.0),
    ("   ott 1920x1080 refract 8", ["ott.obj"], None, 1920, 1080, 8, 1.0),
    ("C4 shell+cube+ott 3840x2160 refract 8", ["shell.obj", "cube.obj", "ott.obj"],
     ([xf(0, 0, 0), xf(0, 0, -4.0), xf(0, 0, 4.0)], [0, 1, 2]), 3840, 2160, 8, 1.6),
    ("C5 monkey x1024 3840x2160 refract 16", ["monkey.obj"],
     ([xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)], [0] * 1024), 3840, 2160, 16, 14.0),
]
for label, names, inst, W, H, refr, radius in cases:
    meshes = [load(n) for n in names]
    ids = []
    for m in meshes:
        mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid); ids.append(mid)
    if inst is None:
        r.build_tlas(rr.make_instances(meshes=[ids[0]]))
    else:
        r.build_tlas(rr.make_instances(transforms=inst[0], meshes=[ids[k] for k in inst[1]]))
    r.upload_envmap(env)
    cams = []
    for k in range(16):
        sc = rr.camera_orbit(0.01 * (k + 1))
        sc.camera_loc[0] *= radius; sc.camera_loc[2] *= radius
        if radius > 2: sc.camera_loc[1] = 0.8 * radius
        cams.append(sc)
    out = []
    for depth in (16, 1):
        p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
        for rep in range(5):        # (the kernel choice is measured on the second and third launch of a shape)
            if depth == 1:
                for c in cams[:4]:
                    r.set_camera(c); r.dispatch_rays(W, H, p)
            else:
                r.dispatch_rays_batch(W, H, cams, p)
            if rep == 1: r.kernel_time()
        ms, n = r.kernel_time()
        st = r.stats()
        frames = 1 if depth == 1 else 16
        rays_per_frame = st.rays / frames
        us = ms / n * 1e3 / frames
        out.append("depth %2d: %8.1f us/frame %7.2f Grays/s" % (depth, us, rays_per_frame / us / 1e3))
    print("%-42s %.2f Mrays/frame  bvh depth %2d | %s | %s" % (label, rays_per_frame / 1e6, st.bvh_depth, out[0], out[1]), flush=True)
r.close()
