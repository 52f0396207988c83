"""One-off soak for the two-level path: random scenes of 1-10 instances of 1-3 meshes under random affine transforms
(rotation, non-uniform scale, translation), instance flags and masks; GPU vs the oracle's path-weight mode, bit for bit.
usage: python tools/soak_instances.py [n_frames] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = ("cube.obj", "sphere.obj", "monkey.obj", "shell.obj")
meshes = []
for name in names:
    m = rr.Mesh(); m.load(asset(name)); meshes.append(m)
env = procedural_env(256, 128, seed=4)
r = rr.Renderer(0)
ids = [r.upload_mesh(m.verts, m.indices) for m in meshes]
for i in ids: r.build_blas(i)
r.upload_envmap(env)
def rand_xf():
    a = rng.normal(size=(3, 3)); q, _ = np.linalg.qr(a)
    if rng.random() < 0.3: q = np.eye(3)
    s = np.diag(rng.uniform(0.4, 1.6, 3)) if rng.random() < 0.5 else np.eye(3) * rng.uniform(0.5, 1.5)
    m = np.zeros((3, 4), np.float32); m[:, :3] = (q @ s).astype(np.float32); m[:, 3] = rng.uniform(-3, 3, 3)
    return m
bad = 0; t0 = time.time()
for k in range(n):
    ni = int(rng.integers(1, 11))
    which = rng.integers(0, len(meshes), ni)
    xs = [rand_xf() for _ in range(ni)]
    flags = [int(rng.choice([0, 0, 0, 1, 2])) for _ in range(ni)]
    masks = [int(rng.choice([1, 1, 1, 0xff, 0])) for _ in range(ni)]
    inst = rr.make_instances(transforms=xs, meshes=[ids[w] for w in which], masks=masks, flags=flags)
    r.build_tlas(inst)
    W, H = int(rng.integers(8, 320)), int(rng.integers(8, 200))
    kw = dict(max_refract=int(rng.integers(0, 10)), max_reflect=int(rng.integers(0, 3)))
    sc = rr.camera_orbit(float(rng.uniform(0, 6.3))); sc.camera_loc[0] *= 1.6; sc.camera_loc[2] *= 1.6
    r.set_camera(sc)
    r.dispatch_rays(W, H, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT, **kw))
    rgba, f32 = r.read_frame(want_float=True)
    s = O.Scene()
    for m in meshes: s.add_mesh(m.verts, m.indices)
    oi = np.zeros(ni, O.INSTANCE_DTYPE)
    oi["transform"] = inst["transform"]; oi["id_mask"] = inst["instance_id_mask"]; oi["hitgroup_flags"] = inst["hitgroup_flags"]
    oi["blas"] = which
    s.set_instances(oi); s.set_envmap(env)
    pw = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, accum_mode=1, **kw))
    ok = np.array_equal(f32[..., :3].view(np.uint32), pw["rgb"].view(np.uint32)) and np.array_equal(rgba, pw["rgba8"]) and r.stats().rays == pw["stats"].rays
    if not ok:
        bad += 1
        print("MISMATCH frame", k, ni, W, H, kw, int((rgba != pw["rgba8"]).any(-1).sum()), "pixels", flush=True)
    if k % 50 == 49:
        print("%d frames, %d mismatches, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
print("done: %d frames, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
