"""Soak for launches of many slices (24-72: the launch shapes k_render_lds renders by default, and the kernel-choice
measurements that go with them): random (mesh, orbit start, frame size, bounce limits, ior, launch depth) through
rr_render_orbit, up to four slices of every launch (three random ones and the last) against the oracle's path-weight mode, bit
for bit.  A scene is kept for several launches now and then, so that measured kernel choices render too; the tally of kernels
is printed.  usage: python tools/soak_batches.py [n_launches] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
r = rr.Renderer(0)
meshes = {}
for name in ("cube.obj", "sphere.obj", "monkey.obj", "shell.obj", "ott.obj"):
    m = rr.Mesh(); m.load(asset(name)); meshes[name] = m
env = procedural_env(512, 256, seed=11)
bad = 0
kernels = {}
t0 = time.time()
for k in range(n):
    name = rng.choice(list(meshes))
    m = meshes[name]
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 220))
    ang = float(np.float32(rng.uniform(0, 6.3)))
    depth = int(rng.integers(24, 73))
    kw = dict(max_refract=int(rng.integers(0, 12)), max_reflect=int(rng.integers(0, 4)), ior=float(rng.choice([1.3, 1.5, 1.05, 0.9, 2.4])))
    tone = int(rng.integers(0, 2))
    gflags = rr.DISPATCH_FLOAT_OUTPUT | (rr.DISPATCH_TONEMAP_REINHARD if tone else 0)
    if k == 0 or rng.integers(0, 3) == 0 or name != last:
        r.load_scene(m.verts, m.indices, env)      # (otherwise the scene and its kernel choices stay: later launches run measured choices)
    last = name
    reps = int(rng.integers(1, 4))                  # the second and third launch of a shape are where its kernel is measured
    for _ in range(reps):
        r.render_orbit(W, H, depth, angle=ang, params=rr.default_params(flags=gflags, **kw), frames_per_dispatch=depth)
    kern = r.stats().render_kernel_name.decode().split("<")[0]
    kernels[kern] = kernels.get(kern, 0) + 1
    s = O.Scene(); s.add_mesh(m.verts, m.indices); s.set_envmap(env)
    a = np.float32(ang)
    angles = []
    for j in range(depth):
        angles.append(a); a = np.float32(a + np.float32(0.01))
    for j in sorted(set(int(x) for x in rng.integers(0, depth, 3)) | {depth - 1}):
        rgba, f32 = r.read_frame(want_float=True, slice=j)
        sc = rr.camera_orbit(float(angles[j]))
        pw = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, accum_mode=1, tonemap=tone, **kw))
        ok = np.array_equal(f32[..., :3].view(np.uint32), pw["rgb"].view(np.uint32)) and np.array_equal(rgba, pw["rgba8"])
        if not ok:
            bad += 1
            print("MISMATCH", name, W, H, ang, depth, j, kw, kern, int((rgba != pw["rgba8"]).any(-1).sum()), "pixels", flush=True)
    if k % 10 == 9:
        print("%d launches, %d mismatches, %.0f s, kernels %s" % (k + 1, bad, time.time() - t0, kernels), flush=True)
print("done: %d launches, %d mismatches, kernels %s" % (n, bad, kernels))
sys.exit(1 if bad else 0)
