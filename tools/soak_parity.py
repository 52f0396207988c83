"""One-off soak: random (mesh, camera angle, frame size, bounce limits, ior) frames, GPU vs the oracle's path-weight mode,
bit for bit.  usage: python tools/soak_parity.py [n_frames] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
r = rr.Renderer(0)
meshes = {}
for name in ("cube.obj", "sphere.obj", "monkey.obj", "shell.obj", "ott.obj"):
    m = rr.Mesh(); m.load(asset(name)); meshes[name] = m
env = procedural_env(512, 256, seed=11)
bad = 0
t0 = time.time()
for k in range(n):
    name = rng.choice(list(meshes))
    m = meshes[name]
    W, H = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    ang = float(rng.uniform(0, 6.3))
    kw = dict(max_refract=int(rng.integers(0, 12)), max_reflect=int(rng.integers(0, 4)), ior=float(rng.choice([1.3, 1.5, 1.05, 0.9, 2.4])))
    if k % 7 == 3:  # a larger frame with the camera pulled back: the mesh is small on screen, Depth-1 launches take k_render_paths
        W, H = int(rng.integers(500, 1300)), int(rng.integers(300, 800))
        kw["max_reflect"] = int(rng.integers(0, 3))
    tone = int(rng.integers(0, 2))
    gflags = rr.DISPATCH_FLOAT_OUTPUT | (rr.DISPATCH_TONEMAP_REINHARD if tone else 0)
    r.load_scene(m.verts, m.indices, env)
    sc = rr.camera_orbit(ang)
    if k % 7 == 3:
        sc.camera_loc[0] *= 2.5; sc.camera_loc[2] *= 2.5
    r.set_camera(sc)
    if k % 2:       # every other frame goes through a Depth-3 batch: the high-occupancy builds (8 waves, 16-bit stacks)
        r.dispatch_rays_batch(W, H, [rr.camera_orbit(ang + 1.0), sc, rr.camera_orbit(ang + 2.0)], rr.default_params(flags=gflags, **kw))
        rgba, f32 = r.read_frame(want_float=True, slice=1)
        rays_gpu = None
    else:
        r.dispatch_rays(W, H, rr.default_params(flags=gflags, **kw))
        rgba, f32 = r.read_frame(want_float=True)
        rays_gpu = r.stats().rays
    s = O.Scene(); s.add_mesh(m.verts, m.indices); s.set_envmap(env)
    pw = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, accum_mode=1, tonemap=tone, **kw))
    ok = np.array_equal(f32[..., :3].view(np.uint32), pw["rgb"].view(np.uint32)) and np.array_equal(rgba, pw["rgba8"]) and (rays_gpu is None or rays_gpu == pw["stats"].rays)
    if not ok:
        bad += 1
        print("MISMATCH", name, W, H, ang, kw, int((rgba != pw["rgba8"]).any(-1).sum()), "pixels", flush=True)
    if k % 20 == 19:
        print("%d frames, %d mismatches, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
print("done: %d frames, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
