import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = (
    "import sys, numpy as np\n"
    "sys.path.insert(0, %r)\n"
    "import refraction_raytracing_dxr_amd as rr\n"
    "from refraction_raytracing_dxr_amd.synth import asset, procedural_env\n"
    "r = rr.Renderer(0); out = []; cnt = []\n"
    "for name, kw in (('monkey.obj', dict(max_refract=8)), ('sphere.obj', dict(max_refract=4, max_reflect=1)), ('shell.obj', dict(max_reflect=4)), ('cube.obj', dict())):\n"
    "    m = rr.Mesh(); m.load(asset(name))\n"
    "    r.load_scene(m.verts, m.indices, procedural_env(256, 128, seed=9))\n"
    "    for depth, frames in ((1, 2), (5, 5), (40, 40)):\n"
    "        r.render_orbit(323, 181, frames, angle=0.3, params=rr.default_params(flags=rr.DISPATCH_COLLECT_STATS | rr.DISPATCH_FLOAT_OUTPUT, **kw), frames_per_dispatch=depth)\n"
    "        rgba, f32 = r.read_frame(want_float=True, slice=depth - 1)\n"
    "        st = r.stats(); assert st.traversal_overflow == 0\n"
    "        out += [rgba.view(np.uint32)[..., 0].astype(np.float64), f32[..., 0].astype(np.float64), f32[..., 2].astype(np.float64)]\n"
    "        cnt += [st.rays, st.hits, st.misses, st.terminal_hits, st.tir, st.node_visits, st.tri_tests, st.pixels, st.render_kernel]\n"
    "    r.set_tile_partition(1, 3)\n"
    "    r.render_orbit(323, 181, 3, angle=0.3, params=rr.default_params(**kw), frames_per_dispatch=3)\n"
    "    cnt += [r.stats().rays]\n"
    "    r.set_tile_partition(0, 1)\n"
    "np.save(sys.argv[1], np.stack(out)); print(' '.join(str(c) for c in cnt))\n") % ROOT
import numpy as np
res = {}
for k in ("fused", "paths"):
    env = dict(os.environ, RR_DEBUG_KERNEL=k)
    p = subprocess.run([sys.executable, "-c", code, "/tmp/%s.npy" % k], capture_output=True, text=True, env=env, timeout=600)
    print(k, p.returncode, p.stderr[-500:])
    res[k] = np.load("/tmp/%s.npy" % k)
for i in range(len(res["fused"])):
    d = res["fused"][i] != res["paths"][i]
    if d.any():
        ys, xs = np.nonzero(d)
        print("array", i, "differs in", int(d.sum()), "pixels; first", xs[0], ys[0], res["fused"][i][ys[0], xs[0]], res["paths"][i][ys[0], xs[0]])
