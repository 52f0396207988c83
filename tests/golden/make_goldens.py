"""Regenerates tests/golden/frames.npz from the CPU oracle (brute-force TraceRay, literal recursion).

These are NOT reference outputs (the reference cannot run outside Windows/D3D12 and ships no images);
they pin the oracle itself so that a later edit of oracle/ or of the arithmetic contract shows up as a
diff.  Inputs: the reference's own .obj files and envmap.png, the reference's literals, angle 0.01.
    python tests/golden/make_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O                                   # noqa: E402
import refraction_raytracing_dxr_amd as rr           # noqa: E402  (host-side PNG decode + camera only)

CASES = [("cube.obj", 64, 48, dict()), ("sphere.obj", 64, 64, dict(max_refract=1)), ("monkey.obj", 96, 54, dict(max_refract=8)),
         ("shell.obj", 64, 48, dict()), ("ott.obj", 48, 36, dict(max_refract=8))]


def render(name, w, h, kw):
    env, _ = rr.load_texture(O.asset("envmap.png"), 3)
    v, i = O.mesh_load(O.asset(name))
    s = O.Scene()
    s.add_mesh(v, i)
    s.set_envmap(env)
    M, cam = O.camera(0.01)
    r = s.render(M, cam, w, h, O.default_params(use_bvh=0, **kw))
    return r["rgba8"], r["rgb"], np.array([r["stats"].rays, r["stats"].hits, r["stats"].misses], np.int64)


if __name__ == "__main__":
    out = {}
    for name, w, h, kw in CASES:
        rgba, rgb, cnt = render(name, w, h, kw)
        key = name.split(".")[0]
        out[key + "_rgba8"], out[key + "_rgb"], out[key + "_counts"] = rgba, rgb, cnt
        print(name, w, h, kw, cnt)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
