"""Seeded synthetic inputs for benchmarks and tests (the reference's envmap.hdr is missing from the mount)."""
import os

import numpy as np

# package data: byte-identical copies of the reference's data files (cube / sphere / monkey / shell / ott .obj, envmap.png)
ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def asset(name):
    """path of one of the reference's data files (cube/sphere/monkey/shell/ott .obj, envmap.png) kept as fixtures"""
    return os.path.join(ASSETS, name)


def procedural_env(w=256, h=128, seed=0, peak=16.0):
    """seeded HDR 'studio': smooth gradient + bright soft boxes, values in [0, peak]"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = x / w, y / h
    base = 0.25 + 0.5 * (1.0 - v)[..., None] * np.array([0.9, 0.95, 1.0], np.float32)
    img = np.broadcast_to(base, (h, w, 3)).copy()
    for _ in range(6):
        cu, cv = rng.uniform(0, 1), rng.uniform(0.05, 0.6)
        su, sv = rng.uniform(0.02, 0.08), rng.uniform(0.02, 0.08)
        amp = rng.uniform(2.0, peak)
        col = rng.uniform(0.7, 1.0, 3).astype(np.float32)
        du = np.minimum(np.abs(u - cu), 1 - np.abs(u - cu))
        m = np.exp(-0.5 * ((du / su) ** 2 + ((v - cv) / sv) ** 2)).astype(np.float32)
        img += amp * m[..., None] * col
    img += rng.uniform(0, 0.02, img.shape).astype(np.float32)
    return np.clip(img, 0, peak).astype(np.float32)


def subdivide(verts, levels=1):
    """Midpoint subdivision of an un-indexed triangle list (the layout Mesh::load produces, Mesh.cpp:24-31):
    every triangle becomes four, positions/uvs are edge midpoints (the surface does not move), normals are the
    re-normalised midpoints.  monkey.obj x2 levels = 15 472 triangles, BASELINE's '~16k tri Suzanne' (SURVEY 8d).
    Returns (verts, indices) with indices the identity, as the loader does."""
    from ._capi import VERTEX_DTYPE
    v = np.asarray(verts)
    for _ in range(levels):
        t = v.reshape(-1, 3)
        a, b, c = t[:, 0], t[:, 1], t[:, 2]

        def mid(p, q):
            m = np.zeros(p.shape, VERTEX_DTYPE)
            m["position"] = (p["position"] + q["position"]) * np.float32(0.5)
            m["uv"] = (p["uv"] + q["uv"]) * np.float32(0.5)
            n = p["norm"] + q["norm"]
            ln = np.sqrt((n * n).sum(-1, keepdims=True, dtype=np.float32))
            m["norm"] = np.where(ln > 0, n / np.where(ln > 0, ln, 1), p["norm"]).astype(np.float32)
            return m
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        out = np.empty((t.shape[0], 4, 3), VERTEX_DTYPE)
        out[:, 0, 0], out[:, 0, 1], out[:, 0, 2] = a, ab, ca
        out[:, 1, 0], out[:, 1, 1], out[:, 1, 2] = ab, b, bc
        out[:, 2, 0], out[:, 2, 1], out[:, 2, 2] = ca, bc, c
        out[:, 3, 0], out[:, 3, 1], out[:, 3, 2] = ab, bc, ca
        v = out.reshape(-1)
    return np.ascontiguousarray(v), np.arange(v.shape[0], dtype=np.uint32)
