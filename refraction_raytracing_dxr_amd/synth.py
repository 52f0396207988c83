"""Seeded synthetic inputs for benchmarks and tests (the reference's envmap.hdr is missing from the mount)."""
import os

import numpy as np

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assets")


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
