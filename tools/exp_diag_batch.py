"""Per-wave timeline of one DispatchRays(W, H, Depth) launch of k_render_fused (diagnostic build):
    python tools/exp_diag_batch.py [mesh] [depth] [refract]
Every wave records its start and end on the 100 MHz clock all CUs share, its loop trips and the longest ray chain of its lanes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["RR_DEBUG_DIAG"] = "/tmp/diag_batch.bin"
os.environ.setdefault("RR_DEBUG_KERNEL", "fused")
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
refr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
r = rr.Renderer(0)
m = rr.Mesh(); assert m.load(asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
cams = [rr.camera_orbit(0.01 * (k + 1)) for k in range(depth)]
for _ in range(2):
    r.dispatch_rays_batch(1920, 1080, cams, rr.default_params(max_refract=refr))
    r.wait()
d = np.fromfile("/tmp/diag_batch.bin", dtype=np.uint64)
d = d[: len(d) // 4 * 4].reshape(-1, 4)
d = d[d[:, 1] > 0]
t0 = (d[:, 2] >> np.uint64(32)).astype(np.int64); t1 = (d[:, 3] >> np.uint64(32)).astype(np.int64)
trips = (d[:, 3] & np.uint64(0xffffffff)).astype(np.int64); rays = (d[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0
span = t1.max()
print("%s Depth %d, %d bounces: %d waves, launch span %.1f us = %.1f us per frame" % (name, depth, refr, len(d), span, span / depth))
heavy = trips > 100
print("  waves with > 100 loop trips: %d, duration mean %.1f max %.1f us; start of the last one %.1f us; end of the last one %.1f us" % (
    heavy.sum(), (t1 - t0)[heavy].mean(), (t1 - t0)[heavy].max(), t0[heavy].max(), t1[heavy].max()))
print("  light waves: %d, duration mean %.2f p99 %.2f us; start of the last one %.1f us" % ((~heavy).sum(), (t1 - t0)[~heavy].mean(), np.percentile((t1 - t0)[~heavy], 99), t0[~heavy].max()))
n = 40
edges = np.linspace(0, span, n + 1)
res = [(int(((t0 < edges[k + 1]) & (t1 > edges[k]) & heavy).sum()), int(((t0 < edges[k + 1]) & (t1 > edges[k]) & ~heavy).sum())) for k in range(n)]
print("  waves resident per %.0f us (heavy, light):" % (span / n), res)
# work conservation: sum of trips of waves running in each interval (trips spread evenly over a wave's life)
rate = trips / np.maximum(t1 - t0, 1e-3)
work = [float((rate * np.clip(np.minimum(t1, edges[k + 1]) - np.maximum(t0, edges[k]), 0, None)).sum()) for k in range(n)]
print("  loop trips per interval (k):", [int(w / 1e3) for w in work])
w = np.argsort(t1)[::-1][:6]
print("  last waves to end: start %s end %s trips %s rays %s" % (np.round(t0[w], 1), np.round(t1[w], 1), trips[w], rays[w]))
