"""Where a launch of k_render_lds spends its waves' time (diagnostic build, RR_DEBUG_DIAG): per wave, cycles drawing tickets,
cycles rendering blocks, the node copy, its longest block, and how long before the launch's end it ran out of work.
python tools/exp_diag_lds.py [mesh] [depth] [refract]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["RR_DEBUG_DIAG"] = "/tmp/diag_lds.bin"
os.environ["RR_DEBUG_KERNEL"] = "lds"
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
refr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
r = rr.Renderer(0)
m = rr.Mesh(); m.load(asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
cams = [rr.camera_orbit(0.01 * (k + 1)) for k in range(depth)]
for _ in range(3):
    r.dispatch_rays_batch(1920, 1080, cams, rr.default_params(max_refract=refr)); r.wait()
d = np.fromfile("/tmp/diag_lds.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 3] > 0].astype(np.float64)
wait, render, total, worst, copy = d[:, 0], d[:, 1], d[:, 3], d[:, 4], d[:, 6]
tickets = (d[:, 2].astype(np.uint64) & np.uint64(0xffffffff)).astype(float); blocks = (d[:, 2].astype(np.uint64) >> np.uint64(32)).astype(float)
T = total.max()
us = lambda c: c / 100.0          # in units of 100 ticks of s_memtime (the shares below are what the tool is for)
print("%s Depth %d: %d waves; launch %.0f x100 ticks (longest wave)" % (name, depth, len(d), us(T)))
print("  per wave, x100 ticks: node copy %.1f | tickets %.1f (%.1f draws) | blocks %.1f (%.1f blocks) | other %.1f | idle before the launch ends mean %.1f p50 %.1f p90 %.1f max %.1f" % (
    us(copy.mean()), us(wait.mean()), tickets.mean(), us(render.mean()), blocks.mean(), us((total - wait - render - copy).mean()),
    us((T - total).mean()), us(np.percentile(T - total, 50)), us(np.percentile(T - total, 90)), us((T - total).max())))
print("  longest block of a wave: mean %.1f x100 ticks p90 %.1f p99 %.1f max %.1f" % (us(worst.mean()), us(np.percentile(worst, 90)), us(np.percentile(worst, 99)), us(worst.max())))
print("  share of wave time: rendering %.3f tickets %.3f copy %.3f idle-at-end %.3f" % (render.sum() / (T * len(d)), wait.sum() / (T * len(d)), copy.sum() / (T * len(d)), (T - total).sum() / (T * len(d))))
