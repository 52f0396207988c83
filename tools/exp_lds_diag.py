"""Where the waves of k_render_lds spend their time: RR_DEBUG_DIAG build, per-wave cycles in ticket draws and in blocks.
  python3 tools/exp_lds_diag.py [mesh] [refract] [depth]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["RR_DEBUG_DIAG"] = "/tmp/diag_lds.bin"
os.environ["RR_DEBUG_KERNEL"] = "lds"
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
refr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 64
r = rr.Renderer(0)
m = rr.Mesh(); assert m.load(asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=refr, max_reflect=2, flags=rr.DISPATCH_TIME_KERNEL)
for _ in range(2):
    r.render_orbit(1920, 1080, depth, angle=0.01, params=p, frames_per_dispatch=depth); r.wait()
ms, n = r.kernel_time()
d = np.fromfile("/tmp/diag_lds.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 3] > 0]
wait, rend, total = d[:, 0].astype(float), d[:, 1].astype(float), d[:, 3].astype(float)
tick, blocks = (d[:, 2] & np.uint64(0xffffffff)).astype(float), (d[:, 2] >> np.uint64(32)).astype(float)
print("%s refract %d depth %d: kernel %.1f us/frame (diag build), %d waves" % (name, refr, depth, ms / n * 1e3 / depth, len(d)))
print("wave life (cycles): mean %.0f p50 %.0f p99 %.0f max %.0f" % (total.mean(), *np.percentile(total, [50, 99]), total.max()))
print("share of wave life: ticket draws %.1f %%, blocks %.1f %%, rest (prologue, index math) %.1f %%" % (
    100 * wait.sum() / total.sum(), 100 * rend.sum() / total.sum(), 100 * (total - wait - rend).sum() / total.sum()))
print("tickets %.0f (%.0f cycles each), blocks %.0f (%.0f cycles each); per wave: tickets mean %.0f max %.0f, blocks mean %.0f max %.0f" % (
    tick.sum(), wait.sum() / tick.sum(), blocks.sum(), rend.sum() / max(blocks.sum(), 1), tick.mean(), tick.max(), blocks.mean(), blocks.max()))
idle_end = (total.max() - total)
print("idle at the end (max life - own life): mean %.0f cycles = %.1f %% of the launch" % (idle_end.mean(), 100 * idle_end.mean() / total.max()))
worst, wt = d[:, 4].astype(float), d[:, 5]
wI, wL, wS = (wt & np.uint64(0xfffff)).astype(float), ((wt >> np.uint64(20)) & np.uint64(0xfffff)).astype(float), (wt >> np.uint64(40)).astype(float)
k = np.argsort(worst)[-5:]
print("longest blocks: cycles %s trips I %s L %s passes %s => cycles per trip %s" % (worst[k], wI[k], wL[k], wS[k], np.round(worst[k] / np.maximum(wI[k] + wL[k], 1))))
pro = d[:, 6].astype(float)
print("prologue (node copy + barrier) cycles: mean %.0f p50 %.0f max %.0f; those waves lived %s" % (pro.mean(), np.median(pro), pro.max(), total[k]))
