"""Per-wave timeline of one Depth-1 launch of the path-parallel kernel (k_render_paths, diagnostic build):
    python tools/exp_diag_paths.py [mesh] [refract] [reflect]
Every wave records its start and end (s_memrealtime, 100 MHz) and, per ray level, the lanes alive and the time the level began."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["RR_DEBUG_DIAG"] = "/tmp/diag_paths.bin"
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
name = sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"
refr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
refl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
r = rr.Renderer(0)
m = rr.Mesh(); assert m.load(asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
r.set_camera(rr.camera_orbit(0.01))
for k in range(4):
    r.set_camera(rr.camera_orbit(0.01 * (k + 1)))
    r.dispatch_rays(1920, 1080, rr.default_params(max_refract=refr, max_reflect=refl))
    r.wait()
d = np.fromfile("/tmp/diag_paths.bin", dtype=np.uint64)
d = d[: len(d) // 12 * 12].reshape(-1, 12)
d = d[d[:, 1] > 0]
t0, t1, pp = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64), d[:, 2] == 1
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0          # us
lv = d[:, 4:12].copy().view(np.uint32).reshape(-1, 16)
alive = (lv & 0xff).astype(int)
lvt = (lv >> 8).astype(np.int64)
print("%s %d/%d: %d waves (%d in path-parallel blocks); kernel span %.1f us" % (name, refr, refl, len(d), pp.sum(), t1.max()))
for nm, sel in (("path-parallel", pp), ("background", ~pp)):
    if sel.sum() == 0: continue
    print("  %-13s start p50 %.1f p90 %.1f max %.1f us | duration mean %.1f p90 %.1f p99 %.1f max %.1f us | end max %.1f" % (
        nm, *np.percentile(t0[sel], [50, 90]), t0[sel].max(), (t1 - t0)[sel].mean(), *np.percentile((t1 - t0)[sel], [90, 99]), (t1 - t0)[sel].max(), t1[sel].max()))
order = np.argsort(t1)[::-1][:8]
print("  the eight waves that end last:")
for w in order:
    # level start times relative to the wave's start (the 24-bit field wraps every 0.17 s: differences only)
    ts = [((int(lvt[w, k]) - (int(d[w, 0]) & 0xffffff)) & 0xffffff) / 100.0 if alive[w, k] else -1 for k in range(16)]
    print("    block %5d start %6.1f end %6.1f | alive per level %s | level starts (us after wave start) %s" % (
        int(d[w, 3]), t0[w], t1[w], alive[w, :11].tolist(), [round(x, 1) for x in ts[:11]]))
dur = t1 - t0
order = np.argsort(dur)[::-1][:5]
print("  the five longest waves:")
for w in order:
    ts = [((int(lvt[w, k]) - (int(d[w, 0]) & 0xffffff)) & 0xffffff) / 100.0 if alive[w, k] else -1 for k in range(16)]
    print("    block %5d start %6.1f end %6.1f | alive per level %s | level starts %s" % (int(d[w, 3]), t0[w], t1[w], alive[w, :11].tolist(), [round(x, 1) for x in ts[:11]]))
# occupancy over time: waves resident
edges = np.arange(0, t1.max() + 10, 10.0)
res = [int(((t0 < e + 10) & (t1 > e)).sum()) for e in edges]
print("  resident waves per 10 us:", res)
# alive lanes by level over all pp waves
print("  path-parallel waves with any lane alive, by level:", [(int((alive[pp][:, k] > 0).sum()), round(float(alive[pp][:, k][alive[pp][:, k] > 0].mean()) if (alive[pp][:, k] > 0).any() else 0, 1)) for k in range(11)])
