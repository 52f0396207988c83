"""per-wave timeline of one frame: RR_DEBUG_DIAG=/tmp/diag.bin python tools/exp_diag.py <mesh> <refract> <reflect>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["RR_DEBUG_DIAG"] = "/tmp/diag.bin"
import refraction_raytracing_dxr_amd as rr
import oracle as O
from conftest import procedural_env
name, refr, refl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
r = rr.Renderer(0)
m = rr.Mesh(); m.load(O.asset(name))
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
r.set_camera(rr.camera_orbit(0.01))
for _ in range(3):
    r.dispatch_rays(1920, 1080, rr.default_params(max_refract=refr, max_reflect=refl))
    r.wait()
d = np.fromfile("/tmp/diag.bin", dtype=np.uint64).reshape(-1, 4).astype(np.float64)
d = d[d[:, 1] > 0]
raw0 = np.fromfile("/tmp/diag.bin", dtype=np.uint64).reshape(-1, 4)
raw0 = raw0[raw0[:, 1] > 0][:, 0]
cyc, rays, trips = d[:, 1], d[:, 2], d[:, 3]
if True:
    tS, tL, tI = (raw0 >> np.uint64(40)).astype(float), ((raw0 >> np.uint64(20)) & np.uint64(0xfffff)).astype(float), (raw0 & np.uint64(0xfffff)).astype(float)
    w = np.argmax(cyc)
    print("phase trips: total I/L/S = %.0f / %.0f / %.0f ; worst wave I/L/S = %.0f / %.0f / %.0f" % (tI.sum(), tL.sum(), tS.sum(), tI[w], tL[w], tS[w]))
    st = r.stats() if False else None
print("waves %d" % len(d))
print("wave cycles(ticks): mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (cyc.mean(), *np.percentile(cyc, [50, 90, 99]), cyc.max()))
print("rays/lane max-in-wave: mean %.2f max %d ; trips: mean %.0f p99 %.0f max %.0f" % (rays.mean(), rays.max(), trips.mean(), np.percentile(trips, 99), trips.max()))
heavy = trips > 200
print("heavy waves (>200 trips): %d ; their ticks/trip: mean %.2f  (=> %.0f ns per trip)" % (heavy.sum(), (cyc[heavy] / trips[heavy]).mean(), (cyc[heavy] / trips[heavy]).mean() * 10))
w = np.argsort(cyc)[-5:]
print("5 longest waves: cycles %s trips %s rays %s => cycles/trip %s" % (cyc[w], trips[w], rays[w], np.round(cyc[w]/trips[w])))
print("sum of trips over all waves: %.0f" % trips.sum())
r.dispatch_rays(1920, 1080, rr.default_params(max_refract=refr, max_reflect=refl, flags=rr.DISPATCH_COLLECT_STATS))
st = r.stats()
print("lane utilisation: internal %.1f%% (%.1fM lane visits / %.2fM wave trips), leaf %.1f%% (%.2fM / %.2fM), shading %.1f%% (%.2fM rays / %.3fM passes)" % (
    100 * st.node_visits / (tI.sum() * 64), st.node_visits / 1e6, tI.sum() / 1e6,
    100 * st.tri_tests / (tL.sum() * 64), st.tri_tests / 1e6, tL.sum() / 1e6,
    100 * st.rays / (tS.sum() * 64), st.rays / 1e6, tS.sum() / 1e6))
