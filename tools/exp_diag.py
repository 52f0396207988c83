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
t0 = d[:, 0].min()
start, cyc, rays, trips = d[:, 0] - t0, d[:, 1], d[:, 2], d[:, 3]
end = start + cyc
print("waves %d  kernel span %.0f ticks (s_memtime @100MHz => %.1f us)" % (len(d), end.max(), end.max() / 100.0))
print("wave cycles(ticks): mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (cyc.mean(), *np.percentile(cyc, [50, 90, 99]), cyc.max()))
print("rays/lane max-in-wave: mean %.2f max %d ; trips: mean %.0f p99 %.0f max %.0f" % (rays.mean(), rays.max(), trips.mean(), np.percentile(trips, 99), trips.max()))
heavy = trips > 200
print("heavy waves (>200 trips): %d ; their ticks/trip: mean %.2f  (=> %.0f ns per trip)" % (heavy.sum(), (cyc[heavy] / trips[heavy]).mean(), (cyc[heavy] / trips[heavy]).mean() * 10))
order = np.argsort(end)
print("last 5 waves to finish: end(us) %s  trips %s rays %s" % (np.round(end[order[-5:]] / 100, 1), trips[order[-5:]], rays[order[-5:]]))
# concurrency over time
edges = np.linspace(0, end.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    active = ((start < b) & (end > a)).sum()
    print("  t %6.1f-%6.1f us: %5d waves alive" % (a / 100, b / 100, active))
