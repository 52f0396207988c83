import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
W, H, N = 1920, 1080, 64
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8, max_reflect=2)
out = np.zeros((N, H, W, 4), np.uint8)
r._ck(r._L.rr_host_register(r._h, out.ctypes.data, out.nbytes), "pin")
r.set_frames_in_flight(3)
a = C.c_float(0.01)
for rep in range(8):
    t0 = time.perf_counter()
    r._ck(r._L.rr_render_orbit_to_host(r._h, W, H, C.byref(p), C.byref(a), 0.01, N, 4, rr.host.FOV_Y, rr.host.ASPECT, 1.0, 125.0, out.ctypes.data), "x")
    dt = time.perf_counter() - t0
    print("chunk %d: %.2f ms (%.0f fps)" % (rep, dt * 1e3, N / dt), flush=True)
