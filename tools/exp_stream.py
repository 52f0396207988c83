"""rr_render_orbit_to_host: frames per second delivered to (page-locked) host memory, monkey 1080p 8/2 bounces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
W, H, N = 1920, 1080, 256
m = rr.Mesh(); m.load(asset(sys.argv[1] if len(sys.argv) > 1 else "monkey.obj"))
r = rr.Renderer(0)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
p = rr.default_params(max_refract=8, max_reflect=2)
import numpy as np, ctypes as C
out = np.empty((N, H, W, 4), np.uint8)
for pin in (False, True):
    if pin: r._ck(r._L.rr_host_register(r._h, out.ctypes.data, out.nbytes), "pin")
    for F in (1, 4, 16, 64):
        for fl in (2, 3):
            r.set_frames_in_flight(fl)
            a = C.c_float(0.01)
            for rep in range(2):
                a = C.c_float(0.01); t0 = time.perf_counter()
                r._ck(r._L.rr_render_orbit_to_host(r._h, W, H, C.byref(p), C.byref(a), 0.01, N, F, rr.host.FOV_Y, rr.host.ASPECT, 1.0, 125.0, out.ctypes.data), "x")
                dt = time.perf_counter() - t0
            print("pinned %d  F %2d  regions %d: %.0f fps to host (%.1f GB/s), %.1f us/frame" % (pin, F, fl, N / dt, N * W * H * 4 / dt / 1e9, dt / N * 1e6), flush=True)
    if pin: r._L.rr_host_unregister(r._h, out.ctypes.data)
