import os, sys
sys.path.insert(0, "/root/repo")
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
env = procedural_env(2048, 1024, seed=0)
for kern in ("fused", "paths"):
    os.environ["RR_DEBUG_KERNEL"] = kern
    r = rr.Renderer(0)
    for name in ("monkey.obj", "shell.obj", "sphere.obj"):
        m = rr.Mesh(); m.load(asset(name))
        r.load_scene(m.verts, m.indices, env)
        out = []
        for zoom in (1.6, 1.3, 1.0, 0.8, 0.65):
            cams = []
            for k in range(8):
                sc = rr.camera_orbit(0.01 * (k + 1) * 10); sc.camera_loc[0] *= zoom; sc.camera_loc[1] *= zoom; sc.camera_loc[2] *= zoom
                cams.append(sc)
            p = rr.default_params(max_refract=8, flags=rr.DISPATCH_TIME_KERNEL)
            for rep in range(2):
                for c in cams:
                    r.set_camera(c); r.dispatch_rays(1920, 1080, p)
                if rep == 0: r.kernel_time()
            ms, n = r.kernel_time()
            out.append("zoom %.2f %6.1f us" % (zoom, ms / n * 1e3))
        print("%-6s %-11s %s" % (kern, name, " | ".join(out)), flush=True)
    r.close()
