"""rank 0's de-interleave of one gathered batch (64 frames, 8 ranks): RGBA8 vs RGB8 tiles, kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
from refraction_raytracing_dxr_amd.dist import max_local_tiles, TILE_BYTES, TILE_BYTES_RGB8
W, H, F = 1920, 1080, 64
m = rr.Mesh(); m.load(asset("cube.obj"))
r = rr.Renderer(0)
r.set_stream(torch.cuda.current_stream().cuda_stream)
r.load_scene(m.verts, m.indices, procedural_env(64, 32, seed=0))
r.set_camera(rr.camera_orbit(0.01)); r.dispatch_rays(W, H)                      # (frame size for the context)
frames = torch.zeros(F * H * W * 4, dtype=torch.uint8, device="cuda")
for world in (2, 8):
    mx = max_local_tiles(W, H, world)
    for rgb8, tb in ((False, TILE_BYTES), (True, TILE_BYTES_RGB8)):
        fb = mx * tb
        g = torch.randint(0, 255, (world * F * fb,), dtype=torch.uint8, device="cuda")
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.assemble_frames(g.data_ptr(), world, F * fb, fb, F, W, H, frames.data_ptr(), H * W * 4, rgb8=rgb8)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        gb = (world * F * fb + F * H * W * 4) / 1e9
        print("world %d %s: %.3f ms per %d frames (%.1f us/frame, %.2f TB/s)" % (world, "RGB8 " if rgb8 else "RGBA8", dt * 1e3, F, dt / F * 1e6, gb / dt / 1e3))
