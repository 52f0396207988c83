"""Does overlapping two render launches (lanes) recover the tail a 1/world share of a launch exposes?
One process, one GPU, tile partition (0, world): the load one rank of an N-GPU run sees, no gather.
usage: python tools/exp_lanes.py [world] [F]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd.synth import asset, procedural_env
from refraction_raytracing_dxr_amd.dist import max_local_tiles, TILE_BYTES

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 1920, 1080
m = rr.Mesh(); m.load(asset("monkey.obj"))
r = rr.Renderer(0)
r.set_stream(torch.cuda.current_stream().cuda_stream)
r.load_scene(m.verts, m.indices, procedural_env(2048, 1024, seed=0))
r.set_tile_partition(0, world)
p = rr.default_params(max_refract=8, max_reflect=2, flags=rr.DISPATCH_TILES_RGB8 if os.environ.get("RGB8") else 0)
fb = max_local_tiles(W, H, world) * TILE_BYTES
for F in ([int(sys.argv[2])] if len(sys.argv) > 2 else [8, 16, 64]):
    bufs = [torch.zeros(F * fb, dtype=torch.uint8, device="cuda") for _ in range(3)]
    n = max(4, 1024 // F)
    for lanes in (0, 1, 2, 3):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); a = 0.01
            for b in range(n):
                a = r.render_orbit_sharded(W, H, F, bufs[b % 3].data_ptr(), fb, angle=a, params=p, frames_per_dispatch=F,
                                           lane=None if lanes == 0 else b % lanes)
            r.wait(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("world %d F %3d lanes %d: %.1f us per frame-share, x%d = %.1f us" % (world, F, lanes, dt / (n * F) * 1e6, world, dt / (n * F) * 1e6 * world), flush=True)
