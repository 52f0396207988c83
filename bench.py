#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + refracted/reflected) and fps at 1920x1080 on monkey.obj.

A step = one frame of the reference's drawFrame loop: camera constants for the orbit angle,
DispatchRays over the whole 1920x1080 frame, (N > 1: RCCL gather of the image tiles + de-interleave
on rank 0).  Frames are issued --frames-per-dispatch at a time as the depth slices of one
DispatchRays(W,H,Depth) launch (every frame is fully rendered into its own buffer; Depth 1, the
reference's shape, is reported next to it as depth1_kernel_us).  Workload = BASELINE.json configs[2] (the configuration the metric is quoted on):
monkey.obj (967 triangles, the reference's own file), 8 refraction bounces, 2 reflection bounces,
seeded procedural 2048x1024 HDR env map (the reference's envmap.hdr is missing from the mount),
frame k uses angle 0.01*(k+1) like the reference's `angle += 0.01f`.

Run: python bench.py [--gpus N --steps K --warmup W]; for N > 1 the driver launches it through
torch.distributed.run (one rank per GPU, backend nccl = RCCL over xGMI).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

W, H = 1920, 1080
MAX_REFRACT, MAX_REFLECT = 8, 2
ENV_W, ENV_H = 2048, 1024
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(st):
    """Bytes the traversal/shading must touch (DESIGN.md 'Algorithmic bytes'), priced with the record sizes the
    kernel really reads: 32 B per internal node visit (QNode: both child boxes as fp16 cell counts + child refs),
    48 B per triangle test, 36 B of vertex normals per shaded hit, 12 B env texel per miss, 4 B RGBA8 per
    pixel.  (SURVEY 8d prices a node at 64 B (fp32 boxes) and adds a ray queue and a float accumulator the
    fused kernel does not have; that figure is reported next to this one as survey_formula_bytes_per_ray.)"""
    shaded = st.hits - st.terminal_hits
    return 32 * st.node_visits + 48 * st.tri_tests + 36 * shaded + 12 * st.misses + 4 * st.pixels


def survey_formula_bytes(st):
    return (64 * st.node_visits + 48 * st.tri_tests + 96 * st.secondary + 36 * st.hits + 12 * st.misses
            + 16 * st.pixels)


def host_cores():
    """threads the CPU baseline may really use: cgroup quota if there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(mesh, env, budget_s=12.0):
    """The CPU oracle (oracle/, C, fp32, median-split BVH, one pthread per host core) timed on a
    bounded sample of the same workload: whole 1920x1080 frames of the same orbit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O                      # the checker: only this leg of bench.py touches oracle/
    import refraction_raytracing_dxr_amd as rr
    cores = host_cores()
    s = O.Scene()
    s.add_mesh(mesh.verts, mesh.indices)
    s.set_envmap(env)
    p = O.default_params(use_bvh=1, max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, accum_mode=0)
    rays, frames, t0 = 0, 0, time.perf_counter()
    while True:
        sc = rr.camera_orbit(np.float32(0.01) * (frames + 1))
        M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
        r = s.render(M, cam, W, H, p, threads=cores)
        rays += r["stats"].rays
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s or frames >= 628:
            break
    return {"value": round(rays / el / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full 1920x1080 frames of the same orbit (%.1f s), CPU restatement with its own "
                      "median-split BVH; D3D12 WARP is Windows-only and cannot run here" % (frames, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12288)     # 1.3 s of frames on one GPU: the timed region is not launch jitter
    ap.add_argument("--warmup", type=int, default=192)      # three launches: both render lanes and all three buffer sets of the N>1 pipeline
    ap.add_argument("--depth1", action="store_true", help="also time the reference's shape, one DispatchRays per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subdiv", action="store_true",
                    help="skip the second figure (15 472-triangle monkey); profiling runs use this so that every "
                         "k_render_fused launch in the trace is a launch of the headline workload")
    ap.add_argument("--rotate-root", action="store_true",
                    help="N>1: gather batch b to rank b %% N instead of rank 0 (frames end up spread over the ranks; "
                         "no single GPU takes in every frame).  Off by default: the reference presents from one device")
    ap.add_argument("--frames-per-dispatch", type=int, default=64,
                    help="depth slices per launch (DispatchRays(W,H,Depth)); N>1: also frames per RCCL gather")
    args = ap.parse_args()

    import torch
    import refraction_raytracing_dxr_amd as rr
    from refraction_raytracing_dxr_amd.synth import asset, procedural_env

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("RR_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0                      # rehearsal: every rank shares the one card
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("RR_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse the N>1 path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    mesh = rr.Mesh()
    assert mesh.load(asset("monkey.obj"))
    env = procedural_env(ENV_W, ENV_H, seed=0)
    r = rr.Renderer(local_rank)
    r.set_stream(torch.cuda.current_stream().cuda_stream)       # so torch.cuda.synchronize() covers the kernels
    r.load_scene(mesh.verts, mesh.indices, env)
    params = rr.default_params(max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT)
    K, Wm = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    F = max(1, args.frames_per_dispatch)
    # rehearsal switch: run the N > 1 code path (ShardedFrames + RCCL gather) with a single rank
    force_sharded = world == 1 and os.environ.get("RR_BENCH_FORCE_SHARDED") == "1"
    if force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world == 1 and not force_sharded:
        r.set_tile_partition(0, 1)
        r.render_orbit(W, H, Wm, angle=0.01, params=params, frames_per_dispatch=F)      # warmup, untimed
        barrier()
        t0 = time.perf_counter()
        r.timing_begin()
        r.render_orbit(W, H, K, angle=0.01, params=params, frames_per_dispatch=F)       # EXACTLY K timed steps (frames)
        region_ms = r.timing_end()
        barrier()
        elapsed = time.perf_counter() - t0
        st = r.stats()
        total_rays = st.rays
        overflow = st.traversal_overflow
    else:
        # a rank's share of a launch has 1/world of the blocks: keep launches long enough for their tails not to show
        # (tools/exp_lanes.py, world 8: 113 us/frame-equivalent at 64 frames per launch, 108 at 256)
        Fn = args.frames_per_dispatch * max(1, min(world // 2, 4))
        sf = rr.dist.ShardedFrames(r, W, H, rank, world, torch.device("cuda", local_rank), Fn, always_collective=force_sharded,
                                   rotate_root=args.rotate_root)
        sf.render_orbit(Wm, angle=0.01, params=params)
        barrier()
        t0 = time.perf_counter()
        rays_local = sf.render_orbit(K, angle=0.01, params=params)
        barrier()
        elapsed = time.perf_counter() - t0
        region_ms = None
        t = torch.tensor([elapsed, float(rays_local), float(r.stats().traversal_overflow)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays = int(t[1])
        overflow = int(t[2])

    # ---- roofline of the dominant kernel (k_render_fused), rank 0, single-GPU geometry ---------------------
    roofline = None
    cpu = None
    subdiv = None
    if rank == 0:
        r.set_tile_partition(0, 1)
        # whole launches of F consecutive frames, their start angles spread over the timed orbit (consecutive frames
        # share cache lines, so a launch must hold consecutive frames to be a launch of the timed loop)
        n_launch = max(1, min(K, 512) // F)
        n_prof = n_launch * F
        starts = [0.01 + 0.01 * ((K // n_launch) * j) for j in range(n_launch)]
        for flag in (rr.DISPATCH_COLLECT_STATS, rr.DISPATCH_TIME_KERNEL):
            for j, a0 in enumerate(starts):
                r.render_orbit(W, H, F, angle=a0, frames_per_dispatch=F, params=rr.default_params(
                    max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=flag | (rr.DISPATCH_KEEP_COUNTERS if j else 0)))
            if flag == rr.DISPATCH_COLLECT_STATS:
                sst = r.stats()                                             # exact counters, summed over n_prof frames
        kms, kn = r.kernel_time()                                           # HIP events around each launch, on the launch stream
        bytes_per_launch = algorithmic_bytes(sst) / kn                      # one launch = F frames
        kernel_us = kms / kn * 1e3
        k1ms, k1n = None, 0
        if args.depth1:   # the reference's own shape, one DispatchRays per frame (Depth 1), for comparison
            r.render_orbit(W, H, 32, angle=0.01, frames_per_dispatch=1, params=rr.default_params(
                max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=rr.DISPATCH_TIME_KERNEL))
            k1ms, k1n = r.kernel_time()
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))      # the latest round's PMC passes
        tpath = cands[-1] if cands else ""
        if tpath:
            try:
                tj = json.load(open(tpath))
                if tj.get("frames_per_launch") == F:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "k_render_fused", "kernel_us": round(kernel_us, 2), "frames_per_launch": F,
                    "depth1_kernel_us": round(k1ms / k1n * 1e3, 2) if k1n else None,
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "bytes_per_ray": round(algorithmic_bytes(sst) / sst.rays, 1),
                    "survey_formula_bytes_per_ray": round(survey_formula_bytes(sst) / sst.rays, 1),
                    "node_visits_per_ray": round(sst.node_visits / sst.rays, 2),
                    "tri_tests_per_ray": round(sst.tri_tests / sst.rays, 2),
                    "kernel_grays_per_s": round(sst.rays / kn / (kernel_us * 1e-6) / 1e9, 3)}
        # ---- BASELINE's "~16k tri Suzanne": monkey.obj midpoint-subdivided twice (15 472 tri), same frames ----
        if world == 1 and not args.no_subdiv:
            from refraction_raytracing_dxr_amd.synth import subdivide
            v16, i16 = subdivide(mesh.verts, 2)
            r.load_scene(v16, i16, env)
            n16 = max(F, (min(K, 256) // F) * F)
            r.render_orbit(W, H, F, angle=0.01, params=params, frames_per_dispatch=F)
            torch.cuda.synchronize()
            t16 = time.perf_counter()
            r.render_orbit(W, H, n16, angle=0.01, params=params, frames_per_dispatch=F)
            torch.cuda.synchronize()
            t16 = time.perf_counter() - t16
            s16 = r.stats()
            if s16.traversal_overflow:
                raise SystemExit("traversal stack overflow on the subdivided mesh: result invalid")
            subdiv = {"workload": "monkey.obj midpoint-subdivided x2 (%d tri), otherwise as config.workload" % (len(i16) // 3),
                      "value": round(s16.rays / t16 / 1e6, 2), "unit": "Mrays/s", "fps": round(n16 / t16, 1), "steps": n16}
            r.load_scene(mesh.verts, mesh.indices, env)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(mesh, env)

    if rank == 0:
        if overflow:
            raise SystemExit("traversal stack overflow during the benchmark: result invalid")
        out = {
            "metric": "Mrays/s (primary+refracted) at 1920x1080, monkey.obj",
            "value": round(total_rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "fps": round(K / elapsed, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "monkey.obj (967 tri) 1920x1080, 8 refraction / 2 reflection bounces, ior 1.3, "
                                   "orbit angle 0.01*(k+1), seeded procedural 2048x1024 RGB32F env map",
                       "rays_per_frame": round(total_rays / K, 1),
                       "parallelism": "tiles32x32-roundrobin-x%d%s" % (world, "-rotating-root" if args.rotate_root and world > 1 else ""),
                       "frames_per_dispatch": F if world == 1 else F * max(1, min(world // 2, 4))},
            "device_region_ms_per_step": round(region_ms / K, 5) if region_ms is not None else None,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "monkey_16k": subdiv,
        }
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
