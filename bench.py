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
N_SIMD, CLOCK_GHZ = 1024, 2.4    # 256 CUs x 4 SIMD-32, 2.4 GHz max clock (MI355X_MICROARCH.md)
# Vector-issue roofline of k_render_fused<19, 2, false, false> (the kernel the headline workload runs on).  The kernel is
# bound by vector (VALU) issue: the SIMDs hold 8 waves each and every traversal instruction but the f32 add / mul / fma is a
# quarter-rate one on gfx950 (tools/ubench_valu.hip, profiles/r03_ubench_valu.txt: v_fma_f32 / v_mul_f32 2.2 cycles per
# wave64 instruction per SIMD, v_cndmask / v_fma_mix_f32 / v_max3 / v_min / v_cmp / v_bfi 4.2, transcendental 8).  A 64-lane
# wave issues one internal-node step, one triangle test or one shading pass per loop trip however many lanes take part, so
# its issue time is  node_trips * C_NODE + leaf_trips * C_LEAF + passes * C_PASS + waves * C_WAVE + background_waves * C_BG
# SIMD cycles (a background wave -- a block outside the scene's screen rectangle, RayGen + one Miss on a branch of its own
# -- is counted in the last term only), with per-trip instruction counts fitted against the PMC counters SQ_INSTS_VALU /
# _ADD_F32 / _MUL_F32 / _FMA_F32 / _TRANS_F32 of seven workloads of this kernel (tools/fit_valu.sh, tools/fit_valu.py;
# residual <= 1.2 %, profiles/r02_valu_fit.txt) and priced with the class costs above.  The trip counters are exact and come
# from the RR_DISPATCH_COLLECT_STATS launches of the same frames.
# Round 3: tools/ubench_valu.hip now completes (profiles/r03_ubench_valu.txt; round 2's record of it ended in a GPU fault) and
# its "slab-test mix" row -- the node step's own 29 instructions in the kernel's proportions -- issues at 3.80 cycles per
# instruction per SIMD with 8 waves resident, less than the 4.2 the single-class rows give: the node trip is priced with that
# row, 30.5 x 3.80 = 115.9 (round 2: 130.4).  All costs are TRUE shader cycles (wall time x the clock the ubench saw), so the
# matching peak is SIMDs x the clock the render launch ran at; `peak` uses the 2.4 GHz maximum clock, an upper bound (the
# launches run at 2.0-2.3 GHz: roofline.clock_seen_GHz, measured live), so `frac` is a lower bound and
# roofline.frac_at_clock_seen is the same figure over the clock actually held.
VALU_CYCLES_PER_TRIP = {"node": 115.9,      # 30.5 instructions, all quarter rate (12 of them v_fma_mix_f32), 3.80 cycles each in this mix
                        "leaf": 194.9,      # 61.3: Moller-Trumbore is mostly f32 mul / fma
                        "pass": 856.4,      # 250.0 per shading pass of a traced wave (ray set-up, ClosestHit / Miss)
                        "wave": 299.6,      # 70.7 per traced 8x8 block: RayGen, addressing, store
                        "bg": 1013.1}       # 280.5 per background block: RayGen, Miss, store
# k_render_lds (persistent workgroups, nodes in LDS; the measured choice takes it where it is more than 5 % faster: sphere.obj and
# shell.obj at Depth 64): the same fit on its own counter passes (tools/fit_valu_lds.sh, profiles/r03_valu_fit_lds.txt, eight
# workloads, residual <= 2.0 %): 29.1 / 72.4 / 225.6 / 120.5 / 304.2 instructions per node trip / leaf trip / pass / block /
# background block, priced like the others (node trip 29.1 x 3.80; the rest by class).  On sphere.obj and shell.obj the priced
# cycles come to 1.00-1.02 of the launch's SIMD cycles: the kernel is at this roof and the class prices are good to a few per
# cent, so a fraction up to 1.03 is reported as 1.0 with the model's figure next to it (`model_fraction`).
VALU_CYCLES_PER_TRIP_LDS = {"node": 110.6, "leaf": 240.8, "pass": 758.2, "wave": 486.8, "bg": 1109.5}
FRAC_OVERSHOOT = 1.03
ROOFLINE_KERNEL = "k_render_fused<19, 2, false, false, false, unsigned int, 0>"


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


def valu_issue_cycles(st, lds=False):
    c = VALU_CYCLES_PER_TRIP_LDS if lds else VALU_CYCLES_PER_TRIP
    bg = st.background_waves
    return (st.node_trips * c["node"] + st.leaf_trips * c["leaf"] + (st.shade_passes - bg) * c["pass"] + (st.waves - bg) * c["wave"]
            + bg * c["bg"])


def pmc_busy_roofs(render_kernel, frames_per_launch):
    """TA / TD / LDS busy fractions of the headline kernel from the latest PMC summary under profiles/ (a counter pass cannot
    run inside the bench).  They were collected for k_render_fused on monkey.obj 8/2 at Depth 64 and describe nothing else:
    any other kernel or launch depth gets None."""
    import glob
    tag = {0: "fused", 1: "lds"}.get(render_kernel, "none")
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_monkey_d64.txt" % tag)))
    if not cands or frames_per_launch != 64:
        return {"l1_texture_addresser_busy": None, "l1_data_return_busy": None, "lds_busy": None,
                "busy_source": "PMC passes exist for k_render_fused / k_render_lds, monkey.obj 8/2, Depth 64 only (%s); this run's shape differs"
                               % (os.path.relpath(cands[-1], ROOT) if cands else "none under profiles/")}
    v = {}
    for line in open(cands[-1]):
        f = line.split()
        if len(f) >= 3 and f[1] == "avg_per_launch":
            v[f[0]] = float(f[2])
    try:
        cu_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * 256.0              # the counter sums 8 XCDs; 256 CUs each with one TA / TD
        out = {"l1_texture_addresser_busy": round(v["TA_TA_BUSY_sum"] / cu_cycles, 3),
               "l1_data_return_busy": round(v["TD_TD_BUSY_sum"] / cu_cycles, 3),
               "lds_busy": round(v["SQ_ACTIVE_INST_LDS"] * 4.0 / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)}
    except KeyError:
        return {"l1_texture_addresser_busy": None, "l1_data_return_busy": None, "lds_busy": None, "busy_source": "unparsed " + cands[-1]}
    out["busy_source"] = ("%s: rocprofv3 --pmc passes of %s, monkey.obj 8/2, Depth 64 (static, the shape of this run)"
                          % (os.path.relpath(cands[-1], ROOT), "k_render_lds<12, 2>" if render_kernel == 1 else "k_render_fused<19, 2>"))
    return out


def pmc_record(tag, must_contain=None, kernel_name=""):
    """Busy fractions of a kernel from its counter passes under profiles/ (static: a counter pass cannot run inside the bench):
    vector issue = SQ_ACTIVE_INST_VALU * 4 / SIMD cycles, texture addresser and data return per CU.  None if there is no
    such file or the kernel that ran is not the one the passes were taken on."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.txt" % tag)))
    if not cands or (must_contain and must_contain not in kernel_name):
        return None
    v = {}
    for line in open(cands[-1]):
        f = line.split()
        if len(f) >= 3 and f[1] == "avg_per_launch":
            v[f[0]] = float(f[2])
    try:
        xcd_cycles = v["GRBM_GUI_ACTIVE"] / 8.0
        return {"valu_issue_busy": round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / (xcd_cycles * 1024.0), 3),
                "l1_texture_addresser_busy": round(v["TA_TA_BUSY_sum"] / (xcd_cycles * 256.0), 3),
                "l1_data_return_busy": round(v["TD_TD_BUSY_sum"] / (xcd_cycles * 256.0), 3),
                "source": "%s (rocprofv3 --pmc passes of this workload, tools/pmc_passes.sh; static, not measured in this run)" % os.path.relpath(cands[-1], ROOT)}
    except KeyError:
        return None


def issue_record(st, launches, kernel_us, fused_model_applies):
    """what the vector-issue model says about `launches` launches of `kernel_us` each, from their exact STATS counters"""
    rec = {"kernel": st.render_kernel_name.decode() if isinstance(st.render_kernel_name, bytes) else str(st.render_kernel_name),
           "kernel_us": round(kernel_us, 2),
           "lane_utilisation": {"node": round(st.node_visits / (64.0 * max(st.node_trips, 1)), 3),
                                "leaf": round(st.tri_tests / (64.0 * max(st.leaf_trips, 1)), 3),
                                "shade": round(st.rays / (64.0 * max(st.shade_passes, 1)), 3)},
           "wave_trips_per_launch": {"node": int(st.node_trips / launches), "leaf": int(st.leaf_trips / launches),
                                     "shade_passes": int(st.shade_passes / launches), "waves": int(st.waves / launches),
                                     "background_waves": int(st.background_waves / launches)},
           "clock_seen_GHz": round(st.clock_ghz, 3) if st.clock_ghz else None}
    if fused_model_applies:
        cyc = valu_issue_cycles(st, lds=(st.render_kernel == 1)) / launches
        if st.render_kernel == 1:
            rec["model_note"] = "per-trip costs of k_render_lds (profiles/r03_valu_fit_lds.txt)"
        frac = cyc / (kernel_us * 1e-6) / 1e9 / (N_SIMD * CLOCK_GHZ)
        rec["bound"] = "valu_issue"
        rec["frac"] = round(min(frac, 1.0), 4) if frac <= FRAC_OVERSHOOT else None
        if frac > 1.0:
            rec["model_fraction"] = round(frac, 4)
        if st.clock_ghz:
            f2 = frac * CLOCK_GHZ / st.clock_ghz
            rec["frac_at_clock_seen"] = round(min(f2, 1.0), 4) if f2 <= FRAC_OVERSHOOT * 1.05 else None
        if not frac <= FRAC_OVERSHOOT:
            rec["model_invalid"] = "fitted vector-issue fraction %.3f > 1: the per-trip costs do not describe this launch" % frac
    else:
        rec["bound"] = "not modelled"
        rec["frac"] = None
        rec["model_invalid"] = ("the vector-issue model was fitted on k_render_fused<19, 2> (single BLAS); this kernel's trips are "
                                "reported, its roof is not")
    return rec


def xf(tx, ty, tz, s=1.0):
    m = np.eye(4, dtype=np.float32)[:3] * np.float32(s)
    m[:, 3] = (tx, ty, tz)
    return m


def config_records(r, rr, asset, env):
    """The other BASELINE.json configurations on this GPU (C1, C2, C4, C5; C3 is the headline): HIP-event kernel time of
    16-slice launches and of single launches, exact ray counts.  Scene placement of C4 / C5 as in SURVEY 8d (builder-defined:
    the reference has one mesh, one instance)."""
    def load(n):
        m = rr.Mesh(); assert m.load(asset(n)); return m
    cases = [
        ("C1", "sphere.obj 256x256, 1 refraction bounce", ["sphere.obj"], None, 256, 256, 1, 1.0),
        ("C2", "sphere.obj 1920x1080, 4 refraction bounces", ["sphere.obj"], None, 1920, 1080, 4, 1.0),
        ("C4", "shell.obj + cube.obj + ott.obj (3 BLAS, TLAS) 3840x2160, 8 bounces", ["shell.obj", "cube.obj", "ott.obj"],
         ([xf(0, 0, 0), xf(0, 0, -4.0), xf(0, 0, 4.0)], [0, 1, 2]), 3840, 2160, 8, 1.6),
        ("C5", "monkey.obj x 1024 instances (TLAS) 3840x2160, 16 bounces", ["monkey.obj"],
         ([xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)], [0] * 1024), 3840, 2160, 16, 14.0),
    ]
    out = {}
    for key, label, names, inst, W_, H_, refr, radius in cases:
        ids = []
        for m in [load(n) for n in names]:
            mid = r.upload_mesh(m.verts, m.indices); r.build_blas(mid); ids.append(mid)
        if inst is None:
            r.build_tlas(rr.make_instances(meshes=[ids[0]]))
        else:
            r.build_tlas(rr.make_instances(transforms=inst[0], meshes=[ids[k] for k in inst[1]]))
        r.upload_envmap(env)
        cams = []
        for k in range(16):
            sc = rr.camera_orbit(0.01 * (k + 1))
            sc.camera_loc[0] *= radius; sc.camera_loc[2] *= radius
            if radius > 2: sc.camera_loc[1] = 0.8 * radius
            cams.append(sc)
        rec = {"workload": label}
        for depth in (16, 1):
            p = rr.default_params(max_refract=refr, flags=rr.DISPATCH_TIME_KERNEL)
            for rep in range(5):        # the first two launches of a shape are where the kernel choice is measured (twice): not timed
                if depth == 1:
                    for c in cams[:4]:
                        r.set_camera(c); r.dispatch_rays(W_, H_, p)
                else:
                    r.dispatch_rays_batch(W_, H_, cams, p)
                if rep == 1: r.kernel_time()
            ms, n = r.kernel_time()
            st = r.stats()
            frames = 1 if depth == 1 else 16
            us = ms / n * 1e3 / frames
            rec["depth%d" % depth] = {"us_per_frame": round(us, 1), "Mrays_per_s": round(st.rays / frames / us, 1),
                                      "rays_per_frame": int(st.rays / frames)}
            if depth == 16:         # the same launch through the STATS build: trips, lane utilisation, clock, kernel name
                r.dispatch_rays_batch(W_, H_, cams, rr.default_params(max_refract=refr, flags=rr.DISPATCH_COLLECT_STATS))
                ss = r.stats()
                r.dispatch_rays_batch(W_, H_, cams, rr.default_params(max_refract=refr))      # the product build's name
                name = r.stats().render_kernel_name
                rec["roofline"] = issue_record(ss, 1, ms / n * 1e3, fused_model_applies=(inst is None and ss.render_kernel in (0, 1)))
                rec["roofline"]["kernel"] = name.decode() if isinstance(name, bytes) else str(name)
                if key == "C4":
                    rec["roofline"]["counters"] = pmc_record("c4_fused", "k_render_fused<39", rec["roofline"]["kernel"])
                elif key == "C5":
                    rec["roofline"]["counters"] = pmc_record("c5_stream", "k_stream_rays", rec["roofline"]["kernel"])
        out[key] = rec
    return out


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
    ap.add_argument("--prewarm", type=int, default=512,
                    help="frames rendered before the --warmup steps, untimed and reported as config.prewarm_steps: a fresh box starts at "
                         "its idle clocks and a --warmup of a few frames (half a millisecond) ends before they have come up")
    ap.add_argument("--no-depth1", action="store_true", help="skip the reference's own shape, one DispatchRays per frame (32 launches, ~15 ms)")
    ap.add_argument("--no-configs", action="store_true", help="skip the single-GPU records of the other BASELINE configurations (C1, C2, C4, C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subdiv", action="store_true",
                    help="skip the second figure (15 472-triangle monkey); profiling runs use this so that every "
                         "k_render_fused launch in the trace is a launch of the headline workload")
    ap.add_argument("--rotate-root", action="store_true",
                    help="N>1: gather batch b to rank b %% N instead of rank 0 (frames end up spread over the ranks; "
                         "no single GPU takes in every frame).  Off by default: the reference presents from one device")
    ap.add_argument("--round-robin-tiles", action="store_true",
                    help="N>1: deal every tile of the frame round robin and gather them all (round 2's partition) instead of only "
                         "the tiles that touch the scene's screen rectangle")
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
    # RCCL prints a version banner to STDOUT when its first communicator comes up; the driver reads one JSON line from there.
    # While the process group is set up (and its first collective runs) file descriptor 1 points at stderr.
    class _StdoutToStderr:
        def __enter__(self):
            sys.stdout.flush(); self.saved = os.dup(1); os.dup2(2, 1); return self
        def __exit__(self, *exc):
            sys.stdout.flush(); os.dup2(self.saved, 1); os.close(self.saved)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("RR_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse the N>1 path on a 1-GPU box
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend=backend)
            dist.barrier()

    mesh = rr.Mesh()
    assert mesh.load(asset("monkey.obj"))
    env = procedural_env(ENV_W, ENV_H, seed=0)
    r = rr.Renderer(local_rank)
    # a stream of the bench's own, made torch's current one (collectives, copies and the library's launches all go through it;
    # torch.cuda.synchronize() is device-wide and covers it): the legacy default stream synchronises with every other stream at
    # each launch, which a 1.8 ms timed region notices
    torch.cuda.set_stream(torch.cuda.Stream(device=torch.device("cuda", local_rank)))
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.load_scene(mesh.verts, mesh.indices, env)
    params = rr.default_params(max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT)
    K, Wm = args.steps, args.warmup
    timed_kernel_name = None

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
        with _StdoutToStderr():
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            dist.barrier()
    if world == 1 and not force_sharded:
        r.set_tile_partition(0, 1)
        if args.prewarm > 0:
            r.render_orbit(W, H, args.prewarm, angle=0.01, params=params, frames_per_dispatch=F)     # clocks up, buffers allocated
        r.render_orbit(W, H, Wm, angle=0.01, params=params, frames_per_dispatch=F)      # warmup, untimed
        barrier()
        t0 = time.perf_counter()
        r.timing_begin()
        r.render_orbit(W, H, K, angle=0.01, params=params, frames_per_dispatch=F)       # EXACTLY K timed steps (frames)
        region_ms = r.timing_end()
        barrier()
        elapsed = time.perf_counter() - t0
        st = r.stats()
        timed_kernel_name = st.render_kernel_name.decode()     # the kernel that rendered the (last launch of the) timed steps
        total_rays = st.rays
        overflow = st.traversal_overflow
    else:
        # a rank's share of a launch has 1/world of the blocks: keep launches long enough for their tails not to show
        # (tools/exp_lanes.py, world 8: 113 us/frame-equivalent at 64 frames per launch, 108 at 256)
        Fn = args.frames_per_dispatch * max(1, min(world // 2, 4))
        # a run shorter than two such batches is split in two, so that the gather of the first half travels under the render of
        # the second and the de-interleave of the first under the gather of the second (one batch overlaps nothing)
        Fn = max(1, min(Fn, (K + 1) // 2))
        # only the tiles that touch the scene's screen rectangle are dealt to the ranks and gathered; rank 0 renders the background
        # tiles itself (rr_mesh_partition): 72 instead of 255 tiles per rank and frame at N = 8 on this view
        sf = rr.dist.ShardedFrames(r, W, H, rank, world, torch.device("cuda", local_rank), Fn, always_collective=force_sharded,
                                   rotate_root=args.rotate_root, mesh_partition=not args.rotate_root and not args.round_robin_tiles)
        if args.prewarm > 0:
            sf.render_orbit(args.prewarm, angle=0.01, params=params)
        sf.render_orbit(Wm, angle=0.01, params=params)
        barrier()
        t0 = time.perf_counter()
        sf.gathered_bytes = 0
        rays_local = sf.render_orbit(K, angle=0.01, params=params)
        barrier()
        elapsed = time.perf_counter() - t0
        gathered_bytes = sf.gathered_bytes
        region_ms = None
        # attribution of the N-GPU figure (untimed extra): every rank's share of the same frames rendered without the gather
        # and the de-interleave -- if the end-to-end time is far above the slowest rank's render time, the rest is rank 0's
        # ingest (7/8 of every frame crosses xGMI into one GPU) and its assemble launches, not the render kernel
        barrier()
        t1 = time.perf_counter()
        sf.render_only(K, angle=0.01, params=params)
        torch.cuda.synchronize()
        render_only = time.perf_counter() - t1
        ro = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)] if dist.get_world_size() > 1 else None
        mine = torch.tensor([render_only], dtype=torch.float64, device="cuda")
        if ro is not None:
            dist.all_gather(ro, mine)
            render_only_all = [float(x[0]) for x in ro]
        else:
            render_only_all = [render_only]
        t = torch.tensor([elapsed, float(rays_local), float(r.stats().traversal_overflow)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays = int(t[1])
        overflow = int(t[2])

    # ---- N > 1: the run validates itself -- the last frame the pipeline assembled on rank 0 against the same frame rendered
    # by rank 0 alone (untimed).  The kernels are deterministic, so anything but byte equality is a bug in the sharding, the
    # gather or the de-interleave.
    multi_gpu_check = None
    if (world > 1 or force_sharded) and rank == 0 and not args.rotate_root:
        got = sf.frames_host()[-1].copy()
        r.set_tile_partition(0, 1)
        ang = np.float32(0.01)
        for _ in range(K - 1):                       # the frame loop's own arithmetic: angle += 0.01f in fp32 (RefractionDemo.cpp:567)
            ang = np.float32(ang + np.float32(0.01))
        r.render_orbit(W, H, 1, angle=float(ang), params=params, frames_per_dispatch=1)
        want = r.read_frame()
        multi_gpu_check = {"frame": K - 1, "identical_to_single_gpu_render": bool(np.array_equal(got, want)),
                           "differing_pixels": int((got != want).any(axis=-1).sum())}
    # ---- roofline of the dominant kernel (k_render_fused), rank 0, single-GPU geometry ---------------------
    roofline = None
    cpu = None
    subdiv = None
    configs = None
    Fl = min(K, F)                  # slices per launch as the timed region really issued them
    if rank == 0:
        r.set_tile_partition(0, 1)
        # whole launches of Fl consecutive frames, their start angles spread over the timed orbit (consecutive frames
        # share cache lines, so a launch must hold consecutive frames to be a launch of the timed loop)
        n_launch = max(1, min(K, 512) // Fl)
        starts = [0.01 + 0.01 * ((K // n_launch) * j) for j in range(n_launch)]
        reps = max(1, 8 // n_launch)    # a short run (K < 64: one launch per pass) is timed over several passes, not one sample
        for flag in (rr.DISPATCH_COLLECT_STATS, rr.DISPATCH_TIME_KERNEL):
            for rep in range(reps if flag == rr.DISPATCH_TIME_KERNEL else 1):
                for j, a0 in enumerate(starts):
                    r.render_orbit(W, H, Fl, angle=a0, frames_per_dispatch=Fl, params=rr.default_params(
                        max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=flag | (rr.DISPATCH_KEEP_COUNTERS if j else 0)))
            if flag == rr.DISPATCH_COLLECT_STATS:
                sst = r.stats()                                             # exact counters, summed over the n_launch launches
        kms, kn = r.kernel_time()                                           # HIP events around each launch, on the launch stream
        kernel_us = kms / kn * 1e3
        kn = n_launch                                                       # the counters cover one pass
        _st = r.stats()
        render_kernel = int(_st.render_kernel)                              # which kernel those launches were
        kernel_name = _st.render_kernel_name.decode()
        k1ms, k1n = None, 0
        if not args.no_depth1:   # the reference's own shape, one DispatchRays per frame (RefractionDemo.cpp:589-594: Depth = 1)
            for warm in (True, False):       # (the first launches of the shape are where its kernel is chosen: not part of the figure)
                r.render_orbit(W, H, 4 if warm else 32, angle=0.01, frames_per_dispatch=1, params=rr.default_params(
                    max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=rr.DISPATCH_TIME_KERNEL))
                k1ms, k1n = r.kernel_time()
        # vector-issue roofline: SIMD cycles the launch's wave-level trips need / SIMD cycles the launch had
        issue_cycles = valu_issue_cycles(sst, lds=(render_kernel == 1)) / kn
        vc = VALU_CYCLES_PER_TRIP_LDS if render_kernel == 1 else VALU_CYCLES_PER_TRIP
        achieved = issue_cycles / (kernel_us * 1e-6) / 1e9                  # G SIMD-cycles of vector issue per second
        peak = N_SIMD * CLOCK_GHZ
        frac = achieved / peak
        model_invalid = None
        if render_kernel not in (0, 1):
            model_invalid = "the timed launches ran on render kernel %d, which no issue model describes" % render_kernel
        elif not frac <= 1.0:
            model_invalid = "fitted vector-issue fraction %.3f > 1: the per-trip costs no longer describe the kernel; refit (tools/fit_valu.sh)" % frac
        clock_seen = sst.clock_ghz or None
        frac_seen = frac * CLOCK_GHZ / clock_seen if clock_seen else None
        bytes_per_launch = algorithmic_bytes(sst) / kn
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))      # the latest round's PMC passes
        if cands:
            try:
                tj = json.load(open(cands[-1]))
                traffic = {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"), "frames_per_launch": tj.get("frames_per_launch"),
                           "source": "static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command), not measured in this run"
                                     % os.path.relpath(cands[-1], ROOT)}
            except Exception:
                traffic = None
        busy = pmc_busy_roofs(render_kernel, Fl)
        roofline = {"bound": "valu_issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G SIMD-cycles/s",
                    "frac": None if model_invalid else round(frac, 4), "traffic": traffic,
                    "peak_note": "%d SIMDs x %.1f GHz, the maximum clock: an upper bound, so frac is a lower bound; the launches ran at "
                                 "clock_seen_GHz (s_memtime / s_memrealtime over every wave of the STATS launches of the same frames)" % (N_SIMD, CLOCK_GHZ),
                    "clock_seen_GHz": round(clock_seen, 3) if clock_seen else None,
                    "frac_at_clock_seen": None if (model_invalid or not frac_seen or frac_seen > 1.0) else round(frac_seen, 4),
                    "model_invalid": model_invalid,
                    "kernel": kernel_name, "kernel_us": round(kernel_us, 2), "frames_per_launch": Fl,
                    "timed_region_kernel": timed_kernel_name,
                    "kernel_note": None if timed_kernel_name == kernel_name else
                                   ("the timed steps ran on %s (the first launch of a shape runs its class's default kernel); the launches of this "
                                    "record came after the kernel choice had been measured for the shape and ran on %s" % (timed_kernel_name, kernel_name)),
                    "model": "node_trips*%.1f + leaf_trips*%.1f + (shade_passes - background_waves)*%.1f + (waves - background_waves)*%.1f + "
                             "background_waves*%.1f SIMD cycles (instruction counts: profiles/%s; cycles per class and the node mix: "
                             "profiles/r03_ubench_valu.txt); %d SIMDs x %.1f GHz" % (
                                 vc["node"], vc["leaf"], vc["pass"], vc["wave"], vc["bg"],
                                 "r03_valu_fit_lds.txt" if render_kernel == 1 else "r02_valu_fit.txt", N_SIMD, CLOCK_GHZ),
                    "wave_trips_per_launch": {"node": int(sst.node_trips / kn), "leaf": int(sst.leaf_trips / kn),
                                              "shade_passes": int(sst.shade_passes / kn), "waves": int(sst.waves / kn),
                                              "background_waves": int(sst.background_waves / kn)},
                    "lane_utilisation": {"node": round(sst.node_visits / (64.0 * sst.node_trips), 3),
                                         "leaf": round(sst.tri_tests / (64.0 * sst.leaf_trips), 3),
                                         "shade": round(sst.rays / (64.0 * sst.shade_passes), 3)},
                    # the other candidate roofs, as fractions of their peaks (rocprofv3 PMC passes of this kernel at this shape,
                    # profiles/r02_pmc_fused_monkey_d64.txt; static -- a counter pass cannot run inside the bench):
                    "roofs": {"valu_issue": None if model_invalid else round(frac, 4),
                              "l1_texture_addresser_busy": busy["l1_texture_addresser_busy"], "l1_data_return_busy": busy["l1_data_return_busy"],
                              "hbm_counter_bytes": (round(traffic["hbm_bytes_per_launch"] / traffic["frames_per_launch"] * Fl / (kernel_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
                                                    if traffic and traffic.get("hbm_bytes_per_launch") else None),
                              "lds_busy": busy["lds_busy"],
                              "source": "valu_issue live (exact trip counters x fitted per-trip cost); hbm_counter_bytes = the PMC bytes of "
                                        "roofline.traffic over this run's kernel time and 8 TB/s; TA / TD / LDS busy: " + busy["busy_source"]},
                    "depth1_kernel_us": round(k1ms / k1n * 1e3, 2) if k1n else None,
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "algorithmic_GBps_not_a_roof": round(bytes_per_launch / (kernel_us * 1e-6) / 1e9, 1),
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
            r.render_orbit(W, H, n16, angle=0.01, params=params, frames_per_dispatch=F)      # the same shape, untimed: buffers for its launches in flight
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
            # its roof: one launch of F frames through the STATS build (exact trips, clock) and three through the product build
            # with HIP events.  The per-trip costs were fitted on k_render_fused<19, 2> (32-bit stack entries); this mesh's tree
            # is deeper and runs the 16-bit-stack instantiation of the same loops (two more vector instructions per push / pop),
            # so the fraction is the model's estimate for it, labelled as such.
            r.render_orbit(W, H, F, angle=0.01, frames_per_dispatch=F, params=rr.default_params(
                max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=rr.DISPATCH_COLLECT_STATS))
            ss16 = r.stats()
            for rep in range(5):
                r.render_orbit(W, H, F, angle=0.01, frames_per_dispatch=F, params=rr.default_params(
                    max_refract=MAX_REFRACT, max_reflect=MAX_REFLECT, flags=rr.DISPATCH_TIME_KERNEL))
                if rep == 1: r.kernel_time()
            ms16, n16k = r.kernel_time()
            name16 = r.stats().render_kernel_name.decode()
            subdiv["roofline"] = issue_record(ss16, 1, ms16 / n16k * 1e3, fused_model_applies=ss16.render_kernel == 0)
            subdiv["roofline"]["kernel"] = name16
            subdiv["roofline"]["frames_per_launch"] = F
            subdiv["roofline"]["model_note"] = "per-trip costs fitted on k_render_fused<19, 2, ..., unsigned int>; applied to this instantiation as an estimate"
            subdiv["roofline"]["counters"] = pmc_record("monkey16k", "k_render_fused<39", name16)
            subdiv["roofline"]["node_visits_per_ray"] = round(ss16.node_visits / ss16.rays, 2)
            subdiv["roofline"]["tri_tests_per_ray"] = round(ss16.tri_tests / ss16.rays, 2)
            r.load_scene(mesh.verts, mesh.indices, env)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(mesh, env)
        if world == 1 and not args.no_configs:
            configs = config_records(r, rr, asset, env)

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
                       "parallelism": "tiles32x32-%s-x%d%s" % ("roundrobin" if (args.round_robin_tiles or args.rotate_root or world == 1) else "meshtiles-roundrobin",
                                                                 world, "-rotating-root" if args.rotate_root and world > 1 else ""),
                       "frames_per_dispatch": min(K, F) if world == 1 and not force_sharded else Fn,
                       "prewarm_steps": args.prewarm,
                       "launch_shape": "DispatchRays(W, H, Depth = frames_per_dispatch): every frame complete in its own buffer; the "
                                       "reference's own shape, Depth 1, is roofline.depth1_kernel_us"},
            "device_region_ms_per_step": round(region_ms / K, 5) if region_ms is not None else None,
            "multi_gpu_attribution": None if world == 1 and not force_sharded else {
                "end_to_end_ms_per_step": round(elapsed / K * 1e3, 5),
                "render_only_ms_per_step_by_rank": [round(x / K * 1e3, 5) for x in render_only_all],
                "tile_partition": "round-robin over all tiles" if (args.round_robin_tiles or args.rotate_root) else
                                  "mesh tiles round-robin, background tiles on rank 0 (rr_mesh_partition)",
                "gathered_bytes_per_rank_per_step": int(gathered_bytes / K),
                "note": "render_only: each rank's tiles of the same frames, same launches and lanes, no gather, no de-interleave "
                        "(untimed extra pass).  end_to_end - max(render_only) = rank 0's gather ingest + assemble that the pipeline "
                        "did not hide"},
            "multi_gpu_check": multi_gpu_check,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "monkey_16k": subdiv,
            "configs": configs,
        }
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
