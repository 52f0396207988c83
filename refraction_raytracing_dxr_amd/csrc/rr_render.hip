// rr_render.hip -- the DispatchRays stand-in (RefractionDemo.cpp:580-594) for gfx950.
//
// One launch renders a frame: each lane owns a pixel and runs RayGen (RayTracing.hlsl:42-64),
// then walks that pixel's whole ray tree depth-first -- ClosestHit (hlsl:79-125) spawns the
// refracted child (followed immediately) and the reflected child (parked in registers),
// Miss (hlsl:127-137) adds weight*texel.  The recursive "color += w * child.color" of the
// shader becomes a path-weight sum taken in the same leaf order, so results are deterministic
// and need no atomics, queues or second launch.  A wave covers an 8x8 pixel block (Morton lane
// order) for BVH / env-map coherence; a 256-thread block covers a 32x8 strip of a 32x32 tile,
// tiles are dealt round-robin to ranks (multi-GPU sharding), and the block->tile map keeps each
// XCD on a contiguous run of tiles.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "rr_render_common.h"

namespace rr {

// WPS: waves per SIMD the instantiation is built for (0: what its stack size leaves room for, see rr_render_common.h)
template <int STACK, int PEND, bool STATS, bool TLAS, bool DIAG = false, class E = uint32_t, int WPS = 0>
__global__ __launch_bounds__(256, WPS ? WPS : TLAS ? RR_TLAS_WAVES_PER_SIMD(STACK) : sizeof(E) == 2 ? 8 : RR_FUSED_WAVES_PER_SIMD(STACK)) void k_render_fused(SceneDev sc, DispatchDev a)
{
    __shared__ uint32_t diag_trips[12];    // per wave: internal trips, leaf trips, shading passes
    const unsigned long long diag_t0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long diag_rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (DIAG && threadIdx.x < 12) diag_trips[threadIdx.x] = 0;
    if (DIAG) __syncthreads();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;      // wave: uniform, so that everything derived from it is scalar
    E* stk = reinterpret_cast<E*>(lds) + wave * (STACK * 64) + lane;

    const BlockPos bp = wave_block_pos(a, blockIdx.x * 4u + wave);
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
    const uint32_t x = bp.x0 + lx, y = bp.y0 + ly;
    const bool valid = bp.tile_ok && x < a.W && y < a.H;
    const CamDev& cb = a.cams[bp.frame];                              // wave-uniform: scalar loads

    LaneStats st;
    stats_clock_begin<STATS>(st);
    st.blocks = bp.tile_ok ? 1u : 0u;
    if (STATS && bp.tile_ok && !(DIAG || (bp.x0 + 8u > a.hx0 && bp.x0 < a.hx1 && bp.y0 + 8u > a.hy0 && bp.y0 < a.hy1))) st.bg_blocks = 1u;
    if (valid) {
        st.pixels = 1;
        const bool may_hit = DIAG || (bp.x0 + 8u > a.hx0 && bp.x0 < a.hx1 && bp.y0 + 8u > a.hy0 && bp.y0 < a.hy1);
        f3 acc;
        if (!may_hit) {
            // A block outside the scene's screen rectangle (nine in ten on the reference's scenes): RayGen and one Miss, with none
            // of the ray-tree machinery -- no traversal state, no parked rays, nothing spilled -- and the same arithmetic:
            // payload.color = 0 + 1 * texel (hlsl:57-62, 127-137)
            const f3 D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
            st.rays = 1;
            if (STATS) { st.miss = 1; if (first_active_lane()) st.passes = 1; }
            const f3 e = env_lookup(sc, D);
            acc = mk3(fmaf(1.0f, e.x, 0.0f), fmaf(1.0f, e.y, 0.0f), fmaf(1.0f, e.z, 0.0f));
        } else {
            RegPark<PEND> park;
            acc = render_pixel<STATS, TLAS, DIAG, E, GlobalNodes>(sc, a, cb, x, y, true, stk, GlobalNodes{}, park, st, Diag{ &diag_trips[wave], 4 });
        }
        // What the store needs is read from the kernel's argument block AGAIN here instead of being kept across the renderer:
        // the renderer keeps ~70 scalars live, the compiler allows itself 80 at eight waves per SIMD and moves the rest through
        // vector lanes -- 43 vector instructions per wave before this, 13 now.  (Arguments lie in the block in order, each at
        // its own alignment: `a` follows `sc`.  The asm keeps the compiler from recognising the loads as the ones it already
        // did at the top; every frame-parity test would fail on a wrong offset.)
        typedef const __attribute__((address_space(4))) DispatchDev* KA;
        KA ap = (KA)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                     ((sizeof(SceneDev) + alignof(DispatchDev) - 1) / alignof(DispatchDev)) * alignof(DispatchDev));
        asm volatile("" : "+s"(ap));
        const uint32_t k_compact = ap->compact_out, k_W = ap->W, k_tonemap = ap->tonemap;
        const size_t o = k_compact == 0u ? (size_t)y * k_W + x
                                         : (size_t)bp.tile_local * (TILE * TILE) + (bp.py0 + ly) * TILE + (bp.px0 + lx);
        uint32_t* const out_rgba8 = bp.bg ? ap->out_bg + (size_t)bp.frame * ap->bg_stride : ap->out_rgba8 + (size_t)bp.frame * ap->frame_stride;
        float4* const out_f32 = ap->out_f32 ? ap->out_f32 + (size_t)bp.frame * ap->frame_stride : nullptr;
        DispatchDev a2;
        a2.tonemap = k_tonemap; a2.compact_out = k_compact;
        store_pixel(a2, out_rgba8, out_f32, o, acc);
    }

    if (DIAG) {
        uint32_t mx = st.rays;
        for (int off = 32; off > 0; off >>= 1) { uint32_t v = __shfl_xor(mx, off, 64); mx = v > mx ? v : mx; }
        if (lane == 0) {
            unsigned long long* d = a.diag + (size_t)(blockIdx.x * 4u + wave) * 4;
            d[0] = ((unsigned long long)diag_trips[8 + wave] << 40) | ((unsigned long long)diag_trips[4 + wave] << 20) | diag_trips[wave];
            d[1] = __builtin_amdgcn_s_memtime() - diag_t0; d[2] = mx | ((diag_rt0 & 0xffffffffull) << 32);
            d[3] = (unsigned long long)(diag_trips[wave] + diag_trips[4 + wave]) | ((__builtin_amdgcn_s_memrealtime() & 0xffffffffull) << 32);   // start / end on the 100 MHz clock all CUs share
        }
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// ---------------------------------------------------------------------------------------------------
// Path-parallel form, for launches of one or two slices (the reference's own shape is DispatchRays(W,H,1),
// RefractionDemo.cpp:589-594).  Such a launch lasts as long as its most expensive wave: in k_render_fused a lane walks its
// pixel's whole ray tree, up to 19 rays one after the other on monkey.obj, and the wave that owns that pixel makes 1 400
// loop trips while the rest of the chip is idle (Depth 1: 460 us per frame against 82 at Depth 64).  The tree only
// branches while count < max_reflect; for max_reflect <= 2 it therefore has at most four root-to-leaf paths -- refract or
// reflect at the primary hit, refract or reflect at the next, refractions only from there on -- and each ends in at most
// one leaf (a Miss, weight * texel; a terminal hit or a total internal reflection ends it with nothing).  Here FOUR lanes
// share a pixel, one in each wave of the workgroup, wave p following path p (bit 1: reflect at count 0, bit 0: reflect at
// count 1) for the 64 pixels of the block; the four leaves are then
// summed in the recursion's order -- TT, TR, RT, RR, the order in which k_render_fused reaches them -- with the same
// fma sequence, a path without a leaf contributing fma(0, 0, acc) = acc.  So the frame is bit-identical, the longest chain of
// dependent rays drops from 19 to 2 + the refraction limit, and a block's work spreads over four waves.  The primary ray is
// traced by all four lanes and the two count-1 rays by two each (more work, which a launch of one slice has room for);
// counters count a shared ray once.
// A workgroup is an 8x8 pixel block inside the scene's screen rectangle; the blocks outside the rectangle follow in the
// same launch as 32x8 strips, one Miss per pixel without a trace.  Measured and rejected (monkey.obj 1080p, Depth 1, 8/2
// bounces, us per frame; this form: 244): the four paths of a pixel in adjacent lanes of one wave, the first form of this
// kernel (271: refracted and reflected rays in one wave diverge at once); waves that retire as they finish, the last one
// summing, on 16-bit stacks so that more workgroups fit a CU (265; ott.obj 708 against 634) and, the other way, fewer
// workgroups per CU (6: 292, 4: 328) -- the chains of dependent rays that decide the launch slow down when more waves
// compete, the bulk of the frame when fewer run; s_setprio by ray depth (no effect: the waves deep in a chain are the oldest
// on their SIMD anyway).
struct PathLeaf { float w; f3 e; };

// Called by all 64 lanes of all four waves of the workgroup (valid: the lane has a pixel; it contains workgroup barriers).
// ww: the wave's stack region as 32-bit words; xch: wave 1's.  Levels with few rays left are traced by groups of lanes
// (trace_blas_group): 2 lanes per ray from 32 rays down, 4 from 16.
template <bool STATS, bool TLAS, class E>
__device__ __forceinline__ PathLeaf render_path(const SceneDev& sc, const DispatchDev& a, const CamDev& cb, uint32_t x, uint32_t y, bool valid,
                                                uint32_t path, E* stk, uint32_t* ww, uint32_t* xch, uint32_t lane, LaneStats& st, uint32_t* diag_lv = nullptr)
{
    PathLeaf leaf; leaf.w = 0.0f; leaf.e = mk3(0.0f, 0.0f, 0.0f);
    f3 O = mk3(cb.cam[0], cb.cam[1], cb.cam[2]);
    f3 D = valid ? camera_ray_dir(cb.M, a.sx[x], a.sy[y]) : mk3(0.0f, 0.0f, 1.0f);
    float w = 1.0f;
    uint32_t count = 0;
    bool outside = true, alive = valid;
    float tmin = a.tmin_p, tmax = a.tmax_p;
    for (uint32_t level = 0;; ++level) {
        const unsigned long long m_alive = __ballot(alive);
        if (level >= 2u && m_alive == 0ull) break;          // (levels 0 and 1 hold workgroup barriers: every wave goes through them)
        const int n_alive = __popcll(m_alive);
        // the lane that accounts for this ray (and owns its leaf, should it be one): rays at count 0 are shared by the four
        // lanes of the pixel, rays at count 1 by the two with the same first turn
        const bool owner = level == 0u ? path == 0u : level == 1u ? (path & 1u) == 0u : true;
        if (diag_lv && alive) {      // diagnostic builds: lanes alive at this level, time at which it starts
            if (first_active_lane()) diag_lv[level < 15u ? level : 15u] = (uint32_t)n_alive | ((uint32_t)__builtin_amdgcn_s_memrealtime() << 8);
        }
        HitRec h;
        h.t = tmax; h.hit = false; h.prim = 0; h.leaf = 0; h.inst = 0; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f;
        TravCounters cnt; cnt.nodes = 0; cnt.tris = 0;
        const uint32_t cullf = outside ? CULL_BACK : CULL_FRONT;
        if (level < 2u) {
            // A shared ray is traced ONCE: by wave 0 at level 0 (the primary ray of the block's 64 pixels), by waves 0 and 2 at
            // level 1 (the refracted and the reflected child); the closest hit -- t, leaf, instance -- goes to the other waves of
            // the workgroup through LDS (`xch`: rows of wave 1's stack region, which does not trace before level 2) and each
            // shades it for its own path: the hit attributes and the shading are the same arithmetic on the same operands.
            uint32_t* const row = xch + (level == 0u ? 0u : 3u + 3u * (path >> 1)) * 64u + lane;
            if (owner) {
                if (alive) trace_scene<STATS, TLAS, E, GlobalNodes>(sc, O, D, tmin, tmax, cullf, h, stk, cnt);
                row[0] = __float_as_uint(h.t); row[64] = h.hit ? h.leaf : 0xffffffffu; row[128] = h.inst;
            }
            __syncthreads();
            if (!owner && alive) {
                const uint32_t l = row[64];
                if (l != 0xffffffffu) {
                    h.t = __uint_as_float(row[0]); h.leaf = l; h.inst = row[128]; h.hit = true;
                    if (!TLAS) hit_attributes(sc.blas0.tris, O, D, h);
                    else {
                        const InstDev& in = sc.insts[h.inst];
                        f3 Oh = O, Dh = D;
                        if (!in.identity) { Oh = xform_point(in.inv, O); Dh = xform_dir(in.inv, D); }
                        hit_attributes(sc.pool_tris, Oh, Dh, h);
                    }
                }
            }
            if (level == 1u) __syncthreads();               // wave 1's stack region is a stack again from here on
        } else if (!TLAS && a.group_trace != 0u && n_alive <= 16) {
            trace_blas_group<4, STATS, E>(sc.blas0, alive, m_alive, O, D, tmin, tmax, cullf, h, stk, ww, lane, cnt);
            if (STATS) { st.cnt.nodes += cnt.nodes; st.cnt.tris += cnt.tris; cnt.nodes = 0; cnt.tris = 0; }
        } else if (!TLAS && a.group_trace != 0u && n_alive <= 32) {
            trace_blas_group<2, STATS, E>(sc.blas0, alive, m_alive, O, D, tmin, tmax, cullf, h, stk, ww, lane, cnt);
            if (STATS) { st.cnt.nodes += cnt.nodes; st.cnt.tris += cnt.tris; cnt.nodes = 0; cnt.tris = 0; }
        } else if (alive) {
            trace_scene<STATS, TLAS, E, GlobalNodes>(sc, O, D, tmin, tmax, cullf, h, stk, cnt);
        }
        if (STATS) { st.cnt.node_trips += cnt.node_trips; st.cnt.leaf_trips += cnt.leaf_trips; }
        if (alive) {
            if (owner) { ++st.rays; if (STATS) { st.cnt.nodes += cnt.nodes; st.cnt.tris += cnt.tris; } }
            if (STATS && first_active_lane()) ++st.passes;
            if (!h.hit) {                                             // Miss
                if (owner) { if (STATS) ++st.miss; leaf.w = w; leaf.e = env_lookup(sc, D); }
                alive = false;
            } else {
                if (STATS && owner) ++st.hits;
                if ((int)count >= a.max_refract) { if (STATS && owner) ++st.term; alive = false; }      // hlsl:82, payload.color stays 0
                else {
                    const f3 N = shading_normal<TLAS>(sc, h);
                    const f3 X = mk3(fmaf(h.t, D.x, O.x), fmaf(h.t, D.y, O.y), fmaf(h.t, D.z, O.z));
                    const f3 Nf = outside ? N : neg3(N);
                    const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);
                    const float b = 1.0f - dot3(D, Nf);
                    const float b2 = b * b, b4 = b2 * b2;
                    const float R = (R0 * (1.0f - R0)) * (b4 * b);
                    const float eta = outside ? a.inv_ior : a.ior;
                    f3 d1;
                    const bool refr = refract_ray(d1, D, Nf, eta);
                    if (STATS && owner && !refr) ++st.tir;
                    const bool refl = (int)count < a.max_reflect;
                    // which child this lane follows: the reflected one where its path says so (count 0: bit 1, count 1: bit 0), else the
                    // refracted one; k_render_fused follows the refracted child and parks the reflected one, or, without a refracted
                    // child, follows the reflected one directly -- that one is then the node's only subtree, and it belongs to the
                    // "reflect" lanes here as well (the "refract" lanes have no leaf below this node)
                    const bool turn = count == 0u ? (path & 2u) != 0u : count == 1u ? (path & 1u) != 0u : false;
                    const uint32_t c1 = count + 1u;
                    tmin = a.tmin_s; tmax = a.tmax_s;
                    O = X;
                    if (!turn) {
                        if (!refr) alive = false;
                        else { D = d1; w = w * (1.0f - R); count = c1; outside = !outside; }
                    } else {
                        if (!refl) alive = false;
                        else { D = normalize3(reflect_ray(D, Nf)); w = w * R; count = c1; }
                    }
                }
            }
        }
    }
    return leaf;
}

template <int STACK, bool STATS, bool TLAS, bool DIAG = false>
__global__ __launch_bounds__(256, TLAS ? 5 : 8) void k_render_paths(SceneDev sc, DispatchDev a, uint32_t n_pp_blocks, uint32_t rect_bw)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t diag_lv[4][16];        // diagnostic builds: per wave and ray level, lanes alive | start time << 8
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;      // wave: uniform, so that everything derived from it is scalar
    const unsigned long long diag_t0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (DIAG) { if (lane < 16u) diag_lv[wave][lane] = 0u; }
    LaneStats st;
    stats_clock_begin<STATS>(st);
    if (blockIdx.x < n_pp_blocks) {
        // (Measured and rejected, round 3: starting last launch's slowest blocks first -- as cost classes, whose empty workgroups cost
        // 30 ns each, and as the previous launch's completion order walked backwards: 250 us per frame against 245 on monkey.obj,
        // 450 against 397 on sphere.obj.  A chain of dependent rays runs at 220 us under the full chip's load and at 150 us on an
        // idle one, so a slow block gains nothing from starting while everything else does, and raster order keeps neighbours in L2.)
        const uint32_t frame = blockIdx.x % a.n_frames, b = blockIdx.x / a.n_frames;
        uint32_t* stk = lds + wave * (STACK * 64) + lane;
        // wave p follows path p of the block's 64 pixels: the lanes of a wave then trace rays of one kind (all refracted twice,
        // all reflected then refracted, ...), which stay closer together than the four paths of one pixel do
        const uint32_t path = wave;
        const uint32_t x = a.hx0 + (b % rect_bw) * 8u + compact1by1(lane), y = a.hy0 + (b / rect_bw) * 8u + compact1by1(lane >> 1);
        const bool valid = x < a.W && y < a.H;
        st.blocks = 1u;                 // (a quarter of an 8x8 block: the per-wave cost of the issue model does not apply to this kernel)
        if (valid && path == 0u) st.pixels = 1;
        const PathLeaf lf = render_path<STATS, TLAS, uint32_t>(sc, a, a.cams[frame], x, y, valid, path, stk, lds + wave * (STACK * 64), lds + 1 * (STACK * 64), lane, st,
                                                               DIAG ? diag_lv[wave] : nullptr);
        // the pixel's colour: its leaves in the recursion's order, handed over through the (now idle) stack space
        float* const mine = reinterpret_cast<float*>(lds + wave * (STACK * 64)) + lane;
        mine[0] = lf.w; mine[64] = lf.e.x; mine[128] = lf.e.y; mine[192] = lf.e.z;
        __syncthreads();
        if (wave == 0u && valid) {
            f3 acc = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float* src = reinterpret_cast<const float*>(lds + p * (STACK * 64)) + lane;
                const float w = src[0], ex = src[64], ey = src[128], ez = src[192];
                acc.x = fmaf(w, ex, acc.x); acc.y = fmaf(w, ey, acc.y); acc.z = fmaf(w, ez, acc.z);
            }
            store_pixel(a, a.out_rgba8 + (size_t)frame * a.frame_stride, a.out_f32 ? a.out_f32 + (size_t)frame * a.frame_stride : nullptr,
                        (size_t)y * a.W + x, acc);
        }
    } else {                                                       // a 32x8 strip outside the rectangle: Miss only
        const BlockPos bp = wave_block_pos(a, (blockIdx.x - n_pp_blocks) * 4u + wave);
        const uint32_t x = bp.x0 + compact1by1(lane), y = bp.y0 + compact1by1(lane >> 1);
        const bool in_rect = bp.x0 >= a.hx0 && bp.x0 < a.hx1 && bp.y0 >= a.hy0 && bp.y0 < a.hy1;
        if (bp.tile_ok && !in_rect && x < a.W && y < a.H) {
            const CamDev& cb = a.cams[bp.frame];
            const f3 D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
            st.pixels = 1; st.rays = 1; if (STATS) st.miss = 1;
            const f3 e = env_lookup(sc, D);
            const f3 acc = mk3(fmaf(1.0f, e.x, 0.0f), fmaf(1.0f, e.y, 0.0f), fmaf(1.0f, e.z, 0.0f));
            store_pixel(a, a.out_rgba8 + (size_t)bp.frame * a.frame_stride, a.out_f32 ? a.out_f32 + (size_t)bp.frame * a.frame_stride : nullptr,
                        (size_t)y * a.W + x, acc);
        }
    }
    if (DIAG && a.diag) {       // per wave: start, end (100 MHz), block kind, then the 16 level words
        unsigned long long* d = a.diag + (size_t)(blockIdx.x * 4u + wave) * 12;
        if (lane == 0) { d[0] = diag_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = blockIdx.x < n_pp_blocks ? 1ull : 0ull; d[3] = blockIdx.x; }
        if (lane < 8u) d[4 + lane] = (unsigned long long)diag_lv[wave][2 * lane] | ((unsigned long long)diag_lv[wave][2 * lane + 1] << 32);
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// ---------------------------------------------------------------------------------------------------
// The same renderer with the BLAS's nodes in LDS, for meshes whose whole node array fits beside the traversal stacks
// (the reference's meshes up to shell.obj: 24-49 KB of QNodes).  Persistent workgroups of NW waves: each copies the node
// array into its LDS once, then its waves draw tickets -- a few 8x8 pixel blocks each, shared inside the workgroup -- from
// counters until none are left.  Why: in the L1-fed kernel the texture addresser / L1 / data-return path is the
// busiest unit after the VALU (TA 84 %, TD 97 % busy, 276 cycles per request; waves spend 53 % of their life in s_waitcnt:
// profiles/r02_pmc_fused.txt); a divergent 32-byte node costs the CU's L1 56 cycles and its LDS 25 (tools/ubench_nodefetch.hip),
// and a lone wave's trip shrinks from an L1 round trip to an LDS one, which is what the tail of a Depth-1 launch is made of.
// Order of work (LdsDispatch): first the screen rectangle of the mesh in 32x8 strips, then everything else in 32x32 tiles.
// Tickets come from n_queues counters per phase (rr_types.h); a wave whose own queue is empty drains the others, so every
// block is rendered whatever the placement of workgroups is.
// NW waves per workgroup, WGS workgroups per CU (NW * WGS / 4 waves per SIMD); stack entries are 16 bits (node index or
// ~leaf index: an LDS-resident array has fewer than 5 120 nodes).  After the last block the last wave to leave
// zeroes the ticket words, so the next launch on the same slot needs no memset.
// k_render_lds renders unsharded dispatches only, and its blocks come from its own two phases: no tile partitions, no launch
// order of tiles -- every scalar that stays live across the renderer is one that may end up being moved through vector lanes.
// block j (0..15, row-major 8x8 blocks) of phase 2 ticket u = tile * n_frames + slice
// (divisions by launch constants are a multiply-high with a reciprocal from the host: a scalar division would be done in the
// vector unit, twenty-odd instructions each)
__device__ __forceinline__ BlockPos lds_tile_block(const DispatchDev& a, const LdsDispatch& q, uint32_t u, uint32_t j)
{
    BlockPos p;
    const uint32_t tile = a.n_frames == 1u ? u : __umulhi(u, q.div_frames);
    p.frame = u - tile * a.n_frames;
    p.tile_local = tile;
    p.tile_ok = tile < a.n_local_tiles;
    p.bg = false;
    const uint32_t ty = a.tiles_x == 1u ? tile : __umulhi(tile, q.div_tiles_x), tx = tile - ty * a.tiles_x;
    p.px0 = (j & 3u) * 8u; p.py0 = (j >> 2) * 8u;
    p.x0 = tx * TILE + p.px0; p.y0 = ty * TILE + p.py0;
    return p;
}

// block j (0..3) of phase 1 ticket u = strip * n_frames + slice: strip = a 32x8 run of the scene's screen rectangle, row-major
__device__ __forceinline__ BlockPos lds_rect_block(const DispatchDev& a, const LdsDispatch& q, uint32_t u, uint32_t j)
{
    BlockPos p;
    const uint32_t strip = a.n_frames == 1u ? u : __umulhi(u, q.div_frames), per_row = q.rect_bw >> 2;
    p.frame = u - strip * a.n_frames;
    const uint32_t row = per_row == 1u ? strip : __umulhi(strip, q.div_per_row), col = strip - row * per_row;
    p.tile_local = 0u; p.tile_ok = true; p.bg = false;
    p.px0 = 0u; p.py0 = 0u;
    p.x0 = q.rx0 + col * 32u + j * 8u; p.y0 = q.ry0 + row * 8u;
    return p;
}

template <int NW, int WGS, bool STATS, bool DIAG = false>
__global__ __launch_bounds__(NW * 64, NW * WGS / 4) void k_render_lds(SceneDev sc, DispatchDev a, LdsDispatch q)
{
    typedef uint16_t E;
    __shared__ uint32_t diag_tr[3 * 16];        // diagnostic builds: per wave internal trips, leaf trips, shading passes
    __shared__ uint32_t wg_arrived;             // waves of this workgroup that have finished
    __shared__ unsigned long long wg_share[16]; // per wave: ((ticket + 1) | tile << 31) << 32 | next block of the ticket it drew that is still to be rendered
    if (threadIdx.x == 0) wg_arrived = 0u;      // (ordered before its first use by the barrier behind the node copy)
    if (threadIdx.x < 16) wg_share[threadIdx.x] = 0ull;
    const unsigned long long diag_t0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long diag_wait = 0ull, diag_render = 0ull, diag_n = 0ull;       // cycles in ticket draws / in blocks, tickets | blocks << 32
    unsigned long long diag_worst = 0ull, diag_worst_trips = 0ull;                // the wave's longest block: cycles, its trips (I | L << 20 | S << 40)
    if (DIAG && threadIdx.x < 48) diag_tr[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    {   // the BLAS's nodes -> LDS (q.node_bytes is a multiple of 32)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(sc.blas0.nodes);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = threadIdx.x; i < q.node_bytes / 16u; i += NW * 64) dst[i] = src[i];
    }
    __syncthreads();
    const unsigned long long diag_t1 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;      // wave: uniform, so that everything derived from it is scalar
    const LdsNodes ns{ reinterpret_cast<const char*>(lds) };
    MemPark park{ q.park + (size_t)(blockIdx.x * NW + wave) * ((size_t)q.park_slots * 8 * 64) + lane };
    E* stk = reinterpret_cast<E*>(reinterpret_cast<char*>(lds) + q.node_bytes) + wave * (q.stack_entries * 64u) + lane;
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);

    LaneStats st;
    stats_clock_begin<STATS>(st);
    auto in_rect = [&](const BlockPos& bp) { return bp.x0 >= q.rx0 && bp.x0 < q.rx1 && bp.y0 >= q.ry0 && bp.y0 < q.ry1; };
    // The wave's work loop.  Everything that decides WHICH block comes next is wave-uniform (scalar); the renderer itself is
    // instantiated once, at the bottom of the loop (its code is ~10 KB: several inlined copies thrash the instruction cache).
    // a wave's own queue: its XCD's number (queue i holds tickets i, i + n_queues, ...: with eight queues an XCD keeps to
    // every eighth strip and slice, which its L2 rewards -- monkey.obj Depth 64, 90 us per frame against 104 with queues
    // entered by wave number)
    const uint32_t NQ = q.n_queues;
    uint32_t home = (blockIdx.x * NW + wave) % NQ;
    if (q.home_xcc) { uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); home = (xcc & 7u) % NQ; }
    // Little scalar state on purpose: the renderer below keeps ~70 scalars live, and what does not fit the scalar registers is
    // moved through vector lanes -- vector instructions (the first form of this loop, which looked through 64 counters at a
    // time and kept a strip's BlockPos, cost 491 of them in 1 362).
    uint32_t phase = 0, qi = home, tried = 0;  // tried: queues of this phase the wave has found empty
    for (;;) {
        BlockPos bp;
        bool have = false;
        while (!have) {
            // A ticket is several blocks next to each other, which share triangles and texels in the CU's L1: a 32x8 strip of the
            // scene's screen rectangle in phase 1, a whole 32x32 tile of what lies outside it in phase 2 (the background costs a
            // microsecond per block: eight counters hand out ~600 tickets per microsecond, and tickets of four blocks had the
            // background phases ask for twice that).  One wave rendering a ticket's blocks one after the other would be a chain
            // that many blocks long, and a launch cannot end before its longest chain has (1 ms on monkey.obj, on top of every
            // launch, when this kernel did that).  So the wave that draws a ticket keeps its first block and leaves the others in
            // its slot of wg_share for whichever wave of the workgroup needs a block next -- its own slot first, then the others':
            // one LDS read of all slots, one LDS atomic.
            {
                uint32_t got = 0xffffffffu, got_u = 0u;
                for (;;) {                                      // (again only after losing a block to another wave)
                    const unsigned long long mine = lane < (uint32_t)NW ? wg_share[lane] : 0ull;      // every slot at once, one per lane
                    const uint32_t mh = (uint32_t)(mine >> 32);
                    const unsigned long long m = __ballot(mh != 0u && (uint32_t)mine < ((mh & 0x80000000u) ? 16u : 4u));
                    if (m == 0ull) break;
                    const unsigned long long from_own = m >> wave << wave;                       // the wave's own slot first, then the next ones
                    const uint32_t sl = (uint32_t)__ffsll((long long)(from_own ? from_own : m)) - 1u;
                    unsigned long long v = 0ull;
                    if (lane == 0) v = atomicAdd(&wg_share[sl], 1ull);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
                    if (hi != 0u && lo < ((hi & 0x80000000u) ? 16u : 4u)) { got = lo; got_u = hi; break; }
                }
                if (got != 0xffffffffu) {
                    if (got_u & 0x80000000u) { bp = lds_tile_block(a, q, (got_u & 0x7fffffffu) - 1u, got); have = bp.tile_ok && !in_rect(bp); }
                    else { bp = lds_rect_block(a, q, got_u - 1u, got); have = true; }
                    continue;
                }
            }
            if (phase >= 2u) break;
            const uint32_t total = phase == 0u ? q.p1_tickets : q.p2_tickets;
            uint32_t* const cnt = q.tickets + phase * LDS_QUEUES * 16u;
            bool drew = false;
            uint32_t u = 0;
            while (total != 0u && tried < NQ) {
                const uint32_t n_tickets = total > qi ? (total - qi + NQ - 1u) / NQ : 0u;   // tickets qi, qi + NQ, ...
                // a queue other than the wave's own is looked at before it is drawn from (a counter only ever grows: a queue
                // seen empty stays empty), so the waves that find everything drained add no atomics to the last ones' wait
                uint32_t seen = 0u;
                if (tried != 0u) {
                    if (lane == 0) seen = __hip_atomic_load(&cnt[qi * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = __builtin_amdgcn_readfirstlane(seen);
                }
                if (seen < n_tickets) {
                    uint32_t t = 0;
                    const unsigned long long dw0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
                    if (lane == 0) t = atomicAdd(&cnt[qi * 16u], 1u);
                    t = __builtin_amdgcn_readfirstlane(t);
                    if (DIAG) { diag_wait += __builtin_amdgcn_s_memtime() - dw0; diag_n += 1ull; }
                    if (t < n_tickets) { u = qi + NQ * t; drew = true; break; }
                }
                ++tried;
                qi = qi + 1u == NQ ? 0u : qi + 1u;
            }
            if (!drew) { ++phase; qi = home; tried = 0u; continue; }
            // block 0 of the ticket is this wave's, the rest is for the workgroup
            if (phase == 0u) {
                if (lane == 0) wg_share[wave] = ((unsigned long long)(u + 1u) << 32) | 1ull;
                bp = lds_rect_block(a, q, u, 0u);
                have = true;
            } else {
                if (lane == 0) wg_share[wave] = ((unsigned long long)((u + 1u) | 0x80000000u) << 32) | 1ull;
                bp = lds_tile_block(a, q, u, 0u);
                have = bp.tile_ok && !in_rect(bp);
            }
        }
        if (!have) break;
        st.blocks += 1u;
        const unsigned long long dr0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
        const uint32_t dI0 = DIAG ? diag_tr[wave] : 0u, dL0 = DIAG ? diag_tr[16 + wave] : 0u, dS0 = DIAG ? diag_tr[32 + wave] : 0u;
        const uint32_t x = bp.x0 + lx, y = bp.y0 + ly;
        const bool may_hit = bp.x0 + 8u > a.hx0 && bp.x0 < a.hx1 && bp.y0 + 8u > a.hy0 && bp.y0 < a.hy1;
        if (STATS && !may_hit) st.bg_blocks += 1u;
        if (x < a.W && y < a.H) {
            const CamDev& cb = a.cams[bp.frame];
            st.pixels += 1;
            const f3 acc = render_pixel<STATS, false, DIAG, E, LdsNodes>(sc, a, cb, x, y, may_hit, stk, ns, park, st, Diag{ DIAG ? &diag_tr[wave] : nullptr, 16 });
            const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x
                                                : (size_t)bp.tile_local * (TILE * TILE) + (bp.py0 + ly) * TILE + (bp.px0 + lx);
            store_pixel(a, a.out_rgba8 + (size_t)bp.frame * a.frame_stride,
                        a.out_f32 ? a.out_f32 + (size_t)bp.frame * a.frame_stride : nullptr, o, acc);
        }
        if (DIAG) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - dr0;
            diag_render += dt; diag_n += 1ull << 32;
            if (dt > diag_worst) {
                diag_worst = dt;
                diag_worst_trips = (unsigned long long)(diag_tr[wave] - dI0) | ((unsigned long long)(diag_tr[16 + wave] - dL0) << 20) |
                                   ((unsigned long long)(diag_tr[32 + wave] - dS0) << 40);
            }
        }
    }
    if (DIAG && lane == 0) {
        unsigned long long* d = a.diag + (size_t)(blockIdx.x * NW + wave) * 8;
        d[0] = diag_wait; d[1] = diag_render; d[2] = diag_n; d[3] = __builtin_amdgcn_s_memtime() - diag_t0;
        d[4] = diag_worst; d[5] = diag_worst_trips; d[6] = diag_t1 - diag_t0; d[7] = diag_tr[wave] | ((unsigned long long)diag_tr[16 + wave] << 32);
    }
    flush_stats<STATS>(a, st, blockIdx.x * NW + wave, lane);
    // Every ticket a wave drew came back before it arrives here, so the last workgroup to arrive sees all queues drained and
    // zeroes the words for the next launch.  One arrival per workgroup (its waves count in LDS first): one per wave would be
    // 6 144 atomics on one word.
    if (lane == 0) {
        const uint32_t w_arrived = atomicAdd(&wg_arrived, 1u);
        if (w_arrived + 1u == (uint32_t)NW) {
            const uint32_t arrived = atomicAdd(&q.tickets[2u * LDS_QUEUES * 16u], 1u);
            if (arrived + 1u == gridDim.x) {
                for (uint32_t k = 0; k <= 2u * LDS_QUEUES; ++k) atomicExch(&q.tickets[k * 16u], 0u);
            }
        }
    }
}

// GenerateCameraRay's per-column and per-row screen coordinates (RayTracing.hlsl:29-33): out[0..W) = sx, out[W..W+H) = sy
__global__ __launch_bounds__(256) void k_screen_tables(float* out, uint32_t W, uint32_t H)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < W) out[i] = screen_coord(i, W, false);
    else if (i < W + H) out[i] = screen_coord(i - W, H, true);
}

// Miss in isolation, for the parity tests (rr_env_lookup): dirs / rgb are n x 3 floats
__global__ __launch_bounds__(256) void k_env_lookup(SceneDev sc, const float* dirs, uint32_t n, float* rgb)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const f3 e = env_lookup(sc, mk3(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]));
    rgb[i * 3] = e.x; rgb[i * 3 + 1] = e.y; rgb[i * 3 + 2] = e.z;
}

// TraceRay in isolation, for the parity tests (rr_trace_rays)
template <int STACK, bool TLAS>
__global__ __launch_bounds__(256) void k_trace_rays(SceneDev sc, const rr_ray_dev* rays, uint32_t n, rr_hit_dev* hits,
                                                    uint32_t* error_flag)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;      // wave: uniform, so that everything derived from it is scalar
    uint32_t* stk = lds + wave * (STACK * 64) + lane;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4* q = reinterpret_cast<const float4*>(rays + i);
    float4 o = q[0], d = q[1];
    uint32_t flags = rays[i].flags;
    HitRec h;
    TravCounters cnt; cnt.nodes = 0; cnt.tris = 0;
    uint32_t err = 0;
    trace_scene<false, TLAS>(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, d.w, flags, h, stk, cnt);
    rr_hit_dev r;
    r.hit = h.hit ? 1u : 0u;
    r.t = h.hit ? h.t : d.w;
    r.u = h.hit ? h.U / h.ad : 0.0f;
    r.v = h.hit ? h.V / h.ad : 0.0f;
    r.prim = h.prim; r.inst = h.inst;
    hits[i] = r;
    if (err) atomicOr(error_flag, 1u);
}

// rank 0 after the RCCL gather: [world][max_tiles][32*32] RGBA8 -> W*H raster
__global__ __launch_bounds__(256) void k_assemble_tiles(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame,
                                                        uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                                        uint32_t world, uint32_t max_tiles)
{
    const uint32_t tile = blockIdx.x >> 2, strip = blockIdx.x & 3u;
    if (tile >= n_tiles) return;
    const uint32_t rank = tile % world, tile_local = tile / world;
    const uint32_t px = threadIdx.x & 31u, py = strip * 8u + (threadIdx.x >> 5);
    const uint32_t x = (tile % tiles_x) * TILE + px, y = (tile / tiles_x) * TILE + py;
    if (x < W && y < H)
        frame[(size_t)y * W + x] = gathered[((size_t)rank * max_tiles + tile_local) * (TILE * TILE) + py * TILE + px];
}

__global__ __launch_bounds__(256) void k_assemble_frames(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frames,
                                                         uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                                         uint32_t world, size_t rank_stride, size_t frame_stride,
                                                         size_t out_stride)
{
    const uint32_t tile = blockIdx.x >> 2, strip = blockIdx.x & 3u, f = blockIdx.y;
    if (tile >= n_tiles) return;
    const uint32_t rank = tile % world, tile_local = tile / world;
    const uint32_t px = threadIdx.x & 31u, py = strip * 8u + (threadIdx.x >> 5);
    const uint32_t x = (tile % tiles_x) * TILE + px, y = (tile / tiles_x) * TILE + py;
    if (x < W && y < H)
        frames[f * out_stride + (size_t)y * W + x] =
            gathered[rank * rank_stride + f * frame_stride + (size_t)tile_local * (TILE * TILE) + py * TILE + px];
}

// the same for RGB8 tiles (3 bytes per pixel in the gathered buffers); strides in bytes.  One workgroup per tile and
// frame, one thread per group of four pixels of a tile row: 12 contiguous bytes in, one 16-byte store out (rank 0 runs
// this for every gathered batch while it also renders its own share, so it is written for bandwidth).
// vec4: W % 4 == 0 (a group never straddles the right edge and its raster address is 16-byte aligned).
template <bool VEC4>
__global__ __launch_bounds__(256) void k_assemble_frames_rgb8(const uint8_t* __restrict__ gathered, uint32_t* __restrict__ frames,
                                                              uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                                              uint32_t world, size_t rank_stride_b, size_t frame_stride_b,
                                                              size_t out_stride)
{
    const uint32_t tile = blockIdx.x, f = blockIdx.y;
    if (tile >= n_tiles) return;
    const uint32_t rank = tile % world, tile_local = tile / world;
    const uint32_t py = threadIdx.x >> 3, px = (threadIdx.x & 7u) * 4u;
    const uint32_t x = (tile % tiles_x) * TILE + px, y = (tile / tiles_x) * TILE + py;
    if (y >= H || x >= W) return;
    const uint8_t* p = gathered + rank * rank_stride_b + f * frame_stride_b + ((size_t)tile_local * (TILE * TILE) + py * TILE + px) * 3;
    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p);                 // 12-byte group: 4-byte aligned
    const uint32_t w0 = p4[0], w1 = p4[1], w2 = p4[2];
    uint4 o;
    o.x = (w0 & 0x00ffffffu) | 0xff000000u;
    o.y = (((w0 >> 24) | (w1 << 8)) & 0x00ffffffu) | 0xff000000u;
    o.z = (((w1 >> 16) | (w2 << 16)) & 0x00ffffffu) | 0xff000000u;
    o.w = (w2 >> 8) | 0xff000000u;
    uint32_t* dst = frames + f * out_stride + (size_t)y * W + x;
    if (VEC4) {
        *reinterpret_cast<uint4*>(dst) = o;
    } else {
        dst[0] = o.x;
        if (x + 1 < W) dst[1] = o.y;
        if (x + 2 < W) dst[2] = o.z;
        if (x + 3 < W) dst[3] = o.w;
    }
}


// the mesh-tile partition's de-interleave (rr_mesh_partition): a tile inside the rectangle comes from the gathered buffer of the
// rank it was dealt to, any other tile from rank 0's own background tiles.  RGB8 in, RGBA8 rasters out; one workgroup per tile
// and frame, one thread per group of four pixels of a tile row.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_assemble_frames_mesh_rgb8(const uint8_t* __restrict__ gathered, const uint8_t* __restrict__ bg,
                                                                   uint32_t* __restrict__ frames, uint32_t W, uint32_t H, MeshPartDev mp,
                                                                   size_t rank_stride_b, size_t frame_stride_b, size_t bg_stride_b, size_t out_stride)
{
    const uint32_t tile = blockIdx.x, f = blockIdx.y;
    if (tile >= mp.n_tiles) return;
    const uint32_t tx = tile % mp.tiles_x, ty = tile / mp.tiles_x;
    const uint8_t* src;
    if (mp.rect_w == 0u || (tx >= mp.rect_x0 && tx < mp.rect_x0 + mp.rect_w && ty >= mp.rect_y0 && ty < mp.rect_y0 + mp.rect_h)) {
        const uint32_t i = mp.rect_w == 0u ? tile : (ty - mp.rect_y0) * mp.rect_w + (tx - mp.rect_x0);
        uint32_t rank, slot;
        mesh_deal_owner(i, mp.world, mp.rounds, rank, slot);
        src = gathered + (size_t)rank * rank_stride_b + f * frame_stride_b + (size_t)slot * (TILE * TILE * 3);
    } else {
        const uint32_t per_row = mp.tiles_x - mp.rect_w;
        uint32_t j;
        if (ty < mp.rect_y0) j = ty * mp.tiles_x + tx;
        else if (ty < mp.rect_y0 + mp.rect_h) j = mp.rect_y0 * mp.tiles_x + (ty - mp.rect_y0) * per_row + (tx < mp.rect_x0 ? tx : tx - mp.rect_w);
        else j = mp.rect_y0 * mp.tiles_x + mp.rect_h * per_row + (ty - mp.rect_y0 - mp.rect_h) * mp.tiles_x + tx;
        src = bg + f * bg_stride_b + (size_t)j * (TILE * TILE * 3);
    }
    const uint32_t py = threadIdx.x >> 3, px = (threadIdx.x & 7u) * 4u;
    const uint32_t x = tx * TILE + px, y = ty * TILE + py;
    if (y >= H || x >= W) return;
    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(src + (py * TILE + px) * 3);
    const uint32_t w0 = p4[0], w1 = p4[1], w2 = p4[2];
    uint4 o;
    o.x = (w0 & 0x00ffffffu) | 0xff000000u;
    o.y = (((w0 >> 24) | (w1 << 8)) & 0x00ffffffu) | 0xff000000u;
    o.z = (((w1 >> 16) | (w2 << 16)) & 0x00ffffffu) | 0xff000000u;
    o.w = (w2 >> 8) | 0xff000000u;
    uint32_t* dst = frames + f * out_stride + (size_t)y * W + x;
    if (VEC4) *reinterpret_cast<uint4*>(dst) = o;
    else {
        dst[0] = o.x;
        if (x + 1 < W) dst[1] = o.y;
        if (x + 2 < W) dst[2] = o.z;
        if (x + 3 < W) dst[3] = o.w;
    }
}

hipError_t launch_assemble_frames_mesh_rgb8(const uint8_t* gathered, const uint8_t* bg, uint32_t* frames, uint32_t W, uint32_t H, const MeshPartDev& mp,
                                            size_t rank_stride_b, size_t frame_stride_b, size_t bg_stride_b, size_t out_stride, uint32_t n_frames, hipStream_t s)
{
    if (mp.n_tiles == 0 || n_frames == 0) return hipSuccess;
    const bool vec4 = (W % 4u) == 0u && (out_stride % 4u) == 0u && (reinterpret_cast<uintptr_t>(frames) % 16u) == 0u;
    if (vec4) hipLaunchKernelGGL(k_assemble_frames_mesh_rgb8<true>, dim3(mp.n_tiles, n_frames), dim3(256), 0, s, gathered, bg, frames, W, H, mp,
                                 rank_stride_b, frame_stride_b, bg_stride_b, out_stride);
    else      hipLaunchKernelGGL(k_assemble_frames_mesh_rgb8<false>, dim3(mp.n_tiles, n_frames), dim3(256), 0, s, gathered, bg, frames, W, H, mp,
                                 rank_stride_b, frame_stride_b, bg_stride_b, out_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ launchers
// name of the render kernel instantiation the calling thread launched last (rr_stats::render_kernel_name)
static thread_local char g_kernel_name[96] = "";
const char* last_render_kernel_name() { return g_kernel_name; }
#define RR_NAME(...) snprintf(g_kernel_name, sizeof g_kernel_name, __VA_ARGS__)
template <int STACK, int PEND, bool TLAS>
static hipError_t launch_fused_spt(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(uint32_t);
    RR_NAME("k_render_fused<%d, %d, %s, %s, false, unsigned int, 0>", STACK, PEND, stats ? "true" : "false", TLAS ? "true" : "false");
    if (stats) hipLaunchKernelGGL((k_render_fused<STACK, PEND, true, TLAS>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_fused<STACK, PEND, false, TLAS>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    return hipGetLastError();
}

template <int NW, int WGS>
static hipError_t launch_lds_nw(const SceneDev& sc, const DispatchDev& a, const LdsDispatch& q, size_t lds, int n_cus, bool stats, hipStream_t s)
{
    static const hipError_t attr = [] {       // more than 64 KB of dynamic LDS has to be asked for, once per instantiation
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_lds<NW, WGS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_lds<NW, WGS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        return e;
    }();
    if (attr != hipSuccess) return attr;
    const dim3 grid((uint32_t)n_cus * WGS), block(NW * 64);
    RR_NAME("k_render_lds<%d, %d, %s, false>", NW, WGS, stats ? "true" : "false");
    if (stats) hipLaunchKernelGGL((k_render_lds<NW, WGS, true>), grid, block, lds, s, sc, a, q);
    else       hipLaunchKernelGGL((k_render_lds<NW, WGS, false>), grid, block, lds, s, sc, a, q);
    return hipGetLastError();
}

// Workgroup shapes of k_render_lds; a shape fits when the node array plus its stacks fit the CU's LDS WGS times (limit per
// workgroup: 160 KiB / WGS, less a little for the allocation granule).  12 waves x 2 workgroups (6 waves per SIMD, 80
// registers) is the one that pays: 16 x 2 (8 waves, 64 registers) spills 28 words per lane and 8 192 waves' scratch no
// longer fits the L2 (monkey.obj Depth 64: 90 us per frame against 83), 16 x 1 leaves 4 waves per SIMD (109 us).
int lds_kernel_shape(uint32_t node_bytes, uint32_t stack_entries, size_t* lds_bytes, int min_shape)
{
    static const struct { int nw, wgs; } shapes[] = { { 12, 2 }, { 16, 2 }, { 16, 1 } };
    for (int i = min_shape < 0 ? 0 : min_shape; i < 3; ++i) {
        const size_t need = (size_t)node_bytes + (size_t)shapes[i].nw * stack_entries * 64 * sizeof(uint16_t);
        if (need <= (size_t)(160 * 1024) / shapes[i].wgs - 512) { if (lds_bytes) *lds_bytes = need; return i; }
        if (min_shape <= 0) break;       // product path: the first shape or none
    }
    return -1;
}

hipError_t launch_render_lds(const SceneDev& sc, const DispatchDev& a, LdsDispatch q, int n_cus, bool stats, hipStream_t s, int min_shape)
{
    if (a.n_blocks == 0) return hipSuccess;
    size_t lds = 0;
    const int shape = lds_kernel_shape(q.node_bytes, q.stack_entries, &lds, min_shape);
    if (shape < 0) return hipErrorInvalidValue;
    {   // one queue per workgroup of the shape that will run (never more than LDS_QUEUES)
        static const int wgs[] = { 2, 2, 1 };
        const uint32_t grid = (uint32_t)n_cus * (uint32_t)wgs[shape];
        if (q.n_queues > grid) q.n_queues = grid;
        if (q.n_queues > LDS_QUEUES) q.n_queues = LDS_QUEUES;
        if (q.n_queues < 1u) q.n_queues = 1u;
    }
    q.p2_tickets = a.n_local_tiles * a.n_frames;        // phase 2: one 32x32 tile of one slice per ticket (blocks inside the rectangle are skipped)
    // phase 1: the rectangle, widened to whole 32-pixel columns, in 32x8 strips of one slice each
    if (q.rx1 > q.rx0 && q.ry1 > q.ry0) {
        q.rx0 &= ~31u;
        q.rx1 = (q.rx1 + 31u) & ~31u;
    }
    const bool rect = q.rx1 > q.rx0 && q.ry1 > q.ry0;
    q.rect_bw = rect ? (q.rx1 - q.rx0) / 8u : 0u;
    q.p1_tickets = !rect ? 0u : (q.rect_bw / 4u) * ((q.ry1 - q.ry0) / 8u) * a.n_frames;
    if ((uint64_t)q.p2_tickets * a.n_frames >= 0xffffffffull || (uint64_t)q.p1_tickets * a.n_frames >= 0xffffffffull) return hipErrorInvalidValue;   // (multiply-high divisions)
    q.div_frames = (uint32_t)(0x100000000ull / a.n_frames) + 1u;
    q.div_tiles_x = (uint32_t)(0x100000000ull / a.tiles_x) + 1u;
    q.div_per_row = rect ? (uint32_t)(0x100000000ull / (q.rect_bw / 4u)) + 1u : 0u;
    if (a.diag) {       // diagnostic build (RR_DEBUG_DIAG): per-wave cycles in ticket draws and in blocks; 12x2 shape
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_lds<12, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (attr != hipSuccess) return attr;
        const size_t l12 = (size_t)q.node_bytes + (size_t)12 * q.stack_entries * 64 * sizeof(uint16_t);
        hipLaunchKernelGGL((k_render_lds<12, 2, false, true>), dim3((uint32_t)n_cus * 2), dim3(12 * 64), l12, s, sc, a, q);
        return hipGetLastError();
    }
    if (shape == 0) return launch_lds_nw<12, 2>(sc, a, q, lds, n_cus, stats, s);
    if (shape == 1) return launch_lds_nw<16, 2>(sc, a, q, lds, n_cus, stats, s);
    return launch_lds_nw<16, 1>(sc, a, q, lds, n_cus, stats, s);
}

template <int STACK, int PEND>
static hipError_t launch_fused_sp(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    if (!sc.single_identity) return launch_fused_spt<STACK, PEND, true>(sc, a, stats, s);
    return launch_fused_spt<STACK, PEND, false>(sc, a, stats, s);
}

// reference-scene kernel on 16-bit stack entries (meshes below 32 768 triangles, trees of 20..39 levels): 39 entries are
// 19 968 B per workgroup, eight workgroups per CU
template <int PEND>
static hipError_t launch_fused_s16(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * 39 * 64 * sizeof(uint16_t);
    RR_NAME("k_render_fused<39, %d, %s, false, false, unsigned short, 0>", PEND, stats ? "true" : "false");
    if (stats) hipLaunchKernelGGL((k_render_fused<39, PEND, true, false, false, uint16_t>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_fused<39, PEND, false, false, false, uint16_t>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    return hipGetLastError();
}

// Two-level scenes whose stack entries fit 16 bits (fewer than 32 768 pool nodes and triangles + instances) and whose trees are
// at most 30 levels deep: 15 KB of stacks per workgroup instead of 31, so the LDS no longer caps the kernel at five waves per
// SIMD, and the build for seven (72 registers, 64 words through scratch) is the fastest -- the 1 024-monkey grid at 2160p,
// Depth 16: 7.51 ms per frame with 32-bit stacks (five waves), 7.23 / 6.95 / 7.24 ms built for six / seven / eight.
// (Parking the reflected rays in LDS instead of registers -- 96 registers, five waves, a third of the spills -- measured 7.76.)
// Trees of 31..39 levels (C4: ott.obj under a TLAS) get the 39-entry build for five waves: 934 us per frame at Depth 16
// against 979 on 32-bit stacks (four waves), 993 / 1 051 built for six / seven; launches of one or two slices stay on the
// 32-bit build (2.14 against 2.44 ms).
template <int STACK, int WPS>
static hipError_t launch_fused_tlas16(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(uint16_t);
    RR_NAME("k_render_fused<%d, 2, %s, true, false, unsigned short, %d>", STACK, stats ? "true" : "false", WPS);
    if (stats) hipLaunchKernelGGL((k_render_fused<STACK, 2, true, true, false, uint16_t, WPS>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_fused<STACK, 2, false, true, false, uint16_t, WPS>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    return hipGetLastError();
}

hipError_t launch_render_fused(const SceneDev& sc, const DispatchDev& a, int stack, int pend, bool stats, hipStream_t s, bool stack16)
{
    if (a.n_blocks == 0) return hipSuccess;
#ifndef RR_TLAS30_WPS
#define RR_TLAS30_WPS 7
#endif
#ifndef RR_TLAS39_WPS
#define RR_TLAS39_WPS 5
#endif
    if (stack16 && !a.diag && !sc.single_identity && pend <= 2 && stack <= 30) return launch_fused_tlas16<30, RR_TLAS30_WPS>(sc, a, stats, s);
    if (stack16 && !a.diag && !sc.single_identity && pend <= 2 && stack <= 39) return launch_fused_tlas16<39, RR_TLAS39_WPS>(sc, a, stats, s);
    if (stack16 && !a.diag && sc.single_identity && stack <= 39)
        return pend <= 2 ? launch_fused_s16<2>(sc, a, stats, s) : launch_fused_s16<8>(sc, a, stats, s);
    if (a.diag) {       // diagnostic build of the reference-scene kernel (RR_DEBUG_DIAG; never used by the product path)
        if (stack <= 19) hipLaunchKernelGGL((k_render_fused<19, 2, false, false, true>), dim3(a.n_blocks), dim3(256), 4 * 19 * 64 * 4, s, sc, a);
        else hipLaunchKernelGGL((k_render_fused<31, 2, false, false, true>), dim3(a.n_blocks), dim3(256), 4 * 31 * 64 * 4, s, sc, a);
        return hipGetLastError();
    }
    if (stack <= 19 && pend <= 2) return launch_fused_sp<19, 2>(sc, a, stats, s);
    if (stack <= 22 && pend <= 2) return launch_fused_sp<22, 2>(sc, a, stats, s);
    if (stack <= 26 && pend <= 2) return launch_fused_sp<26, 2>(sc, a, stats, s);
    if (stack <= 31) return pend <= 2 ? launch_fused_sp<31, 2>(sc, a, stats, s) : launch_fused_sp<31, 8>(sc, a, stats, s);
    if (stack <= 39) return pend <= 2 ? launch_fused_sp<39, 2>(sc, a, stats, s) : launch_fused_sp<39, 8>(sc, a, stats, s);
    return pend <= 2 ? launch_fused_sp<64, 2>(sc, a, stats, s) : launch_fused_sp<64, 8>(sc, a, stats, s);
}

template <int STACK, bool TLAS>
static void launch_trace_st(const SceneDev& sc, const rr_ray_dev* rays, uint32_t n, rr_hit_dev* hits, uint32_t* err, hipStream_t s)
{
    hipLaunchKernelGGL((k_trace_rays<STACK, TLAS>), dim3((n + 255u) / 256u), dim3(256), 4 * STACK * 64 * 4, s, sc, rays, n, hits, err);
}

hipError_t launch_trace_rays(const SceneDev& sc, const rr_ray_dev* rays, uint32_t n, rr_hit_dev* hits, uint32_t* err,
                             int stack, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (stack <= 31) { if (sc.single_identity) launch_trace_st<31, false>(sc, rays, n, hits, err, s); else launch_trace_st<31, true>(sc, rays, n, hits, err, s); }
    else             { if (sc.single_identity) launch_trace_st<64, false>(sc, rays, n, hits, err, s); else launch_trace_st<64, true>(sc, rays, n, hits, err, s); }
    return hipGetLastError();
}

template <int STACK, bool TLAS>
static hipError_t launch_paths_st(const SceneDev& sc, const DispatchDev& a, uint32_t n_pp, uint32_t rect_bw, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(uint32_t);
    const dim3 grid(n_pp + a.n_blocks);
    RR_NAME("k_render_paths<%d, %s, %s, false>", STACK, stats ? "true" : "false", TLAS ? "true" : "false");
    if (a.diag && !TLAS) { hipLaunchKernelGGL((k_render_paths<STACK, false, false, true>), grid, dim3(256), lds, s, sc, a, n_pp, rect_bw); return hipGetLastError(); }
    if (stats) hipLaunchKernelGGL((k_render_paths<STACK, true, TLAS>), grid, dim3(256), lds, s, sc, a, n_pp, rect_bw);
    else       hipLaunchKernelGGL((k_render_paths<STACK, false, TLAS>), grid, dim3(256), lds, s, sc, a, n_pp, rect_bw);
    return hipGetLastError();
}

// unsharded raster frames, max_reflect <= 2, stack <= 39 entries; the rectangle DispatchDev::hx0..hy1 is not empty
hipError_t launch_render_paths(const SceneDev& sc, const DispatchDev& a, int stack, bool stats, hipStream_t s)
{
    const uint32_t rect_bw = (a.hx1 - a.hx0) / 8u, rect_bh = (a.hy1 - a.hy0) / 8u;
    const uint32_t n_pp = rect_bw * rect_bh * a.n_frames;
    if (!sc.single_identity) return stack <= 19 ? launch_paths_st<19, true>(sc, a, n_pp, rect_bw, stats, s) : launch_paths_st<39, true>(sc, a, n_pp, rect_bw, stats, s);
    return stack <= 19 ? launch_paths_st<19, false>(sc, a, n_pp, rect_bw, stats, s) : launch_paths_st<39, false>(sc, a, n_pp, rect_bw, stats, s);
}

hipError_t launch_screen_tables(float* out, uint32_t W, uint32_t H, hipStream_t s)
{
    hipLaunchKernelGGL(k_screen_tables, dim3((W + H + 255u) / 256u), dim3(256), 0, s, out, W, H);
    return hipGetLastError();
}

hipError_t launch_env_lookup(const SceneDev& sc, const float* dirs, uint32_t n, float* rgb, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_env_lookup, dim3((n + 255u) / 256u), dim3(256), 0, s, sc, dirs, n, rgb);
    return hipGetLastError();
}

hipError_t launch_assemble_tiles(const uint32_t* gathered, uint32_t* frame, uint32_t W, uint32_t H, uint32_t tiles_x,
                                 uint32_t n_tiles, uint32_t world, uint32_t max_tiles, hipStream_t s)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_assemble_tiles, dim3(n_tiles * 4u), dim3(256), 0, s, gathered, frame, W, H, tiles_x, n_tiles, world,
                       max_tiles);
    return hipGetLastError();
}

hipError_t launch_assemble_frames(const uint32_t* gathered, uint32_t* frames, uint32_t W, uint32_t H, uint32_t tiles_x,
                                  uint32_t n_tiles, uint32_t world, size_t rank_stride, size_t frame_stride,
                                  size_t out_stride, uint32_t n_frames, hipStream_t s)
{
    if (n_tiles == 0 || n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_assemble_frames, dim3(n_tiles * 4u, n_frames), dim3(256), 0, s, gathered, frames, W, H, tiles_x,
                       n_tiles, world, rank_stride, frame_stride, out_stride);
    return hipGetLastError();
}

hipError_t launch_assemble_frames_rgb8(const uint8_t* gathered, uint32_t* frames, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                       uint32_t world, size_t rank_stride_b, size_t frame_stride_b, size_t out_stride, uint32_t n_frames,
                                       hipStream_t s)
{
    if (n_tiles == 0 || n_frames == 0) return hipSuccess;
    const bool vec4 = (W % 4u) == 0u && (out_stride % 4u) == 0u && (reinterpret_cast<uintptr_t>(frames) % 16u) == 0u;
    if (vec4) hipLaunchKernelGGL(k_assemble_frames_rgb8<true>, dim3(n_tiles, n_frames), dim3(256), 0, s, gathered, frames, W, H, tiles_x,
                                 n_tiles, world, rank_stride_b, frame_stride_b, out_stride);
    else      hipLaunchKernelGGL(k_assemble_frames_rgb8<false>, dim3(n_tiles, n_frames), dim3(256), 0, s, gathered, frames, W, H, tiles_x,
                                 n_tiles, world, rank_stride_b, frame_stride_b, out_stride);
    return hipGetLastError();
}



} // namespace rr
