// rr_render_exp.hip -- render-kernel EXPERIMENTS, kept for the measurements DESIGN.md quotes; compiled only into builds made
// with RR_EXPERIMENTAL=1 (refraction_raytracing_dxr_amd/_build.py) and selected there with RR_DEBUG_KERNEL=async / wavefront /
// refill.  None of them is on the product path; all of them render the same bits as k_render_fused (tests, same switch).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "rr_render_common.h"

namespace rr {

// ---------------------------------------------------------------------------------------------------
// Pixel refill, for scenes whose pixels differ wildly in their number of rays (the instanced TLAS scenes: 1 to 40 rays per
// pixel on the 1 024-monkey grid, where a wave of k_render_fused makes 6.6 shading passes for 2.8 rays per pixel and only
// 42 % of its lanes hold a live pixel in a pass).  A wave owns a column of four 8x8 blocks of a 32x32 tile (256 pixels) and
// its lanes take the next pixel of the column as soon as theirs is finished, so every pass works on 64 live pixels until
// the column runs out.  A lane's pixel is still rendered exactly as in k_render_fused -- RayGen, the tree depth-first, leaves
// summed in the recursion's order -- only the moment it starts differs, so frames and counters are bit-identical.
// Parked reflected rays live in LDS (a lane's slot moves on with it), stacks hold 16-bit entries where the scene allows.
template <int STACK, bool STATS, bool TLAS, class E, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_render_refill(SceneDev sc, DispatchDev a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    E* stk = reinterpret_cast<E*>(lds) + wave * (STACK * 64) + lane;
    LdsPark park{ lds + (4u * STACK * 64u * (uint32_t)sizeof(E)) / 4u + wave * (2u * 8u * 64u) + lane };

    // workgroup = one 32x32 tile of one slice (slices interleaved, as everywhere); wave w = the 8-pixel-wide column w
    const uint32_t frame = blockIdx.x % a.n_frames;
    const uint32_t tile_local = blockIdx.x / a.n_frames;
    if (tile_local >= a.n_local_tiles) return;
    const uint32_t tile = tile_local * a.tile_world + a.tile_rank;
    const uint32_t x0 = (tile % a.tiles_x) * TILE + wave * 8u, y0 = (tile / a.tiles_x) * TILE;
    const CamDev& cb = a.cams[frame];
    uint32_t* const out_rgba8 = a.out_rgba8 + (size_t)frame * a.frame_stride;
    float4* const out_f32 = a.out_f32 ? a.out_f32 + (size_t)frame * a.frame_stride : nullptr;
    const bool may_hit = x0 + 8u > a.hx0 && x0 < a.hx1 && y0 + 32u > a.hy0 && y0 < a.hy1;

    LaneStats st;
    st.blocks = 4u;
    constexpr uint32_t TOTAL = 4u * 64u;            // pixels of the column
    uint32_t cursor = 0;                            // next pixel of the column to hand out (wave-uniform)
    bool alive = false;
    uint32_t pix = 0;                               // the lane's pixel: index in the column (block j = pix >> 6, Morton position pix & 63)
    RayState r;
    r.O = r.D = mk3(0.0f, 0.0f, 0.0f); r.w = 0.0f; r.tmin = r.tmax = 0.0f; r.count = 0; r.outside = true;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    int np = 0;
    bool fresh = false;                             // the lane's ray is a primary ray (may be skipped outside the scene's rectangle)
    for (;;) {
        // ---- refill: lanes without a pixel take the next ones of the column
        if (cursor < TOTAL) {
            const unsigned long long need = __ballot(!alive);
            if (need) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                const uint32_t mine = cursor + rank;
                if (!alive && mine < TOTAL) {
                    pix = mine;
                    const uint32_t x = x0 + compact1by1(pix & 63u), y = y0 + (pix >> 6) * 8u + compact1by1((pix & 63u) >> 1);
                    if (x < a.W && y < a.H) {
                        r = primary_ray(a, cb, x, y);
                        acc = mk3(0.0f, 0.0f, 0.0f); np = 0;
                        alive = true; fresh = true;
                        st.pixels += 1;
                    }
                }
                cursor += (uint32_t)__popcll(need);
            }
        }
        if (__ballot(alive) == 0ull) { if (cursor >= TOTAL) break; else continue; }
        // ---- one ray per live lane
        if (alive) {
            HitRec h;
            const uint32_t yb = y0 + (pix >> 6) * 8u;                       // the lane's 8x8 block: the same skip rule as k_render_fused
            if (!fresh || (may_hit && yb + 8u > a.hy0 && yb < a.hy1))
                trace_scene<STATS, TLAS, E, GlobalNodes>(sc, r.O, r.D, r.tmin, r.tmax, r.outside ? CULL_BACK : CULL_FRONT, h, stk, st.cnt);
            else h.hit = false;
            fresh = false;
            ++st.rays;
            if (STATS && first_active_lane()) ++st.passes;
            if (!shade_ray<STATS, TLAS>(sc, a, h, r, acc, np, park, st)) {
                const uint32_t lx = compact1by1(pix & 63u), ly = (pix >> 6) * 8u + compact1by1((pix & 63u) >> 1);
                const uint32_t x = x0 + lx, y = y0 + ly;
                const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x : (size_t)tile_local * (TILE * TILE) + ly * TILE + (wave * 8u + lx);
                store_pixel(a, out_rgba8, out_f32, o, acc);
                alive = false;
            }
        }
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// ---------------------------------------------------------------------------------------------------
// Lane-asynchronous form of the same renderer, for the reference's scene (one identity instance).
//
// k_render_fused keeps the 64 lanes of a wave in lock step: every lane traces "its" ray to the end,
// then all lanes shade, then all trace the next ray -- so each step costs the slowest lane's
// traversal (measured on monkey.obj: ~4x more loop trips than the longest lane needs).  Here every
// lane runs its own pixel's depth-first ray tree as a little state machine (at an internal node /
// holding a leaf / ray finished, waiting to be shaded) and the WAVE picks, each trip, the phase most
// of its lanes are waiting for: internal-node step, triangle step or shading step.  Lanes never wait
// for each other's rays; the expensive shading code runs when a majority needs it.  Arithmetic per
// ray and the order of a pixel's leaves are unchanged, so results are bit-identical to k_render_fused.
template <int STACK, int PEND, bool STATS, bool DIAG = false>
__global__ __launch_bounds__(256) void k_render_async(SceneDev sc, DispatchDev a)
{
    const unsigned long long diag_t0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t diag_trips = 0, diag_tI = 0, diag_tL = 0, diag_tS = 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* stk = lds + wave * (STACK * 64) + lane;

    // Depth slices are interleaved block by block (block b renders slice b % Depth): measured on MI355X,
    // mixing the slices keeps every CU on a blend of cheap background waves and expensive mesh waves
    // (monkey.obj 1080p, Depth 16: 193 us/frame interleaved, 238 us slice-after-slice, 747 us at Depth 1).
    const uint32_t frame = blockIdx.x % a.n_frames;
    uint32_t tile_local, strip;
    block_to_tile(blockIdx.x / a.n_frames, tile_local, strip);
    const bool tile_ok = tile_local < a.n_local_tiles;
    const uint32_t tile = tile_local * a.tile_world + a.tile_rank;
    const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
    const uint32_t px = wave * 8u + lx, py = strip * 8u + ly;         // inside the 32x32 tile
    const uint32_t x = tx * TILE + px, y = ty * TILE + py;
    const bool valid = tile_ok && x < a.W && y < a.H;
    const CamDev& cb = a.cams[frame];                                 // wave-uniform: scalar loads
    uint32_t* const out_rgba8 = a.out_rgba8 + (size_t)frame * a.frame_stride;
    float4* const out_f32 = a.out_f32 ? a.out_f32 + (size_t)frame * a.frame_stride : nullptr;

    uint32_t n_rays = 0, n_hits = 0, n_miss = 0, n_term = 0, n_tir = 0, n_nodes = 0, n_tris = 0;
    uint32_t err = 0;
    const QNode* __restrict__ nodes = sc.blas0.nodes;

    // ---- per-lane state -------------------------------------------------------------------------
    bool alive = valid;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    PendRay pend[PEND];
    int np = 0;
    f3 O = mk3(cb.cam[0], cb.cam[1], cb.cam[2]);                    // RayGen, RayTracing.hlsl:42-60
    f3 D = valid ? camera_ray_dir(cb.M, a.sx[x], a.sy[y]) : mk3(1.0f, 0.0f, 0.0f);
    float w = 1.0f;
    uint32_t count = 0;
    bool outside = true;
    float tmin = a.tmin_p;
    BoxRay br = box_ray(O, D, sc.blas0.scale, sc.blas0.grid);
    HitRec h;
    h.t = a.tmax_p; h.hit = false; h.prim = 0; h.leaf = 0; h.inst = 0; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f;
    int node = 0;
    uint32_t* top = stk;

#ifndef RR_SHADE_SHIFT
#define RR_SHADE_SHIFT 1
#endif
    for (;;) {
        // ---- travel: lanes whose ray is not finished.  The phase ends for the wave once 1/2^RR_SHADE_SHIFT of
        // them are waiting to be shaded (scalar count of exec, as in trace_blas).
        const int n_trav = __popcll(__ballot(alive && node != TRAV_DONE));
        while (alive && node != TRAV_DONE) {
            {
                const int part = n_trav >> RR_SHADE_SHIFT;
                if (__popcll(__ballot(1)) + (part > 1 ? part : 1) <= n_trav) break;
            }
            if (DIAG) ++diag_trips;
            const int n_in = __popcll(__ballot(node >= 0));
            while (node >= 0) {
                if (leaf_phase_due(n_in)) break;
                if (DIAG) ++diag_tI;
                const NodeQ q = load_node(nodes, node);
                if (STATS) ++n_nodes;
                node = node_step(br, q, tmin, h.t, top, stk);
            }
            if (node < 0 && node != TRAV_DONE) {
                if (DIAG) ++diag_tL;
                if (STATS) ++n_tris;
                tri_test(sc.blas0.tris, (uint32_t)~node, O, D, tmin, outside ? CULL_BACK : CULL_FRONT, 0u, h);
                if (top > stk) { top -= STACK_STRIDE; node = (int)*top; } else node = TRAV_DONE;
            }
        }
        {
            const bool wantS = alive && node == TRAV_DONE;
            if (DIAG && __ballot(wantS)) ++diag_tS;
            // ---- shading step: Miss / ClosestHit for every lane whose ray is finished -------------------
            if (wantS) {
                ++n_rays;
                bool have_next = false;
                if (!h.hit) {                                             // Miss, hlsl:127-137
                    if (STATS) ++n_miss;
                    const f3 e = env_lookup(sc, D);
                    acc.x = fmaf(w, e.x, acc.x); acc.y = fmaf(w, e.y, acc.y); acc.z = fmaf(w, e.z, acc.z);
                } else {                                                  // ClosestHit, hlsl:79-125
                    if (STATS) ++n_hits;
                    if ((int)count < a.max_refract) {
                        hit_attributes(sc.blas0.tris, O, D, h);
                        const f3 N = shading_normal<false>(sc, h);
                        const f3 X = mk3(fmaf(h.t, D.x, O.x), fmaf(h.t, D.y, O.y), fmaf(h.t, D.z, O.z));
                        const f3 Nf = outside ? N : neg3(N);
                        const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);
                        const float b = 1.0f - dot3(D, Nf);
                        const float b2 = b * b, b4 = b2 * b2;
                        const float R = (R0 * (1.0f - R0)) * (b4 * b);
                        const float eta = outside ? a.inv_ior : a.ior;
                        f3 d1;
                        const bool refr = refract_ray(d1, D, Nf, eta);
                        if (STATS && !refr) ++n_tir;
                        const bool refl = (int)count < a.max_reflect;
                        f3 d2 = mk3(0.0f, 0.0f, 0.0f);
                        if (refl) d2 = normalize3(reflect_ray(D, Nf));
                        const uint32_t c1 = count + 1u;
                        O = X;
                        if (refr) {
                            if (refl) {
                                PendRay p;
                                p.ox = X.x; p.oy = X.y; p.oz = X.z; p.dx = d2.x; p.dy = d2.y; p.dz = d2.z;
                                p.w = w * R; p.meta = c1 | (outside ? 0x10000u : 0u);
#pragma unroll
                                for (int k = 0; k < PEND; ++k) if (k == np) pend[k] = p;
                                ++np;
                            }
                            D = d1; w = w * (1.0f - R); count = c1; outside = !outside;
                            have_next = true;
                        } else if (refl) {
                            D = d2; w = w * R; count = c1;
                            have_next = true;
                        }
                    } else if (STATS) {
                        ++n_term;
                    }
                }
                if (!have_next && np > 0) {
                    --np;
                    PendRay p = pend[0];
#pragma unroll
                    for (int k = 1; k < PEND; ++k) if (k == np) p = pend[k];
                    O = mk3(p.ox, p.oy, p.oz); D = mk3(p.dx, p.dy, p.dz); w = p.w;
                    count = p.meta & 0xffffu; outside = (p.meta & 0x10000u) != 0u;
                    have_next = true;
                }
                if (have_next) {                                          // TraceRay(child, [1e-3, 1000])
                    tmin = a.tmin_s;
                    br = box_ray(O, D, sc.blas0.scale, sc.blas0.grid);
                    h.t = a.tmax_s; h.hit = false; h.prim = 0; h.leaf = 0; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f;
                    node = 0; top = stk;
                } else {                                                  // RenderTarget[xy] = float4(color,1)
                    const uint32_t packed = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xff000000u;
                    const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x
                                                         : (size_t)tile_local * (TILE * TILE) + py * TILE + px;
                    if (a.compact_out == 2u) {
                        uint8_t* p3 = reinterpret_cast<uint8_t*>(out_rgba8) + o * 3;
                        p3[0] = (uint8_t)packed; p3[1] = (uint8_t)(packed >> 8); p3[2] = (uint8_t)(packed >> 16);
                    } else
                    out_rgba8[o] = packed;
                    if (out_f32) out_f32[o] = make_float4(acc.x, acc.y, acc.z, 1.0f);
                    alive = false;
                }
            }
        }
        if (__ballot(alive) == 0ull) break;
    }

    if (DIAG) {
        uint32_t mx = n_rays;
        for (int off = 32; off > 0; off >>= 1) { uint32_t v = __shfl_xor(mx, off, 64); mx = v > mx ? v : mx; }
        if (lane == 0) {
            unsigned long long* d = a.diag + (size_t)(blockIdx.x * 4u + wave) * 4;
            d[0] = ((unsigned long long)diag_tI << 40) | ((unsigned long long)diag_tL << 20) | diag_tS;
            d[1] = __builtin_amdgcn_s_memtime() - diag_t0; d[2] = mx; d[3] = diag_trips;
        }
    }
    uint32_t wr = wave_reduce_add(n_rays);
    if (lane == 0 && wr) atomicAdd(&a.ray_shards[(blockIdx.x * 4u + wave) & (RAY_SHARDS - 1)], wr);
    if (err) atomicOr(a.error_flag, 1u);
    if (STATS) {
        uint32_t v;
        v = wave_reduce_add(n_hits);  if (lane == 0 && v) atomicAdd(&a.counters[C_HITS], (unsigned long long)v);
        v = wave_reduce_add(n_miss);  if (lane == 0 && v) atomicAdd(&a.counters[C_MISSES], (unsigned long long)v);
        v = wave_reduce_add(n_term);  if (lane == 0 && v) atomicAdd(&a.counters[C_TERMINAL], (unsigned long long)v);
        v = wave_reduce_add(n_tir);   if (lane == 0 && v) atomicAdd(&a.counters[C_TIR], (unsigned long long)v);
        v = wave_reduce_add(n_nodes); if (lane == 0 && v) atomicAdd(&a.counters[C_NODES], (unsigned long long)v);
        v = wave_reduce_add(n_tris);  if (lane == 0 && v) atomicAdd(&a.counters[C_TRIS], (unsigned long long)v);
        v = wave_reduce_add(valid ? 1u : 0u); if (lane == 0 && v) atomicAdd(&a.counters[C_PRIMARY], (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------------------------------
// Experimental queue-per-bounce ("wavefront path tracing") form of the same renderer, for comparison with the
// fused kernel (RR_DEBUG_KERNEL=wavefront).  One kernel per ray generation: rays of bounce g are read from a queue,
// traced and shaded; children go to the queue of bounce g+1 (one atomic per wave and child kind).  A pixel's ray
// tree branches only while count < max_reflect, so with max_reflect <= 2 it has at most four leaves; leaf k (in the
// recursion's depth-first order: refraction before reflection) writes its (weight, texel) to slot k of the pixel, and
// a last kernel sums the four slots in order -- the same fma sequence as the fused kernel, bit for bit, whatever
// order the queues were filled in.
struct WfRay { f3 O, D; float w; uint32_t pix, count, slot; bool outside; };

__device__ __forceinline__ void wf_store(float4* q, uint32_t i, const WfRay& r)
{
    q[(size_t)i * 3 + 0] = make_float4(r.O.x, r.O.y, r.O.z, r.w);
    q[(size_t)i * 3 + 1] = make_float4(r.D.x, r.D.y, r.D.z, __uint_as_float(r.pix));
    q[(size_t)i * 3 + 2] = make_float4(__uint_as_float(r.count | (r.outside ? 0x10000u : 0u) | (r.slot << 20)), 0.0f, 0.0f, 0.0f);
}
__device__ __forceinline__ WfRay wf_load(const float4* q, uint32_t i)
{
    const float4 a = q[(size_t)i * 3 + 0], b = q[(size_t)i * 3 + 1], c = q[(size_t)i * 3 + 2];
    WfRay r;
    r.O = mk3(a.x, a.y, a.z); r.w = a.w; r.D = mk3(b.x, b.y, b.z); r.pix = __float_as_uint(b.w);
    const uint32_t m = __float_as_uint(c.x);
    r.count = m & 0xffffu; r.outside = (m & 0x10000u) != 0u; r.slot = m >> 20;
    return r;
}
// append the rays of the lanes with `have` set: one atomic per wave
__device__ __forceinline__ void wf_push(const WfBuffers& wf, int gen, bool have, const WfRay& r, uint32_t* err)
{
    const unsigned long long m = __ballot(have);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const int first = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == first) base = atomicAdd(&wf.counts[gen], (uint32_t)__popcll(m));
    base = __shfl(base, first, 64);
    const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (have) { if (idx < wf.cap) wf_store(wf.q[gen & 1], idx, r); else *err = 1u; }
}

// ClosestHit for one lane: emits up to two children (refracted first).  Returns false for a terminal hit.
__device__ __forceinline__ void wf_shade_hit(const SceneDev& sc, const DispatchDev& a, const WfRay& in, const HitRec& h,
                                             bool& refr, bool& refl, WfRay& c1, WfRay& c2)
{
    refr = false; refl = false;
    if ((int)in.count >= a.max_refract) return;                       // hlsl:82 (payload.color stays 0)
    const f3 N = shading_normal<false>(sc, h);
    const f3 X = mk3(fmaf(h.t, in.D.x, in.O.x), fmaf(h.t, in.D.y, in.O.y), fmaf(h.t, in.D.z, in.O.z));
    const f3 Nf = in.outside ? N : neg3(N);
    const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);
    const float b = 1.0f - dot3(in.D, Nf);
    const float b2 = b * b, b4 = b2 * b2;
    const float R = (R0 * (1.0f - R0)) * (b4 * b);
    const float eta = in.outside ? a.inv_ior : a.ior;
    f3 d1;
    refr = refract_ray(d1, in.D, Nf, eta);
    refl = (int)in.count < a.max_reflect;
    f3 d2 = mk3(0.0f, 0.0f, 0.0f);
    if (refl) d2 = normalize3(reflect_ray(in.D, Nf));
    const uint32_t bit = in.count < 2u ? (2u >> in.count) : 0u;       // the reflected branch at depth 0 / 1 owns slots 2,3 / 1,3
    c1.O = X; c1.D = d1; c1.pix = in.pix; c1.count = in.count + 1u; c1.slot = in.slot;
    c2.O = X; c2.D = d2; c2.pix = in.pix; c2.count = in.count + 1u; c2.slot = in.slot | bit; c2.outside = in.outside;
    if (refr) { c1.w = in.w * (1.0f - R); c1.outside = !in.outside; c2.w = in.w * R; }
    else { c1.w = 0.0f; c1.outside = in.outside; c2.w = in.w * R; }
}

template <int STACK>
__global__ __launch_bounds__(256, RR_FUSED_WAVES_PER_SIMD(STACK)) void k_wf_primary(SceneDev sc, DispatchDev a, WfBuffers wf)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* stk = lds + wave * (STACK * 64) + lane;
    const uint32_t frame = blockIdx.x % a.n_frames;
    uint32_t tile_local, strip;
    block_to_tile(blockIdx.x / a.n_frames, tile_local, strip);
    const bool tile_ok = tile_local < a.n_local_tiles;
    const uint32_t tile = tile_local * a.tile_world + a.tile_rank;
    const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
    const uint32_t px = wave * 8u + lx, py = strip * 8u + ly;
    const uint32_t x = tx * TILE + px, y = ty * TILE + py;
    const bool valid = tile_ok && x < a.W && y < a.H;
    const CamDev& cb = a.cams[frame];
    uint32_t err = 0;
    bool refr = false, refl = false;
    WfRay c1, c2;
    c1.O = c1.D = c2.O = c2.D = mk3(0.0f, 0.0f, 0.0f); c1.w = c2.w = 0.0f; c1.pix = c2.pix = 0u; c1.count = c2.count = 0u;
    c1.slot = c2.slot = 0u; c1.outside = c2.outside = true;
    if (valid) {
        WfRay r;
        r.O = mk3(cb.cam[0], cb.cam[1], cb.cam[2]);
        r.D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
        r.w = 1.0f; r.count = 0u; r.slot = 0u; r.outside = true;
        r.pix = (uint32_t)((size_t)frame * a.frame_stride + (size_t)y * a.W + x);
        HitRec h;
        TravCounters cnt; cnt.nodes = 0; cnt.tris = 0;
        trace_scene<false, false>(sc, r.O, r.D, a.tmin_p, a.tmax_p, CULL_BACK, h, stk, cnt);
        if (!h.hit) {                                            // the pixel's only leaf: acc = fma(1, texel, 0)
            const f3 e = env_lookup(sc, r.D);
            const f3 acc = mk3(fmaf(1.0f, e.x, 0.0f), fmaf(1.0f, e.y, 0.0f), fmaf(1.0f, e.z, 0.0f));
            a.out_rgba8[r.pix] = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xff000000u;
        } else {
            wf_shade_hit(sc, a, r, h, refr, refl, c1, c2);
            if (!refr && !refl) a.out_rgba8[r.pix] = 0xff000000u;                    // no child: black
            else {
                const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float4* sl = wf.slots + (size_t)r.pix * 4;
                sl[0] = z; sl[1] = z; sl[2] = z; sl[3] = z;
            }
        }
    }
    // covered pixels and their children (wave-aggregated appends)
    {
        const bool cov = refr || refl;
        const unsigned long long m = __ballot(cov);
        if (m) {
            const int first = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if ((int)lane == first) base = atomicAdd(&wf.counts[63], (uint32_t)__popcll(m));
            base = __shfl(base, first, 64);
            if (cov) wf.hit_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c2.pix;
        }
    }
    wf_push(wf, 1, refr, c1, &err);
    wf_push(wf, 1, refl, c2, &err);
    if (err) atomicOr(a.error_flag, 1u);
}

template <int STACK>
__global__ __launch_bounds__(256, RR_FUSED_WAVES_PER_SIMD(STACK)) void k_wf_bounce(SceneDev sc, DispatchDev a, WfBuffers wf, int gen)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* stk = lds + wave * (STACK * 64) + lane;
    uint32_t n = wf.counts[gen];
    n = n < wf.cap ? n : wf.cap;
    const float4* qin = wf.q[gen & 1];
    uint32_t err = 0;
    for (uint32_t base = (blockIdx.x * 4u + wave) * 64u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + lane;
        bool refr = false, refl = false;
        WfRay c1, c2;
        c1.O = c1.D = c2.O = c2.D = mk3(0.0f, 0.0f, 0.0f); c1.w = c2.w = 0.0f; c1.pix = c2.pix = 0u; c1.count = c2.count = 0u;
        c1.slot = c2.slot = 0u; c1.outside = c2.outside = true;
        if (i < n) {
            const WfRay r = wf_load(qin, i);
            HitRec h;
            TravCounters cnt; cnt.nodes = 0; cnt.tris = 0;
            trace_scene<false, false>(sc, r.O, r.D, a.tmin_s, a.tmax_s, r.outside ? CULL_BACK : CULL_FRONT, h, stk, cnt);
            if (!h.hit) {
                const f3 e = env_lookup(sc, r.D);
                wf.slots[(size_t)r.pix * 4 + r.slot] = make_float4(r.w, e.x, e.y, e.z);
            } else {
                wf_shade_hit(sc, a, r, h, refr, refl, c1, c2);
            }
        }
        wf_push(wf, gen + 1, refr, c1, &err);
        wf_push(wf, gen + 1, refl, c2, &err);
    }
    if (err) atomicOr(a.error_flag, 1u);
}

__global__ __launch_bounds__(256) void k_wf_resolve(DispatchDev a, WfBuffers wf)
{
    const uint32_t n = wf.counts[63];
    if (blockIdx.x == 0 && threadIdx.x == 0) {                   // TraceRay calls of the dispatch: primaries + every queued ray
        uint32_t rays = a.W * a.H * a.n_frames;
        for (int g = 1; g < 63; ++g) rays += wf.counts[g];
        atomicAdd(&a.ray_shards[0], rays);
    }
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t pix = wf.hit_list[i];
        const float4* sl = wf.slots + (size_t)pix * 4;
        f3 acc = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 s = sl[k];                              // an unused slot holds w = 0: fma(0, 0, acc) == acc
            acc.x = fmaf(s.x, s.y, acc.x); acc.y = fmaf(s.x, s.z, acc.y); acc.z = fmaf(s.x, s.w, acc.z);
        }
        a.out_rgba8[pix] = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xff000000u;
    }
}

template <int STACK, bool TLAS, class E, int WPS>
static hipError_t launch_refill_ste(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(E) + (size_t)4 * 2 * 8 * 64 * 4;
    const dim3 grid(a.n_local_tiles * a.n_frames);
    if (stats) hipLaunchKernelGGL((k_render_refill<STACK, true, TLAS, E, WPS>), grid, dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_refill<STACK, false, TLAS, E, WPS>), grid, dim3(256), lds, s, sc, a);
    return hipGetLastError();
}

// max_reflect <= 2; stack <= 39 entries; stack16: every stack entry of the scene fits 16 bits
hipError_t launch_render_refill(const SceneDev& sc, const DispatchDev& a, int stack, bool stats, hipStream_t s, bool stack16)
{
    if (a.n_local_tiles == 0) return hipSuccess;
    if (sc.single_identity) {
        if (stack16) return launch_refill_ste<39, false, uint16_t, 8>(sc, a, stats, s);
        return stack <= 19 ? launch_refill_ste<19, false, uint32_t, 6>(sc, a, stats, s) : launch_refill_ste<39, false, uint32_t, 4>(sc, a, stats, s);
    }
    if (stack16) return launch_refill_ste<39, true, uint16_t, 5>(sc, a, stats, s);
    return stack <= 31 ? launch_refill_ste<31, true, uint32_t, 4>(sc, a, stats, s) : launch_refill_ste<39, true, uint32_t, 3>(sc, a, stats, s);
}


template <int STACK, int PEND>
static hipError_t launch_async_sp(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(uint32_t);
    if (stats) hipLaunchKernelGGL((k_render_async<STACK, PEND, true>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_async<STACK, PEND, false>), dim3(a.n_blocks), dim3(256), lds, s, sc, a);
    return hipGetLastError();
}


hipError_t launch_render_async(const SceneDev& sc, const DispatchDev& a, int stack, int pend, bool stats, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    if (a.diag) {
        hipLaunchKernelGGL((k_render_async<31, 2, false, true>), dim3(a.n_blocks), dim3(256), 4 * 31 * 64 * 4, s, sc, a);
        return hipGetLastError();
    }
    if (stack <= 31) return pend <= 2 ? launch_async_sp<31, 2>(sc, a, stats, s) : launch_async_sp<31, 8>(sc, a, stats, s);
    return pend <= 2 ? launch_async_sp<64, 2>(sc, a, stats, s) : launch_async_sp<64, 8>(sc, a, stats, s);
}

template <int STACK>
static hipError_t launch_wavefront_s(const SceneDev& sc, const DispatchDev& a, const WfBuffers& wf, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(uint32_t);
    hipError_t e = hipMemsetAsync(wf.counts, 0, 64 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wf_primary<STACK>), dim3(a.n_blocks), dim3(256), lds, s, sc, a, wf);
    const uint32_t persistent = 256u * 6u;
    for (int g = 1; g <= a.max_refract && g < 62; ++g)
        hipLaunchKernelGGL((k_wf_bounce<STACK>), dim3(persistent), dim3(256), lds, s, sc, a, wf, g);
    hipLaunchKernelGGL(k_wf_resolve, dim3(persistent), dim3(256), 0, s, a, wf);
    return hipGetLastError();
}

hipError_t launch_render_wavefront(const SceneDev& sc, const DispatchDev& a, const WfBuffers& wf, int stack, hipStream_t s)
{
    if (stack <= 26) return launch_wavefront_s<26>(sc, a, wf, s);
    if (stack <= 31) return launch_wavefront_s<31>(sc, a, wf, s);
    return launch_wavefront_s<64>(sc, a, wf, s);
}

// ---------------------------------------------------------------------------------------------------
// Lane-asynchronous renderer for scenes beyond the reference's single instance (a TLAS over many instances).  EXPERIMENT
// (RR_DEBUG_KERNEL=scene-async): on the 1 024-monkey grid it cuts the internal-node trips from 63 M to 23 M per frame
// (lane utilisation 17 % -> 45 %) and still takes 12.9 ms against k_render_fused's 8.9 -- a leaf step (triangle test,
// instance entry and exit, each run in turn for the few lanes that need it) and a shading pass cost ten to twenty times an
// internal-node step, and this form makes as many of them as the lock-step one (thresholds from 1/4 to 7/8 tried).
// There the rays of a wave differ enormously in length -- on the 1 024-monkey grid a shading pass of k_render_fused lasts
// 95 loop trips for rays that need 29 on average (a ray skimming the grid crosses dozens of instance boxes), and only 17 % of
// the lanes of an internal-node step do anything.  Here every lane runs its own pixel as a little machine -- at an
// internal node / holding a leaf or the end of a subtree / ray finished, to be shaded / without a pixel -- and the WAVE
// picks, each trip, the one kind of step most of its lanes are waiting for: an internal-node step, a leaf step (triangle
// test, instance entry or exit) or a shading pass (ClosestHit / Miss, next ray or next pixel of the wave's 256-pixel
// column).  A lane never waits for another lane's ray.  Every lane still performs exactly the operations of
// trace_scene<TLAS> and shade_ray for its own rays, in the same order, so frames and counters are bit-identical to
// k_render_fused's.
template <int STACK, bool STATS, class E, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_render_scene_async(SceneDev sc, DispatchDev a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    E* const stk = reinterpret_cast<E*>(lds) + wave * (STACK * 64) + lane;
    LdsPark park{ lds + (4u * STACK * 64u * (uint32_t)sizeof(E)) / 4u + wave * (2u * 8u * 64u) + lane };

    // workgroup = one 32x32 tile of one slice (slices interleaved); wave w = its 8-pixel-wide column w (256 pixels)
    const uint32_t frame = blockIdx.x % a.n_frames;
    const uint32_t tile_local = blockIdx.x / a.n_frames;
    if (tile_local >= a.n_local_tiles) return;
    const uint32_t tile = tile_local * a.tile_world + a.tile_rank;
    const uint32_t x0 = (tile % a.tiles_x) * TILE + wave * 8u, y0 = (tile / a.tiles_x) * TILE;
    const CamDev& cb = a.cams[frame];
    uint32_t* const out_rgba8 = a.out_rgba8 + (size_t)frame * a.frame_stride;
    float4* const out_f32 = a.out_f32 ? a.out_f32 + (size_t)frame * a.frame_stride : nullptr;
    const bool col_may_hit = x0 + 8u > a.hx0 && x0 < a.hx1;

    constexpr int TRAV_FIN = (int)0x80000001;       // the lane's ray is finished: shade it
    constexpr uint32_t NO_INST = 0xffffffffu;
    constexpr uint32_t TOTAL = 4u * 64u;            // pixels of the column
    const QNode* __restrict__ nodes = sc.pool_nodes;

    LaneStats st;
    st.blocks = 4u;
    uint32_t cursor = 0;                            // next pixel of the column to hand out (wave-uniform)
    bool alive = false;
    uint32_t pix = 0;
    RayState r;
    r.O = r.D = mk3(0.0f, 0.0f, 0.0f); r.w = 0.0f; r.tmin = r.tmax = 0.0f; r.count = 0; r.outside = true;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    int np = 0;
    // traversal state of the lane's ray
    int node = TRAV_FIN;
    E* top = stk;
    const E* floor = stk;
    uint32_t cur = NO_INST, cull = 0;
    f3 Oc = r.O, Dc = r.D;
    BoxRay br = box_ray(r.O, mk3(1.0f, 1.0f, 1.0f), sc.scale, sc.grid);
    HitRec best;
    best.t = 0.0f; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = best.V = 0.0f; best.ad = 1.0f;

    auto start_ray = [&](bool traced) {             // TraceRay(r): set the traversal up (traced = false: a Miss without a trace)
        best.t = r.tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f; best.ad = 1.0f;
        cull = r.outside ? CULL_BACK : CULL_FRONT;
        cur = NO_INST; Oc = r.O; Dc = r.D; top = stk; floor = stk;
        br = box_ray(r.O, r.D, sc.scale, sc.grid);
        node = traced ? 0 : TRAV_FIN;
    };

    for (;;) {
        // ---- lanes without a pixel take the next ones of the column
        if (cursor < TOTAL) {
            const unsigned long long need = __ballot(!alive);
            if (need) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                const uint32_t mine = cursor + rank;
                if (!alive && mine < TOTAL) {
                    pix = mine;
                    const uint32_t yb = y0 + (pix >> 6) * 8u;
                    const uint32_t x = x0 + compact1by1(pix & 63u), y = yb + compact1by1((pix & 63u) >> 1);
                    if (x < a.W && y < a.H) {
                        r = primary_ray(a, cb, x, y);
                        acc = mk3(0.0f, 0.0f, 0.0f); np = 0;
                        alive = true;
                        st.pixels += 1;
                        start_ray(col_may_hit && yb + 8u > a.hy0 && yb < a.hy1);     // the same skip rule as k_render_fused, per 8x8 block
                    }
                }
                cursor += (uint32_t)__popcll(need);
            }
        }
        const unsigned long long m_alive = __ballot(alive);
        if (m_alive == 0ull) { if (cursor >= TOTAL) break; else continue; }
        const unsigned long long m_fin = __ballot(alive && node == TRAV_FIN);
        const unsigned long long m_node = __ballot(alive && node >= 0);
        const unsigned long long m_leaf = m_alive & ~m_fin & ~m_node;             // holding a leaf / at the end of a subtree
        const int n_alive = __popcll(m_alive), n_fin = __popcll(m_fin), n_node = __popcll(m_node), n_leaf = __popcll(m_leaf);
        const int n_trav = n_node + n_leaf;
        // which step: shade once a quarter of the live lanes wait for it (or nothing else is left); among the travelling lanes a
        // leaf step once a quarter of them hold one (or none is at an internal node)
        if (n_fin > 0 && (n_trav == 0 || n_fin * 8 >= n_alive * (int)a.async_shade_num)) {
            if (alive && node == TRAV_FIN) {
                if (best.hit) {                         // the ray in the space of the instance that was hit, as at its entry
                    const InstDev& in = sc.insts[best.inst];
                    f3 Oh = r.O, Dh = r.D;
                    if (!in.identity) { Oh = xform_point(in.inv, r.O); Dh = xform_dir(in.inv, r.D); }
                    hit_attributes(sc.pool_tris, Oh, Dh, best);
                }
                ++st.rays;
                if (STATS && first_active_lane()) ++st.passes;
                if (shade_ray<STATS, true>(sc, a, best, r, acc, np, park, st)) {
                    start_ray(true);
                } else {
                    const uint32_t lx = compact1by1(pix & 63u), ly = (pix >> 6) * 8u + compact1by1((pix & 63u) >> 1);
                    const uint32_t x = x0 + lx, y = y0 + ly;
                    const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x : (size_t)tile_local * (TILE * TILE) + ly * TILE + (wave * 8u + lx);
                    store_pixel(a, out_rgba8, out_f32, o, acc);
                    alive = false;
                }
            }
        } else if (n_node > 0 && n_leaf * 8 < n_trav * (int)a.async_leaf_num) {
            if (alive && node >= 0) {                   // internal-node step
                const NodeQ q = load_node(nodes, node);
                if (STATS) { st.cnt.nodes++; if (first_active_lane()) st.cnt.node_trips++; }
                node = node_step(br, q, r.tmin, best.t, top, floor);
            }
        } else {
            if (alive && node < 0 && node != TRAV_FIN) {    // leaf step
                if (STATS && first_active_lane()) st.cnt.leaf_trips++;
                if (node == TRAV_DONE) {
                    if (cur == NO_INST) node = TRAV_FIN;
                    else {                              // leave the instance
                        cur = NO_INST; Oc = r.O; Dc = r.D; cull = r.outside ? CULL_BACK : CULL_FRONT; floor = stk;
                        br = box_ray(r.O, r.D, sc.scale, sc.grid);
                        if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_FIN;
                    }
                } else {
                    const uint32_t L = (uint32_t)~node;
                    if (L < sc.n_pool_tris) {
                        if (STATS) st.cnt.tris++;
                        tri_test(sc.pool_tris, L, Oc, Dc, r.tmin, cull, cur, best);
                        if (top > floor) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
                    } else {
                        const uint32_t ii = L - sc.n_pool_tris;
                        const InstDev& in = sc.insts[ii];
                        if (in.mask & 0xffu) {                                            // InstanceInclusionMask 0xff
                            uint32_t f = r.outside ? CULL_BACK : CULL_FRONT;
                            if (in.flags & 0x1u) f &= ~(CULL_BACK | CULL_FRONT);          // TRIANGLE_CULL_DISABLE
                            else if (in.flags & 0x2u) {                                    // TRIANGLE_FRONT_COUNTERCLOCKWISE
                                if (f & CULL_BACK) f = (f & ~CULL_BACK) | CULL_FRONT;
                                else if (f & CULL_FRONT) f = (f & ~CULL_FRONT) | CULL_BACK;
                            }
                            cull = f; cur = ii; floor = top;
                            if (!in.identity) { Oc = xform_point(in.inv, r.O); Dc = xform_dir(in.inv, r.D); }
                            br = box_ray(Oc, Dc, in.scale, in.grid);
                            node = (int)in.root;
                        } else if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
                    }
                }
            }
        }
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}


// ---------------------------------------------------------------------------------------------------
// The same lane-asynchronous renderer as a STREAMING kernel (RR_DEBUG_KERNEL=scene-stream): persistent waves that pull 8x8
// pixel blocks from one ticket counter for the whole launch (numbered as k_render_fused's wave-blocks), a lane taking the next
// pixel of the wave's current block as soon as its own is finished -- so a wave has no tail of its own, only the launch has
// one --, parked rays in registers / scratch instead of LDS and the 30-entry 16-bit stacks, i.e. the LDS and register budget
// of the seven-wave lock-step build, and several internal-node steps per vote.
template <int STACK, bool STATS, class E, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_render_scene_stream(SceneDev sc, DispatchDev a, uint32_t* ticket)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    E* const stk = reinterpret_cast<E*>(lds) + wave * (STACK * 64) + lane;
    RegPark<2> park;

    constexpr int TRAV_FIN = (int)0x80000001;       // the lane's ray is finished: shade it
    constexpr uint32_t NO_INST = 0xffffffffu;
    const QNode* __restrict__ nodes = sc.pool_nodes;
    const uint32_t total_blocks = a.n_blocks * 4u;  // wave-blocks of the launch

    LaneStats st;
    // the block the wave is handing out: its position, the pixels not yet taken
    uint32_t res_left = 0, res_x0 = 0, res_y0 = 0, res_frame = 0;
    bool res_may_hit = false, pool_empty = false;
    bool alive = false;
    uint32_t xy = 0, frame = 0;                     // the lane's pixel
    RayState r;
    r.O = r.D = mk3(0.0f, 0.0f, 0.0f); r.w = 0.0f; r.tmin = r.tmax = 0.0f; r.count = 0; r.outside = true;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    int np = 0;
    int node = TRAV_FIN;
    E* top = stk;
    const E* floor = stk;
    uint32_t cur = NO_INST, cull = 0;
    f3 Oc = r.O, Dc = r.D;
    BoxRay br = box_ray(r.O, mk3(1.0f, 1.0f, 1.0f), sc.scale, sc.grid);
    HitRec best;
    best.t = 0.0f; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = best.V = 0.0f; best.ad = 1.0f;

    auto start_ray = [&](bool traced) {
        best.t = r.tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f; best.ad = 1.0f;
        cull = r.outside ? CULL_BACK : CULL_FRONT;
        cur = NO_INST; Oc = r.O; Dc = r.D; top = stk; floor = stk;
        br = box_ray(r.O, r.D, sc.scale, sc.grid);
        node = traced ? 0 : TRAV_FIN;
    };

    for (;;) {
        // ---- lanes without a pixel take the next ones of the wave's block; an empty block is replaced from the pool
        unsigned long long need = __ballot(!alive);
        while (need != 0ull && !(pool_empty && res_left == 0u)) {
            if (res_left == 0u) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(ticket, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= total_blocks) { pool_empty = true; break; }
                const BlockPos bp = wave_block_pos(a, t);
                if (!bp.tile_ok) continue;
                res_x0 = bp.x0; res_y0 = bp.y0; res_frame = bp.frame; res_left = 64u;
                res_may_hit = bp.x0 + 8u > a.hx0 && bp.x0 < a.hx1 && bp.y0 + 8u > a.hy0 && bp.y0 < a.hy1;
                if (STATS) st.blocks += 1u;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t first = 64u - res_left;              // pixels of the block already handed out
            if (!alive && rank < res_left) {
                const uint32_t pix = first + rank;
                const uint32_t x = res_x0 + compact1by1(pix), y = res_y0 + compact1by1(pix >> 1);
                if (x < a.W && y < a.H) {
                    xy = x | (y << 16); frame = res_frame;
                    r = primary_ray(a, a.cams[res_frame], x, y);
                    acc = mk3(0.0f, 0.0f, 0.0f); np = 0;
                    alive = true;
                    st.pixels += 1;
                    start_ray(res_may_hit);
                }
            }
            const uint32_t taken = (uint32_t)__popcll(need) < res_left ? (uint32_t)__popcll(need) : res_left;
            res_left -= taken;
            need = __ballot(!alive);
            if (res_left != 0u) break;              // lanes still without a pixel got an off-screen one: they ask again next trip
        }
        const unsigned long long m_alive = __ballot(alive);
        if (m_alive == 0ull) { if (pool_empty && res_left == 0u) break; else continue; }
        const unsigned long long m_fin = __ballot(alive && node == TRAV_FIN);
        const unsigned long long m_node = __ballot(alive && node >= 0);
        const unsigned long long m_leaf = m_alive & ~m_fin & ~m_node;
        const int n_alive = __popcll(m_alive), n_fin = __popcll(m_fin), n_node = __popcll(m_node), n_leaf = __popcll(m_leaf);
        const int n_trav = n_node + n_leaf;
        if (n_fin > 0 && (n_trav == 0 || n_fin * 8 >= n_alive * (int)a.async_shade_num)) {
            if (alive && node == TRAV_FIN) {
                if (best.hit) {
                    const InstDev& in = sc.insts[best.inst];
                    f3 Oh = r.O, Dh = r.D;
                    if (!in.identity) { Oh = xform_point(in.inv, r.O); Dh = xform_dir(in.inv, r.D); }
                    hit_attributes(sc.pool_tris, Oh, Dh, best);
                }
                ++st.rays;
                if (STATS && first_active_lane()) ++st.passes;
                if (shade_ray<STATS, true>(sc, a, best, r, acc, np, park, st)) {
                    start_ray(true);
                } else {
                    const uint32_t x = xy & 0xffffu, y = xy >> 16;
                    store_pixel(a, a.out_rgba8 + (size_t)frame * a.frame_stride, a.out_f32 ? a.out_f32 + (size_t)frame * a.frame_stride : nullptr,
                                (size_t)y * a.W + x, acc);
                    alive = false;
                }
            }
        } else if (n_node > 0 && n_leaf * 8 < n_trav * (int)a.async_leaf_num) {
            // internal-node steps, until the lanes holding a leaf (or the end of a subtree) reach the share at which the leaf step runs
            if (alive && node >= 0) {
                const int stop = (n_trav * (8 - (int)a.async_leaf_num) + 7) / 8;       // lanes at internal nodes below which the vote changes
                do {
                    const NodeQ q = load_node(nodes, node);
                    if (STATS) { st.cnt.nodes++; if (first_active_lane()) st.cnt.node_trips++; }
                    node = node_step(br, q, r.tmin, best.t, top, floor);
                } while (node >= 0 && __popcll(__ballot(1)) > stop);
            }
        } else {
            if (alive && node < 0 && node != TRAV_FIN) {    // leaf step
                if (STATS && first_active_lane()) st.cnt.leaf_trips++;
                if (node == TRAV_DONE) {
                    if (cur == NO_INST) node = TRAV_FIN;
                    else {
                        cur = NO_INST; Oc = r.O; Dc = r.D; cull = r.outside ? CULL_BACK : CULL_FRONT; floor = stk;
                        br = box_ray(r.O, r.D, sc.scale, sc.grid);
                        if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_FIN;
                    }
                } else {
                    const uint32_t L = (uint32_t)~node;
                    if (L < sc.n_pool_tris) {
                        if (STATS) st.cnt.tris++;
                        tri_test(sc.pool_tris, L, Oc, Dc, r.tmin, cull, cur, best);
                        if (top > floor) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
                    } else {
                        const uint32_t ii = L - sc.n_pool_tris;
                        const InstDev& in = sc.insts[ii];
                        if (in.mask & 0xffu) {
                            uint32_t f = r.outside ? CULL_BACK : CULL_FRONT;
                            if (in.flags & 0x1u) f &= ~(CULL_BACK | CULL_FRONT);
                            else if (in.flags & 0x2u) {
                                if (f & CULL_BACK) f = (f & ~CULL_BACK) | CULL_FRONT;
                                else if (f & CULL_FRONT) f = (f & ~CULL_FRONT) | CULL_BACK;
                            }
                            cull = f; cur = ii; floor = top;
                            if (!in.identity) { Oc = xform_point(in.inv, r.O); Dc = xform_dir(in.inv, r.D); }
                            br = box_ray(Oc, Dc, in.scale, in.grid);
                            node = (int)in.root;
                        } else if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
                    }
                }
            }
        }
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

template <int STACK, class E, int WPS>
static hipError_t launch_scene_stream_se(const SceneDev& sc, const DispatchDev& a, uint32_t* ticket, int n_cus, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(E);
    const dim3 grid((uint32_t)n_cus * WPS);
    hipError_t e = hipMemsetAsync(ticket, 0, 4, s);
    if (e != hipSuccess) return e;
    if (stats) hipLaunchKernelGGL((k_render_scene_stream<STACK, true, E, WPS>), grid, dim3(256), lds, s, sc, a, ticket);
    else       hipLaunchKernelGGL((k_render_scene_stream<STACK, false, E, WPS>), grid, dim3(256), lds, s, sc, a, ticket);
    return hipGetLastError();
}

// scenes with a TLAS, unsharded raster frames, max_reflect <= 2, every stack entry fits 16 bits, trees of at most 30 levels
hipError_t launch_render_scene_stream(const SceneDev& sc, const DispatchDev& a, uint32_t* ticket, int n_cus, int waves, bool stats, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    if (waves >= 7) return launch_scene_stream_se<30, uint16_t, 7>(sc, a, ticket, n_cus, stats, s);
    if (waves == 6) return launch_scene_stream_se<30, uint16_t, 6>(sc, a, ticket, n_cus, stats, s);
    return launch_scene_stream_se<30, uint16_t, 5>(sc, a, ticket, n_cus, stats, s);
}

template <int STACK, class E, int WPS>
static hipError_t launch_scene_async_se(const SceneDev& sc, const DispatchDev& a, bool stats, hipStream_t s)
{
    const size_t lds = (size_t)4 * STACK * 64 * sizeof(E) + (size_t)4 * 2 * 8 * 64 * 4;
    const dim3 grid(a.n_local_tiles * a.n_frames);
    if (stats) hipLaunchKernelGGL((k_render_scene_async<STACK, true, E, WPS>), grid, dim3(256), lds, s, sc, a);
    else       hipLaunchKernelGGL((k_render_scene_async<STACK, false, E, WPS>), grid, dim3(256), lds, s, sc, a);
    return hipGetLastError();
}

// scenes with a TLAS; max_reflect <= 2; stack <= 39 entries; stack16: every stack entry of the scene fits 16 bits
hipError_t launch_render_scene_async(const SceneDev& sc, const DispatchDev& a, int stack, bool stats, hipStream_t s, bool stack16)
{
    if (a.n_local_tiles == 0) return hipSuccess;
    if (stack16) return stack <= 31 ? launch_scene_async_se<31, uint16_t, 4>(sc, a, stats, s) : launch_scene_async_se<39, uint16_t, 4>(sc, a, stats, s);
    return stack <= 31 ? launch_scene_async_se<31, uint32_t, 3>(sc, a, stats, s) : launch_scene_async_se<39, uint32_t, 3>(sc, a, stats, s);
}


} // namespace rr
