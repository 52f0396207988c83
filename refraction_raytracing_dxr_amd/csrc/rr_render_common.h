// rr_render_common.h -- device code shared by the render kernels (rr_render.hip, rr_render_stream.hip): a pixel's ray tree as the lanes walk it (RayGen, ClosestHit / Miss, the parked
// reflected rays), the frame store, the counters, and the numbering of a dispatch's 8x8 pixel blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rr_device.h"
#include "rr_launch.h"

// Stack sizes and waves per SIMD go together: a workgroup's four stacks take STACK KiB of LDS and a CU lets about
// 156 KiB be allocated (tools/ubench_occupancy.hip: six workgroups are resident up to 26 624 B each, five up to
// 31 744 B, four up to 40 960 B -- 32 768 B already drops to four).  So the instantiations are 26 entries (6 waves
// per SIMD, 80 VGPRs), 31 (5 waves, 96 VGPRs), 39 (4 waves) and 64 (2 waves).  Measured at Depth 64, us/frame on
// monkey / sphere: 4 waves 123 / 214, 5 waves 108 / 189, 6 waves (15 words spilled) 104 / 183.
#ifndef RR_FUSED_WAVES_PER_SIMD
#define RR_FUSED_WAVES_PER_SIMD(STACK) ((STACK) <= 19 ? 8 : (STACK) <= 22 ? 7 : (STACK) <= 26 ? 6 : (STACK) <= 31 ? 5 : (STACK) <= 39 ? 4 : 2)
#endif
#ifndef RR_TLAS_WAVES_PER_SIMD
#define RR_TLAS_WAVES_PER_SIMD(STACK) ((STACK) <= 31 ? 5 : (STACK) <= 39 ? 4 : 2)      // C5: 9.70 -> 9.35 ms with 5 (96 VGPRs, spills)
#endif

namespace rr {


struct PendRay {
    float ox, oy, oz, dx, dy, dz, w;
    uint32_t meta;          // count | outside << 16
};

__device__ __forceinline__ uint32_t compact1by1(uint32_t v)
{
    v &= 0x55555555u;
    v = (v ^ (v >> 1)) & 0x33333333u;
    v = (v ^ (v >> 2)) & 0x0f0f0f0fu;
    return v;
}

// Strip-slot s of a slice -> (tile, strip): the four 32x8 strips of a tile sit eight slots apart and consecutive tiles in
// consecutive slots.  In a launch of ONE slice workgroups b and b+8 go to the same XCD (round-robin dispatch), so a tile's
// strips share an XCD (BVH subtrees and env-map lines in that XCD's L2) while consecutive tiles go to consecutive XCDs:
// coverage is centre-heavy (the mesh fills ~7 % of the frame but owns a third of the rays), so giving an XCD a contiguous
// image region would leave most of the chip idle.  With Depth slices interleaved (workgroup b renders slice b % Depth) the XCD
// of a workgroup is (slot * Depth + slice) % 8: for Depth a multiple of 8 an XCD keeps to every eighth SLICE instead and sees
// all of its tiles and strips.
__device__ __forceinline__ void block_to_tile(uint32_t b, uint32_t& tile_local, uint32_t& strip)
{
    const uint32_t xcd = b & 7u, slot = b >> 3;
    tile_local = (slot >> 2) * 8u + xcd;
    strip = slot & 3u;
}

// per-lane tallies of one wave's work (reduced and added to the dispatch counters once, when the wave ends)
struct LaneStats {
    uint32_t rays = 0, hits = 0, miss = 0, term = 0, tir = 0, pixels = 0;
    uint32_t passes = 0;            // wave-level shading passes (ray rounds), STATS builds
    uint32_t blocks = 0;            // 8x8 pixel blocks this wave rendered (wave-uniform)
    uint32_t bg_blocks = 0;         // of those, background blocks rendered on the RayGen + Miss branch (wave-uniform)
    unsigned long long clk0 = 0, rt0 = 0;   // STATS builds: the wave's start on the shader clock and on the 100 MHz reference
    TravCounters cnt = { 0, 0 };
};
template <bool STATS> __device__ __forceinline__ void stats_clock_begin(LaneStats& st)
{
    if (STATS) { st.clk0 = __builtin_amdgcn_s_memtime(); st.rt0 = __builtin_amdgcn_s_memrealtime(); }
}

// One pixel: RayGen (RayTracing.hlsl:42-64), then the pixel's whole ray tree depth-first -- ClosestHit (hlsl:79-125)
// spawns the refracted child (followed at once) and the reflected child (parked in registers), Miss (hlsl:127-137)
// adds weight * texel.  Returns the pixel's colour, the sum of its leaves in the recursion's order.
// Where a pixel's parked reflected rays wait.  RegPark: in registers (PEND slots of 8 words: they stay live through every
// traversal and are most of what the 64-register builds spill).  MemPark: in a per-wave slab of device memory, field by
// field, 64 lanes to a 256-byte row -- written once per spawned reflection and read once when it is resumed (0.3 times per
// ray), so the traversal loops carry 8 * PEND registers less.
template <int PEND>
struct RegPark {
    PendRay pend[PEND];
    __device__ __forceinline__ void put(int k, const PendRay& p)
    {
#pragma unroll
        for (int i = 0; i < PEND; ++i) if (i == k) pend[i] = p;
    }
    __device__ __forceinline__ PendRay get(int k) const
    {
        PendRay p = pend[0];
#pragma unroll
        for (int i = 1; i < PEND; ++i) if (i == k) p = pend[i];
        return p;
    }
};
struct MemPark {
    uint32_t* base;         // this lane's column of the wave's slab: word f of slot k at base[(k * 8 + f) * 64]
    __device__ __forceinline__ void put(int k, const PendRay& p)
    {
        uint32_t* q = base + (size_t)k * (8 * 64);
        q[0 * 64] = __float_as_uint(p.ox); q[1 * 64] = __float_as_uint(p.oy); q[2 * 64] = __float_as_uint(p.oz);
        q[3 * 64] = __float_as_uint(p.dx); q[4 * 64] = __float_as_uint(p.dy); q[5 * 64] = __float_as_uint(p.dz);
        q[6 * 64] = __float_as_uint(p.w);  q[7 * 64] = p.meta;
    }
    __device__ __forceinline__ PendRay get(int k) const
    {
        const uint32_t* q = base + (size_t)k * (8 * 64);
        PendRay p;
        p.ox = __uint_as_float(q[0 * 64]); p.oy = __uint_as_float(q[1 * 64]); p.oz = __uint_as_float(q[2 * 64]);
        p.dx = __uint_as_float(q[3 * 64]); p.dy = __uint_as_float(q[4 * 64]); p.dz = __uint_as_float(q[5 * 64]);
        p.w = __uint_as_float(q[6 * 64]);  p.meta = q[7 * 64];
        return p;
    }
};

// a lane's current ray and what TraceRay is called with for it
struct RayState {
    f3 O, D;
    float w, tmin, tmax;
    uint32_t count;
    bool outside;
};

// RayGen (RayTracing.hlsl:42-60): payload {color 0, mask 1, outside true, count 0}, CULL_BACK, [1e-4, 100]
__device__ __forceinline__ RayState primary_ray(const DispatchDev& a, const CamDev& cb, uint32_t x, uint32_t y)
{
    RayState r;
    r.O = mk3(cb.cam[0], cb.cam[1], cb.cam[2]);
    r.D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
    r.w = 1.0f; r.count = 0; r.outside = true;
    r.tmin = a.tmin_p; r.tmax = a.tmax_p;
    return r;
}

// ClosestHit (hlsl:79-125) / Miss (hlsl:127-137) for the ray just traced: adds a leaf to acc, or spawns the refracted child
// (followed at once) and the reflected one (parked).  Returns false once the pixel's tree is exhausted; otherwise r is the
// next ray of the pixel in the recursion's depth-first order.
template <bool STATS, bool TLAS, class PK>
__device__ __forceinline__ bool shade_ray(const SceneDev& sc, const DispatchDev& a, const HitRec& h, RayState& r, f3& acc, int& np,
                                          PK& park, LaneStats& st)
{
    bool have_next = false;
    if (!h.hit) {                                             // Miss
        if (STATS) ++st.miss;
        f3 e = env_lookup(sc, r.D);
        acc.x = fmaf(r.w, e.x, acc.x); acc.y = fmaf(r.w, e.y, acc.y); acc.z = fmaf(r.w, e.z, acc.z);
    } else {                                                  // ClosestHit
        if (STATS) ++st.hits;
        if ((int)r.count < a.max_refract) {                   // hlsl:82
            f3 N = shading_normal<TLAS>(sc, h);
            f3 X = mk3(fmaf(h.t, r.D.x, r.O.x), fmaf(h.t, r.D.y, r.O.y), fmaf(h.t, r.D.z, r.O.z));   // hlsl:88
            f3 Nf = r.outside ? N : neg3(N);
            const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);   // hlsl:92
            float b = 1.0f - dot3(r.D, Nf);                   // hlsl:93, pow(b,5) = b*b*b*b*b
            float b2 = b * b, b4 = b2 * b2;
            float R = (R0 * (1.0f - R0)) * (b4 * b);
            float eta = r.outside ? a.inv_ior : a.ior;        // hlsl:95
            f3 d1;
            bool refr = refract_ray(d1, r.D, Nf, eta);
            if (STATS && !refr) ++st.tir;
            bool refl = (int)r.count < a.max_reflect;         // hlsl:110
            f3 d2 = mk3(0.0f, 0.0f, 0.0f);
            if (refl) d2 = normalize3(reflect_ray(r.D, Nf));  // hlsl:113
            const uint32_t c1 = r.count + 1u;
            r.tmin = a.tmin_s; r.tmax = a.tmax_s;
            r.O = X;
            if (refr) {
                if (refl) {                                   // park the reflected child
                    PendRay p;
                    p.ox = X.x; p.oy = X.y; p.oz = X.z; p.dx = d2.x; p.dy = d2.y; p.dz = d2.z;
                    p.w = r.w * R; p.meta = c1 | (r.outside ? 0x10000u : 0u);
                    park.put(np, p);
                    ++np;
                }
                r.D = d1; r.w = r.w * (1.0f - R); r.count = c1; r.outside = !r.outside;   // hlsl:103-107
                have_next = true;
            } else if (refl) {
                r.D = d2; r.w = r.w * R; r.count = c1;                                // hlsl:118-122
                have_next = true;
            }
        } else if (STATS) {
            ++st.term;                                        // payload.color stays 0
        }
    }
    if (!have_next) {
        if (np == 0) return false;
        --np;
        const PendRay p = park.get(np);
        r.O = mk3(p.ox, p.oy, p.oz); r.D = mk3(p.dx, p.dy, p.dz); r.w = p.w;
        r.count = p.meta & 0xffffu; r.outside = (p.meta & 0x10000u) != 0u;
        r.tmin = a.tmin_s; r.tmax = a.tmax_s;
    }
    return true;
}

// One pixel: RayGen, then the pixel's whole ray tree depth-first.  Returns the pixel's colour, the sum of its leaves in the
// recursion's order.  may_hit (wave-uniform): false for a block outside the screen rectangle of the scene
// (DispatchDev::hx0..hy1) -- its primary rays are Misses by construction and are not traced.
template <bool STATS, bool TLAS, bool DIAG, class E, class NS, class PK>
__device__ __forceinline__ f3 render_pixel(const SceneDev& sc, const DispatchDev& a, const CamDev& cb, uint32_t x, uint32_t y,
                                           bool may_hit, E* stk, const NS ns, PK& park, LaneStats& st, const Diag dg)
{
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    int np = 0;
    RayState r = primary_ray(a, cb, x, y);
    for (;;) {
        HitRec h;
        if (may_hit)
            trace_scene<STATS, TLAS, E, NS>(sc, r.O, r.D, r.tmin, r.tmax, r.outside ? CULL_BACK : CULL_FRONT, h, stk, st.cnt,
                                            DIAG ? dg : Diag{ nullptr }, ns);
        else h.hit = false;
        may_hit = true;
        ++st.rays;
        if (STATS && first_active_lane()) ++st.passes;
        if (DIAG) diag_trip(dg, 2);
        if (!shade_ray<STATS, TLAS>(sc, a, h, r, acc, np, park, st)) break;
    }
    return acc;
}

// RenderTarget[xy] = float4(color,1) -> R8G8B8A8_UNORM (hlsl:62); o: element index inside the slice
__device__ __forceinline__ void store_pixel(const DispatchDev& a, uint32_t* out_rgba8, float4* out_f32, size_t o, f3 acc)
{
    f3 c = acc;
    if (a.tonemap) {            // c / (1 + c) with c clamped to the largest finite float (so that +inf gives 1); NaN and c <= 0 end as 0 in unorm8
        c = mk3(fminf(acc.x, 3.4028234663852886e38f), fminf(acc.y, 3.4028234663852886e38f), fminf(acc.z, 3.4028234663852886e38f));
        c = mk3(acc.x > 0.0f ? c.x / (1.0f + c.x) : 0.0f, acc.y > 0.0f ? c.y / (1.0f + c.y) : 0.0f, acc.z > 0.0f ? c.z / (1.0f + c.z) : 0.0f);
    }
    const uint32_t packed = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | 0xff000000u;
    if (a.compact_out == 2u) {               // RGB8 tiles for the gather: alpha is always 255, not worth a link byte
        uint8_t* p3 = reinterpret_cast<uint8_t*>(out_rgba8) + o * 3;
        p3[0] = (uint8_t)packed; p3[1] = (uint8_t)(packed >> 8); p3[2] = (uint8_t)(packed >> 16);
    } else {
        out_rgba8[o] = packed;
    }
    if (out_f32) out_f32[o] = make_float4(acc.x, acc.y, acc.z, 1.0f);
}

// the wave's tallies -> dispatch counters: one sharded add per wave for the ray count, the rest only in STATS builds
template <bool STATS>
__device__ __forceinline__ void flush_stats(const DispatchDev& a, const LaneStats& st, uint32_t shard, uint32_t lane)
{
    uint32_t wr = wave_reduce_add(st.rays);
    if (lane == 0 && wr) atomicAdd(&a.ray_shards[shard & (RAY_SHARDS - 1)], wr);
    if (STATS) {
        uint32_t v;
        v = wave_reduce_add(st.hits);  if (lane == 0 && v) atomicAdd(&a.counters[C_HITS], (unsigned long long)v);
        v = wave_reduce_add(st.miss);  if (lane == 0 && v) atomicAdd(&a.counters[C_MISSES], (unsigned long long)v);
        v = wave_reduce_add(st.term);  if (lane == 0 && v) atomicAdd(&a.counters[C_TERMINAL], (unsigned long long)v);
        v = wave_reduce_add(st.tir);   if (lane == 0 && v) atomicAdd(&a.counters[C_TIR], (unsigned long long)v);
        v = wave_reduce_add(st.cnt.nodes); if (lane == 0 && v) atomicAdd(&a.counters[C_NODES], (unsigned long long)v);
        v = wave_reduce_add(st.cnt.tris);  if (lane == 0 && v) atomicAdd(&a.counters[C_TRIS], (unsigned long long)v);
        v = wave_reduce_add(st.pixels); if (lane == 0 && v) atomicAdd(&a.counters[C_PRIMARY], (unsigned long long)v);
        v = wave_reduce_add(st.cnt.node_trips); if (lane == 0 && v) atomicAdd(&a.counters[C_NODE_TRIPS], (unsigned long long)v);
        v = wave_reduce_add(st.cnt.leaf_trips); if (lane == 0 && v) atomicAdd(&a.counters[C_LEAF_TRIPS], (unsigned long long)v);
        v = wave_reduce_add(st.passes); if (lane == 0 && v) atomicAdd(&a.counters[C_PASSES], (unsigned long long)v);
        if (lane == 0 && st.blocks) atomicAdd(&a.counters[C_WAVES], (unsigned long long)st.blocks);
        if (lane == 0 && st.bg_blocks) atomicAdd(&a.counters[C_BG_WAVES], (unsigned long long)st.bg_blocks);
        if (lane == 0 && (st.blocks || st.rays)) {
            atomicAdd(&a.counters[C_CLK_TICKS], (unsigned long long)(__builtin_amdgcn_s_memtime() - st.clk0));
            atomicAdd(&a.counters[C_CLK_REAL], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - st.rt0));
        }
    }
}

// wave-block wb of the dispatch -> its slice (frame) and the 8x8 pixel block it covers.  Four consecutive wave-blocks are
// the 32x8 strip one 256-thread workgroup of k_render_fused renders; strips of consecutive frames follow each other (the
// depth slices are interleaved: mixing the slices keeps every CU on a blend of cheap background waves and expensive mesh
// waves -- monkey.obj 1080p, Depth 16: 193 us/frame interleaved, 238 us slice after slice).
// i-th tile of a launch in the order "rectangle first" (DispatchDev::rt_*): everything is wave-uniform scalar arithmetic
__device__ __forceinline__ uint32_t tile_in_launch_order(const DispatchDev& a, uint32_t i, uint32_t limit)
{
    if (a.rt_w == 0u || i >= limit) return i;
    const uint32_t n_rect = a.rt_w * a.rt_h;
    if (i < n_rect) {
        const uint32_t row = a.rt_w == 1u ? i : __umulhi(i, a.rt_div_w);
        return (a.rt_y0 + row) * a.tiles_x + a.rt_x0 + (i - row * a.rt_w);
    }
    uint32_t j = i - n_rect;
    const uint32_t above = a.rt_y0 * a.tiles_x;
    if (j < above) return j;
    j -= above;
    const uint32_t per_row = a.tiles_x - a.rt_w, beside = per_row * a.rt_h;
    if (j < beside) {
        const uint32_t row = per_row == 1u ? j : __umulhi(j, a.rt_div_o), c = j - row * per_row;
        return (a.rt_y0 + row) * a.tiles_x + (c < a.rt_x0 ? c : c + a.rt_w);
    }
    return (a.rt_y0 + a.rt_h) * a.tiles_x + (j - beside);
}

struct BlockPos { uint32_t frame, tile_local, x0, y0, px0, py0; bool tile_ok, bg; };
__device__ __forceinline__ BlockPos wave_block_pos(const DispatchDev& a, uint32_t wb)
{
    BlockPos p;
    const uint32_t blk = wb >> 2, wave = wb & 3u;
    p.frame = blk % a.n_frames;
    uint32_t strip;
    block_to_tile(blk / a.n_frames, p.tile_local, strip);
    p.tile_ok = p.tile_local < a.n_local_tiles;
    p.bg = false;
    uint32_t tile;
    if (a.mesh_part) {          // the rank's mesh tiles, then (rank 0) the background tiles: rr_mesh_partition
        const uint32_t k = p.tile_local;
        p.bg = k >= a.n_mesh_local;
        const uint32_t i = p.bg ? a.n_rect_tiles + (k - a.n_mesh_local) : mesh_deal_index(a.tile_rank, k, a.tile_world, a.mesh_rounds);
        tile = tile_in_launch_order(a, i < a.n_tiles ? i : 0u, a.n_tiles);
        p.tile_local = p.bg ? k - a.n_mesh_local : k;
    } else {
        p.tile_local = tile_in_launch_order(a, p.tile_local, a.n_local_tiles);
        tile = p.tile_local * a.tile_world + a.tile_rank;
    }
    const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    p.px0 = wave * 8u; p.py0 = strip * 8u;                         // inside the 32x32 tile
    p.x0 = tx * TILE + p.px0; p.y0 = ty * TILE + p.py0;
    return p;
}

struct LdsPark {
    uint32_t* base;         // this lane's column of the wave's slab in LDS: word f of slot k at base[(k * 8 + f) * 64]
    __device__ __forceinline__ void put(int k, const PendRay& p)
    {
        uint32_t* q = base + k * (8 * 64);
        q[0 * 64] = __float_as_uint(p.ox); q[1 * 64] = __float_as_uint(p.oy); q[2 * 64] = __float_as_uint(p.oz);
        q[3 * 64] = __float_as_uint(p.dx); q[4 * 64] = __float_as_uint(p.dy); q[5 * 64] = __float_as_uint(p.dz);
        q[6 * 64] = __float_as_uint(p.w);  q[7 * 64] = p.meta;
    }
    __device__ __forceinline__ PendRay get(int k) const
    {
        const uint32_t* q = base + k * (8 * 64);
        PendRay p;
        p.ox = __uint_as_float(q[0 * 64]); p.oy = __uint_as_float(q[1 * 64]); p.oz = __uint_as_float(q[2 * 64]);
        p.dx = __uint_as_float(q[3 * 64]); p.dy = __uint_as_float(q[4 * 64]); p.dz = __uint_as_float(q[5 * 64]);
        p.w = __uint_as_float(q[6 * 64]);  p.meta = q[7 * 64];
        return p;
    }
};


} // namespace rr
