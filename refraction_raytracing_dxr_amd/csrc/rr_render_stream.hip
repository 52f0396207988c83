// rr_render_stream.hip -- the DispatchRays stand-in for scenes whose rays diverge: a TLAS over many instances.
//
// k_render_fused walks a pixel's whole ray tree in one lane, 64 pixels in lock-step.  On the reference's scene (one BLAS) that
// keeps 55 % of the lanes busy; on a TLAS stress scene (BASELINE configs[4]: 1 024 instances) the 64 rays of a wave leave for
// different instances and a traversal pass lasts as long as its longest ray: 17 % of the lanes take part in a node step.
// Here the recursion of RayTracing.hlsl:79-125 is cut at every TraceRay instead: the rays of recursion depth g ("generation"
// g: payload.count == g) live in a queue in HBM, one kernel per generation traces and shades them, and the children go to the
// queue of generation g + 1.  Inside a generation a lane is bound to nothing: as soon as its ray is finished it takes the next
// one of its wave's share of the queue, so the wave has no idle lanes waiting for a long ray (lane utilisation of the node
// steps 17 % -> ~50 %), and nothing of a pixel's state has to stay in registers while a ray is traced -- what made the
// lane-asynchronous experiments of round 2 (every lane a state machine over its pixel's tree) lose what they gained.
//
//   generation 0   k_stream_primary: RayGen (hlsl:42-60) and the primary rays of the 8x8 pixel blocks of the tiles that touch the
//                  scene's screen rectangle, in lock-step (primary rays stay together); every pixel gets a mark: its colour is
//                  the texel in slot 0 (Miss), black (no leaf below it), or the sum of its four leaf slots, zeroed here (it has
//                  children).  k_stream_background: the other blocks, one Miss per pixel without TraceRay (as
//                  k_render_fused's background waves).
//   generation g   k_stream_rays: ClosestHit (hlsl:79-125) pushes the refracted and the reflected child; Miss (hlsl:127-137)
//                  writes (weight, texel) to the ray's leaf slot.  The tree only branches while count < max_reflect <= 2, so
//                  a pixel has at most four root-to-leaf paths; slot = path bits (reflect at count 0: 2, at count 1: 1), which
//                  is the order the recursion reaches the leaves in.
//   resolve        stores every pixel of those blocks; acc = fma(w, texel, acc) over the four slots in order: the very fma sequence of k_render_fused and of the
//                  oracle's path-weight mode, whatever order the queues were filled in -- frames and counters are bit-identical.
//
// Queues: 48 B per ray (origin, weight | direction, pixel ordinal | count, inside/outside, slot).  A wave reserves SQ_BLK
// entries at a time from the generation's head counter (one returning atomic per 1 024 rays, not per push: a single word takes
// ~88 of those per microsecond, MI355X_MICROARCH.md "dequeue") and records how many rays each 64-entry chunk of its blocks holds;
// the next generation's waves take chunks w, w + W, w + 2W, ... of the reserved range and hand their rays out to lanes as those
// fall idle.  No atomics on the consuming side, none on the image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "rr_render_common.h"

#ifndef RR_STREAM_WPS
#define RR_STREAM_WPS 6
#endif

namespace rr {

constexpr uint32_t SQ_BLK = 1024;               // queue entries a wave reserves at a time (16 chunks of 64)
constexpr uint32_t SQ_NONE = 0xffffffffu;       // no block reserved yet
constexpr uint32_t SQ_DEAD = 0xfffffffeu;       // the queue is full: the error flag is set, pushes are dropped
constexpr int TRAV_FIN = (int)0x80000001;       // the lane's ray is finished: shade it
constexpr int TRAV_IDLE = (int)0x80000002;      // the lane has no ray
// StreamDev::pending: what generation 0 found for a pixel (every pixel of the ray kernels' blocks is stored by the resolve kernel)
constexpr uint8_t PIX_BLACK = 0, PIX_ONE_LEAF = 1, PIX_LEAVES = 2;

// ---- producing side (wave-uniform state: base / used of the block being filled) -----------------------------------------
__device__ __forceinline__ void sq_finalize(const StreamDev& s, uint32_t qsel, uint32_t base, uint32_t used, uint32_t lane)
{
    if (lane < SQ_BLK / 64u) {
        const uint32_t lo = lane * 64u;
        s.fill[qsel][(base >> 6) + lane] = used > lo ? (used - lo < 64u ? used - lo : 64u) : 0u;
    }
}

// append the rays of the lanes with `have` set to the queue of generation gen_out.  Called by all lanes of the wave together.
__device__ __forceinline__ void sq_push(const StreamDev& s, uint32_t gen_out, uint32_t& base, uint32_t& used, bool have, f3 O, f3 D,
                                        float w, uint32_t cov, uint32_t meta, uint32_t lane, uint32_t* error_flag)
{
    const unsigned long long m = __ballot(have);
    if (m == 0ull) return;
    const uint32_t k = (uint32_t)__popcll(m);
    const uint32_t qsel = gen_out & 1u;
    if (base == SQ_NONE || (base != SQ_DEAD && used + k > SQ_BLK)) {
        if (base != SQ_NONE) sq_finalize(s, qsel, base, used, lane);
        const int first = __ffsll((long long)__ballot(1)) - 1;
        uint32_t b = 0;
        if ((int)lane == first) b = atomicAdd(&s.heads[gen_out], SQ_BLK);
        b = (uint32_t)__builtin_amdgcn_readlane((int)b, first);
        if (b > s.cap - SQ_BLK) { if ((int)lane == first) atomicOr(error_flag, 1u); b = SQ_DEAD; }
        base = b; used = 0u;
    }
    if (base != SQ_DEAD && have) {
        const uint32_t idx = base + used + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        float4* q = s.q[qsel] + (size_t)idx * 3;
        q[0] = make_float4(O.x, O.y, O.z, w);
        q[1] = make_float4(D.x, D.y, D.z, __uint_as_float(cov));
        q[2] = make_float4(__uint_as_float(meta), 0.0f, 0.0f, 0.0f);
    }
    used += k;
}

// ClosestHit (RayTracing.hlsl:79-125) for one ray with count < max_refract: the refracted child c1 (followed first by the
// recursion) and the reflected child c2.  slot: the leaf slot the ray's own subtree starts at; the reflected branch at count 0 /
// 1 owns slots 2,3 / 1,3.
struct StreamChild { f3 D; float w; uint32_t meta; };
__device__ __forceinline__ void stream_shade_hit(const SceneDev& sc, const DispatchDev& a, f3 O, f3 D, float w, uint32_t count, bool outside,
                                                 uint32_t slot, const HitRec& h, f3& X, bool& refr, bool& refl, StreamChild& c1, StreamChild& c2)
{
    const f3 N = shading_normal<true>(sc, h);                                                  // hlsl:83-86
    X = mk3(fmaf(h.t, D.x, O.x), fmaf(h.t, D.y, O.y), fmaf(h.t, D.z, O.z));                    // hlsl:88
    const f3 Nf = outside ? N : neg3(N);
    const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);                                            // hlsl:92
    const float b = 1.0f - dot3(D, Nf);                                                        // hlsl:93
    const float b2 = b * b, b4 = b2 * b2;
    const float R = (R0 * (1.0f - R0)) * (b4 * b);
    const float eta = outside ? a.inv_ior : a.ior;                                             // hlsl:95
    f3 d1;
    refr = refract_ray(d1, D, Nf, eta);
    refl = (int)count < a.max_reflect;                                                         // hlsl:110
    f3 d2 = mk3(0.0f, 0.0f, 0.0f);
    if (refl) d2 = normalize3(reflect_ray(D, Nf));                                             // hlsl:113
    const uint32_t bit = count < 2u ? (2u >> count) : 0u;
    const uint32_t c1n = count + 1u;
    // meta: count | 0x10000 if the ray runs INSIDE the mesh | slot << 20.  The refracted ray changes sides (hlsl:103-107)
    c1.D = d1; c1.w = w * (1.0f - R); c1.meta = c1n | (outside ? 0x10000u : 0u) | (slot << 20);     // 0x10000 = META_INSIDE
    c2.D = d2; c2.w = w * R;          c2.meta = c1n | (outside ? 0u : 0x10000u) | ((slot | bit) << 20);
}

// ---------------------------------------------------------------------------------------------------
// Generation 0: RayGen (RayTracing.hlsl:42-60) and the primary rays of the blocks of the tiles that touch the rectangle, traced in
// LOCK-STEP, one 8x8 pixel block per wave trip -- the 64 primary rays of a block are the one kind of ray that stays together
// (lane utilisation of the node steps 60-80 %), so the refill machinery would only add its queue round trip to them.  Persistent
// waves draw blocks from the generation's ticket counters; the children go to queue 1 through the wave's block reservation.
// Every pixel gets its mark for the resolve kernel: its colour is the texel in slot 0 (Miss), black (no leaf below it), or the sum
// of its four leaf slots, zeroed here (it has children).  Pixel ordinal = wave-block * 64 + Morton position.
constexpr uint32_t META_INSIDE = 0x10000u;

template <int STACK, bool STATS, class E, int WPS>
__global__ __launch_bounds__(256, WPS) void k_stream_primary(SceneDev sc, DispatchDev a, StreamDev s)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    E* const stk = reinterpret_cast<E*>(lds) + wave * (STACK * 64) + lane;
    uint32_t* const tickets = s.next;                                   // generation 0's counters
    LaneStats st;
    stats_clock_begin<STATS>(st);
    uint32_t shard;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(shard));
    shard &= 7u;
    uint32_t shards_left = 8u;
    uint32_t ob = SQ_NONE, ou = 0u;
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
    for (;;) {
        uint32_t wb = 0xffffffffu;
        while (shards_left != 0u) {
            uint32_t t = 0;
            if (lane == 0u) t = atomicAdd(&tickets[shard * 16u], 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            const uint32_t c = t * 8u + shard;
            if (c < s.n_rect_wb) { wb = c; break; }
            shard = (shard + 1u) & 7u; --shards_left;
        }
        if (wb == 0xffffffffu) break;
        const BlockPos bp = wave_block_pos(a, wb);
        if (!bp.tile_ok) continue;
        const uint32_t x = bp.x0 + lx, y = bp.y0 + ly;
        const bool valid = x < a.W && y < a.H;
        const bool may_hit = bp.x0 + 8u > a.hx0 && bp.x0 < a.hx1 && bp.y0 + 8u > a.hy0 && bp.y0 < a.hy1;
        st.blocks += 1u;
        if (STATS && !may_hit) st.bg_blocks += 1u;
        const uint32_t cov = wb * 64u + lane;
        bool refr = false, refl = false;
        f3 X = mk3(0.0f, 0.0f, 0.0f);
        StreamChild c1, c2;
        c1.D = c2.D = X; c1.w = c2.w = 0.0f; c1.meta = c2.meta = 0u;
        if (valid) {
            const CamDev& cb = a.cams[bp.frame];
            const f3 O = mk3(cb.cam[0], cb.cam[1], cb.cam[2]);
            const f3 D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
            st.pixels += 1; ++st.rays;
            if (STATS && first_active_lane()) ++st.passes;
            HitRec h;
            h.hit = false;
            if (may_hit) trace_scene<STATS, true, E, GlobalNodes>(sc, O, D, a.tmin_p, a.tmax_p, CULL_BACK, h, stk, st.cnt);
            if (!h.hit) {                                               // the pixel's only leaf: payload.color = 0 + 1 * texel
                if (STATS) ++st.miss;
                const f3 e = env_lookup(sc, D);
                s.slots[(size_t)cov * 4] = make_float4(1.0f, e.x, e.y, e.z);
                s.pending[cov] = PIX_ONE_LEAF;
            } else {
                if (STATS) ++st.hits;
                if (0 < a.max_refract) {                                // hlsl:82 at count 0
                    stream_shade_hit(sc, a, O, D, 1.0f, 0u, true, 0u, h, X, refr, refl, c1, c2);
                    if (STATS && !refr) ++st.tir;
                } else if (STATS) ++st.term;                            // payload.color stays 0
                if (refr || refl) {
                    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    float4* sl = s.slots + (size_t)cov * 4;
                    sl[0] = z; sl[1] = z; sl[2] = z; sl[3] = z;
                    s.pending[cov] = PIX_LEAVES;
                } else s.pending[cov] = PIX_BLACK;                      // no leaf below this pixel
            }
        }
        sq_push(s, 1u, ob, ou, refr, X, c1.D, c1.w, cov, c1.meta, lane, a.error_flag);
        sq_push(s, 1u, ob, ou, refl, X, c2.D, c2.w, cov, c2.meta, lane, a.error_flag);
    }
    if (ob != SQ_NONE && ob != SQ_DEAD) sq_finalize(s, 1u, ob, ou, lane);
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// ---------------------------------------------------------------------------------------------------
// One ray kernel (generations 1 and up).  MODE 1: a generation whose rays still branch (count < max_reflect): both children go
// to the next queue.  MODE 2: every later generation at once --
// from count == max_reflect on a ray has at most ONE child (the refracted one), so the lane follows the chain itself: the child
// overwrites the parent's queue entry (the lane owns it) and is traced next, nothing is queued, no generation has to end before
// the next begins.
// Work is handed out dynamically, 64 rays (one chunk) at a time: chunk numbers come from eight ticket counters, one per XCD
// (chunk = ticket * 8 + counter), a wave starts on its XCD's counter and moves on to the others once that is exhausted.
//
// The wave's loop: every lane is in one of six states -- at an internal node, at a triangle leaf, at an instance leaf, at the
// end of an instance's subtree, finished (to be shaded), without a ray -- and a step of one kind is only issued when enough
// lanes wait for it (a divergent `if` over all kinds would issue every body for a handful of lanes each: the lock-step kernel's
// leaf part costs it a third of its time on the 1 024-instance scene).  The world-space slab constants of a ray are
// parked in LDS when it first enters an instance and fetched back when it leaves one.
constexpr uint32_t ST_DONE = 0x80000000u, ST_FIN = 0x80000001u, ST_IDLE = 0x80000002u;     // == TRAV_DONE / TRAV_FIN / TRAV_IDLE as unsigned
constexpr uint32_t F_ENTERED = 1u;              // in `cull`: the ray has been inside an instance (Oc / Dc are no longer the world-space ray)
constexpr int BR_SAVE_WORDS = 10;               // kn, kf, inv (9 floats) + the three direction signs

template <int STACK, bool STATS, class E, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_stream_rays(SceneDev sc, DispatchDev a, StreamDev s, uint32_t gen, float tmin, float tmax)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    constexpr uint32_t WAVE_WORDS = (STACK * 64 * sizeof(E)) / 4 + BR_SAVE_WORDS * 64;
    E* const stk = reinterpret_cast<E*>(lds + wave * WAVE_WORDS) + lane;
    uint32_t* const brs = lds + wave * WAVE_WORDS + (STACK * 64 * sizeof(E)) / 4 + lane;       // word k of the lane's parked slab constants at brs[k * 64]
    constexpr uint32_t NO_INST = 0xffffffffu, NO_CHUNK = 0xffffffffu;
    const QNode* __restrict__ nodes = sc.pool_nodes;
    float4* const qin = s.q[gen & 1u];
    const uint32_t* __restrict__ fin = s.fill[gen & 1u];
    uint32_t n_chunks;
    { uint32_t h = s.heads[gen]; h = h < s.cap ? h : s.cap; n_chunks = h >> 6; }
    uint32_t* const tickets = s.next + (size_t)gen * (8u * 16u);
    const uint32_t n_tris = sc.n_pool_tris;

    LaneStats st;
    stats_clock_begin<STATS>(st);
    uint32_t shard;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(shard));
    shard &= 7u;
    uint32_t shards_left = 8u;                      // counters not yet seen exhausted
    uint32_t chunk = NO_CHUNK, off = 0u, nv = 0u;   // the chunk being handed out: `off` of its nv rays are taken
    uint32_t ob = SQ_NONE, ou = 0u;                 // the block of the next generation's queue this wave is filling

    // the lane's ray: only what the traversal needs.  rid: its entry in the queue (reloaded when the ray is shaded or enters a second instance)
    int node = TRAV_IDLE;
    uint32_t rid = 0u;
    f3 Oc = mk3(0.0f, 0.0f, 0.0f), Dc = mk3(0.0f, 0.0f, 1.0f);
    BoxRay br = box_ray(Oc, Dc, sc.scale, sc.grid);
    E* top = stk;
    const E* floor = stk;
    uint32_t cur = NO_INST, cull = 0u, wcull = 0u;  // cull: the cull flags in force (| F_ENTERED); wcull: the world-space ray's
    HitRec best;
    best.t = 0.0f; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = best.V = 0.0f; best.ad = 1.0f;

    auto start_ray = [&](f3 O, f3 D, bool inside, bool traced) {
        best.t = tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f; best.ad = 1.0f;
        wcull = inside ? CULL_FRONT : CULL_BACK;
        cull = wcull; cur = NO_INST; Oc = O; Dc = D; top = stk; floor = stk;
        br = box_ray(O, D, sc.scale, sc.grid);
        node = traced ? 0 : TRAV_FIN;
    };
    // a subtree is exhausted (node == TRAV_DONE): at the top level, or with nothing left on the top level's stack, the ray is finished
    auto settle = [&]() { if ((uint32_t)node == ST_DONE && (cur == NO_INST || floor == stk)) node = TRAV_FIN; };

    // One trip of the wave's loop: internal-node steps while enough lanes are descending, then one step of each other kind that
    // enough lanes wait for (triangle tests, instances entered, instances left, shading + refill).  `serve`: lanes a step needs
    // to be worth issuing; when no kind has that many, the one most lanes wait for runs alone.
    for (;;) {
        const bool more = shards_left != 0u || (chunk != NO_CHUNK && off < nv);
        int pick = -1;                                  // nothing is worth issuing: the step most lanes wait for runs alone
        int serve, need_lanes, node_need;
        {
            const uint32_t un = (uint32_t)node;
            const bool at_leaf = un > ST_IDLE;
            const int n_node = __popcll(__ballot(node >= 0)), n_tri = __popcll(__ballot(at_leaf && ~un < n_tris)), n_ent = __popcll(__ballot(at_leaf && ~un >= n_tris)),
                      n_exit = __popcll(__ballot(un == ST_DONE)), n_fin = __popcll(__ballot(un == ST_FIN)), n_idle = __popcll(__ballot(un == ST_IDLE));
            const int n_shade = n_fin + (more ? n_idle : 0);
            if (n_node + n_tri + n_ent + n_exit + n_shade == 0) break;
            const int n_alive = 64 - (more ? 0 : n_idle);
            serve = (n_alive * (int)(a.async_leaf_num & 0xffu) + 15) / 16;
            need_lanes = (n_alive * (int)a.async_shade_num + 15) / 16;
            node_need = (a.async_leaf_num >> 8) ? serve * (int)(a.async_leaf_num >> 8) / 2 : serve * 2;     // (experiments: the node loop's share in halves of `serve`)
            if (n_node < node_need && n_tri < serve && n_ent < serve && n_exit < serve && (n_shade < need_lanes || n_shade == 0)) {
                pick = 0; int mx = n_node;
                if (n_tri > mx) { mx = n_tri; pick = 1; }
                if (n_ent > mx) { mx = n_ent; pick = 2; }
                if (n_exit > mx) { mx = n_exit; pick = 3; }
                if (n_shade > mx) { mx = n_shade; pick = 4; }
            }
        }
        // ---- internal-node steps
        {
            const int n_node = __popcll(__ballot(node >= 0));
            if (n_node > 0 && (n_node >= node_need || pick == 0)) {
                if (node >= 0) {
                    const int stop = pick == 0 ? (n_node + 1) / 2 : node_need - 1;         // go on while at least 2 * serve lanes descend
                    do {
                        const NodeQ q = load_node(nodes, node);
                        if (STATS) { st.cnt.nodes++; if (first_active_lane()) st.cnt.node_trips++; }
                        node = node_step(br, q, tmin, best.t, top, floor);
                    } while (node >= 0 && __popcll(__ballot(1)) > stop);
                    settle();
                }
            }
        }
        // ---- triangle tests
        {
            const uint32_t un = (uint32_t)node;
            const bool mine = un > ST_IDLE && ~un < n_tris;
            const int n = __popcll(__ballot(mine));
            if (n > 0 && (n >= serve || pick == 1)) {
                if (mine) {
                    if (STATS) { st.cnt.tris++; if (first_active_lane()) st.cnt.leaf_trips++; }
                    tri_test(sc.pool_tris, ~un, Oc, Dc, tmin, cull & (CULL_BACK | CULL_FRONT), cur, best);
                    if (top > floor) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else { node = TRAV_DONE; settle(); }
                }
            }
        }
        // ---- instances entered: the lane's ray goes to the instance's object space (t is preserved: the direction is not renormalised)
        {
            const uint32_t un = (uint32_t)node;
            const bool mine = un > ST_IDLE && ~un >= n_tris;
            const int n = __popcll(__ballot(mine));
            if (n > 0 && (n >= serve || pick == 2)) {
                if (mine) {
                    if (STATS && first_active_lane()) st.cnt.leaf_trips++;
                    const uint32_t ii = ~un - n_tris;
                    const InstDev& in = sc.insts[ii];
                    if (in.mask & 0xffu) {                                      // InstanceInclusionMask 0xff
                        f3 O = Oc, D = Dc;
                        if (cull & F_ENTERED) {                                 // Oc / Dc are a former instance's: the world-space ray again
                            const float4 q0 = qin[(size_t)rid * 3], q1 = qin[(size_t)rid * 3 + 1];
                            O = mk3(q0.x, q0.y, q0.z); D = mk3(q1.x, q1.y, q1.z);
                        } else {                                                // first instance: park the world-space slab constants
                            brs[0 * 64] = __float_as_uint(br.inv.x); brs[1 * 64] = __float_as_uint(br.inv.y); brs[2 * 64] = __float_as_uint(br.inv.z);
                            brs[3 * 64] = __float_as_uint(br.kn.x);  brs[4 * 64] = __float_as_uint(br.kn.y);  brs[5 * 64] = __float_as_uint(br.kn.z);
                            brs[6 * 64] = __float_as_uint(br.kf.x);  brs[7 * 64] = __float_as_uint(br.kf.y);  brs[8 * 64] = __float_as_uint(br.kf.z);
                            brs[9 * 64] = (br.sx ? 1u : 0u) | (br.sy ? 2u : 0u) | (br.sz ? 4u : 0u);
                        }
                        uint32_t f = wcull;
                        if (in.flags & 0x1u) f &= ~(CULL_BACK | CULL_FRONT);                  // TRIANGLE_CULL_DISABLE
                        else if (in.flags & 0x2u) {                                            // TRIANGLE_FRONT_COUNTERCLOCKWISE
                            if (f & CULL_BACK) f = (f & ~CULL_BACK) | CULL_FRONT;
                            else if (f & CULL_FRONT) f = (f & ~CULL_FRONT) | CULL_BACK;
                        }
                        cull = f | F_ENTERED; cur = ii; floor = top;
                        if (!in.identity) { Oc = xform_point(in.inv, O); Dc = xform_dir(in.inv, D); } else { Oc = O; Dc = D; }
                        br = box_ray(Oc, Dc, in.scale, in.grid);
                        node = (int)in.root;
                    } else if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_FIN;
                }
            }
        }
        // ---- instances left (with top-level entries still to visit): the parked world-space slab constants come back
        {
            const bool mine = (uint32_t)node == ST_DONE;
            const int n = __popcll(__ballot(mine));
            if (n > 0 && (n >= serve || pick == 3)) {
                if (mine) {
                    if (STATS && first_active_lane()) st.cnt.leaf_trips++;
                    br.inv = mk3(__uint_as_float(brs[0 * 64]), __uint_as_float(brs[1 * 64]), __uint_as_float(brs[2 * 64]));
                    br.kn = mk3(__uint_as_float(brs[3 * 64]), __uint_as_float(brs[4 * 64]), __uint_as_float(brs[5 * 64]));
                    br.kf = mk3(__uint_as_float(brs[6 * 64]), __uint_as_float(brs[7 * 64]), __uint_as_float(brs[8 * 64]));
                    const uint32_t sg = brs[9 * 64];
                    br.sx = (sg & 1u) != 0u; br.sy = (sg & 2u) != 0u; br.sz = (sg & 4u) != 0u;
                    cur = NO_INST; cull = wcull | F_ENTERED; floor = stk;
                    top -= STACK_STRIDE; node = StackCodec<E>::dec(*top);       // (settle() left this state only to lanes whose top-level stack is not empty)
                }
            }
        }
        // ---- shading pass: every finished lane shades its ray; then the lanes without a ray take new ones
        {
            const int n_fin = __popcll(__ballot((uint32_t)node == ST_FIN));
            const int n_idle = more ? __popcll(__ballot((uint32_t)node == ST_IDLE)) : 0;
            const int n_shade = n_fin + n_idle;
            if (n_shade > 0 && (n_shade >= need_lanes || pick == 4)) {
                bool refr = false, refl = false;
                f3 X = mk3(0.0f, 0.0f, 0.0f);
                StreamChild c1, c2;
                c1.D = c2.D = X; c1.w = c2.w = 0.0f; c1.meta = c2.meta = 0u;
                uint32_t cov = 0u;
                if ((uint32_t)node == ST_FIN) {
                    const float4 q0 = qin[(size_t)rid * 3], q1 = qin[(size_t)rid * 3 + 1];
                    const uint32_t m = __float_as_uint(qin[(size_t)rid * 3 + 2].x);
                    const f3 O = mk3(q0.x, q0.y, q0.z), D = mk3(q1.x, q1.y, q1.z);
                    const float w = q0.w;
                    cov = __float_as_uint(q1.w);
                    const uint32_t count = m & 0xffffu, slot = (m >> 20) & 3u;
                    const bool outside = (m & META_INSIDE) == 0u;
                    ++st.rays;
                    if (STATS && first_active_lane()) ++st.passes;
                    node = TRAV_IDLE;
                    if (!best.hit) {                                            // Miss (hlsl:127-137)
                        if (STATS) ++st.miss;
                        const f3 e = env_lookup(sc, D);
                        s.slots[(size_t)cov * 4 + slot] = make_float4(w, e.x, e.y, e.z);
                    } else {
                        if (STATS) ++st.hits;
                        const InstDev& in = sc.insts[best.inst];                // the ray in the space of the instance that was hit
                        f3 Oh = O, Dh = D;
                        if (!in.identity) { Oh = xform_point(in.inv, O); Dh = xform_dir(in.inv, D); }
                        hit_attributes(sc.pool_tris, Oh, Dh, best);
                        if ((int)count < a.max_refract) {                       // hlsl:82
                            stream_shade_hit(sc, a, O, D, w, count, outside, slot, best, X, refr, refl, c1, c2);
                            if (STATS && !refr) ++st.tir;
                        } else if (STATS) ++st.term;                            // payload.color stays 0
                        if (MODE == 2 && refr) {                                // the chain goes on in this lane, in this queue entry
                            qin[(size_t)rid * 3] = make_float4(X.x, X.y, X.z, c1.w);
                            qin[(size_t)rid * 3 + 1] = make_float4(c1.D.x, c1.D.y, c1.D.z, __uint_as_float(cov));
                            qin[(size_t)rid * 3 + 2] = make_float4(__uint_as_float(c1.meta), 0.0f, 0.0f, 0.0f);
                            start_ray(X, c1.D, (c1.meta & META_INSIDE) != 0u, true);
                        }
                    }
                }
                if (MODE != 2) {
                    sq_push(s, gen + 1u, ob, ou, refr, X, c1.D, c1.w, cov, c1.meta, lane, a.error_flag);
                    sq_push(s, gen + 1u, ob, ou, refl, X, c2.D, c2.w, cov, c2.meta, lane, a.error_flag);
                }
                // ---- refill
                unsigned long long need = __ballot(node == TRAV_IDLE);
                while (need != 0ull) {
                    if (chunk == NO_CHUNK || off >= nv) {                       // next chunk: a ticket from the wave's counter, then the others'
                        chunk = NO_CHUNK;
                        while (shards_left != 0u) {
                            const int first = __ffsll((long long)__ballot(1)) - 1;
                            uint32_t t = 0;
                            if ((int)lane == first) t = atomicAdd(&tickets[shard * 16u], 1u);
                            t = (uint32_t)__builtin_amdgcn_readlane((int)t, first);
                            const uint32_t c = t * 8u + shard;
                            if (c < n_chunks) { chunk = c; break; }
                            shard = (shard + 1u) & 7u; --shards_left;
                        }
                        if (chunk == NO_CHUNK) break;
                        nv = fin[chunk]; off = 0u;
                        if (nv == 0u) continue;
                    }
                    const uint32_t left = nv - off;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    if (node == TRAV_IDLE && rank < left) {
                        const uint32_t r = chunk * 64u + off + rank;
                        const uint32_t m = __float_as_uint(qin[(size_t)r * 3 + 2].x);
                        const float4 q0 = qin[(size_t)r * 3], q1 = qin[(size_t)r * 3 + 1];
                        rid = r;
                        start_ray(mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), (m & META_INSIDE) != 0u, true);
                    }
                    const uint32_t asked = (uint32_t)__popcll(need);
                    off += asked < left ? asked : left;
                    need = __ballot(node == TRAV_IDLE);
                }
            }
        }
    }
    if (MODE != 2 && ob != SQ_NONE && ob != SQ_DEAD) sq_finalize(s, (gen + 1u) & 1u, ob, ou, lane);
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// the blocks of the tiles that do not touch the scene's screen rectangle: RayGen + one Miss per pixel, no TraceRay
template <bool STATS>
__global__ __launch_bounds__(256) void k_stream_background(SceneDev sc, DispatchDev a, StreamDev s)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t wb = s.n_rect_wb + blockIdx.x * 4u + wave;
    LaneStats st;
    stats_clock_begin<STATS>(st);
    if (wb < a.n_blocks * 4u) {
        const BlockPos bp = wave_block_pos(a, wb);
        const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
        const uint32_t x = bp.x0 + lx, y = bp.y0 + ly;
        st.blocks = bp.tile_ok ? 1u : 0u;
        if (STATS) st.bg_blocks = st.blocks;
        if (bp.tile_ok && x < a.W && y < a.H) {
            const CamDev& cb = a.cams[bp.frame];
            const f3 D = camera_ray_dir(cb.M, a.sx[x], a.sy[y]);
            st.pixels = 1; st.rays = 1;
            if (STATS) { st.miss = 1; if (first_active_lane()) st.passes = 1; }
            const f3 e = env_lookup(sc, D);
            const f3 acc = mk3(fmaf(1.0f, e.x, 0.0f), fmaf(1.0f, e.y, 0.0f), fmaf(1.0f, e.z, 0.0f));
            const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x : (size_t)bp.tile_local * (TILE * TILE) + (bp.py0 + ly) * TILE + (bp.px0 + lx);
            uint32_t* const out = bp.bg ? a.out_bg + (size_t)bp.frame * a.bg_stride : a.out_rgba8 + (size_t)bp.frame * a.frame_stride;
            store_pixel(a, out, a.out_f32 ? a.out_f32 + (size_t)bp.frame * a.frame_stride : nullptr, o, acc);
        }
    }
    flush_stats<STATS>(a, st, blockIdx.x * 4u + wave, lane);
}

// every pixel of the ray kernels' blocks: black, its one texel, or the sum of its four leaf slots in the recursion's order
__global__ __launch_bounds__(256) void k_stream_resolve(DispatchDev a, StreamDev s)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t wb = blockIdx.x * 4u + wave;
    if (wb >= s.n_rect_wb) return;
    const BlockPos bp = wave_block_pos(a, wb);
    const uint32_t lx = compact1by1(lane), ly = compact1by1(lane >> 1);
    const uint32_t x = bp.x0 + lx, y = bp.y0 + ly;
    if (!bp.tile_ok || x >= a.W || y >= a.H) return;
    const uint32_t cov = wb * 64u + lane;
    const uint8_t kind = s.pending[cov];
    const float4* sl = s.slots + (size_t)cov * 4;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    if (kind == PIX_ONE_LEAF) {
        const float4 v = sl[0];
        acc = mk3(fmaf(v.x, v.y, 0.0f), fmaf(v.x, v.z, 0.0f), fmaf(v.x, v.w, 0.0f));
    } else if (kind == PIX_LEAVES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = sl[k];                              // an unused slot holds w = 0: fma(0, 0, acc) == acc
            acc.x = fmaf(v.x, v.y, acc.x); acc.y = fmaf(v.x, v.z, acc.y); acc.z = fmaf(v.x, v.w, acc.z);
        }
    }
    const size_t o = a.compact_out == 0u ? (size_t)y * a.W + x : (size_t)bp.tile_local * (TILE * TILE) + (bp.py0 + ly) * TILE + (bp.px0 + lx);
    uint32_t* const out = bp.bg ? a.out_bg + (size_t)bp.frame * a.bg_stride : a.out_rgba8 + (size_t)bp.frame * a.frame_stride;     // (a group of eight tiles may end in background tiles)
    store_pixel(a, out, a.out_f32 ? a.out_f32 + (size_t)bp.frame * a.frame_stride : nullptr, o, acc);
}

// ------------------------------------------------------------------------------------ launcher
static thread_local char g_stream_name[96] = "";
const char* last_stream_kernel_name() { return g_stream_name; }

template <int STACK, bool STATS, int WPS>
static hipError_t launch_stream_sw(const SceneDev& sc, const DispatchDev& a, const StreamDev& s, uint32_t n_wg, hipStream_t st)
{
    typedef uint16_t E;
    const size_t lds = (size_t)4 * (STACK * 64 * sizeof(E) + BR_SAVE_WORDS * 64 * 4);
    const uint32_t total_wb = a.n_blocks * 4u;
    hipError_t e = hipMemsetAsync(s.heads, 0, (STREAM_MAX_GEN + STREAM_MAX_GEN * 8u * 16u) * sizeof(uint32_t), st);     // heads, then the ticket counters
    if (e != hipSuccess) return e;
    if (s.n_rect_wb < total_wb)
        hipLaunchKernelGGL((k_stream_background<STATS>), dim3((total_wb - s.n_rect_wb + 3u) / 4u), dim3(256), 0, st, sc, a, s);
    if (s.n_rect_wb == 0u) {        // (a rank without mesh tiles: the background kernel was the whole launch)
        snprintf(g_stream_name, sizeof g_stream_name, "k_stream_background");
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_stream_primary<STACK, STATS, E, WPS>), dim3(n_wg), dim3(256), (size_t)4 * STACK * 64 * sizeof(E), st, sc, a, s);
    // generations that still branch push to the next queue; from count == max_reflect on one kernel follows every chain to its end
    const int chain_gen = a.max_reflect < 1 ? 1 : a.max_reflect;
    int launches = 0;
    for (int g = 1; g <= a.max_refract && g < chain_gen; ++g, ++launches)
        hipLaunchKernelGGL((k_stream_rays<STACK, STATS, E, 1, WPS>), dim3(n_wg), dim3(256), lds, st, sc, a, s, (uint32_t)g, a.tmin_s, a.tmax_s);
    if (chain_gen <= a.max_refract) {
        hipLaunchKernelGGL((k_stream_rays<STACK, STATS, E, 2, WPS>), dim3(n_wg), dim3(256), lds, st, sc, a, s, (uint32_t)chain_gen, a.tmin_s, a.tmax_s);
        ++launches;
    }
    hipLaunchKernelGGL(k_stream_resolve, dim3((s.n_rect_wb + 3u) / 4u), dim3(256), 0, st, a, s);
    snprintf(g_stream_name, sizeof g_stream_name, "k_stream_primary + k_stream_rays<%d, %s, unsigned short, 1|2, %d> x %d", STACK, STATS ? "true" : "false", WPS, launches);
    return hipGetLastError();
}

// two-level scenes whose stack entries fit 16 bits, trees of at most 39 levels, max_reflect <= 2, max_refract < STREAM_MAX_GEN - 1
hipError_t launch_render_stream(const SceneDev& sc, const DispatchDev& a, const StreamDev& s, int stack, uint32_t n_wg, bool stats, hipStream_t st, int waves)
{
    if (a.n_blocks == 0) return hipSuccess;
    // (built for six waves per SIMD: the eight-wave build spilled into every loop header, those for seven and five measured up to
    // 8 % slower on C5 when the kernel was tuned; -DRR_STREAM_WPS=<n> rebuilds it for another number)
    (void)waves;
    if (stack <= 30) return stats ? launch_stream_sw<30, true, RR_STREAM_WPS>(sc, a, s, n_wg, st) : launch_stream_sw<30, false, RR_STREAM_WPS>(sc, a, s, n_wg, st);
    return stats ? launch_stream_sw<39, true, RR_STREAM_WPS>(sc, a, s, n_wg, st) : launch_stream_sw<39, false, RR_STREAM_WPS>(sc, a, s, n_wg, st);
}

} // namespace rr
