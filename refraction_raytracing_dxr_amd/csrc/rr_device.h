// rr_device.h -- gfx950 device code shared by the render and trace kernels:
// fp32 vector helpers in the specified operation order, the DXR TraceRay stand-in
// (LDS-stack BVH2 traversal + scaled Moller-Trumbore with face culling), and the
// ClosestHit / Miss shading of RayTracing.hlsl:66-137.
//
// Arithmetic contract (DESIGN.md "Arithmetic"): compiled with -ffp-contract=off; every
// fused multiply-add is an explicit fmaf; divide and sqrt are the correctly rounded
// IEEE forms hipcc emits by default.  The box test is exempt (it only has to be
// conservative); everything that decides a hit or a colour follows the written order.
#pragma once
#include <hip/hip_runtime.h>
#include "rr_types.h"

namespace rr {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float inv = 1.0f / sqrtf(dot3(a, a));
    return scale3(a, inv);
}

// ---- spec'd atan2 / acos (Cephes single-precision minimax forms; HLSL's are implementation-defined)
// The oracle writes them with branches (oracle/rr_oracle.c: rro_atanf_pos, rro_atan2f, rro_asinf_core, rro_acosf).  Here
// every lane evaluates ONE division chain, ONE square root and ONE polynomial, and the branches become selects of operands
// and results: a wave whose lanes fall into different ranges would otherwise run every branch body in turn (the Miss
// shader is a third of all vector instructions of a frame).  Each lane still performs exactly the operations of the
// branch the oracle takes for its argument, so the bits are the same:
//   atan range reduction   -(1/q) == (-1)/q,  (q-1)/(q+1) as is,  q == q/1   (IEEE division is exact in sign and by 1)
//   acos                   0.5*(1+x) == 0.5*(1-|x|) for x < 0; the argument handed to the asin kernel is <= 0.5 in all
//                          three ranges, so its own "> 0.5" branch is never taken
__device__ __forceinline__ float rr_atan2f(float y, float x)
{
    const float q = fabsf(y) / fabsf(x);
    const bool big = q > 2.414213562373095f, mid = !big && q > 0.4142135623730950f;
    const float num = big ? -1.0f : mid ? q - 1.0f : q;
    const float den = big ? q : mid ? q + 1.0f : 1.0f;
    const float y0 = big ? 1.5707963267948966f : mid ? 0.7853981633974483f : 0.0f;
    const float xr = num / den;
    const float z = xr * xr;
    const float p = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f;
    const float r = p * z * xr + xr;
    float a = y0 + r;
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    float res = y < 0.0f ? -a : a;
    if (x == 0.0f) res = y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f;
    if (y == 0.0f) res = (x > 0.0f || (x == 0.0f && !__builtin_signbit(x))) ? y
                                                                             : (__builtin_signbit(y) ? -3.14159265358979323846f : 3.14159265358979323846f);
    if (x != x || y != y) res = __builtin_nanf("");
    return res;
}

__device__ __forceinline__ float rr_acosf(float x)
{
    const float ax = fabsf(x);
    const bool outer = ax > 0.5f;
    const float a = outer ? sqrtf(0.5f * (1.0f - ax)) : ax;
    const float z = a * a;
    const float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
                     + 1.6666752422e-1f);
    const float r = p * z * a + a;
    const float two_r = 2.0f * r;
    float res = outer ? (x < 0.0f ? 3.14159265358979323846f - two_r : two_r) : 1.5707963267948966f - (x < 0.0f ? -r : r);
    if (!(x >= -1.0f && x <= 1.0f)) res = __builtin_nanf("");
    return res;
}

// D3D ftou: truncate, NaN/negative -> 0, overflow -> 0xffffffff: what v_cvt_u32_f32 does
__device__ __forceinline__ uint32_t ftou(float f)
{
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// typed UAV store to R8G8B8A8_UNORM (RefractionDemo.cpp:431): NaN -> 0, clamp, floor(x*255 + 0.5) (the conversion truncates,
// which is the floor of a positive number)
// Written without branches (every wave ends with three of these): max(x, 0) is 0 for a NaN and for anything below 0 -- the
// conversion of 0 * 255 + 0.5 is 0 --, min(.., 1) gives 1 * 255 + 0.5 = 255.5 -> 255 for anything from 1 up, and a value in
// between goes through the very operations the oracle performs (oracle/rr_oracle.c: rro_unorm8).
__device__ __forceinline__ uint32_t unorm8(float x)
{
    float c;
    asm("v_max_f32 %0, %1, 0" : "=v"(c) : "v"(x));
    asm("v_min_f32 %0, %1, 1.0" : "=v"(c) : "v"(c));
    return ftou(c * 255.0f + 0.5f);
}

// ---- TraceRay ------------------------------------------------------------------------------
struct HitRec {
    float t, U, V, ad;       // barycentric numerators scaled by |det|; u = U/ad, v = V/ad
    uint32_t prim;           // PrimitiveIndex()
    uint32_t leaf;           // index into TriRec/NrmRec (LBVH leaf order)
    uint32_t inst;
    bool hit;
};

// nodes, tris: per-lane visits.  node_trips, leaf_trips: wave-level loop trips (STATS builds only; counted by the lowest
// active lane of each trip): the vector unit issues an internal-node step or a triangle test once per trip whatever the
// number of lanes that take part, so these -- not the lane counts -- are what its time is spent on.
struct TravCounters { uint32_t nodes, tris, node_trips = 0, leaf_trips = 0; };
__device__ __forceinline__ bool first_active_lane()
{
    return (int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(1)) - 1;
}

constexpr uint32_t CULL_BACK = 0x10u, CULL_FRONT = 0x20u;

// ---- box-test ray setup ----------------------------------------------------------------------------
// The box test is exempt from the arithmetic contract: it only has to be CONSERVATIVE (never cull a
// box that holds a triangle the exact-order triangle test would accept).  Slabs are evaluated as
// t = fma(q, cell*inv, (org-O)*inv) with a hardware reciprocal.  With oi = (org-O)*inv the computed t is
// within (|q*cell*inv| + |oi|) * (2^-22 + 2^-23) of the exact slab distance (reciprocal, the two products, the
// fma); |q*cell| <= extent, so moving the near planes earlier and the far planes later by
// pad = |oi|*1e-6 + |inv|*eps_w (the box grows by eps_w >= 5e-6*extent world units, see box_ray), folded into
// the per-ray constants, covers it with a wide margin and the test is a plain tn <= tf.  A (nearly) zero direction component
// gets inv = +-1e20: pad is then huge, so the slab on that axis only rejects origins clearly outside
// it -- rays lying exactly in a box face stay conservative.
//
// Planes are fp16 numbers of cells around the centre of the BLAS bounds (QGrid): plane = org + q*cell, so
// t = q*(cell*inv) + (org - O)*inv; the extra rounding of cell*inv (<= extent*|inv|*2^-24) is far inside
// the |inv|*eps_w term (eps_w >= 1e-5 * largest |coordinate| >= 5e-6 * extent).
struct BoxRay {
    f3 inv;          // cell/D (approximate): t per grid step
    f3 kn, kf;       // additive constants of the near / far plane of each axis: (org-O)/D -/+ pad
    bool sx, sy, sz; // D < 0 on that axis: the hi plane is the near one
};
__device__ __forceinline__ void box_axis(float o, float d, float eps_w, float org, float cell, float& inv_g, float& kn, float& kf, bool& neg)
{
    const float dg = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    const float inv = __builtin_amdgcn_rcpf(dg);
    const float oi = (org - o) * inv;
    const float pad = fmaf(fabsf(oi), 1e-6f, fabsf(inv) * eps_w);
    kn = oi - pad;
    kf = oi + pad;
    inv_g = inv * cell;
    neg = inv < 0.0f;
}
// scene_scale: largest |coordinate| of the geometry the ray is traced against.  Boxes are grown by
// 1e-5 of the larger of that and the ray origin's magnitude: the fp32 triangle test accepts points
// a few ulps outside a triangle's edge (e.g. a ray running exactly along the symmetry plane of a
// mirrored mesh, hitting the shared edges), and the box test must not cull those.
__device__ __forceinline__ BoxRay box_ray(f3 O, f3 D, float scene_scale, const QGrid& g)
{
    const float eps_w = 1e-5f * fmaxf(fmaxf(fabsf(O.x), fabsf(O.y)), fmaxf(fabsf(O.z), scene_scale));
    BoxRay r;
    box_axis(O.x, D.x, eps_w, g.org[0], g.cell[0], r.inv.x, r.kn.x, r.kf.x, r.sx);
    box_axis(O.y, D.y, eps_w, g.org[1], g.cell[1], r.inv.y, r.kn.y, r.kf.y, r.sy);
    box_axis(O.z, D.z, eps_w, g.org[2], g.cell[2], r.inv.z, r.kn.z, r.kf.z, r.sz);
    return r;
}

// a QNode as traversal loads it: two 16-byte requests
struct NodeQ { uint4 a, b; };
// node: an internal child ref = BYTE offset of the node inside the array (uniform base + 32-bit lane offset:
// the address needs no VALU work)
__device__ __forceinline__ NodeQ load_node(const QNode* __restrict__ nodes, int node)
{
    const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes) + (uint32_t)node);
    NodeQ n;
    n.a = q[0]; n.b = q[1];
    return n;
}
// Where a BLAS's nodes are read from.  GlobalNodes: the QNode array in HBM, through the vector L1 (two global_load_dwordx4
// per visit).  LdsNodes: a copy of the whole array in the workgroup's LDS (k_render_lds: two ds_read_b128 per visit) -- the
// texture-addresser / L1 path is the busiest unit of the L1-fed kernel (TA busy 84 %, TD 97 %, 276 cycles per request:
// profiles/r02_pmc_fused.txt) and an LDS read returns in a quarter of that time.
struct GlobalNodes {
    __device__ __forceinline__ NodeQ load(const QNode* __restrict__ nodes, int node) const { return load_node(nodes, node); }
};
struct LdsNodes {
    const char* base;       // the copy (an address in LDS: every use is inlined into the kernel that owns the array)
    __device__ __forceinline__ NodeQ load(const QNode* __restrict__, int node) const
    {
        const uint4* q = reinterpret_cast<const uint4*>(base + (uint32_t)node);
        NodeQ n;
        n.a = q[0]; n.b = q[1];
        return n;
    }
};


constexpr int TRAV_DONE = (int)0x80000000;     // neither an internal index (>= 0) nor a leaf (~i with i < 2^31-1)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { v2f r = { x, x }; return r; }
__device__ __forceinline__ v2f mk2(float x, float y) { v2f r = { x, y }; return r; }

// raw min/max instructions (IEEE minNum/maxNum of their operands; written as asm so that the compiler does not
// add a canonicalising v_max x,x per operand per trip)
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// slab test of BOTH children of a node.  The ray's direction signs say which plane of each axis is the near
// one, so six selects on the packed words (both children at once) replace the min/max of the slab test;
// then twelve v_fma_mix_f32 give the near and far distances of both children, tn0/tn1 are the entry distances.
// one plane word = the fp16 plane coordinates of child 0 (low half) and child 1 (high half), in grid cells; each
// distance is one v_fma_mix_f32 (the half is widened inside the FMA: no conversion instruction)
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v as_h2(uint32_t w) { return __builtin_bit_cast(h2v, w); }
#define RR_PLANE2(word, inv, k) mk2(fmaf((float)as_h2(word).x, inv, k), fmaf((float)as_h2(word).y, inv, k))
__device__ __forceinline__ void box2_hit(const BoxRay& r, const NodeQ& n, float tmin, float tmax,
                                         bool& h0, bool& h1, float& tn0, float& tn1)
{
    const uint32_t lox = n.a.x, loy = n.a.y, loz = n.a.z, hix = n.a.w, hiy = n.b.x, hiz = n.b.y;
    const v2f nx = RR_PLANE2(r.sx ? hix : lox, r.inv.x, r.kn.x);
    const v2f ny = RR_PLANE2(r.sy ? hiy : loy, r.inv.y, r.kn.y);
    const v2f nz = RR_PLANE2(r.sz ? hiz : loz, r.inv.z, r.kn.z);
    const v2f fx = RR_PLANE2(r.sx ? lox : hix, r.inv.x, r.kf.x);
    const v2f fy = RR_PLANE2(r.sy ? loy : hiy, r.inv.y, r.kf.y);
    const v2f fz = RR_PLANE2(r.sz ? loz : hiz, r.inv.z, r.kf.z);
    tn0 = vmax2(vmax3(nx.x, ny.x, nz.x), tmin);
    tn1 = vmax2(vmax3(nx.y, ny.y, nz.y), tmin);
    const float tf0 = vmin2(vmin3(fx.x, fy.x, fz.x), tmax);
    const float tf1 = vmin2(vmin3(fx.y, fy.y, fz.y), tmax);
    h0 = tn0 <= tf0;
    h1 = tn1 <= tf1;
}

// per-lane traversal stack in LDS, column layout (entry e of lane l at base[e*64 + l]: conflict-free).  The
// stack pointer is the LDS address of the next free entry; depth never exceeds the tree depth (near child
// followed, far child pushed), and the host picks a stack at least that deep, so there is no overflow check.
constexpr int STACK_STRIDE = 64;
// Stack entries are child refs.  32-bit entries hold them as they are; 16-bit entries (meshes below 32 768 triangles:
// a quarter of the LDS per wave, so deep trees keep eight waves per SIMD) hold a SIGNED half: node index (>= 0) or ~leaf
// (< 0).  A ref is a byte offset (a multiple of 32) or ~leaf, so packing is min(ref >> 5, ref) and unpacking max(v << 5, v)
// of the sign-extended entry: two instructions each way.
template <class E> struct StackCodec;
template <> struct StackCodec<uint32_t> {
    static __device__ __forceinline__ uint32_t enc(int ref) { return (uint32_t)ref; }
    static __device__ __forceinline__ int dec(uint32_t v) { return (int)v; }
};
template <> struct StackCodec<uint16_t> {
    static __device__ __forceinline__ uint16_t enc(int ref) { const int n = ref >> 5; return (uint16_t)(n < ref ? n : ref); }
    static __device__ __forceinline__ int dec(uint16_t v) { const int s = (int)(int16_t)v, n = s << 5; return n > s ? n : s; }
};

// one traversal step at an internal node: returns the next node (near child, or a popped entry, or
// TRAV_DONE) and pushes the far child when both are hit.  top: next free entry, floor: lowest entry that may be popped.
template <class E>
__device__ __forceinline__ int node_step(const BoxRay& br, const NodeQ& n, float tmin, float tmax, E*& top, const E* floor)
{
    bool h0, h1;
    float tn0, tn1;
    box2_hit(br, n, tmin, tmax, h0, h1, tn0, tn1);
    const int c0 = (int)n.b.z, c1 = (int)n.b.w;
    const bool both = h0 && h1, swap = tn1 < tn0;
    const int nearc = (h0 && !(h1 && swap)) ? c0 : c1;
    int next = (h0 || h1) ? nearc : TRAV_DONE;
    if (both) { *top = StackCodec<E>::enc(swap ? c0 : c1); top += STACK_STRIDE; }
    if (!(h0 || h1) && top > floor) { top -= STACK_STRIDE; next = StackCodec<E>::dec(*top); }
    return next;
}

// diagnostic builds count wave-level loop trips in LDS (one word per wave); null in product builds
struct Diag { uint32_t* trips; uint32_t stride = 4; };      // trips[0]: internal-node trips, trips[stride]: leaf trips, trips[2*stride]: shading passes (one word per wave each)
__device__ __forceinline__ void diag_trip(const Diag& d, int which = 0)
{
    if (d.trips) { const unsigned long long m = __ballot(1); if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) d.trips[which * d.stride] += 1u; }
}


// scaled Moller-Trumbore; front-facing <=> det > 0.  Equal-t ties go to the lower
// (instance, primitive) so that the result does not depend on traversal order.
__device__ __forceinline__ void tri_test(const TriRec* __restrict__ tris, uint32_t leaf, f3 O, f3 D, float tmin,
                                         uint32_t cull, uint32_t inst, HitRec& best)
{
    const float4* q = reinterpret_cast<const float4*>(tris + leaf);
    float4 a = q[0], b = q[1], c = q[2];
    f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
    uint32_t prim = __float_as_uint(a.w);
    f3 pv = cross3(D, e2);
    float det = dot3(e1, pv);
    if (cull & CULL_BACK) { if (!(det > 0.0f)) return; }
    else if (cull & CULL_FRONT) { if (!(det < 0.0f)) return; }
    else if (!(det != 0.0f)) return;
    f3 tv = sub3(O, v0);
    float U = dot3(tv, pv);
    f3 qv = cross3(tv, e1);
    float V = dot3(D, qv);
    float T = dot3(e2, qv);
    float ad = det;
    if (det < 0.0f) { U = -U; V = -V; T = -T; ad = -det; }
    if (U < 0.0f || V < 0.0f || U + V > ad) return;
    float t = T / ad;
    if (!(t > tmin)) return;
    // (the barycentrics of the winner are recomputed once, after the traversal: hit_attributes.  Carrying them through
    // the loops costs four registers per lane; an exact tie needs the other triangle's primitive index, a rare load.)
    if (t < best.t || (t == best.t && best.hit && (inst < best.inst || (inst == best.inst && prim < tris[best.leaf].prim)))) {
        best.t = t; best.leaf = leaf; best.inst = inst;
        best.hit = true;
    }
}

// Attributes of the closest hit: the same operations, in the same order, as the test that accepted it (O, D: the ray in
// the space of the triangle's BLAS), so U, V, ad have the very bits tri_test computed.
__device__ __forceinline__ void hit_attributes(const TriRec* __restrict__ tris, f3 O, f3 D, HitRec& best)
{
    const float4* q = reinterpret_cast<const float4*>(tris + best.leaf);
    float4 a = q[0], b = q[1], c = q[2];
    f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
    f3 pv = cross3(D, e2);
    float det = dot3(e1, pv);
    f3 tv = sub3(O, v0);
    float U = dot3(tv, pv);
    f3 qv = cross3(tv, e1);
    float V = dot3(D, qv);
    float ad = det;
    if (det < 0.0f) { U = -U; V = -V; ad = -det; }
    best.U = U; best.V = V; best.ad = ad; best.prim = __float_as_uint(a.w);
}

// "while-while" traversal with an early hand-over: called by the lanes still descending (exec = those lanes, so
// the count is a scalar s_bcnt1 of exec, no VALU work); true once max(2, n_in/4) of the n_in lanes that entered the
// internal-node phase have left it.  Measured on MI355X at Depth 64 against waiting for every lane:
// monkey.obj 122 -> 108 us/frame, ott.obj 218 -> 177, sphere.obj 194 -> 188; thresholds 1..8 and n_in/2..n_in/8
// are all within 2 % of each other.
__device__ __forceinline__ bool leaf_phase_due(int n_in)
{
    const int quarter = n_in >> 2;
    const int thr = quarter > 2 ? quarter : 2;
    return __popcll(__ballot(1)) + thr <= n_in;
}

// One BLAS.  stk: this lane's LDS stack column (entry e at stk[e*64]); sp0: entries already in use
// (two-level traversal leaves the TLAS part of the stack below sp0).
// "while-while" form: the lanes descend internal nodes (near child first, far child pushed) until enough of
// them hold a leaf or have finished (leaf_phase_due); then the (expensive) triangle test is executed once for
// all lanes that hold a leaf.
template <bool STATS, class E, class NS = GlobalNodes>
__device__ __forceinline__ void walk_blas(const QNode* __restrict__ nodes, const TriRec* __restrict__ tris, int root, const BoxRay& br,
                                          f3 O, f3 D, float tmin, uint32_t cull, uint32_t inst, HitRec& best, E* stk,
                                          TravCounters& cnt, const Diag dg = Diag{ nullptr }, const NS ns = NS{})
{
    E* top = stk;
    int node = root;
    for (;;) {
        // internal-node phase.  Lanes drop out as they reach a leaf (or finish); the phase ends for the whole wave
        // once a quarter of the lanes that entered it have dropped out (see leaf_phase_due), not when the last one
        // has: lanes holding a leaf do not wait for the longest descent in the wave.
        const int n_in = __popcll(__ballot(node >= 0));
        while (node >= 0) {
            if (leaf_phase_due(n_in)) break;
            diag_trip(dg);
            const NodeQ q = ns.load(nodes, node);
            if (STATS) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; }
            node = node_step(br, q, tmin, best.t, top, stk);
#ifdef RR_EXP_EXTRA_VALU      // experiment: what do N more VALU instructions per visit cost?
            { float dv = br.inv.x;
#pragma unroll
              for (int k = 0; k < RR_EXP_EXTRA_VALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dv));
              asm volatile("" :: "v"(dv)); }
#endif
        }
        // leaf phase: every lane that holds a leaf tests its triangle and pops
        if (node < 0 && node != TRAV_DONE) {
            diag_trip(dg, 1);
            if (STATS) { cnt.tris++; if (first_active_lane()) cnt.leaf_trips++; }
            tri_test(tris, (uint32_t)~node, O, D, tmin, cull, inst, best);
            if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
        }
        if (__ballot(node != TRAV_DONE) == 0ull) break;
    }
}

template <bool STATS, class E, class NS = GlobalNodes>
__device__ __forceinline__ void trace_blas(const BlasDev& bl, f3 O, f3 D, float tmin, uint32_t cull, uint32_t inst,
                                           HitRec& best, E* stk, TravCounters& cnt, const Diag dg = Diag{ nullptr },
                                           const NS ns = NS{})
{
    const BoxRay br = box_ray(O, D, bl.scale, bl.grid);
    walk_blas<STATS, E, NS>(bl.nodes, bl.tris, 0, br, O, D, tmin, cull, inst, best, stk, cnt, dg, ns);
    if (best.hit) hit_attributes(bl.tris, O, D, best);
}

// ---- group-parallel traversal: G lanes (an aligned group of 2 or 4) walk ONE ray's tree ------------------------------
// For launches whose length is a chain of dependent rays (DispatchRays(W,H,1): k_render_paths): once most lanes of a wave
// have finished their path, a ray level costs as many loop trips as its longest ray has node visits, with a handful of lanes
// taking part.  Here the rays still alive are compacted to the front of the wave and each gets G lanes.  Every lane keeps an
// ordinary depth-first stack of its own (so the tree depth still bounds it); a lane that runs out of work takes the OLDEST
// entry -- the root of the largest untouched subtree -- of its right-hand neighbour's stack (a ring inside the group: work
// spreads from the lane that starts at the root), and the closest hit so far is shared for culling after every trip.  The
// winner -- smallest t, ties to the lower primitive, exactly trace_blas's rule -- is the same triangle whatever the order.
template <int G> __device__ __forceinline__ uint32_t grp_next(uint32_t v);     // v of the next lane of the group's ring
template <int G> __device__ __forceinline__ uint32_t grp_prev(uint32_t v);
template <> __device__ __forceinline__ uint32_t grp_next<4>(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x39, 0xf, 0xf, true); }   // quad_perm:[1,2,3,0]
template <> __device__ __forceinline__ uint32_t grp_prev<4>(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x93, 0xf, 0xf, true); }   // quad_perm:[3,0,1,2]
template <> __device__ __forceinline__ uint32_t grp_next<2>(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true); }   // quad_perm:[1,0,3,2]
template <> __device__ __forceinline__ uint32_t grp_prev<2>(uint32_t v) { return grp_next<2>(v); }
template <int G> __device__ __forceinline__ float grp_min_f(float v)
{
    float m = fminf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xb1, 0xf, 0xf, true)));              // [1,0,3,2]
    if (G == 4) m = fminf(m, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0x4e, 0xf, 0xf, true)));        // [2,3,0,1]
    return m;
}
template <int G> __device__ __forceinline__ uint32_t grp_min_u(uint32_t v)
{
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);
    uint32_t m = v < o ? v : o;
    if (G == 4) { o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4e, 0xf, 0xf, true); m = m < o ? m : o; }
    return m;
}

// Called by ALL 64 lanes of a wave.  alive / m_alive: the lanes that have a ray to trace (at most 64 / G of them); ww: the
// wave's stack region as 32-bit words (at least 9 rows of 64) -- the stacks are empty between two rays, so the region also
// carries the rays to their groups and the results back; stk: the lane's own stack column.  On return the alive lanes hold
// what trace_blas would have given them.
template <int G, bool STATS, class E>
__device__ __forceinline__ void trace_blas_group(const BlasDev& bl, bool alive, unsigned long long m_alive, f3 O, f3 D, float tmin, float tmax,
                                                 uint32_t cull, HitRec& best, E* stk, uint32_t* ww, uint32_t lane, TravCounters& cnt)
{
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_alive >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_alive, 0u));
    const uint32_t n_rays = (uint32_t)__popcll(m_alive);
    if (alive) {
        ww[0 * 64 + rank] = __float_as_uint(O.x); ww[1 * 64 + rank] = __float_as_uint(O.y); ww[2 * 64 + rank] = __float_as_uint(O.z);
        ww[3 * 64 + rank] = __float_as_uint(D.x); ww[4 * 64 + rank] = __float_as_uint(D.y); ww[5 * 64 + rank] = __float_as_uint(D.z);
        ww[6 * 64 + rank] = __float_as_uint(tmax); ww[7 * 64 + rank] = cull; ww[8 * 64 + rank] = __float_as_uint(tmin);
    }
    const uint32_t g = lane / G;
    const bool work = g < n_rays;
    const uint32_t gs = work ? g : 0u;
    const f3 Og = mk3(__uint_as_float(ww[0 * 64 + gs]), __uint_as_float(ww[1 * 64 + gs]), __uint_as_float(ww[2 * 64 + gs]));
    const f3 Dg = mk3(__uint_as_float(ww[3 * 64 + gs]), __uint_as_float(ww[4 * 64 + gs]), __uint_as_float(ww[5 * 64 + gs]));
    const float tmaxg = __uint_as_float(ww[6 * 64 + gs]);
    const uint32_t cullg = ww[7 * 64 + gs];
    const float tming = __uint_as_float(ww[8 * 64 + gs]);        // (a helper lane's own tmin / tmax are those of the level its path ended at)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the rows are stack space again from here on

    const BoxRay br = box_ray(Og, Dg, bl.scale, bl.grid);
    const QNode* __restrict__ nodes = bl.nodes;
    E* const col0 = stk - lane;                                   // entry e of lane l at col0[e * 64 + l]
    E* top = stk;
    E* bot = stk;                                                 // entries below bot have been taken by the neighbour
    int node = (work && (lane % G) == 0u) ? 0 : TRAV_DONE;
    HitRec b;
    b.t = tmaxg; b.hit = false; b.prim = 0; b.leaf = 0; b.inst = 0; b.U = 0.0f; b.V = 0.0f; b.ad = 1.0f;
    float tq = tmaxg;                                             // closest hit of the whole group so far
    for (;;) {
        const int n_in = __popcll(__ballot(node >= 0));
        while (node >= 0) {
            if (leaf_phase_due(n_in)) break;
            const NodeQ q = load_node(nodes, node);
            if (STATS) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; }
            node = node_step(br, q, tming, tq, top, bot);
        }
        if (node < 0 && node != TRAV_DONE) {
            if (STATS) { cnt.tris++; if (first_active_lane()) cnt.leaf_trips++; }
            tri_test(bl.tris, (uint32_t)~node, Og, Dg, tming, cullg, 0u, b);
            if (top > bot) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
        }
        // all 64 lanes again: share the closest hit, hand work round the ring
        tq = grp_min_f<G>(b.t);
        const bool idle = node == TRAV_DONE;                      // (an idle lane's own stack is empty)
        const uint32_t nb_bot = grp_next<G>((uint32_t)(bot - col0)), nb_has = grp_next<G>(top > bot ? 1u : 0u);
        const uint32_t thief_idle = grp_prev<G>(idle ? 1u : 0u);
        if (idle && nb_has) node = StackCodec<E>::dec(col0[nb_bot]);
        if (top > bot && thief_idle) bot += STACK_STRIDE;
        if (__ballot(node != TRAV_DONE) == 0ull) break;
    }
    // the group's closest hit: smallest t, ties to the lower primitive
    const float tw = grp_min_f<G>(b.hit ? b.t : 3.0e38f);
    const bool cand = b.hit && b.t == tw;
    const uint32_t prim = cand ? bl.tris[b.leaf].prim : 0xffffffffu;
    const uint32_t pw = grp_min_u<G>(prim);
    const uint32_t lw = ~grp_min_u<G>((cand && prim == pw) ? ~b.leaf : 0xffffffffu);
    if (work && (lane % G) == 0u) { ww[0 * 64 + g] = __float_as_uint(tw); ww[1 * 64 + g] = tw < 3.0e38f ? lw : 0xffffffffu; }
    best.t = tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f; best.ad = 1.0f;
    if (alive) {
        const uint32_t l = ww[1 * 64 + rank];
        if (l != 0xffffffffu) { best.t = __uint_as_float(ww[0 * 64 + rank]); best.leaf = l; best.hit = true; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (best.hit) hit_attributes(bl.tris, O, D, best);
}

__device__ __forceinline__ f3 xform_point(const float* m, f3 p)
{
    return mk3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
               ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
__device__ __forceinline__ f3 xform_dir(const float* m, f3 p)
{
    return mk3((m[0] * p.x + m[1] * p.y) + m[2] * p.z,
               (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
               (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}

// The record of instance ii.  When every active lane asks for the same instance (the common case of a wave whose rays stay
// together: C4) the index is wave-uniform and the 96 bytes come through the scalar cache once, instead of 64 lanes x 96 B
// through the texture path, which the node fetches already keep > 90 % busy (profiles/r03_pmc_c4_fused.txt).
__device__ __forceinline__ InstDev inst_record(const InstDev* __restrict__ insts, uint32_t ii)
{
    const uint32_t iu = (uint32_t)__builtin_amdgcn_readfirstlane((int)ii);
    InstDev r;
    if (__ballot(ii != iu) == 0ull) r = insts[iu];
    else r = insts[ii];
    return r;
}

// TraceRay(Scene, flags, 0xff, 0,0,0, ray, payload): closest hit over TLAS -> BLAS.
// TLAS = false: the reference's scene, one BLAS.  TLAS = true: the wave walks the top level in world space until every lane
// holds an instance leaf or has finished; the lanes at a leaf then walk their instances (walk_blas: the single-BLAS loop with
// its early hand-over, on the ray's object-space image -- t is preserved, the direction is not renormalised -- with the
// top level's entries left below on the stack) and pop the next top-level entry.  Equal-t ties go to the lower (instance,
// primitive), so the order instances are visited in does not matter.  (Rounds 1-2 walked both levels in ONE loop over the
// flattened pool, so that a lane leaving an instance need not wait for the others; with a leaf part that tests triangles,
// enters and leaves instances it cost more than the waiting: C4 0.96 -> 0.90 ms, C5 on this kernel 6.8 -> 6.1.)
template <bool STATS, bool TLAS, class E = uint32_t, class NS = GlobalNodes>
__device__ __forceinline__ void trace_scene(const SceneDev& sc, f3 O, f3 D, float tmin, float tmax, uint32_t flags,
                                            HitRec& best, E* stk_e, TravCounters& cnt,
                                            const Diag dg = Diag{ nullptr }, const NS ns = NS{})
{
    best.t = tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f;
    best.ad = 1.0f;
    if (!TLAS) {      // the reference's scene: one identity instance, mask 1, flags 0 (RefractionDemo.cpp:324-334)
        trace_blas<STATS, E, NS>(sc.blas0, O, D, tmin, flags, 0u, best, stk_e, cnt, dg, ns);
        return;
    }
    E* stk = stk_e;                         // (16-bit entries: scenes of fewer than 32 768 pool nodes and triangles + instances)
    const QNode* __restrict__ nodes = sc.pool_nodes;
    E* top = stk;
    int node = 0;
    const BoxRay br = box_ray(O, D, sc.scale, sc.grid);
    for (;;) {
        while (node >= 0) {
            const NodeQ q = load_node(nodes, node);
            if (STATS) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; }
            node = node_step(br, q, tmin, best.t, top, stk);
        }
        if (node == TRAV_DONE) break;
        if (STATS && first_active_lane()) cnt.leaf_trips++;
        const uint32_t ii = (uint32_t)~node - sc.n_pool_tris;               // a leaf of the top level is an instance
        const InstDev in = inst_record(sc.insts, ii);
        if (in.mask & 0xffu) {
            uint32_t f = flags;
            if (in.flags & 0x1u) f &= ~(CULL_BACK | CULL_FRONT);
            else if (in.flags & 0x2u) {
                if (f & CULL_BACK) f = (f & ~CULL_BACK) | CULL_FRONT;
                else if (f & CULL_FRONT) f = (f & ~CULL_FRONT) | CULL_BACK;
            }
            f3 Oc = O, Dc = D;
            if (!in.identity) { Oc = xform_point(in.inv, O); Dc = xform_dir(in.inv, D); }
            const BoxRay bi = box_ray(Oc, Dc, in.scale, in.grid);
            walk_blas<STATS, E, GlobalNodes>(nodes, sc.pool_tris, (int)in.root, bi, Oc, Dc, tmin, f, ii, best, top, cnt);
        }
        if (top > stk) { top -= STACK_STRIDE; node = StackCodec<E>::dec(*top); } else node = TRAV_DONE;
    }
    if (best.hit) {
        const InstDev in = inst_record(sc.insts, best.inst);
        f3 Oh = O, Dh = D;
        if (!in.identity) { Oh = xform_point(in.inv, O); Dh = xform_dir(in.inv, D); }
        hit_attributes(sc.pool_tris, Oh, Dh, best);
    }
}

// ---- Miss: RayTracing.hlsl:127-137 -------------------------------------------------------------
__device__ __forceinline__ f3 env_lookup(const SceneDev& sc, f3 r)
{
    if (!sc.env) return mk3(0.0f, 0.0f, 0.0f);
    float at = rr_atan2f(r.x, r.z);
    float ac = rr_acosf(r.y);
    float theta = (float)sc.env_w * (at / 3.14159f + 1.0f) / 2;      // :133
    float phi = (float)sc.env_h * (ac / 3.14159f);                   // :134
    uint32_t ix = ftou(theta), iy = ftou(phi);                       // :135 operator[]: ftou, OOB reads 0
    if (ix >= (uint32_t)sc.env_w || iy >= (uint32_t)sc.env_h) return mk3(0.0f, 0.0f, 0.0f);
    float4 t = sc.env[(size_t)iy * (uint32_t)sc.env_w + ix];
    return mk3(t.x, t.y, t.z);
}

// ---- ReflectRay / RefractRay: RayTracing.hlsl:66-76 ---------------------------------------------
__device__ __forceinline__ f3 reflect_ray(f3 I, f3 N)
{
    float k = 2.0f * dot3(N, I);
    return mk3(I.x - k * N.x, I.y - k * N.y, I.z - k * N.z);
}

__device__ __forceinline__ bool refract_ray(f3& R, f3 I, f3 N, float eta)
{
    float c = dot3(N, I);
    float k = 1.0f - (eta * eta) * (1.0f - c * c);
    if (k < 0.0f) return false;
    float a = eta * c + sqrtf(k);
    R = normalize3(mk3(eta * I.x - a * N.x, eta * I.y - a * N.y, eta * I.z - a * N.z));
    return true;
}

// ---- GenerateCameraRay: RayTracing.hlsl:27-40 ----------------------------------------------------
// screen coordinate of a pixel centre (hlsl:29-33), the same operations for every pixel of a column / row: k_screen_tables
// evaluates them once per column and row, the render kernels read the two tables
__device__ __forceinline__ float screen_coord(uint32_t i, uint32_t n, bool flip)
{
    const float p = (float)i + 0.5f;
    const float s = p / (float)n * 2.0f - 1.0f;
    return flip ? -s : s;
}
__device__ __forceinline__ f3 camera_ray_dir(const float* M, float sx, float sy)
{
    f3 R;
    R.x = (sx * M[0] + sy * M[1]) + M[3];
    R.y = (sx * M[4] + sy * M[5]) + M[7];
    R.z = (sx * M[8] + sy * M[9]) + M[11];
    return normalize3(R);
}

// shading normal of ClosestHit (RayTracing.hlsl:83-86); instance != identity: inverse-transpose
// (extension, the reference's only instance is identity)
template <bool TLAS>
__device__ __forceinline__ f3 shading_normal(const SceneDev& sc, const HitRec& h)
{
    const float4* q = reinterpret_cast<const float4*>((!TLAS ? sc.blas0.nrms : sc.pool_nrms) + h.leaf);
    float4 a = q[0], b = q[1], c = q[2];
    float u = h.U / h.ad, v = h.V / h.ad;
    f3 A = mk3(a.x, a.y, a.z);
    f3 BA = sub3(mk3(b.x, b.y, b.z), A), CA = sub3(mk3(c.x, c.y, c.z), A);
    f3 Nr = mk3(fmaf(v, CA.x, fmaf(u, BA.x, A.x)), fmaf(v, CA.y, fmaf(u, BA.y, A.y)), fmaf(v, CA.z, fmaf(u, BA.z, A.z)));
    const InstDev in = TLAS ? inst_record(sc.insts, h.inst) : InstDev{};
    if (TLAS && !in.identity) {
        const float* w = in.inv;
        Nr = mk3((w[0] * Nr.x + w[4] * Nr.y) + w[8] * Nr.z,
                 (w[1] * Nr.x + w[5] * Nr.y) + w[9] * Nr.z,
                 (w[2] * Nr.x + w[6] * Nr.y) + w[10] * Nr.z);
    }
    return normalize3(Nr);
}

// Sum over the wave, for callers with ALL 64 lanes active (the kernels call it once, at their end): seven DPP adds -- inside
// each row of 16 lanes (row_shr 1, 2, 3, then 4 and 8 under bank masks), then row_bcast 15 / 31 carry the row totals on -- leave
// the total in lane 63.  No LDS round trips (six dependent ds_bpermute were the last 600 cycles of every wave's life).
// Returned wave-uniform.
__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v)
{
    uint32_t t;
    t = v + __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);          // row_shr:1
    t = t + __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);          // row_shr:2
    t = t + __builtin_amdgcn_update_dpp(0u, v, 0x113, 0xf, 0xf, true);          // row_shr:3
    t = t + __builtin_amdgcn_update_dpp(0u, t, 0x114, 0xf, 0xe, true);          // row_shr:4, banks 1..3
    t = t + __builtin_amdgcn_update_dpp(0u, t, 0x118, 0xf, 0xc, true);          // row_shr:8, banks 2..3
    t = t + __builtin_amdgcn_update_dpp(0u, t, 0x142, 0xa, 0xf, true);          // row_bcast:15 into rows 1 and 3
    t = t + __builtin_amdgcn_update_dpp(0u, t, 0x143, 0xc, 0xf, true);          // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
}

} // namespace rr
