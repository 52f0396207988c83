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
__device__ __forceinline__ float atan_pos(float x)
{
    float y0;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else { y0 = 0.0f; }
    float z = x * x;
    float p = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f;
    float r = p * z * x + x;
    return y0 + r;
}

__device__ __forceinline__ float rr_atan2f(float y, float x)
{
    if (x != x || y != y) return __builtin_nanf("");
    if (y == 0.0f) {
        if (x > 0.0f || (x == 0.0f && !__builtin_signbit(x))) return y;
        return __builtin_signbit(y) ? -3.14159265358979323846f : 3.14159265358979323846f;
    }
    if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f;
    float a = atan_pos(fabsf(y) / fabsf(x));
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

__device__ __forceinline__ float asin_core(float a)
{
    float z, x;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); x = sqrtf(z); }
    else { x = a; z = x * x; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f);
    float r = p * z * x + x;
    if (flag) { r = r + r; r = 1.5707963267948966f - r; }
    return r;
}

__device__ __forceinline__ float rr_acosf(float x)
{
    if (!(x >= -1.0f && x <= 1.0f)) return __builtin_nanf("");
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_core(sqrtf(0.5f * (1.0f - x)));
    float s = asin_core(fabsf(x));
    if (x < 0.0f) s = -s;
    return 1.5707963267948966f - s;
}

// D3D ftou: truncate, NaN/negative -> 0, overflow -> 0xffffffff
__device__ __forceinline__ uint32_t ftou(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// typed UAV store to R8G8B8A8_UNORM (RefractionDemo.cpp:431)
__device__ __forceinline__ uint32_t unorm8(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 1.0f) return 255u;
    return (uint32_t)floorf(x * 255.0f + 0.5f);
}

// ---- TraceRay ------------------------------------------------------------------------------
struct HitRec {
    float t, U, V, ad;       // barycentric numerators scaled by |det|; u = U/ad, v = V/ad
    uint32_t prim;           // PrimitiveIndex()
    uint32_t leaf;           // index into TriRec/NrmRec (LBVH leaf order)
    uint32_t inst;
    bool hit;
};

struct TravCounters { uint32_t nodes, tris; };

constexpr uint32_t CULL_BACK = 0x10u, CULL_FRONT = 0x20u;

// ---- box-test ray setup ----------------------------------------------------------------------------
// The box test is exempt from the arithmetic contract: it only has to be CONSERVATIVE (never cull a
// box that holds a triangle the exact-order triangle test would accept).  Slabs are evaluated as
// t = fma(plane, inv, -(O*inv)) with a hardware reciprocal; the rounding of that form
// (<= |t|*2^-22 + |O*inv|*2^-24) is covered by moving the near planes earlier and the far planes later
// by pad = |O*inv|*2^-22 + |inv|*eps_w (i.e. the box grows by eps_w world units, see box_ray), folded
// into the per-ray constants, plus a relative 2^-20 on t_far.  A (nearly) zero direction component
// gets inv = +-1e20: pad is then huge, so the slab on that axis only rejects origins clearly outside
// it -- rays lying exactly in a box face stay conservative.
struct BoxRay {
    f3 inv;        // 1/D (approximate)
    f3 klo, khi;   // additive constants for the lo / hi planes: -(O*inv) -/+ sign(inv)*pad
};
__device__ __forceinline__ void box_axis(float o, float d, float eps_w, float& inv, float& klo, float& khi)
{
    const float dg = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    inv = __builtin_amdgcn_rcpf(dg);
    const float oi = -(o * inv);
    const float pad = copysignf(fmaf(fabsf(oi), 2.4e-7f, fabsf(inv) * eps_w), inv);
    klo = oi - pad;
    khi = oi + pad;
}
// scene_scale: largest |coordinate| of the geometry the ray is traced against.  Boxes are grown by
// 1e-5 of the larger of that and the ray origin's magnitude: the fp32 triangle test accepts points
// a few ulps outside a triangle's edge (e.g. a ray running exactly along the symmetry plane of a
// mirrored mesh, hitting the shared edges), and the box test must not cull those.
__device__ __forceinline__ BoxRay box_ray(f3 O, f3 D, float scene_scale)
{
    const float eps_w = 1e-5f * fmaxf(fmaxf(fabsf(O.x), fabsf(O.y)), fmaxf(fabsf(O.z), scene_scale));
    BoxRay r;
    box_axis(O.x, D.x, eps_w, r.inv.x, r.klo.x, r.khi.x);
    box_axis(O.y, D.y, eps_w, r.inv.y, r.klo.y, r.khi.y);
    box_axis(O.z, D.z, eps_w, r.inv.z, r.klo.z, r.khi.z);
    return r;
}

constexpr int TRAV_DONE = (int)0x80000000;     // neither an internal index (>= 0) nor a leaf (~i with i < 2^31-1)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { v2f r = { x, x }; return r; }
__device__ __forceinline__ v2f mk2(float x, float y) { v2f r = { x, y }; return r; }

// slab test of BOTH children of a node (q0,q1,q2 = the node's first three 16-byte words);
// six v_pk_fma_f32 give the twelve plane distances, tn0/tn1 are the entry distances
__device__ __forceinline__ void box2_hit(const BoxRay& r, const float4 q0, const float4 q1, const float4 q2, float tmin, float tmax,
                                         bool& h0, bool& h1, float& tn0, float& tn1)
{
    const v2f ax = __builtin_elementwise_fma(mk2(q0.x, q0.y), splat2(r.inv.x), splat2(r.klo.x));
    const v2f ay = __builtin_elementwise_fma(mk2(q0.z, q0.w), splat2(r.inv.y), splat2(r.klo.y));
    const v2f az = __builtin_elementwise_fma(mk2(q1.x, q1.y), splat2(r.inv.z), splat2(r.klo.z));
    const v2f bx = __builtin_elementwise_fma(mk2(q1.z, q1.w), splat2(r.inv.x), splat2(r.khi.x));
    const v2f by = __builtin_elementwise_fma(mk2(q2.x, q2.y), splat2(r.inv.y), splat2(r.khi.y));
    const v2f bz = __builtin_elementwise_fma(mk2(q2.z, q2.w), splat2(r.inv.z), splat2(r.khi.z));
    tn0 = fmaxf(fmaxf(fmaxf(fminf(ax.x, bx.x), fminf(ay.x, by.x)), fminf(az.x, bz.x)), tmin);
    tn1 = fmaxf(fmaxf(fmaxf(fminf(ax.y, bx.y), fminf(ay.y, by.y)), fminf(az.y, bz.y)), tmin);
    const float tf0 = fminf(fminf(fminf(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x)), fmaxf(az.x, bz.x)), tmax);
    const float tf1 = fminf(fminf(fminf(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y)), fmaxf(az.y, bz.y)), tmax);
    h0 = tn0 <= tf0 * 1.000001f;
    h1 = tn1 <= tf1 * 1.000001f;
}

// per-lane traversal stack in LDS, column layout (entry e of lane l at base[e*64 + l]: conflict-free)
struct Stack32 {
    uint32_t* p;
    __device__ __forceinline__ void push(int sp, int v) const { p[sp * 64] = (uint32_t)v; }
    __device__ __forceinline__ int pop(int sp) const { return (int)p[sp * 64]; }
};
// 16-bit entries (half the LDS) for BVHs with < 32768 nodes and leaves: internal i -> i, leaf ~l -> 0x8000 | l
struct Stack16 {
    uint16_t* p;
    __device__ __forceinline__ void push(int sp, int v) const { p[sp * 64] = v >= 0 ? (uint16_t)v : (uint16_t)(0x8000u | (uint32_t)~v); }
    __device__ __forceinline__ int pop(int sp) const { const uint32_t u = p[sp * 64]; return (u & 0x8000u) ? ~(int)(u & 0x7fffu) : (int)u; }
};

// one traversal step at an internal node: returns the next node (near child, or a popped entry, or
// TRAV_DONE) and pushes the far child when both are hit.  stk: this lane's LDS column, sp0: stack floor.
template <bool CHECK, class StackT>
__device__ __forceinline__ int node_step(const BoxRay& br, const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                         float tmin, float tmax, const StackT stk, int& sp, int sp0, int cap, uint32_t& err)
{
    bool h0, h1;
    float tn0, tn1;
    box2_hit(br, q0, q1, q2, tmin, tmax, h0, h1, tn0, tn1);
    const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
    const bool both = h0 && h1, swap = tn1 < tn0;
    const int nearc = (h0 && !(h1 && swap)) ? c0 : c1;
    int next = (h0 || h1) ? nearc : TRAV_DONE;
    if (both) {
        if (!CHECK || sp < cap) { stk.push(sp, swap ? c0 : c1); ++sp; } else err = 1u;
    }
    if (!(h0 || h1) && sp > sp0) { --sp; next = stk.pop(sp); }
    return next;
}

// diagnostic builds count wave-level loop trips in LDS (one word per wave); null in product builds
struct Diag { uint32_t* trips; };      // trips[0]: internal-node trips, trips[4]: leaf trips (one word per wave each)
__device__ __forceinline__ void diag_trip(const Diag& d, int which = 0)
{
    if (d.trips) { const unsigned long long m = __ballot(1); if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) d.trips[which * 4] += 1u; }
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
    if (t < best.t || (t == best.t && best.hit && (inst < best.inst || (inst == best.inst && prim < best.prim)))) {
        best.t = t; best.U = U; best.V = V; best.ad = ad; best.prim = prim; best.leaf = leaf; best.inst = inst;
        best.hit = true;
    }
}

// One BLAS.  stk: this lane's LDS stack column (entry e at stk[e*64]); sp0: entries already in use
// (two-level traversal leaves the TLAS part of the stack below sp0).
// "while-while" form: all lanes first descend internal nodes (near child first, far child pushed)
// until every lane of the wave holds a leaf or has finished; only then is the (expensive) triangle
// test executed, once, for all lanes that hold a leaf.
template <int STACK, bool STATS, class StackT = Stack32>
__device__ __forceinline__ void trace_blas(const BlasDev& bl, f3 O, f3 D, float tmin, uint32_t cull, uint32_t inst,
                                           HitRec& best, const StackT stk, int sp0, uint32_t* err, TravCounters& cnt,
                                           const Diag dg = Diag{ nullptr })
{
    const BoxRay br = box_ray(O, D, bl.scale);
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(bl.nodes);
    int sp = sp0;
    int node = 0;
    for (;;) {
        while (node >= 0) {
            diag_trip(dg);
            const float4* q = nodes + (uint32_t)node * 4u;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            if (STATS) cnt.nodes++;
            node = node_step<true>(br, q0, q1, q2, q3, tmin, best.t, stk, sp, sp0, STACK, *err);
        }
        if (node == TRAV_DONE) break;
        diag_trip(dg, 1);
        if (STATS) cnt.tris++;
        tri_test(bl.tris, (uint32_t)~node, O, D, tmin, cull, inst, best);
        if (sp > sp0) { --sp; node = stk.pop(sp); } else node = TRAV_DONE;
    }
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

// TraceRay(Scene, flags, 0xff, 0,0,0, ray, payload): closest hit over TLAS -> BLAS.
// TLAS = false: the reference's scene, one BLAS.  TLAS = true: one loop over the flattened node pool;
// reaching an instance leaf swaps the lane's ray for its object-space image (t is preserved: the
// direction is not renormalised) and remembers the stack level, exhausting that level swaps it back.
template <int STACK, bool STATS, bool TLAS>
__device__ __forceinline__ void trace_scene(const SceneDev& sc, f3 O, f3 D, float tmin, float tmax, uint32_t flags,
                                            HitRec& best, uint32_t* stk, uint32_t* err, TravCounters& cnt,
                                            const Diag dg = Diag{ nullptr })
{
    best.t = tmax; best.hit = false; best.prim = 0; best.leaf = 0; best.inst = 0; best.U = 0.0f; best.V = 0.0f;
    best.ad = 1.0f;
    if (!TLAS) {      // the reference's scene: one identity instance, mask 1, flags 0 (RefractionDemo.cpp:324-334)
        trace_blas<STACK, STATS>(sc.blas0, O, D, tmin, flags, 0u, best, Stack32{ stk }, 0, err, cnt, dg);
        return;
    }
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(sc.pool_nodes);
    const Stack32 st{ stk };
    constexpr uint32_t NO_INST = 0xffffffffu;
    BoxRay br = box_ray(O, D, sc.scale);
    f3 Oc = O, Dc = D;                      // the ray in the space of the level being walked
    uint32_t cull = flags, cur = NO_INST;
    int sp = 0, floor = 0, node = 0;        // floor: stack level at which the current instance was entered
    uint32_t e = 0;
    for (;;) {
        while (node >= 0) {
            const float4* q = nodes + (uint32_t)node * 4u;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            if (STATS) cnt.nodes++;
            node = node_step<true>(br, q0, q1, q2, q3, tmin, best.t, st, sp, floor, STACK, e);
        }
        if (node == TRAV_DONE) {
            if (cur == NO_INST) break;
            cur = NO_INST; Oc = O; Dc = D; cull = flags; floor = 0;          // leave the instance
            br = box_ray(O, D, sc.scale);
            if (sp > 0) { --sp; node = st.pop(sp); continue; }
            break;
        }
        const uint32_t L = (uint32_t)~node;
        if (L < sc.n_pool_tris) {
            if (STATS) cnt.tris++;
            tri_test(sc.pool_tris, L, Oc, Dc, tmin, cull, cur, best);
            if (sp > floor) { --sp; node = st.pop(sp); } else node = TRAV_DONE;
        } else {
            const uint32_t ii = L - sc.n_pool_tris;
            const InstDev& in = sc.insts[ii];
            if (in.mask & 0xffu) {                                            // InstanceInclusionMask 0xff
                uint32_t f = flags;
                if (in.flags & 0x1u) f &= ~(CULL_BACK | CULL_FRONT);          // TRIANGLE_CULL_DISABLE
                else if (in.flags & 0x2u) {                                    // TRIANGLE_FRONT_COUNTERCLOCKWISE
                    if (f & CULL_BACK) f = (f & ~CULL_BACK) | CULL_FRONT;
                    else if (f & CULL_FRONT) f = (f & ~CULL_FRONT) | CULL_BACK;
                }
                cull = f; cur = ii; floor = sp;
                if (!in.identity) { Oc = xform_point(in.inv, O); Dc = xform_dir(in.inv, D); }
                br = box_ray(Oc, Dc, in.scale);
                node = (int)in.root;
            } else if (sp > 0) { --sp; node = st.pop(sp); } else node = TRAV_DONE;
        }
    }
    if (e) *err = 1u;
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
__device__ __forceinline__ f3 camera_ray_dir(const float* M, uint32_t x, uint32_t y, uint32_t W, uint32_t H)
{
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float sx = px / (float)W * 2.0f - 1.0f;
    float sy = py / (float)H * 2.0f - 1.0f;
    sy = -sy;
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
    if (TLAS && !sc.insts[h.inst].identity) {
        const float* w = sc.insts[h.inst].inv;
        Nr = mk3((w[0] * Nr.x + w[4] * Nr.y) + w[8] * Nr.z,
                 (w[1] * Nr.x + w[5] * Nr.y) + w[9] * Nr.z,
                 (w[2] * Nr.x + w[6] * Nr.y) + w[10] * Nr.z);
    }
    return normalize3(Nr);
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

} // namespace rr
