// rr_bvh_build.hip -- BuildRaytracingAccelerationStructure stand-in (RefractionDemo.cpp:277-356).
//
// The reference hands Vertices/Indices (Mesh.cpp:39-53) to the driver and gets an opaque BVH back.
// Here the BVH is a Morton-sorted LBVH built on the GPU:
//   primitive boxes + scene bounds -> 30-bit Morton code of the box centre, made unique by the
//   primitive index (64-bit key) -> bitonic sort (LDS-resident for <= 2048-key runs) -> Karras
//   radix-tree hierarchy -> bottom-up box refit with one arrival counter per internal node ->
//   pack into 64-byte nodes that carry BOTH child boxes, so traversal fetches one record per visit.
// The same builder makes the TLAS (primitives = instance world boxes).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include "rr_launch.h"

namespace rr {

// order-preserving float <-> uint for atomicMin/atomicMax
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

__device__ __forceinline__ void reduce_scene_box(const float lo[3], const float hi[3], bool valid, uint32_t* scene_box)
{
    const float inf = __builtin_huge_valf();
    for (int k = 0; k < 3; ++k) {
        float l = wave_min(valid ? lo[k] : inf), h = wave_max(valid ? hi[k] : -inf);
        if ((threadIdx.x & 63u) == 0u) {
            atomicMin(&scene_box[k], f2ord(l));
            atomicMax(&scene_box[3 + k], f2ord(h));
        }
    }
}

__global__ __launch_bounds__(256) void k_init_build(BuildBuffers b)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 3u) { b.scene_box[i] = 0xffffffffu; b.scene_box[3 + i] = 0u; }
    if (i == 0u) *b.depth = 0u;
    if (b.n > 1u && i < b.n - 1u) b.visit[i] = 0u;
}

// Vertices[Indices[3p+k]].position, 32-byte stride (Mesh.cpp:44-45)
__global__ __launch_bounds__(256) void k_tri_boxes(const float* __restrict__ verts, const uint32_t* __restrict__ idx,
                                                   uint32_t n, BuildBuffers b)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool valid = p < n;
    float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    if (valid) {
        const float* A = verts + (size_t)idx[3 * p + 0] * 8;
        const float* B = verts + (size_t)idx[3 * p + 1] * 8;
        const float* C = verts + (size_t)idx[3 * p + 2] * 8;
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(A[k], fminf(B[k], C[k]));
            hi[k] = fmaxf(A[k], fmaxf(B[k], C[k]));
            b.prim_box[(size_t)p * 6 + k] = lo[k];
            b.prim_box[(size_t)p * 6 + 3 + k] = hi[k];
        }
    }
    reduce_scene_box(lo, hi, valid, b.scene_box);
}

// instance world box = box of the 8 transformed BLAS-bounds corners (object->world 3x4)
__global__ __launch_bounds__(256) void k_inst_boxes(const InstDev* __restrict__ insts, const float* __restrict__ xforms,
                                                    const float* __restrict__ blas_bounds, uint32_t n, BuildBuffers b)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool valid = p < n;
    const float inf = __builtin_huge_valf();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    if (valid) {
        const float* m = xforms + (size_t)p * 12;
        const float* bb = blas_bounds + (size_t)p * 6;
        const bool ident = insts[p].identity != 0u;
        for (int c = 0; c < 8; ++c) {
            float x = (c & 1) ? bb[3] : bb[0], y = (c & 2) ? bb[4] : bb[1], z = (c & 4) ? bb[5] : bb[2];
            float wx = x, wy = y, wz = z;
            if (!ident) {
                wx = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
                wy = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
                wz = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
            }
            lo[0] = fminf(lo[0], wx); lo[1] = fminf(lo[1], wy); lo[2] = fminf(lo[2], wz);
            hi[0] = fmaxf(hi[0], wx); hi[1] = fmaxf(hi[1], wy); hi[2] = fmaxf(hi[2], wz);
        }
        // the corners were rounded by the transform: grow the box a little so it still contains the instance
        for (int k = 0; k < 3; ++k) {
            lo[k] -= fmaf(fabsf(lo[k]), 1e-5f, 1e-7f); hi[k] += fmaf(fabsf(hi[k]), 1e-5f, 1e-7f);
            b.prim_box[(size_t)p * 6 + k] = lo[k]; b.prim_box[(size_t)p * 6 + 3 + k] = hi[k];
        }
    }
    reduce_scene_box(lo, hi, valid, b.scene_box);
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_morton_keys(BuildBuffers b)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= b.n_pad) return;
    if (p >= b.n) { b.keys[p] = ~0ull; return; }
    uint32_t code = 0;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float slo = ord2f(b.scene_box[k]), shi = ord2f(b.scene_box[3 + k]);
        float c = 0.5f * (b.prim_box[(size_t)p * 6 + k] + b.prim_box[(size_t)p * 6 + 3 + k]);
        float ext = shi - slo;
        float t = ext > 0.0f ? (c - slo) / ext : 0.0f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)t;
    }
    code = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
    b.keys[p] = ((unsigned long long)code << 32) | (unsigned long long)p;
}

// ---- bitonic sort of 64-bit keys (ascending), n_pad a power of two ---------------------------
constexpr uint32_t SORT_CH = 2048;      // keys per workgroup run (16 KB of LDS), 1024 threads

__device__ __forceinline__ void cswap(unsigned long long& a, unsigned long long& b, bool asc)
{
    if ((a > b) == asc) { unsigned long long t = a; a = b; b = t; }
}

// all (k, j) stages with k <= min(SORT_CH, N) inside LDS
__global__ __launch_bounds__(1024) void k_bitonic_block(unsigned long long* keys, uint32_t N)
{
    __shared__ unsigned long long s[SORT_CH];
    const uint32_t base = blockIdx.x * SORT_CH, t = threadIdx.x;
    const uint32_t cnt = N < SORT_CH ? N : SORT_CH;
    if (t < cnt) s[t] = keys[base + t];
    if (t + 1024u < cnt) s[t + 1024u] = keys[base + t + 1024u];
    __syncthreads();
    for (uint32_t k = 2; k <= cnt; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            uint32_t i = 2u * t - (t & (j - 1u));           // lower index of pair t at stride j
            if (i + j < cnt) {
                bool asc = ((base + i) & k) == 0u;
                unsigned long long a = s[i], c = s[i + j];
                cswap(a, c, asc);
                s[i] = a; s[i + j] = c;
            }
            __syncthreads();
        }
    }
    if (t < cnt) keys[base + t] = s[t];
    if (t + 1024u < cnt) keys[base + t + 1024u] = s[t + 1024u];
}

// one global stage (j >= SORT_CH)
__global__ __launch_bounds__(256) void k_bitonic_global(unsigned long long* keys, uint32_t N, uint32_t k, uint32_t j)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= N / 2u) return;
    const uint32_t i = 2u * t - (t & (j - 1u));
    const bool asc = (i & k) == 0u;
    unsigned long long a = keys[i], c = keys[i + j];
    cswap(a, c, asc);
    keys[i] = a; keys[i + j] = c;
}

// the stages j = SORT_CH/2 .. 1 of an outer k > SORT_CH, inside LDS
__global__ __launch_bounds__(1024) void k_bitonic_tail(unsigned long long* keys, uint32_t N, uint32_t k)
{
    __shared__ unsigned long long s[SORT_CH];
    const uint32_t base = blockIdx.x * SORT_CH, t = threadIdx.x;
    s[t] = keys[base + t];
    s[t + 1024u] = keys[base + t + 1024u];
    __syncthreads();
    for (uint32_t j = SORT_CH >> 1; j > 0; j >>= 1) {
        uint32_t i = 2u * t - (t & (j - 1u));
        bool asc = ((base + i) & k) == 0u;
        unsigned long long a = s[i], c = s[i + j];
        cswap(a, c, asc);
        s[i] = a; s[i + j] = c;
        __syncthreads();
    }
    keys[base + t] = s[t];
    keys[base + t + 1024u] = s[t + 1024u];
}

// ---- Karras 2012: one thread per internal node ------------------------------------------------
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));       // keys are unique
}

__global__ __launch_bounds__(256) void k_karras(BuildBuffers b)
{
    const int n = (int)b.n;
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const unsigned long long* keys = b.keys;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == gamma) ? ~gamma : gamma;              // ~leaf (sorted position) or internal index
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    b.child[2 * i] = left;
    b.child[2 * i + 1] = right;
    b.parent[left >= 0 ? left : (n - 1) + ~left] = i;
    b.parent[right >= 0 ? right : (n - 1) + ~right] = i;
    if (i == 0) b.parent[0] = -1;
}

// ---- PLOC (Meister & Bittner 2018): agglomerative clustering in Morton order -------------------
// The D3D12 build flag the reference passes is PREFER_FAST_TRACE (RefractionDemo.cpp:286): for meshes of up
// to PLOC_MAX_PRIMS triangles the Karras radix tree is replaced by this bottom-up builder, which merges
// mutually-nearest neighbours (surface area of the merged box, search radius PLOC_RADIUS in the Morton
// order) until one cluster is left.  One workgroup; every step is deterministic (ids come from scans,
// not atomics).  Output conventions are those of k_karras + k_refit: child[], parent[], node_box[].
#ifndef RR_PLOC_RADIUS
#define RR_PLOC_RADIUS 16
#endif
constexpr int PLOC_RADIUS = RR_PLOC_RADIUS;

__device__ __forceinline__ float merged_area(const float* a, const float* c)
{
    const float dx = fmaxf(a[3], c[3]) - fminf(a[0], c[0]);
    const float dy = fmaxf(a[4], c[4]) - fminf(a[1], c[1]);
    const float dz = fmaxf(a[5], c[5]) - fminf(a[2], c[2]);
    return dx * dy + dy * dz + dz * dx;
}

// exclusive scan of one packed u32 per thread over the 1024-thread workgroup; returns the total
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t& excl, uint32_t* lds /* 16 words */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += t; }
    if (lane == 63u) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < 16u; ++w) { const uint32_t t = lds[w]; if (w < wave) base += t; total += t; }
    __syncthreads();
    excl = base + inc - v;
    return total;
}

__global__ __launch_bounds__(1024) void k_ploc(BuildBuffers b)
{
    __shared__ uint32_t scan_lds[16];
    const int n = (int)b.n;
    const int tid = (int)threadIdx.x;
    uint32_t* A = b.ploc;
    uint32_t* B = b.ploc + b.n;
    int* NN = reinterpret_cast<int*>(b.visit);
    for (int i = tid; i < n; i += 1024) {
        const uint32_t prim = (uint32_t)(b.keys[i] & 0xffffffffull);
        for (int k = 0; k < 6; ++k) b.node_box[(size_t)(n - 1 + i) * 6 + k] = b.prim_box[(size_t)prim * 6 + k];
        A[i] = (uint32_t)(n - 1 + i);
    }
    __syncthreads();
    int m = n;
    int next_id = n - 2;                              // internal ids are handed out downwards: the last merge is node 0
    bool force_pair = false;
    while (m > 1) {
        // 1. nearest neighbour of every cluster within the radius
        for (int i = tid; i < m; i += 1024) {
            const float* bu = b.node_box + (size_t)A[i] * 6;
            float best = __builtin_huge_valf();
            int bj = -1;
            const int j0 = i - PLOC_RADIUS < 0 ? 0 : i - PLOC_RADIUS, j1 = i + PLOC_RADIUS > m - 1 ? m - 1 : i + PLOC_RADIUS;
            for (int j = j0; j <= j1; ++j) {
                if (j == i) continue;
                const float a = merged_area(bu, b.node_box + (size_t)A[j] * 6);
                if (bj < 0 || a < best) { best = a; bj = j; }     // total: the first candidate stands unless a smaller area turns up (areas that overflow or are NaN compare false)
            }
            NN[i] = bj;                                            // m >= 2: there is always a candidate
        }
        __syncthreads();
        if (force_pair && tid == 0) { NN[0] = 1; NN[1] = 0; }      // the round before merged nothing: clusters 0 and 1 merge now
        __syncthreads();
        // 2. mutual pairs merge (the lower position keeps the new cluster), everything else survives as is
        const int chunk = (m + 1023) / 1024;
        const int c0 = tid * chunk, c1 = c0 + chunk < m ? c0 + chunk : m;
        uint32_t cnt = 0;                             // low 16 bits: clusters kept, high 16: merges
        for (int i = c0; i < c1; ++i) {
            const int j = NN[i];
            const bool mutual = NN[j] == i;
            if (!mutual || i < j) cnt += 1u;
            if (mutual && i < j) cnt += 0x10000u;
        }
        uint32_t excl;
        const uint32_t total = block_scan_1024(cnt, excl, scan_lds);
        uint32_t pos = excl & 0xffffu, mrg = excl >> 16;
        for (int i = c0; i < c1; ++i) {
            const int j = NN[i];
            const bool mutual = NN[j] == i;
            if (mutual && i < j) {
                const int id = next_id - (int)mrg;
                const uint32_t ul = A[i], ur = A[j];
                const float* bl = b.node_box + (size_t)ul * 6;
                const float* br = b.node_box + (size_t)ur * 6;
                for (int k = 0; k < 3; ++k) {
                    b.node_box[(size_t)id * 6 + k] = fminf(bl[k], br[k]);
                    b.node_box[(size_t)id * 6 + 3 + k] = fmaxf(bl[3 + k], br[3 + k]);
                }
                b.child[2 * id] = ul >= (uint32_t)(n - 1) ? ~(int)(ul - (uint32_t)(n - 1)) : (int)ul;
                b.child[2 * id + 1] = ur >= (uint32_t)(n - 1) ? ~(int)(ur - (uint32_t)(n - 1)) : (int)ur;
                b.parent[ul] = id;
                b.parent[ur] = id;
                B[pos++] = (uint32_t)id;
                ++mrg;
            } else if (!mutual) {
                B[pos++] = A[i];
            }
        }
        next_id -= (int)(total >> 16);
        m = (int)(total & 0xffffu);
        force_pair = (total >> 16) == 0u;             // no mutual pair (possible only among equal or unordered areas): never loop without progress
        __syncthreads();
        uint32_t* t = A; A = B; B = t;
        __syncthreads();
    }
    if (tid == 0) b.parent[0] = -1;
}

// ---- bottom-up refit: the second thread to arrive at a node owns it ---------------------------
__global__ __launch_bounds__(256) void k_refit(BuildBuffers b)
{
    const int n = (int)b.n;
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const uint32_t prim = (uint32_t)(b.keys[i] & 0xffffffffull);
    float box[6];
    for (int k = 0; k < 6; ++k) { box[k] = b.prim_box[(size_t)prim * 6 + k]; b.node_box[(size_t)(n - 1 + i) * 6 + k] = box[k]; }
    if (n == 1) return;
    int cur = b.parent[n - 1 + i];
    uint32_t depth = 1;
    while (cur >= 0) {
        // publish this subtree's box, then take a ticket (agent scope: another CU/XCD may own the sibling)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&b.visit[cur], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int l = b.child[2 * cur], r = b.child[2 * cur + 1];
        const size_t il = (size_t)(l >= 0 ? l : (n - 1) + ~l), ir = (size_t)(r >= 0 ? r : (n - 1) + ~r);
        for (int k = 0; k < 3; ++k) {
            box[k] = fminf(__builtin_nontemporal_load(&b.node_box[il * 6 + k]), __builtin_nontemporal_load(&b.node_box[ir * 6 + k]));
            box[3 + k] = fmaxf(__builtin_nontemporal_load(&b.node_box[il * 6 + 3 + k]), __builtin_nontemporal_load(&b.node_box[ir * 6 + 3 + k]));
        }
        for (int k = 0; k < 6; ++k) b.node_box[(size_t)cur * 6 + k] = box[k];
        cur = b.parent[cur];
        ++depth;
    }
    (void)depth;
}

__global__ __launch_bounds__(256) void k_depth(BuildBuffers b)
{
    const int n = (int)b.n;
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    uint32_t d = 0;
    if (i < n) {
        d = 1;
        if (n > 1) for (int cur = b.parent[n - 1 + i]; cur >= 0; cur = b.parent[cur]) ++d;
    }
    for (int o = 32; o > 0; o >>= 1) { uint32_t v = __shfl_xor(d, o, 64); d = v > d ? v : d; }
    if ((threadIdx.x & 63u) == 0u && d) atomicMax(b.depth, d);
}

__global__ __launch_bounds__(256) void k_pack_nodes(BuildBuffers b)
{
    const int n = (int)b.n;
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    const float inf = __builtin_huge_valf();
    if (n == 1) {
        if (i == 0) {                                   // single primitive: node 0 = {leaf 0, empty}
            BvhNode nd;
            nd.lox[0] = b.node_box[0]; nd.loy[0] = b.node_box[1]; nd.loz[0] = b.node_box[2];
            nd.hix[0] = b.node_box[3]; nd.hiy[0] = b.node_box[4]; nd.hiz[0] = b.node_box[5];
            nd.lox[1] = nd.loy[1] = nd.loz[1] = inf; nd.hix[1] = nd.hiy[1] = nd.hiz[1] = -inf;
            const int leaf0 = b.leaf_ref_prim ? ~(int)b.leaf_base : ~0;
            nd.c[0] = leaf0; nd.c[1] = leaf0; nd.pad[0] = 0; nd.pad[1] = 0;
            b.nodes[0] = nd;
        }
        return;
    }
    if (i >= n - 1) return;
    int l = b.child[2 * i], r = b.child[2 * i + 1];
    const size_t il = (size_t)(l >= 0 ? l : (n - 1) + ~l), ir = (size_t)(r >= 0 ? r : (n - 1) + ~r);
    if (b.leaf_ref_prim) {
        if (l < 0) l = ~(int)(b.leaf_base + (uint32_t)(b.keys[~l] & 0xffffffffull));
        if (r < 0) r = ~(int)(b.leaf_base + (uint32_t)(b.keys[~r] & 0xffffffffull));
    }
    BvhNode nd;
    const float* bl = b.node_box + il * 6;
    const float* br = b.node_box + ir * 6;
    nd.lox[0] = bl[0]; nd.loy[0] = bl[1]; nd.loz[0] = bl[2]; nd.hix[0] = bl[3]; nd.hiy[0] = bl[4]; nd.hiz[0] = bl[5];
    nd.lox[1] = br[0]; nd.loy[1] = br[1]; nd.loz[1] = br[2]; nd.hix[1] = br[3]; nd.hiy[1] = br[4]; nd.hiz[1] = br[5];
    nd.c[0] = l; nd.c[1] = r; nd.pad[0] = 0; nd.pad[1] = 0;
    b.nodes[i] = nd;
}

// leaf-ordered triangle + normal records (Vertices[Indices[3*prim+k]], RayTracing.hlsl:83-85)
__global__ __launch_bounds__(256) void k_pack_tris(const float* __restrict__ verts, const uint32_t* __restrict__ idx,
                                                   BuildBuffers b, TriRec* tris, NrmRec* nrms)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= b.n) return;
    const uint32_t prim = (uint32_t)(b.keys[i] & 0xffffffffull);
    const float* A = verts + (size_t)idx[3 * prim + 0] * 8;
    const float* B = verts + (size_t)idx[3 * prim + 1] * 8;
    const float* C = verts + (size_t)idx[3 * prim + 2] * 8;
    TriRec t;
    NrmRec m;
    for (int k = 0; k < 3; ++k) {
        t.v0[k] = A[k]; t.e1[k] = B[k] - A[k]; t.e2[k] = C[k] - A[k];
        m.nA[k] = A[3 + k]; m.nB[k] = B[3 + k]; m.nC[k] = C[3 + k];
    }
    t.prim = prim; t.pad1 = 0; t.pad2 = 0;
    m.pad0 = 0; m.pad1 = 0; m.pad2 = 0;
    tris[i] = t;
    nrms[i] = m;
}

// fp32 node -> traversal node with fp16 planes on the grid g; child refs optionally rebased into the scene pool.
// fp16 planes in cell units around the centre of the grid (|x| <= 32768 < 65504): directed rounding outward, after a
// guard for the fp32 rounding of (v - org)/cell.  The spacing of fp16 grows with |x| (1 cell below 2048 cells from the
// centre, 16 cells at the faces of the bounds: 2.4e-4 of the extent), always outward, which is all the box test needs.
__device__ __forceinline__ uint32_t q_lo(float v, float org, float cell)
{
    const float x = (v - org) / cell - 0.05f;
    return (uint32_t)__half_as_ushort(__float2half_rd(x));
}
__device__ __forceinline__ uint32_t q_hi(float v, float org, float cell)
{
    const float x = (v - org) / cell + 0.05f;
    return (uint32_t)__half_as_ushort(__float2half_ru(x));
}
__global__ __launch_bounds__(256) void k_quantize_nodes(QNode* __restrict__ dst, const BvhNode* __restrict__ src, uint32_t n, QGrid g,
                                                        uint32_t node_off, uint32_t tri_off)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const BvhNode s = src[i];
    QNode q;
    q.lox = q_lo(s.lox[0], g.org[0], g.cell[0]) | (q_lo(s.lox[1], g.org[0], g.cell[0]) << 16);
    q.loy = q_lo(s.loy[0], g.org[1], g.cell[1]) | (q_lo(s.loy[1], g.org[1], g.cell[1]) << 16);
    q.loz = q_lo(s.loz[0], g.org[2], g.cell[2]) | (q_lo(s.loz[1], g.org[2], g.cell[2]) << 16);
    q.hix = q_hi(s.hix[0], g.org[0], g.cell[0]) | (q_hi(s.hix[1], g.org[0], g.cell[0]) << 16);
    q.hiy = q_hi(s.hiy[0], g.org[1], g.cell[1]) | (q_hi(s.hiy[1], g.org[1], g.cell[1]) << 16);
    q.hiz = q_hi(s.hiz[0], g.org[2], g.cell[2]) | (q_hi(s.hiz[1], g.org[2], g.cell[2]) << 16);
    // internal refs become BYTE offsets into the (pooled) QNode array; leaf refs stay ~index
    for (int k = 0; k < 2; ++k) q.c[k] = s.c[k] >= 0 ? (int)(((uint32_t)s.c[k] + node_off) * (uint32_t)sizeof(QNode)) : ~(int)((uint32_t)~s.c[k] + tri_off);
    dst[i] = q;
}

__global__ __launch_bounds__(256) void k_env_pad(const float* __restrict__ rgb, float4* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = make_float4(rgb[3 * (size_t)i], rgb[3 * (size_t)i + 1], rgb[3 * (size_t)i + 2], 1.0f);
}

// ------------------------------------------------------------------------------------ launchers
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

hipError_t launch_tri_setup(const void* verts, const uint32_t* idx, uint32_t n_tris, const BuildBuffers& b, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_build, dim3(cdiv(b.n > 3u ? b.n : 3u, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_tri_boxes, dim3(cdiv(n_tris, 256u)), dim3(256), 0, s, (const float*)verts, idx, n_tris, b);
    return hipGetLastError();
}

hipError_t launch_inst_setup(const InstDev* insts, const float* xforms_and_bounds, uint32_t n, const BuildBuffers& b, hipStream_t s)
{
    // xforms_and_bounds: n*12 floats of object->world transforms followed by n*6 floats of BLAS bounds
    hipLaunchKernelGGL(k_init_build, dim3(cdiv(b.n > 3u ? b.n : 3u, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_inst_boxes, dim3(cdiv(n, 256u)), dim3(256), 0, s, insts, xforms_and_bounds,
                       xforms_and_bounds + (size_t)n * 12, n, b);
    return hipGetLastError();
}

hipError_t launch_lbvh(const BuildBuffers& b, hipStream_t s)
{
    const uint32_t N = b.n_pad;
    hipLaunchKernelGGL(k_morton_keys, dim3(cdiv(N, 256u)), dim3(256), 0, s, b);
    if (N > 1u) {
        hipLaunchKernelGGL(k_bitonic_block, dim3(N <= SORT_CH ? 1u : N / SORT_CH), dim3(1024), 0, s, b.keys, N);
        for (uint32_t k = SORT_CH * 2u; k <= N && k != 0u; k <<= 1) {
            for (uint32_t j = k >> 1; j >= SORT_CH; j >>= 1)
                hipLaunchKernelGGL(k_bitonic_global, dim3(cdiv(N / 2u, 256u)), dim3(256), 0, s, b.keys, N, k, j);
            hipLaunchKernelGGL(k_bitonic_tail, dim3(N / SORT_CH), dim3(1024), 0, s, b.keys, N, k);
        }
    }
    if (b.n > 1u) hipLaunchKernelGGL(k_karras, dim3(cdiv(b.n - 1u, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_refit, dim3(cdiv(b.n, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_depth, dim3(cdiv(b.n, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_pack_nodes, dim3(cdiv(b.n > 1u ? b.n - 1u : 1u, 256u)), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_ploc(const BuildBuffers& b, hipStream_t s)
{
    // keys must already be sorted (launch_lbvh's first half); builds child/parent/node_box, then depth + pack
    const uint32_t N = b.n_pad;
    hipLaunchKernelGGL(k_morton_keys, dim3(cdiv(N, 256u)), dim3(256), 0, s, b);
    if (N > 1u) {
        hipLaunchKernelGGL(k_bitonic_block, dim3(N <= SORT_CH ? 1u : N / SORT_CH), dim3(1024), 0, s, b.keys, N);
        for (uint32_t k = SORT_CH * 2u; k <= N && k != 0u; k <<= 1) {
            for (uint32_t j = k >> 1; j >= SORT_CH; j >>= 1)
                hipLaunchKernelGGL(k_bitonic_global, dim3(cdiv(N / 2u, 256u)), dim3(256), 0, s, b.keys, N, k, j);
            hipLaunchKernelGGL(k_bitonic_tail, dim3(N / SORT_CH), dim3(1024), 0, s, b.keys, N, k);
        }
    }
    hipLaunchKernelGGL(k_ploc, dim3(1), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(k_depth, dim3(cdiv(b.n, 256u)), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_pack_nodes, dim3(cdiv(b.n > 1u ? b.n - 1u : 1u, 256u)), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_pack_tris(const void* verts, const uint32_t* idx, const BuildBuffers& b, TriRec* tris, NrmRec* nrms,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_tris, dim3(cdiv(b.n, 256u)), dim3(256), 0, s, (const float*)verts, idx, b, tris, nrms);
    return hipGetLastError();
}

hipError_t launch_quantize_nodes(QNode* dst, const BvhNode* src, uint32_t n_nodes, const QGrid& g, uint32_t node_off, uint32_t tri_off,
                                 hipStream_t s)
{
    if (n_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize_nodes, dim3(cdiv(n_nodes, 256u)), dim3(256), 0, s, dst, src, n_nodes, g, node_off, tri_off);
    return hipGetLastError();
}

hipError_t launch_env_pad(const float* rgb, float4* out, uint32_t n_texels, hipStream_t s)
{
    hipLaunchKernelGGL(k_env_pad, dim3(cdiv(n_texels, 256u)), dim3(256), 0, s, rgb, out, n_texels);
    return hipGetLastError();
}

} // namespace rr
