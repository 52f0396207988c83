// Where should a 32-byte BVH node come from?  Per-lane dependent fetches of 32 B (two 16-byte requests, as a QNode visit)
// from a small table (the whole monkey.obj / sphere.obj BVH is 24-31 KB) through the vector L1 (global_load_dwordx4 x2)
// versus from a copy in LDS (ds_read_b128 x2), at 8 waves per SIMD, with the lanes of a wave fully divergent, drawing from
// 8 nodes, or all on one node, and with 0 or 32 VALU instructions of "slab test" per visit.
// Prints shader cycles per wave-visit per CU: the budget the CU's four SIMDs share.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench_nodefetch.hip -o /tmp/ubench_nodefetch && /tmp/ubench_nodefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool LDS, int SPREAD, int FILL>       // SPREAD: 0 = every lane its own random node, 1 = 8 nodes per wave, 2 = one node per wave
__global__ void __launch_bounds__(1024, 8) k(const uint4* __restrict__ nodes, unsigned n_nodes, int iters, float* out, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) uint4 tab[];
    if (LDS) {
        for (unsigned i = threadIdx.x; i < n_nodes * 2; i += blockDim.x) tab[i] = nodes[i];
        __syncthreads();
    }
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned s = (SPREAD == 0 ? (blockIdx.x * blockDim.x + threadIdx.x) : SPREAD == 1 ? wave * 8u + (lane & 7u) : wave) * 2654435761u + 12345u;
    float acc = 0.f, f0 = lane, f1 = 1.0001f, f2 = 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned node = (s >> 8) % n_nodes;
        uint4 a, b;
        if (LDS) { a = tab[node * 2]; b = tab[node * 2 + 1]; }
        else { a = nodes[node * 2]; b = nodes[node * 2 + 1]; }
        acc += __uint_as_float(a.x) + __uint_as_float(b.w);
#pragma unroll
        for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
        s += (__float_as_uint(acc) & 1u) + (a.y & 1u);                 // the next address depends on the data, as in a traversal
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + f0;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <bool LDS, int SPREAD, int FILL>
static void run(const uint4* d, unsigned n_nodes, float* o, unsigned long long* c)
{
    // global: 8 blocks of 256 per CU; LDS: 2 blocks of 1024 per CU sharing one table copy each -> 8 waves per SIMD either way
    const int threads = LDS ? 1024 : 256, blocks = LDS ? 512 : 2048, iters = 1000;
    const size_t lds = LDS ? (size_t)n_nodes * 32 : 19968;
    hipFuncSetAttribute((const void*)k<LDS, SPREAD, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<LDS, SPREAD, FILL><<<blocks, threads, lds>>>(d, n_nodes, iters, o, c);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), c, (size_t)waves * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double cyc_per_wave = sum / waves;                         // each wave's own loop time, in shader cycles
    const double per_cu = cyc_per_wave / iters / 32.0;               // 32 waves share a CU's L1 / LDS
    printf("%-6s %-9s fill %2d : %.3f ms, %.0f cycles per visit per wave, %.1f cycles per wave-visit per CU (%.1f per SIMD)\n",
           LDS ? "LDS" : "L1", SPREAD == 0 ? "divergent" : SPREAD == 1 ? "8 nodes" : "1 node", FILL, ms, cyc_per_wave / iters, per_cu, per_cu * 4);
}

int main()
{
    const unsigned n_nodes = 966;
    std::vector<uint4> h((size_t)n_nodes * 2, make_uint4(1, 2, 3, 4));
    uint4* d; float* o; unsigned long long* c;
    hipMalloc(&d, h.size() * 16); hipMalloc(&o, (size_t)2048 * 256 * 4); hipMalloc(&c, (size_t)8192 * 8);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    run<false, 0, 0>(d, n_nodes, o, c);  run<true, 0, 0>(d, n_nodes, o, c);
    run<false, 1, 0>(d, n_nodes, o, c);  run<true, 1, 0>(d, n_nodes, o, c);
    run<false, 2, 0>(d, n_nodes, o, c);  run<true, 2, 0>(d, n_nodes, o, c);
    run<false, 0, 32>(d, n_nodes, o, c); run<true, 0, 32>(d, n_nodes, o, c);
    run<false, 1, 32>(d, n_nodes, o, c); run<true, 1, 32>(d, n_nodes, o, c);
    run<false, 2, 32>(d, n_nodes, o, c); run<true, 2, 32>(d, n_nodes, o, c);
    return 0;
}
