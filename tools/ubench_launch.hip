// micro-benchmark: what does an (almost) empty grid of the render kernel's shape cost on MI355X?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o /tmp/ubench_launch && /tmp/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_empty(unsigned* out, int work)
{
    extern __shared__ unsigned lds[];
    unsigned v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = v * 1664525u + 1013904223u;
    if (work < 0) lds[threadIdx.x] = v;
    if (v == 0x12345u) out[blockIdx.x] = v + lds[0];
}
__global__ __launch_bounds__(256) void k_store(unsigned* out)
{
    out[blockIdx.x * 256 + threadIdx.x] = threadIdx.x;
}
int main()
{
    unsigned* d; hipMalloc(&d, 64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct { int blocks, threads, lds, work; } cfg[] = {
        {8160, 256, 0, 0}, {8160, 256, 32768, 0}, {8160, 256, 65536, 0}, {32640, 64, 8192, 0}, {2040, 1024, 131072, 0},
        {8160, 256, 32768, 100}, {8160, 256, 32768, 1000}, {8160, 256, 0, 1000}, {1280, 256, 32768, 1000}, {81600, 256, 32768, 0}};
    for (auto& c : cfg) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_empty, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, c.work);
        hipEventRecord(a, 0);
        const int N = 50;
        for (int it = 0; it < N; ++it) hipLaunchKernelGGL(k_empty, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, c.work);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("blocks %6d threads %4d lds %6d work %5d : %8.2f us per launch\n", c.blocks, c.threads, c.lds, c.work, ms / N * 1e3);
    }
    hipEventRecord(a, 0);
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(k_store, dim3(8100), dim3(256), 0, 0, d);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("store 8100x256 u32: %8.2f us per launch\n", ms / 50 * 1e3);
    return 0;
}
