// How many 256-thread blocks of a given dynamic-LDS size are resident per CU on MI355X?
// Each block waits ~20 us (wall clock); time(grid = 256 CUs * k) steps up when k exceeds residency.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_wait(unsigned* out, unsigned long long ticks)
{
    extern __shared__ unsigned lds[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0x7fffffffffffull) { lds[threadIdx.x] = 1; out[0] = lds[0]; }
}
int main()
{
    unsigned* d; (void)hipMalloc(&d, 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    int ldss[] = {0, 8192, 16384, 24576, 26624, 27648, 28672, 30720, 31744, 32256, 32768, 40960, 65536, 81920};
    for (int lds : ldss) {
        (void)hipFuncSetAttribute((const void*)k_wait, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        printf("lds %6d:", lds);
        for (int k = 1; k <= 10; ++k) {
            hipLaunchKernelGGL(k_wait, dim3(256 * k), dim3(256), lds, 0, d, 2000ull);   // 100 MHz ticks: 20 us
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_wait, dim3(256 * k), dim3(256), lds, 0, d, 2000ull);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf(" k%d=%.0fus", k, ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
