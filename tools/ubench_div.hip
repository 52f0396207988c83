// Is x / 3.14159f (RayTracing.hlsl:133-134, the Miss shader's two divisions: IEEE-correct, ~12 vector instructions each as hipcc
// expands them) equal, bit for bit, to   q = x * r;  q + fma(-q, 3.14159f, x) * r   with r = fl(1 / 3.14159f)
// (a multiply and two fmas)?  Checked over ALL 2^32 float bit patterns on the GPU; NaN results compare equal to NaN results.
// Result (MI355X, round 3, profiles/r03_div_pi_check.txt): 3 079 441 mismatches -- the two infinities and 3 079 439 inputs of magnitude up to
// 4.6e-33 (quotients in or next to the denormal range, where the residual fma loses bits) -- rr_atan2f can return such values for rays a hair off the +z axis, so the replacement is NOT bit-exact over
// the shader's inputs and was dropped; the divisions stay.
// Prints the mismatches over all inputs and over |x| <= 4 (what rr_atan2f / rr_acosf can hand to the division).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_div.hip -o ubench_div && ./ubench_div
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void k(unsigned long long* out, uint32_t* first)
{
    const float d = 3.14159f, r = 1.0f / 3.14159f;
    unsigned long long bad = 0, bad_dom = 0;
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) << 8;
    for (uint32_t i = 0; i < 256u; ++i) {
        const uint32_t u = base + i;
        const float x = __uint_as_float(u);
        const float want = x / d;
        const float q = x * r;
        const float got = fmaf(fmaf(-q, d, x), r, q);
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            ++bad;
            if ((u & 0x7fffffffu) < 0x7f800000u) atomicMax(&first[16], u & 0x7fffffffu);
            if (fabsf(x) <= 4.0f) { ++bad_dom; const uint32_t k = atomicAdd(&first[0], 1u); if (k < 15u) first[1 + k] = u; }
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (bad_dom) atomicAdd(&out[1], bad_dom);
}

int main()
{
    unsigned long long* d; uint32_t* f;
    if (hipMalloc(&d, 16) != hipSuccess || hipMalloc(&f, 68) != hipSuccess) return 2;
    hipMemset(d, 0, 16); hipMemset(f, 0, 68);
    k<<<65536, 256>>>(d, f);
    unsigned long long h[2]; uint32_t hf[17];
    if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hf, f, 68, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    printf("x / 3.14159f vs multiply + two fmas over all 2^32 inputs: %llu mismatches; over |x| <= 4: %llu mismatches\n", h[0], h[1]);
    { float x; memcpy(&x, &hf[16], 4); printf("largest finite |x| with a mismatch: %.9g (0x%08x)\n", x, hf[16]); }
    for (uint32_t i = 0; i < hf[0] && i < 15u; ++i) { float x; memcpy(&x, &hf[1 + i], 4); printf("  x = %.9g (0x%08x)\n", x, hf[1 + i]); }
    return 0;
}
