// L1 (TCP) request rate for per-lane divergent loads from a small table: 4 x dwordx4 (64 B node) versus
// 7 x dwordx2 (per-lane plane selection) per node visit.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l1.hip -o /tmp/ubench_l1 && /tmp/ubench_l1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ nodes, unsigned n_nodes, int iters, float* out)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    unsigned sel = (s >> 7) & 7u;                                  // per-lane "octant"
    const char* base = (const char*)nodes;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned node = (s >> 8) % n_nodes;
        if (MODE == 0) {
            const float4* p = nodes + (size_t)node * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else {
            unsigned o = node * 64u;
            unsigned ox = (sel & 1u) * 8u, oy = ((sel >> 1) & 1u) * 8u, oz = ((sel >> 2) & 1u) * 8u;
            float2 a = *(const float2*)(base + o + ox), b = *(const float2*)(base + o + (8u - ox));
            float2 c = *(const float2*)(base + o + 16u + oy), d = *(const float2*)(base + o + 16u + (8u - oy));
            float2 e = *(const float2*)(base + o + 32u + oz), f = *(const float2*)(base + o + 32u + (8u - oz));
            float2 g = *(const float2*)(base + o + 48u);
            acc += a.x + b.y + c.x + d.y + e.x + f.y + g.x;
        }
        s += __float_as_uint(acc) & 1u;                            // dependent chain like a traversal
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main()
{
    for (unsigned n_nodes : { 256u, 1024u, 4096u }) {
        std::vector<float4> h((size_t)n_nodes * 4, make_float4(1, 2, 3, 4));
        float4* d; float* o;
        const int blocks = 256 * 8, iters = 2000;
        hipMalloc(&d, h.size() * 16); hipMalloc(&o, (size_t)blocks * 256 * 4);
        hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, n_nodes, iters, o); else k<1><<<blocks, 256>>>(d, n_nodes, iters, o);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            double visits = (double)blocks * 256 * iters;
            printf("table %4u KB  %s: %.3f ms, %.1f G node-visits/s (lane level), %.2f ns per wave-visit per CU\n", n_nodes * 64 / 1024,
                   mode == 0 ? "4 x dwordx4" : "7 x dwordx2", ms, visits / ms / 1e6, ms * 1e6 / (visits / 64 / 256));
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
