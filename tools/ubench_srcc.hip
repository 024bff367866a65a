// ubench_srcc.hip -- what does the srcC operand of v_mfma_i32_16x16x64_i8 cost on a VALU-issue-bound SIMD?  The 64-antenna kernels
// issue INDEPENDENT MFMAs (one per product, srcC = a seed tuple in VGPRs, 16 different destinations per row tile) between 7-17 VALU
// ops each.  Modes: srcC = the same VGPR tuple for all | srcC = inline 0 | srcC = the destination itself (a running accumulator,
// 16 of them round robin) | one accumulator for all 16 (dependent chain).  K independent v_fma_f32 per MFMA, pinned 1 : K.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_srcc tools/ubench_srcc.hip && tools/ubench_srcc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ __launch_bounds__(256) void srcc_kernel(const v4i* __restrict__ src, float* __restrict__ sink, int iters)
{
    v4i a[2], b[2], d[16];
    a[0] = src[threadIdx.x], a[1] = src[256 + threadIdx.x], b[0] = src[512 + threadIdx.x], b[1] = src[768 + threadIdx.x];
    v4i kc = src[1024 + threadIdx.x];
    asm volatile("" : "+v"(kc));
#pragma unroll
    for (int t = 0; t < 16; t++) d[t] = v4i{0, 0, 0, 0};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 0.5f;
    for (int it = 0; it < iters; it++) {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));   // (not loop-invariant: no hoisting, no CSE across iterations)
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if constexpr (MODE <= 1) asm volatile("" : "+v"(a[t & 1]));       // ... nor inside one
            if constexpr (MODE == 4) {   // 8 chains of 2, chain by chain: a seeded MFMA, then one that continues its accumulator (the general kernel's re / im rows)
                if ((t & 1) == 0) asm volatile("" : "+v"(a[0]));
                d[t >> 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1], b[t & 1], (t & 1) ? d[t >> 1] : kc, 0, 0, 0);
            }
            if constexpr (MODE == 5) {   // the same 8 chains, all first MFMAs before the second ones (the compiler's order)
                if (t < 8) asm volatile("" : "+v"(a[0]));
                d[t & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t >> 3], b[t >> 3], (t >> 3) ? d[t & 7] : kc, 0, 0, 0);
            }
            if constexpr (MODE == 0) d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1], b[(t >> 1) & 1], kc, 0, 0, 0);
            if constexpr (MODE == 1) d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1], b[(t >> 1) & 1], v4i{0, 0, 0, 0}, 0, 0, 0);
            if constexpr (MODE == 2) d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1], b[(t >> 1) & 1], d[t], 0, 0, 0);
            if constexpr (MODE == 3) d[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1], b[(t >> 1) & 1], d[0], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; k++) f[(t * K + k) & 7] = __builtin_fmaf(f[(t * K + k) & 7], m, c);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) asm volatile("" : "+v"(d[t]));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += f[i];
#pragma unroll
    for (int t = 0; t < 16; t++) s += (float)(d[t][0] + d[t][3]);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int K>
void run(const v4i* d_src, float* d_sink, int n_cus, double clk_ghz)
{
    const char* names[] = {"srcC = one VGPR tuple, 16 dsts", "srcC = inline 0, 16 dsts     ", "srcC = dst, 16 accumulators  ", "srcC = dst, 1 accumulator    ",
                           "8 chains of 2, chain by chain", "8 chains of 2, firsts first  "};
    for (int wps : {2, 4}) {
        const int grid = n_cus * wps, iters = 2000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((srcc_kernel<MODE, K>), dim3(grid), dim3(256), 0, 0, d_src, d_sink, iters);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((srcc_kernel<MODE, K>), dim3(grid), dim3(256), 0, 0, d_src, d_sink, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = 5.0 * wps * iters * 16.0;
        printf("%s K=%2d waves/SIMD %d : %6.1f ns per MFMA per SIMD (%5.1f cycles at %.1f GHz)\n", names[MODE], K, wps, ms * 1e6 / mfma_per_simd,
               ms * 1e6 / mfma_per_simd * clk_ghz, clk_ghz);
    }
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    std::vector<int> h(2048 * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)(0x9E3779B9u * (unsigned)(i + 1)) & (int)0xF0F0F0F0;
    v4i* d_src;
    float* d_sink;
    (void)hipMalloc(&d_src, h.size() * 4);
    (void)hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    (void)hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int n = p.multiProcessorCount;
    const double g = 2.0;   // nominal figure for the cycles column only
    run<0, 0>(d_src, d_sink, n, g); run<1, 0>(d_src, d_sink, n, g); run<2, 0>(d_src, d_sink, n, g); run<3, 0>(d_src, d_sink, n, g);
    run<4, 0>(d_src, d_sink, n, g); run<5, 0>(d_src, d_sink, n, g); run<4, 7>(d_src, d_sink, n, g); run<5, 7>(d_src, d_sink, n, g);
    run<0, 7>(d_src, d_sink, n, g); run<1, 7>(d_src, d_sink, n, g); run<2, 7>(d_src, d_sink, n, g); run<3, 7>(d_src, d_sink, n, g);
    run<0, 17>(d_src, d_sink, n, g); run<1, 17>(d_src, d_sink, n, g); run<2, 17>(d_src, d_sink, n, g); run<3, 17>(d_src, d_sink, n, g);
    return 0;
}
