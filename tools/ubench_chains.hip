// ubench_chains.hip -- how many INDEPENDENT accumulator chains does v_mfma_i32_16x16x64_i8 need per wave to run at the pipe's rate,
// at 1 / 2 / 4 waves per SIMD?  (Round 4: the deep classes of fused16_kernel run 4 chains of 8 dependent MFMAs per row tile at two
// waves per SIMD and their MFMA-only skeleton tops out at 0.59 of the nominal peak; fusedg_kernel's, with 32 independent
// accumulators per plane, reaches 0.79.)  Every wave runs `iters` x 32 MFMAs: C chains round-robin, 32 / C dependent steps each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_chains tools/ubench_chains.hip && tools/ubench_chains
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// ORDER 0: round-robin over the C chains (consecutive MFMAs write DIFFERENT accumulators); ORDER 1: chain by chain (each chain's
// 32 / C dependent MFMAs back to back, then the next chain: consecutive MFMAs mostly write the SAME accumulator)
template <int C, int ORDER>
__global__ __launch_bounds__(256) void chains_kernel(const v4i* __restrict__ src, int* __restrict__ sink, int iters)
{
    v4i a[4], b[4], c[C];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = src[(i * 256 + threadIdx.x) & 4095];
        b[i] = src[((i + 4) * 256 + threadIdx.x) & 4095];
    }
#pragma unroll
    for (int t = 0; t < C; t++) c[t] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if constexpr (ORDER == 0) {
#pragma unroll
            for (int k = 0; k < 32 / C; k++)
#pragma unroll
                for (int t = 0; t < C; t++) {
                    c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], c[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int t = 0; t < C; t++)
#pragma unroll
                for (int k = 0; k < 32 / C; k++) {
                    c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], c[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
#pragma unroll
        for (int t = 0; t < C; t++) asm volatile("" : "+v"(c[t]));
    }
    int acc = 0;
#pragma unroll
    for (int t = 0; t < C; t++) acc += c[t][0] + c[t][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int C, int ORDER>
void run(const v4i* d_src, int* d_sink, int n_cus)
{
    for (int wps : {1, 2, 4}) {                       // waves per SIMD = workgroups of 4 waves per CU
        const int grid = n_cus * wps, iters = 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((chains_kernel<C, ORDER>), dim3(grid), dim3(256), 0, 0, d_src, d_sink, iters);
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((chains_kernel<C, ORDER>), dim3(grid), dim3(256), 0, 0, d_src, d_sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 5.0 * grid * 4.0 * iters * 32.0 * 2.0 * 16 * 16 * 64;
        printf("%s chains %2d (x %2d dependent)  waves/SIMD %d : %7.1f TOP/s  (%.3f of 5.0 POP/s)\n", ORDER ? "chain-by-chain" : "round-robin   ", C, 32 / C, wps, ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / 5e15);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::vector<int> h(4096 * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)(0x9E3779B9u * (unsigned)(i + 1)) & (int)0xF0F0F0F0;   // 16 x nibble operands
    v4i* d_src;
    int* d_sink;
    hipMalloc(&d_src, h.size() * 4);
    hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int n = p.multiProcessorCount;
    run<1, 0>(d_src, d_sink, n);
    run<2, 0>(d_src, d_sink, n);
    run<4, 0>(d_src, d_sink, n);
    run<8, 0>(d_src, d_sink, n);
    run<16, 0>(d_src, d_sink, n);
    run<2, 1>(d_src, d_sink, n);
    run<4, 1>(d_src, d_sink, n);
    run<8, 1>(d_src, d_sink, n);
    run<16, 1>(d_src, d_sink, n);
    return 0;
}
