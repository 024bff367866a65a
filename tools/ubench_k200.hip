// ubench_k200.hip -- would packing the 100-antenna contraction as K' = 200 = 3 x 64 + 8 pay?  
// The C5 general kernel runs 100 antennas as 128 with zero weights behind antenna 99: per output row and 16-beam tile a chain of
// FOUR v_mfma_i32_16x16x64_i8 (re|im halves of two 64-antenna k-steps), 22 % of whose products are zero padding.  Packed as
// [re(100) | im(100)] the chain would be three full k-steps and an 8-byte remainder -- which only saves anything if the
// remainder can run on a SHORTER instruction (v_mfma_i32_16x16x32_i8) that costs less than the full one.  This measures exactly
// that, in the kernel's own pattern (docs/PERF_MODEL.md section 2: 8 accumulator chains of 4 per wave, K = 3 VALU ops pinned
// behind every MFMA, 2 waves per SIMD; no memory): chains of 4 full MFMAs against chains of 3 full + 1 short.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_k200 tools/ubench_k200.hip && tools/ubench_k200
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int SHORT_TAIL, int K>
__global__ __launch_bounds__(256) void chain_kernel(const v4i* __restrict__ srca, const v4i* __restrict__ srcb, float* __restrict__ sink, int iters)
{
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = srca[i * 256 + threadIdx.x], b[i] = srcb[i * 256 + threadIdx.x];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 0.5f;
    v4i d[8];
#pragma unroll
    for (int t = 0; t < 8; t++) d[t] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (SHORT_TAIL && k == 3) {
                    const long sa = ((long)a[(t + k) & 3][1] << 32) | (unsigned)a[(t + k) & 3][0];
                    const long sb = ((long)b[(t + 2 * k) & 3][1] << 32) | (unsigned)b[(t + 2 * k) & 3][0];
                    d[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(sa, sb, d[t], 0, 0, 0);
                } else {
                    d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], d[t], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < K; q++) f[(k * K + q) & 7] = __builtin_fmaf(f[(k * K + q) & 7], m, c);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (K) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            }
    }
    float s = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) s += (float)(d[t][0] + d[t][3]);
#pragma unroll
    for (int i = 0; i < 8; i++) s += f[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHORT_TAIL, int K>
double run(const v4i* da, const v4i* db, float* d_sink, int n_cus, int wps)
{
    const int grid = n_cus * wps, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((chain_kernel<SHORT_TAIL, K>), dim3(grid), dim3(256), 0, 0, da, db, d_sink, iters);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((chain_kernel<SHORT_TAIL, K>), dim3(grid), dim3(256), 0, 0, da, db, d_sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double chains_per_simd = 5.0 * wps * iters * 8.0;
    const double ns = ms * 1e6 / chains_per_simd;
    printf("chain of 4: %s  K=%d  waves/SIMD %d : %6.2f ns per chain per SIMD\n", SHORT_TAIL ? "3 x 16x16x64 + 1 x 16x16x32" : "4 x 16x16x64             ", K, wps, ns);
    return ns;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const size_t n = 4 * 256 * 16;
    std::mt19937_64 rng(0xD5A);
    std::vector<signed char> ha(n), hb(n);
    for (size_t i = 0; i < n; i++) ha[i] = (signed char)((rng() & 15) << 4), hb[i] = (signed char)((int)(rng() % 255) - 127);
    signed char *da, *db;
    float* d_sink;
    (void)hipMalloc(&da, n);
    (void)hipMalloc(&db, n);
    (void)hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    (void)hipMemcpy(da, ha.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), n, hipMemcpyHostToDevice);
    const int c = p.multiProcessorCount;
    const v4i *a = (const v4i*)da, *b = (const v4i*)db;
    for (int wps : {2, 4}) {
        const double full = run<0, 3>(a, b, d_sink, c, wps), tail = run<1, 3>(a, b, d_sink, c, wps);
        printf("  -> the short remainder changes the chain's time by %+.1f %% (K = 3, %d waves per SIMD)\n", 100.0 * (tail / full - 1.0), wps);
    }
    for (int wps : {2, 4}) {
        const double full = run<0, 0>(a, b, d_sink, c, wps), tail = run<1, 0>(a, b, d_sink, c, wps);
        printf("  -> MFMAs alone: %+.1f %% (%d waves per SIMD)\n", 100.0 * (tail / full - 1.0), wps);
    }
    return 0;
}
