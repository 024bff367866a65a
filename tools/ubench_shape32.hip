// ubench_shape32.hip -- v_mfma_i32_32x32x32_i8 against v_mfma_i32_16x16x64_i8 in LONG accumulator chains (the deep classes' pattern: 4 chains
// of 8 per row tile) on realistic operands (A = nibble + 8, B = random int8), with K independent v_fma_f32 per 16x16x64-equivalent
// pinned between the MFMAs; 2 and 4 waves per SIMD.  ns per 16x16x64-equivalent (32768 ops) per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_shape32 tools/ubench_shape32.hip && tools/ubench_shape32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int CH, int K>   // SHAPE 16 / 32; CH chains; K VALU per 16x16x64-equivalent
__global__ __launch_bounds__(256) void shape_kernel(const v4i* __restrict__ srca, const v4i* __restrict__ srcb, float* __restrict__ sink, int iters)
{
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = srca[i * 256 + threadIdx.x], b[i] = srcb[i * 256 + threadIdx.x];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 0.5f;
    float s = 0;
    if constexpr (SHAPE == 16) {
        v4i d[CH];
#pragma unroll
        for (int t = 0; t < CH; t++) d[t] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int t = 0; t < CH; t++)
#pragma unroll
                for (int k = 0; k < 32 / CH; k++) {
                    d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], d[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < K; q++) f[(k * K + q) & 7] = __builtin_fmaf(f[(k * K + q) & 7], m, c);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (K) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < CH; t++) s += (float)(d[t][0] + d[t][3]);
    } else {
        v16i d[CH];
#pragma unroll
        for (int t = 0; t < CH; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) d[t][e] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int t = 0; t < CH; t++)
#pragma unroll
                for (int k = 0; k < 16 / CH; k++) {   // 16 MFMAs of twice the work
                    d[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], d[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2 * K; q++) f[(k * K + q) & 7] = __builtin_fmaf(f[(k * K + q) & 7], m, c);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (K) __builtin_amdgcn_sched_group_barrier(0x002, 2 * K, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < CH; t++) s += (float)(d[t][0] + d[t][15]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) s += f[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int CH, int K>
void run(const v4i* da, const v4i* db, float* d_sink, int n_cus)
{
    for (int wps : {2, 4}) {
        const int grid = n_cus * wps, iters = 2000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((shape_kernel<SHAPE, CH, K>), dim3(grid), dim3(256), 0, 0, da, db, d_sink, iters);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((shape_kernel<SHAPE, CH, K>), dim3(grid), dim3(256), 0, 0, da, db, d_sink, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double eq_per_simd = 5.0 * wps * iters * 32.0;
        const double ops = 5.0 * grid * 4.0 * iters * 32.0 * 32768.0;
        printf("%dx%dx%d  chains %d (runs of %2d)  K=%d  waves/SIMD %d : %6.2f ns per 16x16x64-equivalent per SIMD   %.3f of 5.0 POP/s\n", SHAPE, SHAPE, SHAPE == 16 ? 64 : 32, CH,
               (SHAPE == 16 ? 32 : 16) / CH, K, wps, ms * 1e6 / eq_per_simd, ops / (ms * 1e-3) / 5e15);
    }
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const size_t n = 4 * 256 * 16;
    std::mt19937_64 rng(0xD5A);
    std::vector<signed char> ha(n), hb(n);
    for (size_t i = 0; i < n; i++) ha[i] = (signed char)(rng() & 15), hb[i] = (signed char)((int)(rng() % 255) - 127);
    signed char *da, *db;
    float* d_sink;
    (void)hipMalloc(&da, n);
    (void)hipMalloc(&db, n);
    (void)hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    (void)hipMemcpy(da, ha.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), n, hipMemcpyHostToDevice);
    const int c = p.multiProcessorCount;
    const v4i *a = (const v4i*)da, *b = (const v4i*)db;
    run<16, 1, 0>(a, b, d_sink, c); run<32, 1, 0>(a, b, d_sink, c);
    run<16, 4, 0>(a, b, d_sink, c); run<32, 4, 0>(a, b, d_sink, c);
    run<16, 4, 2>(a, b, d_sink, c); run<32, 4, 2>(a, b, d_sink, c);
    run<16, 8, 3>(a, b, d_sink, c); run<32, 8, 3>(a, b, d_sink, c);
    run<16, 16, 7>(a, b, d_sink, c); run<32, 16, 7>(a, b, d_sink, c);
    run<16, 16, 3>(a, b, d_sink, c);   // fusedg_kernel's pattern (32 accumulators in runs of 2; 16 here) against runs of 4 above at the same K
    run<16, 32, 3>(a, b, d_sink, c);   // ... with all 32 accumulators (runs of 1 per iteration of this loop)
    return 0;
}
