// ubench_coexec2.hip -- which MFMA shapes co-execute with fp32 VALU work on gfx950?  Specialised waves: 8-wave
// workgroups, waves 0-3 only MFMA, waves 4-7 only VALU (v_fma_f32), 2 workgroups per CU -> every SIMD holds 2 MFMA
// waves + 2 VALU waves.  Reports MFMA-only, VALU-only and combined wall time; combined == max means full overlap,
// combined == sum means none.  Throw-away measurement tool (not part of the product); results quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// KIND: 0 i8 16x16x64 (gfx950)  1 i8 32x32x32 (gfx950)  2 i8 16x16x32 (gfx942 shape)  3 i8 32x32x16 (gfx942 shape)
//       4 f16 16x16x32 (gfx950) 5 f16 32x32x8 (legacy)   6 f16 16x16x16 (legacy)
template <int KIND, int NM, int NV, bool DO_M, bool DO_V>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
    extern __shared__ char smem[];
    const bool mwave = (threadIdx.x >> 6) < 4;
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    float s = 0;
    if (mwave) {
        if (DO_M) {
            v4i a4 = {(int)threadIdx.x, 2, 3, 4}, b4 = {5, 6, (int)blockIdx.x, 8};
            v2i a2 = {(int)threadIdx.x, 2}, b2 = {5, (int)blockIdx.x};
            v8h ah8, bh8; v4h ah4, bh4;
            for (int i = 0; i < 8; i++) { ah8[i] = (_Float16)(threadIdx.x + i); bh8[i] = (_Float16)(i + 1); }
            for (int i = 0; i < 4; i++) { ah4[i] = (_Float16)(threadIdx.x + i); bh4[i] = (_Float16)(i + 1); }
            long la = ((long)threadIdx.x << 32) | 0x01020304, lb = 0x0506070801020304L + blockIdx.x;
            v4i c4[4]; v16i c16[2]; v4f d4[4]; v16f d16[2];
            for (int i = 0; i < 4; i++) { c4[i] = v4i{i, i, i, i}; d4[i] = v4f{0, 0, 0, 0}; }
            for (int i = 0; i < 2; i++) for (int j = 0; j < 16; j++) { c16[i][j] = j; d16[i][j] = j; }
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    if constexpr (KIND == 0) c4[m % 4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, c4[m % 4], 0, 0, 0);
                    if constexpr (KIND == 1) c16[m % 2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c16[m % 2], 0, 0, 0);
                    if constexpr (KIND == 2) c4[m % 4] = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c4[m % 4], 0, 0, 0);
                    if constexpr (KIND == 3) c16[m % 2] = __builtin_amdgcn_mfma_i32_32x32x16_i8(la, lb, c16[m % 2], 0, 0, 0);
                    if constexpr (KIND == 4) d4[m % 4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah8, bh8, d4[m % 4], 0, 0, 0);
                    if constexpr (KIND == 5) d16[m % 2] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah4, bh4, d16[m % 2], 0, 0, 0);
                    if constexpr (KIND == 6) d4[m % 4] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah4, bh4, d4[m % 4], 0, 0, 0);
                }
            }
            for (int i = 0; i < 4; i++) s += (float)c4[i][0] + d4[i][0];
            for (int i = 0; i < 2; i++) s += (float)c16[i][3] + d16[i][5];
            (void)a2; (void)b2;
        }
    } else if (DO_V) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int v = 0; v < NV; v++) f[v % 16] = __builtin_fmaf(f[v % 16], 1.0001f, 0.5f);
        }
    }
    for (int i = 0; i < 16; i++) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
}

template <int KIND, int NM, int NV, bool DO_M, bool DO_V>
float run(float* d_out, int iters)
{
    const int blocks = 256 * 2 * 8, lds = 72 * 1024;
    auto kern = k<KIND, NM, NV, DO_M, DO_V>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d_out, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int KIND, int NM, int NV>
void compare(const char* name, float* d_out)
{
    const int iters = 1000;
    const float tm = run<KIND, NM, NV, true, false>(d_out, iters), tv = run<KIND, NM, NV, false, true>(d_out, iters);
    const float tb = run<KIND, NM, NV, true, true>(d_out, iters);
    // MFMA cycles each at 2.4 GHz: 16 wave-rounds per SIMD
    printf("%-22s NM=%2d NV=%3d : mfma-only %.3f ms (%.1f cyc/mfma @2.4GHz)  valu-only %.3f ms  both %.3f ms = %.2f x max, %.2f x sum\n",
           name, NM, NV, tm, tm * 1e-3 * 2.4e9 / (16.0 * iters * NM), tv, tb, tb / (tm > tv ? tm : tv), tb / (tm + tv));
}

int main()
{
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 16 * 512 * sizeof(float));
    compare<0, 8, 64>("i8 16x16x64", d_out);
    compare<1, 4, 64>("i8 32x32x32", d_out);
    compare<2, 8, 64>("i8 16x16x32", d_out);
    compare<3, 4, 64>("i8 32x32x16", d_out);
    compare<4, 8, 64>("f16 16x16x32", d_out);
    compare<5, 4, 128>("f16 32x32x8", d_out);
    compare<6, 8, 64>("f16 16x16x16", d_out);
    return 0;
}
