// ubench_coexec.hip -- do int8 MFMA and fp32 VALU instructions overlap on a gfx950 SIMD, and under which conditions?
// Fixed total work, forced occupancy (4 workgroups of 4 waves per CU through the LDS size), wall-clock timing.
// MODE 0: every wave runs NM MFMAs then NV independent FMAs per iteration
// MODE 1: same, but the FMAs consume the MFMA results (like a GEMM epilogue)
// MODE 2: specialised waves: 8-wave workgroups, waves 0-3 only MFMA, waves 4-7 only VALU, so that every SIMD holds
//         2 MFMA waves and 2 VALU waves (2 workgroups per CU); same per-SIMD instruction totals as MODE 0
// Throw-away measurement tool (not part of the product); results are quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NM, int NV, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 512 : 256) void k(float* out, int iters)
{
    extern __shared__ char smem[];
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v4i c[4];
    for (int i = 0; i < 4; i++) c[i] = v4i{i, i, i, i};
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    const bool do_m = MODE != 2 || (threadIdx.x >> 6) < 4;
    const bool do_v = MODE != 2 || (threadIdx.x >> 6) >= 4;
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) c[m % 4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[m % 4], 0, 0, 0);
        }
        if (do_v) {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 16; i++) f[i] += __builtin_bit_cast(v4f, c[i / 4])[i % 4];
            }
#pragma unroll
            for (int v = 0; v < NV; v++) f[v % 16] = __builtin_fmaf(f[v % 16], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += f[i] + (float)c[i / 4][i % 4];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
}

static hipStream_t g_stream = nullptr;  // optionally CU-masked (argv[1] = number of CUs) to take the chip out of its power limit
static int g_cus = 256;

template <int NM, int NV, int MODE>
float run(float* d_out, int iters)
{
    const int blocks = g_cus * 4 * 4;  // 4 resident workgroups per CU, 4 rounds
    const int lds = (MODE == 2 ? 72 : 36) * 1024;  // 160 KB / 36 KB -> 4 workgroups per CU (2 of the 8-wave kind)
    const int threads = MODE == 2 ? 512 : 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NM, NV, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(blocks), dim3(threads), lds, g_stream, d_out, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, g_stream);
        hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(blocks), dim3(threads), lds, g_stream, d_out, iters);
        hipEventRecord(e1, g_stream);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int NM, int NV>
void compare(float* d_out)
{
    const int iters = 1000;
    const float tm = run<NM, 0, 0>(d_out, iters), tv = run<0, NV, 0>(d_out, iters);
    const float t0 = run<NM, NV, 0>(d_out, iters), t1 = run<NM, NV, 1>(d_out, iters);
    const float t2 = run<NM, NV, 2>(d_out, iters);  // specialised waves, same totals per SIMD
    // per-iteration per-wave cycle-equivalents at 16 resident waves per CU: ms -> ns per (wave, iteration) / 4 rounds
    printf("NM=%2d NV=%3d : mfma-only %.3f ms  valu-only %.3f ms  sum %.3f | same-wave independent %.3f (%.2f of sum)  "
           "epilogue-dependent %.3f (%.2f)  specialised waves %.3f (%.2f)\n",
           NM, NV, tm, tv, tm + tv, t0, t0 / (tm + tv), t1, t1 / (tm + tv), t2, t2 / (tm + tv));
}

int main(int argc, char** argv)
{
    if (argc > 1) {
        g_cus = atoi(argv[1]);
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int stride = argc > 2 ? atoi(argv[2]) : 1;  // enable every stride-th CU
        for (int i = 0; i < g_cus; i++) mask[(i * stride) / 32] |= 1u << ((i * stride) % 32);
        if (hipExtStreamCreateWithCUMask(&g_stream, 8, mask) != hipSuccess) { printf("CU mask failed\n"); return 1; }
        printf("CU mask: %d CUs, stride %d\n", g_cus, stride);
    }
    float* d_out;
    hipMalloc(&d_out, 256 * 16 * 512 * sizeof(float));
    compare<8, 32>(d_out);
    compare<8, 64>(d_out);
    compare<8, 128>(d_out);
    compare<16, 128>(d_out);
    compare<8, 256>(d_out);
    return 0;
}
