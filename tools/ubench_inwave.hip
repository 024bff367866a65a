// ubench_inwave.hip -- SYMMETRIC waves (every wave issues MFMAs and fp32 VALU work, like the fused kernel): which
// in-wave schedule lets the VALU work of the W waves of a SIMD run in the shadow of the MFMA pipe?
// tools/ubench_roles.hip showed (profiles/r02_ubench_roles.txt): an MFMA costs the SIMD's vector issue 8 cycles, a
// VGPR-operand VALU op ~2.3-2.6 (an SGPR-operand one 4.4), but an MFMA that has to WAIT for the busy matrix pipe blocks
// vector issue for every wave of the SIMD until the pipe frees -> back-to-back MFMAs degrade to "MFMA time + VALU time".
// Here: per slot [pad][MFMA x G][VALU x G*K]; W waves per SIMD (one workgroup per CU), optional barrier every `bar`
// slots, optional start stagger and static priorities.  Reports cycles per MFMA per SIMD (ideal = max(pipe, 8 + c K)).
// Throw-away measurement tool (not part of the product); results quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// SHAPE 0: 16x16x64 i8, 1: 32x32x32 i8.  K: VALU per MFMA.  G: MFMAs per group (G MFMAs, then G*K VALU).
// PRIO 0: none; 1: static s_setprio by the wave's index on its SIMD (first wave highest); 2: reversed.
// PAD: s_nop wait states in front of every MFMA group (0 = none).
template <int SHAPE, int K, int G, int PRIO, int PAD>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, int iters, int bar, int stagger)
{
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    const int idx = wave >> 2;          // index of this wave on its SIMD
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    float kv1 = 1.0001f, kv2 = 0.5f;
    asm volatile("" : "+v"(kv1), "+v"(kv2));
    v4i a4 = {(int)threadIdx.x * 0x01010101, 0x02030405, 0x03f1e2d3, 0x04a5b6c7}, b4 = {0x05060708, 0x06f7e8d9, (int)blockIdx.x, 0x08192a3b};
    v4i c4[4];
    v16i c16[2];
    for (int i = 0; i < 4; i++) c4[i] = v4i{i, i, i, i};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 16; j++) c16[i][j] = j;
    if (PRIO == 1) {
        if (idx == 0) asm volatile("s_setprio 3");
        if (idx == 1) asm volatile("s_setprio 2");
        if (idx == 2) asm volatile("s_setprio 1");
    }
    if (PRIO == 2) {
        if (idx == 3) asm volatile("s_setprio 3");
        if (idx == 2) asm volatile("s_setprio 2");
        if (idx == 1) asm volatile("s_setprio 1");
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto delay = [&]() {
        for (int d = 0; d < idx * stagger; d++) asm volatile("s_nop 15");   // 16 wait states each
    };
    delay();
    constexpr int SLOTS = 8 / G;   // groups per unrolled body (8 MFMAs per body)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if constexpr (PAD == 4) asm volatile("s_nop 3");
            if constexpr (PAD == 8) asm volatile("s_nop 7");
#pragma unroll
            for (int g = 0; g < G; g++) {
                if constexpr (SHAPE == 0)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c4[(s * G + g) % 4]) : "v"(a4), "v"(b4));
                else
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c16[(s * G + g) % 2]) : "v"(a4), "v"(b4));
            }
#pragma unroll
            for (int v = 0; v < G * K; v++) {   // the detect's mix: 2 fma, 2 mul, 2 add per sample, VGPR operands only
                float& r = f[(s * G * K + v) % 16];
                if (v % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(kv1), "v"(kv2));
                if (v % 3 == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(kv1));
                if (v % 3 == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(kv2));
            }
        }
        if (bar > 0 && (it + 1) % bar == 0) {
            __syncthreads();
            delay();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += f[i];
    for (int i = 0; i < 4; i++) s += (float)c4[i][0];
    for (int i = 0; i < 2; i++) s += (float)c16[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int SHAPE, int K, int G, int PRIO, int PAD>
double run(float* d_out, unsigned long long* d_clk, int W, int bar, int stagger)
{
    const int blocks = 256 * 2, lds = 100 * 1024, threads = 256 * W, iters = 256;
    auto kern = k<SHAPE, K, G, PRIO, PAD>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, 4, bar, stagger);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    std::vector<double> wall;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, iters, bar, stagger);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; b++) {
        unsigned long long mx = 0;
        for (int w = 0; w < 4 * W; w++) mx = std::max(mx, h[b * 16 + w]);
        wall.push_back((double)mx);
    }
    std::sort(wall.begin(), wall.end());
    return wall[wall.size() / 2] / ((double)W * iters * 8);   // cycles per MFMA per SIMD
}

template <int SHAPE, int K, int G, int PRIO, int PAD>
void row(float* d_out, unsigned long long* d_clk)
{
    printf("%s K=%2d G=%d prio=%d pad=%d :", SHAPE ? "32x32x32" : "16x16x64", K, G, PRIO, PAD);
    for (int W = 1; W <= 4; W++) printf("  W=%d %5.1f", W, run<SHAPE, K, G, PRIO, PAD>(d_out, d_clk, W, 0, 0));
    printf(" | bar/2it:");
    for (int W = 2; W <= 4; W++) printf("  W=%d %5.1f", W, run<SHAPE, K, G, PRIO, PAD>(d_out, d_clk, W, 2, 0));
    printf(" | stagger(bar/2it):");
    for (int W = 2; W <= 4; W++) printf("  W=%d %5.1f", W, run<SHAPE, K, G, PRIO, PAD>(d_out, d_clk, W, 2, 2));
    printf(" | stagger(no bar):");
    for (int W = 2; W <= 4; W++) printf("  W=%d %5.1f", W, run<SHAPE, K, G, PRIO, PAD>(d_out, d_clk, W, 0, 2));
    printf("\n");
    fflush(stdout);
}

template <int SHAPE, int K>
void suite(float* d_out, unsigned long long* d_clk)
{
    row<SHAPE, K, 1, 0, 0>(d_out, d_clk);
    row<SHAPE, K, 2, 0, 0>(d_out, d_clk);
    row<SHAPE, K, 4, 0, 0>(d_out, d_clk);
    row<SHAPE, K, 8, 0, 0>(d_out, d_clk);
    row<SHAPE, K, 1, 1, 0>(d_out, d_clk);
    row<SHAPE, K, 4, 1, 0>(d_out, d_clk);
    row<SHAPE, K, 1, 2, 0>(d_out, d_clk);
    row<SHAPE, K, 1, 0, 4>(d_out, d_clk);
    row<SHAPE, K, 1, 0, 8>(d_out, d_clk);
    row<SHAPE, K, 1, 1, 4>(d_out, d_clk);
}

int main()
{
    float* d_out;
    unsigned long long* d_clk;
    (void)hipMalloc(&d_out, 512 * 1024 * sizeof(float));
    (void)hipMalloc(&d_clk, 512 * 16 * 8);
    printf("cycles per MFMA per SIMD; ideal = max(32 or 16.3, 8 + c*K), c ~ 2.4 for W >= 2, ~4.8 for W = 1\n");
    suite<1, 6>(d_out, d_clk);
    suite<1, 9>(d_out, d_clk);
    suite<1, 12>(d_out, d_clk);
    suite<1, 14>(d_out, d_clk);
    suite<0, 3>(d_out, d_clk);
    suite<0, 6>(d_out, d_clk);
    suite<0, 7>(d_out, d_clk);
    return 0;
}
