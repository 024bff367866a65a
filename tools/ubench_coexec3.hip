// ubench_coexec3.hip -- fine-grained interleave: every wave runs (1 x v_mfma_i32_16x16x64_i8 + KV x v_fma_f32) x 8 per
// iteration, the FMAs independent of the MFMA results, program order fixed with volatile inline asm.  Per-SIMD
// occupancy 1, 2 or 4 waves (LDS-limited workgroups of 4 waves).  Reports ns and cycles (at the in-kernel clock) per
// MFMA slot, to read how many VALU instructions hide behind one 16-cycle MFMA.
// Throw-away measurement tool (not part of the product); results quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int KV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk)
{
    extern __shared__ char smem[];
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v4i c[4];
    for (int i = 0; i < 4; i++) c[i] = v4i{i, i, i, i};
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const float k1 = 1.0001f, k2 = 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {   // volatile asm keeps exactly this program order
            if (MF) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[m % 4]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < KV; v++)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(m * KV + v) % 16]) : "v"(k1), "v"(k2));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += f[i] + (float)c[i / 4][i % 4];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KV, bool MF>
void run(float* d_out, unsigned long long* d_clk, int occ)
{
    const int iters = 2000, rounds = 4;
    const int blocks = 256 * occ * rounds;
    const int lds = occ == 1 ? 100 * 1024 : occ == 2 ? 72 * 1024 : 36 * 1024;
    auto kern = k<KV, MF>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, 10, d_clk);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, iters, d_clk);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int i = 0; i < blocks; i++) { ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; cyc[i] = (double)h[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    // per SIMD: occ * rounds waves each doing iters * 8 slots
    const double ns_per_slot = best * 1e6 / ((double)occ * rounds * iters * 8);
    printf("%s KV=%2d occ=%d : %.3f ms  %.2f ns per (MFMA+KV VALU) slot per SIMD = %.1f cyc at in-kernel %.2f GHz ; wave-cycles per slot %.1f\n",
           MF ? "mfma+valu" : "valu-only", KV, occ, best, ns_per_slot, ns_per_slot * ghz[blocks / 2], ghz[blocks / 2],
           cyc[blocks / 2] / (iters * 8.0));
}

int main()
{
    float* d_out; unsigned long long* d_clk;
    (void)hipMalloc(&d_out, 256 * 16 * 256 * sizeof(float));
    (void)hipMalloc(&d_clk, 256 * 16 * 2 * 8);
    for (int occ : {1, 2, 4}) {
        run<0, true>(d_out, d_clk, occ);
        run<1, true>(d_out, d_clk, occ);
        run<2, true>(d_out, d_clk, occ);
        run<3, true>(d_out, d_clk, occ);
        run<4, true>(d_out, d_clk, occ);
        run<6, true>(d_out, d_clk, occ);
        run<8, true>(d_out, d_clk, occ);
        run<16, true>(d_out, d_clk, occ);
        run<4, false>(d_out, d_clk, occ);
        run<8, false>(d_out, d_clk, occ);
        run<16, false>(d_out, d_clk, occ);
    }
    return 0;
}
