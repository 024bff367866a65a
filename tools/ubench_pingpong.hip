// ubench_pingpong.hip -- barrier-separated ROLE ROTATION between the W waves of a SIMD: in every phase exactly one
// wave per SIMD issues an MFMA burst (NB MFMAs, each followed by OWN of its own VALU ops), the other W-1 waves issue
// only VALU ops (their share of the detect work); roles rotate every phase, so after W phases every wave has done one
// burst and (K - OWN) * NB VALU ops outside it.  Compare with tools/ubench_inwave.hip (symmetric, unsynchronised waves).
// Reports cycles per MFMA per SIMD; the issue-port ideal is max(pipe, 8 + c K) with c ~ 2.5.
// Throw-away measurement tool (not part of the product); results quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VALU3(v, r)                                                                                       \
    do {                                                                                                  \
        if ((v) % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(kv1), "v"(kv2));        \
        if ((v) % 3 == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(kv1));                      \
        if ((v) % 3 == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(kv2));                      \
    } while (0)

// SHAPE 0: 16x16x64 i8, 1: 32x32x32 i8.  K: VALU per MFMA overall.  OWN: VALU behind each MFMA inside the burst.
// NB: MFMAs per burst.  BAR: 1 = s_barrier between phases, 0 = none (static rotation only).  NW = waves per SIMD.
template <int SHAPE, int K, int OWN, int NB, int BAR, int NW>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, int iters)
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
    constexpr int REST = (K - OWN) * NB;                       // VALU ops of one rotation outside the burst
    constexpr int PER = NW > 1 ? (REST + NW - 2) / (NW - 1) : REST;   // per non-burst phase
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < NW; p++) {
            if ((p + idx) % NW == 0) {
#pragma unroll
                for (int m = 0; m < NB; m++) {
                    if constexpr (SHAPE == 0)
                        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c4[m % 4]) : "v"(a4), "v"(b4));
                    else
                        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c16[m % 2]) : "v"(a4), "v"(b4));
#pragma unroll
                    for (int v = 0; v < OWN; v++) VALU3(v, f[(m * OWN + v) % 16]);
                }
                if constexpr (NW == 1) {
#pragma unroll
                    for (int v = 0; v < PER; v++) VALU3(v, f[v % 16]);
                }
            } else {
#pragma unroll
                for (int v = 0; v < PER; v++) VALU3(v, f[v % 16]);
            }
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
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

template <int SHAPE, int K, int OWN, int NB, int BAR, int NW>
double run(float* d_out, unsigned long long* d_clk)
{
    const int blocks = 256 * 2, lds = 100 * 1024, threads = 256 * NW, iters = 2048 / NB;
    auto kern = k<SHAPE, K, OWN, NB, BAR, NW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, 4);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    std::vector<double> wall;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; b++) {
        unsigned long long mx = 0;
        for (int w = 0; w < 4 * NW; w++) mx = std::max(mx, h[b * 16 + w]);
        wall.push_back((double)mx);
    }
    std::sort(wall.begin(), wall.end());
    return wall[wall.size() / 2] / ((double)NW * iters * NB);   // cycles per MFMA per SIMD
}

template <int SHAPE, int K, int OWN, int NB>
void row(float* d_out, unsigned long long* d_clk)
{
    printf("%s K=%2d own=%d burst=%2d : barrier  W=2 %5.1f  W=3 %5.1f  W=4 %5.1f | no barrier  W=1 %5.1f  W=2 %5.1f  W=3 %5.1f  W=4 %5.1f\n",
           SHAPE ? "32x32x32" : "16x16x64", K, OWN, NB, run<SHAPE, K, OWN, NB, 1, 2>(d_out, d_clk),
           run<SHAPE, K, OWN, NB, 1, 3>(d_out, d_clk), run<SHAPE, K, OWN, NB, 1, 4>(d_out, d_clk),
           run<SHAPE, K, OWN, NB, 0, 1>(d_out, d_clk), run<SHAPE, K, OWN, NB, 0, 2>(d_out, d_clk),
           run<SHAPE, K, OWN, NB, 0, 3>(d_out, d_clk), run<SHAPE, K, OWN, NB, 0, 4>(d_out, d_clk));
    fflush(stdout);
}

template <int SHAPE, int K>
void suite(float* d_out, unsigned long long* d_clk)
{
    row<SHAPE, K, 0, 8>(d_out, d_clk);
    row<SHAPE, K, 1, 8>(d_out, d_clk);
    row<SHAPE, K, 2, 8>(d_out, d_clk);
    row<SHAPE, K, 3, 8>(d_out, d_clk);
    row<SHAPE, K, 4, 8>(d_out, d_clk);
    row<SHAPE, K, 2, 16>(d_out, d_clk);
    row<SHAPE, K, 4, 16>(d_out, d_clk);
    row<SHAPE, K, 2, 32>(d_out, d_clk);
    row<SHAPE, K, 4, 32>(d_out, d_clk);
}

int main()
{
    float* d_out;
    unsigned long long* d_clk;
    (void)hipMalloc(&d_out, 512 * 1024 * sizeof(float));
    (void)hipMalloc(&d_clk, 512 * 16 * 8);
    printf("cycles per MFMA per SIMD; issue-port ideal = max(32 or 16.3, 8 + 2.5 K)\n");
    suite<1, 12>(d_out, d_clk);
    suite<1, 14>(d_out, d_clk);
    suite<0, 6>(d_out, d_clk);
    suite<0, 7>(d_out, d_clk);
    suite<1, 10>(d_out, d_clk);   // the contracted (5 VALU per sample) detect
    suite<0, 5>(d_out, d_clk);
    return 0;
}
