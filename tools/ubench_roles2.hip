// ubench_roles2.hip -- (sweep of ubench_roles.hip: own fillers x total VALU load)
// ubench_roles.hip -- when do int8 MFMA and fp32 VALU instructions of DIFFERENT waves overlap on one gfx950 SIMD?
// Round 1 (ubench_coexec2) put 2 MFMA-only + 2 VALU-only waves on every SIMD and measured time = sum.  Hypothesis to
// test here: an MFMA that is waiting for the busy matrix pipe sits in the SIMD's vector-issue stage and blocks the
// VALU port for everybody; then overlap needs (a) at most ONE wave per SIMD inside an MFMA burst and (b) that wave not
// attempting its next MFMA before the pipe is free (padding with s_nop / its own few VALU ops).
//
// One workgroup per CU (LDS-limited), 4*(P+Q) waves: waves with (wave/4) < P issue only MFMAs, the others only
// v_fma_f32 (16 independent chains, scalar + inline-constant operands).  Waves w, w+4, w+8 ... share a SIMD.
// For every configuration: MFMA-only, VALU-only (VALU work scaled to take as long as the MFMA work) and both;
// both/max = 1 means full overlap, both/sum = 1 means none.  Cycles are s_memtime ticks (shader clock).
// Throw-away measurement tool (not part of the product); results quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define NOP_CASE(n) if constexpr (PAD == n + 1) asm volatile("s_nop " #n);

template <int PAD>
__device__ __forceinline__ void pad()
{
    NOP_CASE(0) NOP_CASE(1) NOP_CASE(2) NOP_CASE(3) NOP_CASE(4) NOP_CASE(5) NOP_CASE(6) NOP_CASE(7)
    NOP_CASE(8) NOP_CASE(9) NOP_CASE(10) NOP_CASE(11) NOP_CASE(12) NOP_CASE(13) NOP_CASE(14) NOP_CASE(15)
    if constexpr (PAD == 100) asm volatile("s_nop 0\n s_nop 0");
    if constexpr (PAD == 101) asm volatile("s_nop 0\n s_nop 0\n s_nop 0");
}

// SHAPE 0: v_mfma_i32_16x16x64_i8, 1: v_mfma_i32_32x32x32_i8.  PRIO 0: none, 1: VALU waves s_setprio 3, 2: MFMA waves 3.
// OWNV: VALU ops the MFMA wave itself issues behind each of its MFMAs (0..2).
template <int SHAPE, int PAD, int PRIO, int OWNV, int FORM>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, int P, int Q, int iters_m, int iters_v)
{
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    const bool mwave = (wave >> 2) < P;
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    float s = 0;
    float k1 = 1.0001f;
    asm volatile("" : "+s"(k1));
    float kv1 = 1.0001f, kv2 = 0.5f;
    asm volatile("" : "+v"(kv1), "+v"(kv2));
#define VALU_OP(reg)                                                                                             \
    do {                                                                                                         \
        if constexpr (FORM == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(reg) : "s"(k1));                \
        if constexpr (FORM == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(reg) : "v"(kv1), "v"(kv2));      \
        if constexpr (FORM == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(reg) : "v"(kv1));                    \
        if constexpr (FORM == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(reg) : "v"(kv1), "v"(kv2));         \
        if constexpr (FORM == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(reg) : "s"(k1), "v"(kv2));       \
        if constexpr (FORM == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(reg) : "v"(kv1));      \
    } while (0)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mwave) {
        if (PRIO == 2) asm volatile("s_setprio 3");
        v4i a4 = {(int)threadIdx.x * 0x01010101, 0x02030405, 0x03f1e2d3, 0x04a5b6c7}, b4 = {0x05060708, 0x06f7e8d9, (int)blockIdx.x, 0x08192a3b};
        v4i c4[4];
        v16i c16[2];
        for (int i = 0; i < 4; i++) c4[i] = v4i{i, i, i, i};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 16; j++) c16[i][j] = j;
        for (int it = 0; it < iters_m; it++) {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if constexpr (SHAPE == 0)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c4[m % 4]) : "v"(a4), "v"(b4));
                else
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c16[m % 2]) : "v"(a4), "v"(b4));
#pragma unroll
                for (int v = 0; v < OWNV; v++) VALU_OP(f[(m * OWNV + v) % 16]);
                pad<PAD>();
            }
        }
        for (int i = 0; i < 4; i++) s += (float)c4[i][0];
        for (int i = 0; i < 2; i++) s += (float)c16[i][3];
    } else if ((wave >> 2) < P + Q) {
        if (PRIO == 1) asm volatile("s_setprio 3");
        for (int it = 0; it < iters_v; it++) {
#pragma unroll
            for (int v = 0; v < 64; v++) VALU_OP(f[v % 16]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; i++) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        clk[blockIdx.x * 16 + wave] = ((t1 - t0) << 8) | ((hwid >> 4) & 3);   // low byte: SIMD id of the wave
    }
}

struct Res { double wall_cyc; float ms; };

template <int SHAPE, int PAD, int PRIO, int OWNV, int FORM>
Res run(float* d_out, unsigned long long* d_clk, int P, int Q, int im, int iv, int total_waves_per_simd)
{
    const int blocks = 256 * 2;
    const int lds = 100 * 1024;
    const int threads = 256 * total_waves_per_simd;
    auto kern = k<SHAPE, PAD, PRIO, OWNV, FORM>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, P, Q, 4, 4);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    std::vector<unsigned long long> h(blocks * 16);
    std::vector<double> wall;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_clk, P, Q, im, iv);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; b++) {
        unsigned long long mx = 0;
        for (int w = 0; w < 4 * total_waves_per_simd; w++) mx = std::max(mx, h[b * 16 + w] >> 8);
        wall.push_back((double)mx);
    }
    std::sort(wall.begin(), wall.end());
    static bool once = false;
    if (!once && total_waves_per_simd == 4) {
        once = true;
        printf("SIMD id of waves 0..15 of block 0:");
        for (int w = 0; w < 16; w++) printf(" %d", (int)(h[w] & 3));
        printf("   block 300:");
        for (int w = 0; w < 16; w++) printf(" %d", (int)(h[300 * 16 + w] & 3));
        printf("\n");
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return Res{wall[wall.size() / 2], best};
}

template <int SHAPE, int PAD, int PRIO, int OWNV, int FORM>
void config(float* d_out, unsigned long long* d_clk, int P, int Q, double ratio)
{
    const int W = P + Q;
    const int im = 400;
    const double n_m = (double)P * im * 16;                    // MFMAs per SIMD
    const Res m = run<SHAPE, PAD, PRIO, OWNV, FORM>(d_out, d_clk, P, 0, im, 0, W);
    int iv = (int)(ratio * n_m / (Q * 64.0) + 0.5);            // VALU instructions per MFMA (other waves) = ratio
    if (iv < 1) iv = 1;
    // VALU-only with the SAME wave placement as the combined run: MFMA waves present but idle (iters_m = 0)
    const Res v = run<SHAPE, PAD, PRIO, OWNV, FORM>(d_out, d_clk, P, Q, 0, iv, W);
    const double n_v = (double)Q * iv * 64;                    // VALU instructions per SIMD
    const Res b = run<SHAPE, PAD, PRIO, OWNV, FORM>(d_out, d_clk, P, Q, im, iv, W);
    printf("%s form=%d P=%d Q=%d pad=%3d ownv=%d V/M=%5.2f : mfma-only %5.1f cyc/mfma  valu-only %.2f cyc/valu | both %5.1f cyc/mfma-slot = %.2f x max, %.2f x sum ;"
           " model 8M+cV %5.1f  16.3M+cV %5.1f\n",
           SHAPE ? "32x32x32" : "16x16x64", FORM, P, Q, PAD, OWNV, n_v / n_m + OWNV, m.wall_cyc / n_m, v.wall_cyc / n_v, b.wall_cyc / n_m,
           b.wall_cyc / std::max(m.wall_cyc, v.wall_cyc), b.wall_cyc / (m.wall_cyc + v.wall_cyc),
           8.0 + v.wall_cyc / n_m, (SHAPE ? 32.0 : 16.3) + v.wall_cyc / n_m);
    fflush(stdout);
}

template <int SHAPE, int FORM, int OWNV>
void own_row(float* d_out, unsigned long long* d_clk)
{
    // total VALU per MFMA = OWNV (issued by the MFMA waves behind each of their MFMAs) + the VALU-only waves' share
    for (double total : {6.0, 8.0, 10.0, 12.0, 14.0}) {
        if (total - OWNV < 0.5) continue;
        config<SHAPE, 0, 0, OWNV, FORM>(d_out, d_clk, 1, 3, total - OWNV);
        config<SHAPE, 0, 0, OWNV, FORM>(d_out, d_clk, 2, 2, total - OWNV);
        config<SHAPE, 0, 0, OWNV, FORM>(d_out, d_clk, 3, 1, total - OWNV);
    }
}

template <int SHAPE, int FORM>
void shape_suite(float* d_out, unsigned long long* d_clk)
{
    own_row<SHAPE, FORM, 0>(d_out, d_clk);
    own_row<SHAPE, FORM, 1>(d_out, d_clk);
    own_row<SHAPE, FORM, 2>(d_out, d_clk);
    own_row<SHAPE, FORM, 3>(d_out, d_clk);
    own_row<SHAPE, FORM, 4>(d_out, d_clk);
    own_row<SHAPE, FORM, 6>(d_out, d_clk);
}

int main()
{
    float* d_out;
    unsigned long long* d_clk;
    (void)hipMalloc(&d_out, 512 * 1024 * sizeof(float));
    (void)hipMalloc(&d_clk, 512 * 16 * 8);
    printf("specialised waves with OWN fillers: P MFMA waves issue [MFMA, ownv x VALU], Q VALU-only waves supply the rest; V/M = total VALU per MFMA\n");
    shape_suite<1, 1>(d_out, d_clk);
    shape_suite<0, 1>(d_out, d_clk);
    return 0;
}
