// ubench_issue.hip -- measures how int8 MFMA and fp32 VALU instructions share a CDNA4 SIMD's issue/execute resources.
// Each wave loops: 8 x v_mfma_i32_32x32x32_i8 (two 4-deep accumulate chains) + NV VALU ops (plain or packed),
// all on registers.  Reports cycles per iteration per SIMD for several NV and waves/SIMD.  Throw-away measurement
// tool (not part of the product); results are quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NV, int KIND, int NM>  // KIND 0: plain fma, 1: pk_fma, 2: plain mul+add mix ; NM mfma per iter
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v16i c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = i; c1[i] = -i; }
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = 1.0f + i * 0.001f + threadIdx.x;
    v2f g[8];
    for (int i = 0; i < 8; i++) g[i] = v2f{f[2 * i], f[2 * i + 1]};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NM / 2; m++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
        }
        if (KIND == 0) {
#pragma unroll
            for (int v = 0; v < NV; v++) f[v % 16] = __builtin_fmaf(f[v % 16], 1.0001f, 0.5f);
        } else if (KIND == 1) {
#pragma unroll
            for (int v = 0; v < NV; v++) g[v % 8] = __builtin_elementwise_fma(g[v % 8], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f});
        } else {
#pragma unroll
            for (int v = 0; v < NV; v++) f[v % 16] = (v & 1) ? f[v % 16] * 1.0001f : f[v % 16] + 0.5f;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += f[i] + (float)c0[i] + (float)c1[i];
    for (int i = 0; i < 8; i++) s += g[i][0] + g[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, int NM>
void run(const char* name, int waves_per_simd, float* d_out, long long* d_cyc)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND, NM>), dim3(blocks), dim3(256), 0, 0, d_out, 10, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, NM>), dim3(blocks), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= blocks;
    // s_memtime ticks at 100 MHz constant? report both wall-derived and memtime-derived
    printf("%-10s NV=%3d NM=%d waves/SIMD=%d : %8.1f us  -> %7.1f ns/iter/wave-slot ; memtime ticks/iter %.2f ; iter/SIMD-time %.1f ns\n",
           name, NV, NM, waves_per_simd, ms * 1e3, ms * 1e6 / iters, avg / iters, ms * 1e6 / iters / waves_per_simd);
}

int main()
{
    float* d_out; long long* d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 8 * sizeof(long long));
    for (int w : {1, 2, 4}) {
        run<0, 0, 8>("mfma-only", w, d_out, d_cyc);
        run<24, 0, 8>("plain", w, d_out, d_cyc);
        run<48, 0, 8>("plain", w, d_out, d_cyc);
        run<64, 0, 8>("plain", w, d_out, d_cyc);
        run<96, 0, 8>("plain", w, d_out, d_cyc);
        run<128, 0, 8>("plain", w, d_out, d_cyc);
        run<96, 0, 0>("plain-noM", w, d_out, d_cyc);
        run<96, 2, 8>("mul/add", w, d_out, d_cyc);
        run<24, 1, 8>("packed", w, d_out, d_cyc);
        run<48, 1, 8>("packed", w, d_out, d_cyc);
        run<48, 1, 0>("packed-noM", w, d_out, d_cyc);
    }
    return 0;
}
