// ubench_pkadd.hip -- what does a v_pk_add_f32 cost next to v_add_f32 on gfx950, with 1..4 waves per SIMD, operands in
// registers only?  (The shared-window DM kernel does 32 packed adds per wave and channel; is that what it waits for?)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_pkadd tools/ubench_pkadd.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: 32 v_pk_add_f32 per iteration, 1: 64 v_add_f32, 2: 32 v_pk_add + 16 ds_read_b128-sized LDS reads
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, int iters)
{
    __shared__ float lds[16 * 1024];
    v2f acc[32];
    for (int i = 0; i < 32; i++) acc[i] = v2f{(float)threadIdx.x, (float)i};
    v2f x = {1.0f + threadIdx.x * 1e-6f, 0.5f};
    asm volatile("" : "+v"(x));
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float4 v = *reinterpret_cast<const float4*>(&lds[(i * 256 + (threadIdx.x & 63) * 4) & (16 * 1024 - 1)]);
                acc[2 * i] += v2f{v.x, v.y};
                acc[2 * i + 1] += v2f{v.z, v.w};
            }
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] += x;
        } else {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                acc[i].x += x.x;
                acc[i].y += x.y;
            }
        }
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("" : "+v"(acc[i]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v2f s = {0, 0};
    for (int i = 0; i < 32; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&clk, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 3; mode++)
        for (int threads : {256, 512, 1024}) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
                hipDeviceSynchronize();
            }
            unsigned long long h[256];
            hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; i++) avg += (double)h[i];
            avg /= 256;
            const int waves_per_simd = threads / 256;
            printf("mode %d (%s) %d waves/SIMD: %.1f cycles per iteration per wave -> %.2f cycles per SIMD per %s\n", mode,
                   mode == 0 ? "32 v_pk_add_f32" : mode == 1 ? "64 v_add_f32" : "16 ds_read_b128 + 32 v_pk_add_f32", waves_per_simd,
                   avg / iters, avg / iters / (mode == 1 ? 64 : 32) / waves_per_simd * 1.0, mode == 1 ? "v_add_f32" : "v_pk_add_f32");
        }
    return 0;
}
