// ubench_encoding.hip -- does the ENCODING of the 4-bit voltages in the int8 MFMA operand change the rate the chip sustains?
// (The clock under MFMA load follows the operand bits: profiles/r01_ubench_shape.txt -- constants 4.6 POP/s, random int8 3.7.)
// Two accumulator chains per wave issued chain by chain (the pipe's fastest order, ubench_chains.hip), B = uniform random int8
// like calibrated weights; A = the same random nibbles n in [-8, 7] as  16 n | n sign-extended | n + 8 | full random int8 | 0.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_encoding tools/ubench_encoding.hip && tools/ubench_encoding
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <bool SWAP>
__global__ __launch_bounds__(256) void enc_kernel(const v4i* __restrict__ srca, const v4i* __restrict__ srcb, int* __restrict__ sink, int iters)
{
    v4i a[4], b[4], c[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = srca[((blockIdx.x & 3) * 4 + i) * 256 + threadIdx.x];
        b[i] = srcb[((blockIdx.x & 3) * 4 + i) * 256 + threadIdx.x];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                c[t] = SWAP ? __builtin_amdgcn_mfma_i32_16x16x64_i8(b[(t + 2 * k) & 3], a[(t + k) & 3], c[t], 0, 0, 0)
                            : __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + k) & 3], b[(t + 2 * k) & 3], c[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[0][3] + c[1][1] + c[1][2];
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const size_t n_bytes = 16 * 256 * 16;
    std::mt19937_64 rng(0xD5A);
    std::vector<signed char> nib(n_bytes), w(n_bytes), full(n_bytes), enc(n_bytes);
    for (size_t i = 0; i < n_bytes; i++) {
        nib[i] = (signed char)((int)(rng() & 15) - 8);
        w[i] = (signed char)((int)(rng() % 255) - 127);
        full[i] = (signed char)((int)(rng() % 255) - 127);
    }
    signed char *d_a, *d_b;
    int* d_sink;
    hipMalloc(&d_a, n_bytes);
    hipMalloc(&d_b, n_bytes);
    hipMalloc(&d_sink, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    hipMemcpy(d_b, w.data(), n_bytes, hipMemcpyHostToDevice);
    const char* names[] = {"16 x nibble", "sign-extended nibble", "nibble + 8", "random int8", "zero", "nibble + 8, weights + 128 as unsigned bits"};
    for (int rep_all = 0; rep_all < 2; rep_all++)
        for (int mode = 0; mode < 5; mode++) {
            for (size_t i = 0; i < n_bytes; i++)
                enc[i] = mode == 0 ? (signed char)(nib[i] * 16) : mode == 1 ? nib[i] : mode == 2 ? (signed char)(nib[i] + 8) : mode == 3 ? full[i] : 0;
            hipMemcpy(d_a, enc.data(), n_bytes, hipMemcpyHostToDevice);
            for (int swap = 0; swap < 2; swap++)
                for (int wps : {2, 4}) {
                    const int grid = p.multiProcessorCount * wps, iters = 4000;
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0);
                    hipEventCreate(&e1);
                    auto launch = [&]() {
                        if (swap) hipLaunchKernelGGL(enc_kernel<true>, dim3(grid), dim3(256), 0, 0, (const v4i*)d_a, (const v4i*)d_b, d_sink, iters);
                        else hipLaunchKernelGGL(enc_kernel<false>, dim3(grid), dim3(256), 0, 0, (const v4i*)d_a, (const v4i*)d_b, d_sink, iters);
                    };
                    for (int rep = 0; rep < 5; rep++) launch();
                    (void)hipEventRecord(e0);
                    for (int rep = 0; rep < 10; rep++) launch();
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    const double ops = 10.0 * grid * 4.0 * iters * 32.0 * 2.0 * 16 * 16 * 64;
                    printf("A = %-22s as src%c  waves/SIMD %d : %7.1f TOP/s  (%.3f of 5.0 POP/s)\n", names[mode], swap ? 'B' : 'A', wps, ops / (ms * 1e-3) / 1e12,
                           ops / (ms * 1e-3) / 5e15);
                }
        }
    return 0;
}
