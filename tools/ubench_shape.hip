// ubench_shape.hip -- int8 MFMA shape vs sustained rate on random operands (power/clock effect), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 4) void k(const v4i* __restrict__ src, int* out, int iters)
{
    v4i a[4], b[8];
    for (int i = 0; i < 8; i++) b[i] = src[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int i = 0; i < 4; i++) a[i] = src[(i + 3) * 256 + threadIdx.x];
    int acc = 0;
    for (int it = 0; it < iters; it++) {
        if (SHAPE == 32) {
            v16i c0 = {0}, c1 = {0};
#pragma unroll
            for (int m = 0; m < 4; m++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[m], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[4 + m], c1, 0, 0, 0);
            }
            acc += c0[0] + c1[15];
        } else {
            v4i c[8];
#pragma unroll
            for (int t = 0; t < 8; t++) c[t] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int t = 0; t < 8; t++)
                    c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + m) & 3], b[(t + 2 * m) & 7], c[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; t++) acc += c[t][0] + c[t][3];
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int SHAPE>
void run(const char* name, const v4i* d_src, int* d_out)
{
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, 200);
    (void)hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double macs = 262144.0 * iters * blocks * 4;  // per launch
    printf("%-14s ns per 262144-MAC tile per SIMD: avg %.1f best %.1f  -> %.2f POP/s\n", name, sum / 5 * 1e6 / iters / 4,
           best * 1e6 / iters / 4, 2 * macs / (sum / 5 * 1e-3) / 1e15);
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0 const, 1 random bytes, 2 voltage-like (16*nibble) x random
    const size_t n = 256 * 8 * 8 * 256;
    std::vector<v4i> h(n);
    srand(1);
    for (size_t j = 0; j < n; j++)
        for (int i = 0; i < 4; i++) {
            unsigned r = (unsigned)rand() * 2654435761u;
            h[j][i] = mode == 0 ? 0x01010101 : (int)r;
        }
    if (mode == 2)  // first 7*256 vectors feed a[]: make them 16*nibble bytes
        for (size_t j = 0; j < 7 * 256; j++) for (int i = 0; i < 4; i++) h[j][i] &= 0xF0F0F0F0;
    if (mode == 3)  // a[]: sign-extended nibbles (-8..7) in every byte
        for (size_t j = 0; j < 7 * 256; j++) for (int i = 0; i < 4; i++) {
            unsigned w = (unsigned)h[j][i] & 0x0F0F0F0Fu;
            h[j][i] = (int)((((w ^ 0x88888888u) - 0x08080808u) ^ 0x80808080u));
        }
    if (mode == 4)  // a[]: realistic small voltages: gaussian-like nibbles in [-3, 3], times 16
        for (size_t j = 0; j < 7 * 256; j++) for (int i = 0; i < 4; i++) {
            unsigned w = 0;
            for (int b = 0; b < 4; b++) { int v = (rand() % 3) + (rand() % 3) + (rand() % 3) - 3; w |= ((unsigned)(v * 16) & 0xFFu) << (8 * b); }
            h[j][i] = (int)w;
        }
    v4i* d_src; int* d_out;
    (void)hipMalloc(&d_src, n * sizeof(v4i)); (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(int));
    (void)hipMemcpy(d_src, h.data(), n * sizeof(v4i), hipMemcpyHostToDevice);
    printf("operands mode %d\n", mode);
    for (int rep = 0; rep < 2; rep++) {
        run<32>("32x32x32 x8", d_src, d_out);
        run<16>("16x16x64 x16", d_src, d_out);
    }
    return 0;
}
