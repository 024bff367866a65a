// ubench_tile.hip -- one wave-tile of the fused kernel in isolation (registers only unless LDSRD), to price its parts:
//   8 x v_mfma_i32_32x32x32_i8 seeded from a constant vector, then the detect epilogue on the accumulators.
// EPI: 0 none, 1 canonical packed (pk_fma, pk_mul x2, pk_add, chain pk_add), 2 canonical plain, 3 fused-fma plain
//      (2 fma convert + 2 fma accumulate per sample), 4 plain convert-only
// LDSRD: A fragments re-read from LDS each iteration (4 x ds_read_b128).   RANDOM: random operand bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int EPI, bool LDSRD>
__global__ __launch_bounds__(256, 4) void k(const v4i* __restrict__ src, float* out, int iters)
{
    __shared__ v4i lds[4 * 256];
    v4i b[8], a[4];
    for (int i = 0; i < 8; i++) b[i] = src[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int i = 0; i < 4; i++) { a[i] = src[(i + 3) * 256 + threadIdx.x]; lds[i * 256 + threadIdx.x] = a[i]; }
    __syncthreads();
    v16i kc;
    for (int i = 0; i < 16; i++) kc[i] = 0x4B400000;
    asm volatile("" : "+v"(kc));
    v2f carry = {0.f, 0.f};
    float acc1 = 0.f;
    const v2f al = {4.9e-4f, 4.9e-4f}, bi = {-6192.f, -6192.f};
    for (int it = 0; it < iters; it++) {
        if (LDSRD) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = lds[i * 256 + ((threadIdx.x + it) & 255)];
        }
        v16i c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[0], kc, 0, 0, 0);
        v16i c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[4], kc, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < 4; m++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[m], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[4 + m], c1, 0, 0, 0);
        }
        const v16f fr = __builtin_bit_cast(v16f, c0), fi = __builtin_bit_cast(v16f, c1);
        if (EPI == 0) {
            carry[0] += fr[0] + fi[15];
        } else if (EPI == 1) {
            v2f s = carry;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v2f xr = __builtin_elementwise_fma(v2f{fr[2 * i], fr[2 * i + 1]}, al, bi);
                v2f xi = __builtin_elementwise_fma(v2f{fi[2 * i], fi[2 * i + 1]}, al, bi);
                v2f xx = xr * xr, yy = xi * xi;
                s = s + (xx + yy);
            }
            asm volatile("" : "+v"(s));
            carry = s;
        } else if (EPI == 2) {
            float s = acc1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float xr = __builtin_fmaf(fr[i], 4.9e-4f, -6192.f), xi = __builtin_fmaf(fi[i], 4.9e-4f, -6192.f);
                float xx = xr * xr, yy = xi * xi;
                s = s + (xx + yy);
            }
            asm volatile("" : "+v"(s));
            acc1 = s;
        } else if (EPI == 3) {
            float s = acc1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float xr = __builtin_fmaf(fr[i], 4.9e-4f, -6192.f), xi = __builtin_fmaf(fi[i], 4.9e-4f, -6192.f);
                s = __builtin_fmaf(xr, xr, s);
                s = __builtin_fmaf(xi, xi, s);
            }
            asm volatile("" : "+v"(s));
            acc1 = s;
        } else if (EPI == 4) {
            float s = acc1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float xr = fr[i] - 12582912.f, xi = fi[i] - 12582912.f;
                s = __builtin_fmaf(xr, xr, s);
                s = __builtin_fmaf(xi, xi, s);
            }
            asm volatile("" : "+v"(s));
            acc1 = s;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = carry[0] + carry[1] + acc1;
}

template <int EPI, bool LDSRD>
void run(const char* name, int waves_per_simd, const v4i* d_src, float* d_out)
{
    const int iters = 4000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<EPI, LDSRD>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, 200);
    (void)hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<EPI, LDSRD>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-28s lds=%d waves/SIMD=%d : ns per tile per SIMD  avg %.1f  best %.1f\n", name, (int)LDSRD, waves_per_simd,
           sum / 5 * 1e6 / iters / waves_per_simd, best * 1e6 / iters / waves_per_simd);
}

int main(int argc, char** argv)
{
    const bool random = argc > 1;
    const size_t n = 256 * 8 * 8 * 256;
    std::vector<v4i> h(n);
    srand(1);
    for (auto& x : h) for (int i = 0; i < 4; i++) x[i] = random ? (int)((unsigned)rand() * 2654435761u) & (i < 4 ? 0xF0F0F0F0 | 0x0F0F0F0F : 0) : 0x01010101;
    v4i* d_src; float* d_out;
    (void)hipMalloc(&d_src, n * sizeof(v4i)); (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    (void)hipMemcpy(d_src, h.data(), n * sizeof(v4i), hipMemcpyHostToDevice);
    printf("operands: %s\n", random ? "random bits" : "constant 0x01");
    for (int w : {2, 4}) {
        run<0, false>("mfma only", w, d_src, d_out);
        run<0, true>("mfma + lds frag reads", w, d_src, d_out);
        run<1, false>("canonical packed", w, d_src, d_out);
        run<2, false>("canonical plain", w, d_src, d_out);
        run<3, false>("fma-fused plain (4 op/sample)", w, d_src, d_out);
        run<4, false>("sub+fma plain (4 op/sample)", w, d_src, d_out);
        run<2, true>("canonical plain + lds", w, d_src, d_out);
    }
    return 0;
}
