// ubench_tile16.hip -- composite wave-tile cost with the two int8 MFMA shapes (random operands, 4 waves/SIMD):
//   S32: 8 x 32x32x32 + canonical detect on 2x16 accumulators          (the round-1 kernel's tile)
//   S16: 16 x 16x16x64 + canonical detect on 16x(2x4... ) accumulators   (same MACs, same 96 VALU ops, same LDS bytes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float det(float fr, float fi)
{
    const float x = __builtin_fmaf(fr, 4.9e-4f, -6192.f), y = __builtin_fmaf(fi, 4.9e-4f, -6192.f);
    const float xx = x * x, yy = y * y;
    return xx + yy;
}

template <int SHAPE, bool LDSRD, bool EPI>
__global__ __launch_bounds__(256, 4) void k(const v4i* __restrict__ src, float* out, int iters)
{
    __shared__ v4i lds[4 * 256];
    v4i b[8], a[4];
    for (int i = 0; i < 8; i++) b[i] = src[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int i = 0; i < 4; i++) { a[i] = src[(i + 3) * 256 + threadIdx.x] & 0xF0F0F0F0; lds[i * 256 + threadIdx.x] = a[i]; }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    if (SHAPE == 32) {
        v16i kc; for (int i = 0; i < 16; i++) kc[i] = 0x4B400000;
        asm volatile("" : "+v"(kc));
        for (int it = 0; it < iters; it++) {
            if (LDSRD) { _Pragma("unroll") for (int i = 0; i < 4; i++) a[i] = lds[i * 256 + ((threadIdx.x + it) & 255)]; }
            v16i c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[0], kc, 0, 0, 0);
            v16i c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[4], kc, 0, 0, 0);
            _Pragma("unroll") for (int m = 1; m < 4; m++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[m], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[4 + m], c1, 0, 0, 0);
            }
            const v16f fr = __builtin_bit_cast(v16f, c0), fi = __builtin_bit_cast(v16f, c1);
            if (EPI) {
                _Pragma("unroll") for (int i = 0; i < 8; i++) { s0 = s0 + det(fr[2 * i], fi[2 * i]); s1 = s1 + det(fr[2 * i + 1], fi[2 * i + 1]); }
            } else { s0 += fr[0]; s1 += fi[15]; }
            asm volatile("" : "+v"(s0), "+v"(s1));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        v4i kc = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
        asm volatile("" : "+v"(kc));
        for (int it = 0; it < iters; it++) {
            if (LDSRD) { _Pragma("unroll") for (int i = 0; i < 4; i++) a[i] = lds[i * 256 + ((threadIdx.x + it) & 255)]; }
            // 2 row tiles (a[0..1], a[2..3]: k-steps re|im) x 2 column tiles (b[0..3], b[4..7]: re-row k0,k1, im-row k0,k1)
            _Pragma("unroll") for (int rt = 0; rt < 2; rt++) {
                _Pragma("unroll") for (int ct = 0; ct < 2; ct++) {
                    v4i cr = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt], b[4 * ct], kc, 0, 0, 0);
                    v4i ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt], b[4 * ct + 2], kc, 0, 0, 0);
                    cr = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt + 1], b[4 * ct + 1], cr, 0, 0, 0);
                    ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt + 1], b[4 * ct + 3], ci, 0, 0, 0);
                    const v4f fr = __builtin_bit_cast(v4f, cr), fi = __builtin_bit_cast(v4f, ci);
                    if (EPI) {
                        float& s = ct ? s1 : s0;
                        _Pragma("unroll") for (int i = 0; i < 4; i++) s = s + det(fr[i], fi[i]);
                    } else { s0 += fr[0]; s1 += fi[3]; }
                }
            }
            asm volatile("" : "+v"(s0), "+v"(s1));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1;
}

template <int SHAPE, bool LDSRD, bool EPI>
void run(const char* name, const v4i* d_src, float* d_out)
{
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, LDSRD, EPI>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, 200);
    (void)hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDSRD, EPI>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-34s ns per tile per SIMD  avg %.1f  best %.1f\n", name, sum / 5 * 1e6 / iters / 4, best * 1e6 / iters / 4);
}

int main()
{
    const size_t n = 256 * 8 * 8 * 256;
    std::vector<v4i> h(n);
    srand(1);
    for (auto& x : h) for (int i = 0; i < 4; i++) x[i] = (int)((unsigned)rand() * 2654435761u);
    v4i* d_src; float* d_out;
    (void)hipMalloc(&d_src, n * sizeof(v4i)); (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    (void)hipMemcpy(d_src, h.data(), n * sizeof(v4i), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        run<32, false, false>("32x32x32 mfma only", d_src, d_out);
        run<16, false, false>("16x16x64 mfma only", d_src, d_out);
        run<32, true, true>("32x32x32 + lds + canonical detect", d_src, d_out);
        run<16, true, true>("16x16x64 + lds + canonical detect", d_src, d_out);
    }
    return 0;
}
