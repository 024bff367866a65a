// ubench_order.hip -- does the ORDER of 16 x v_mfma_i32_16x16x64_i8 (same operands, same MACs) change the sustained
// rate on random data?  order 0: A-stationary (current kernel: same voltage fragment, weights change);
// order 1: B-stationary (same weight fragment consecutively, voltage fragment changes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(256, 4) void k(const v4i* __restrict__ src, int* out, int iters)
{
    v4i b[8], a[4];
    for (int i = 0; i < 8; i++) b[i] = src[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int i = 0; i < 4; i++) a[i] = src[(i + 3) * 256 + threadIdx.x] & 0xF0F0F0F0;
    int acc = 0;
    for (int it = 0; it < iters; it++) {
        v4i c[2][2][2];  // [rt][ct][rho]
#pragma unroll
        for (int i = 0; i < 8; i++) c[i >> 2][(i >> 1) & 1][i & 1] = v4i{0, 0, 0, 0};
        if (ORDER == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++)
#pragma unroll
                    for (int s = 0; s < 2; s++)
#pragma unroll
                        for (int rho = 0; rho < 2; rho++)
                            c[rt][ct][rho] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt + s], b[4 * ct + 2 * rho + s], c[rt][ct][rho], 0, 0, 0);
        } else {
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int rho = 0; rho < 2; rho++)
#pragma unroll
                        for (int rt = 0; rt < 2; rt++)
                            c[rt][ct][rho] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2 * rt + s], b[4 * ct + 2 * rho + s], c[rt][ct][rho], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) acc += c[i >> 2][(i >> 1) & 1][i & 1][0] + c[i >> 2][(i >> 1) & 1][i & 1][3];
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int ORDER>
void run(const char* name, const v4i* d_src, int* d_out)
{
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ORDER>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, 200);
    (void)hipDeviceSynchronize();
    float sum = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<ORDER>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms;
    }
    printf("%-22s ns per 262144-MAC tile per SIMD: %.1f\n", name, sum / 5 * 1e6 / iters / 4);
}

int main()
{
    const size_t n = 256 * 8 * 8 * 256;
    std::vector<v4i> h(n);
    srand(1);
    for (auto& x : h) for (int i = 0; i < 4; i++) x[i] = (int)((unsigned)rand() * 2654435761u);
    v4i* d_src; int* d_out;
    (void)hipMalloc(&d_src, n * sizeof(v4i)); (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(int));
    (void)hipMemcpy(d_src, h.data(), n * sizeof(v4i), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) { run<0>("A-stationary (current)", d_src, d_out); run<1>("B-stationary", d_src, d_out); }
    return 0;
}
