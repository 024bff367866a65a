// ubench_expand.hip -- variants of the stand-alone 4-bit -> int8 expand (a1), 1 GiB in / 2 GiB out, HBM-bound.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_expand.hip -o /tmp/ue && /tmp/ue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned sext4x4(unsigned nib) { return ((nib ^ 0x88888888u) - 0x08080808u) ^ 0x80808080u; }
__device__ __forceinline__ void expand16(const v4i v, v4i& o0, v4i& o1)
{
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const unsigned w = (unsigned)v[d];
        const unsigned hi = sext4x4((w >> 4) & 0x0F0F0F0Fu), lo = sext4x4(w & 0x0F0F0F0Fu);
        const unsigned e0 = __builtin_amdgcn_perm(lo, hi, 0x05010400u), e1 = __builtin_amdgcn_perm(lo, hi, 0x07030602u);
        if (d < 2) {
            o0[2 * d] = (int)e0;
            o0[2 * d + 1] = (int)e1;
        } else {
            o1[2 * (d - 2)] = (int)e0;
            o1[2 * (d - 2) + 1] = (int)e1;
        }
    }
}
// LD: 0 plain, 1 nontemporal ; ST: 0 plain, 1 nontemporal ; MAP: 0 grid-stride, 1 one vector per thread, 2 contiguous chunk per WG
template <int LD, int ST, int MAP>
__global__ __launch_bounds__(256) void k(const v4i* __restrict__ in, v4i* __restrict__ out, size_t n_vec, size_t per_wg)
{
    auto body = [&](size_t i) {
        const v4i v = LD ? __builtin_nontemporal_load(in + i) : in[i];
        v4i o0, o1;
        expand16(v, o0, o1);
        if (ST) {
            __builtin_nontemporal_store(o0, out + 2 * i);
            __builtin_nontemporal_store(o1, out + 2 * i + 1);
        } else {
            out[2 * i] = o0;
            out[2 * i + 1] = o1;
        }
    };
    if (MAP == 0) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) body(i);
    } else if (MAP == 1) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n_vec) body(i);
    } else {
        const size_t a = (size_t)blockIdx.x * per_wg, b = a + per_wg < n_vec ? a + per_wg : n_vec;
        for (size_t i = a + threadIdx.x; i < b; i += 256) body(i);
    }
}
// each lane writes 32 contiguous bytes?  no: lanes write o0,o1 adjacent (32 B per lane, 2 x 16 B stores at stride 32).
// variant: wave-coalesced stores -- lane l handles input vector pair so that each store instruction covers 1 KiB contiguous
template <int ST>
__global__ __launch_bounds__(256) void k_swz(const v4i* __restrict__ in, v4i* __restrict__ out, size_t n_vec)
{
    // a wave takes 64 consecutive input vectors (1 KiB) -> 2 KiB out; lane l loads vector l, but the two output halves
    // are exchanged through ds_bpermute-free trick: lane l stores o0 of vector l at 2l and o1 at 2l+1 (same as above) --
    // instead load so that stores are contiguous: lane l expands HALF of vectors (l/2 .. ) : vector j = base + l/2,
    // half h = l&1 -> one 16-B store at out[2*j + h] = contiguous 16 B per lane across the wave (1 KiB per instruction),
    // two passes cover 64 vectors; the load is an 8-byte load (half a vector) per lane: contiguous 512 B per instruction.
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * blockDim.x / 64;
    for (size_t base = wave * 64; base < n_vec; base += n_waves * 64) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t half = (base + p * 32) * 2 + lane;   // index in units of 8 input bytes
            if (half >= n_vec * 2) break;
            const v2i v = __builtin_nontemporal_load(reinterpret_cast<const v2i*>(in) + half);
            v4i o;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const unsigned w = (unsigned)(d ? v.y : v.x);
                const unsigned hi = sext4x4((w >> 4) & 0x0F0F0F0Fu), lo = sext4x4(w & 0x0F0F0F0Fu);
                o[2 * d] = (int)__builtin_amdgcn_perm(lo, hi, 0x05010400u);
                o[2 * d + 1] = (int)__builtin_amdgcn_perm(lo, hi, 0x07030602u);
            }
            if (ST)
                __builtin_nontemporal_store(o, out + half);
            else
                out[half] = o;
        }
    }
}

template <typename F>
float time_it(F f, int reps = 20)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ms;
    for (int i = 0; i < reps; i++) {
        (void)hipEventRecord(a);
        f();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float t;
        (void)hipEventElapsedTime(&t, a, b);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}
#include <algorithm>
int main()
{
    const size_t nbytes = (size_t)1 << 30, n_vec = nbytes / 16;
    v4i *in, *out;
    (void)hipMalloc(&in, nbytes);
    (void)hipMalloc(&out, 2 * nbytes);
    (void)hipMemset(in, 0x5a, nbytes);
    auto report = [&](const char* name, float ms) { printf("%-52s %.3f ms  %.2f TB/s algorithmic (3 B per input byte)\n", name, ms, 3.0 * nbytes / (ms * 1e-3) / 1e12); };
    for (int rnd = 0; rnd < 2; rnd++) {
        for (unsigned grid : {16384u, 65536u, 262144u}) {
            char nm[96];
            snprintf(nm, 96, "grid-stride nt-load nt-store grid %u", grid);
            report(nm, time_it([&] { hipLaunchKernelGGL((k<1, 1, 0>), dim3(grid), dim3(256), 0, 0, in, out, n_vec, 0); }));
        }
        report("grid-stride plain load, plain store, 65536", time_it([&] { hipLaunchKernelGGL((k<0, 0, 0>), dim3(65536), dim3(256), 0, 0, in, out, n_vec, 0); }));
        report("grid-stride nt load, plain store, 65536", time_it([&] { hipLaunchKernelGGL((k<1, 0, 0>), dim3(65536), dim3(256), 0, 0, in, out, n_vec, 0); }));
        report("grid-stride plain load, nt store, 65536", time_it([&] { hipLaunchKernelGGL((k<0, 1, 0>), dim3(65536), dim3(256), 0, 0, in, out, n_vec, 0); }));
        report("one vector per thread, nt/nt", time_it([&] { hipLaunchKernelGGL((k<1, 1, 1>), dim3((unsigned)(n_vec / 256)), dim3(256), 0, 0, in, out, n_vec, 0); }));
        report("one vector per thread, plain/plain", time_it([&] { hipLaunchKernelGGL((k<0, 0, 1>), dim3((unsigned)(n_vec / 256)), dim3(256), 0, 0, in, out, n_vec, 0); }));
        for (unsigned wgs : {2048u, 8192u, 32768u}) {
            char nm[96];
            snprintf(nm, 96, "contiguous chunk per workgroup nt/nt, %u workgroups", wgs);
            report(nm, time_it([&] { hipLaunchKernelGGL((k<1, 1, 2>), dim3(wgs), dim3(256), 0, 0, in, out, n_vec, n_vec / wgs); }));
        }
        report("wave-contiguous 16-B stores (8-B loads), nt, 65536", time_it([&] { hipLaunchKernelGGL((k_swz<1>), dim3(65536), dim3(256), 0, 0, in, out, n_vec); }));
        report("wave-contiguous 16-B stores (8-B loads), plain, 65536", time_it([&] { hipLaunchKernelGGL((k_swz<0>), dim3(65536), dim3(256), 0, 0, in, out, n_vec); }));
    }
    return 0;
}
