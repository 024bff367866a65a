// vmm_probe.cpp -- does this HIP stack map ONE physical allocation at two consecutive virtual ranges (a ring without wrap-around
// for kernels that read a contiguous window), and does it stay correct when such rings are created and destroyed repeatedly?
//   hipcc -O2 --offload-arch=gfx950 tools/vmm_probe.cpp -o tools/vmm_probe; ./tools/vmm_probe <mode> on the GPU box
//   mode 0: create / use / destroy, 6 times in a row      1: + a hipMalloc / hipFree between two rings
//   mode 2: + hipDeviceSynchronize after the unmaps       3: the virtual range is never reused (address hints move upwards)
//   mode 4: the virtual range is reserved once and kept; only the physical memory is created / mapped / unmapped / released
//   mode 5: ONE arena of address space reserved once; every ring takes the next, never-used stretch of it (no address is ever mapped twice)
//   mode 6: as 5, and the rings' physical memory is never released either (tells stale translations from stale physical pages)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define TRY(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("FAIL %s: %s\n", #x, hipGetErrorString(e_));                                 \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

__global__ void fill(float* p, size_t n, float base)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = base + (float)i;
}
__global__ void check(const float* p, size_t n, float base, int* bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != base + (float)i) atomicAdd(bad, 1);
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    int dev = 0, vmm = 0;
    TRY(hipSetDevice(dev));
    TRY(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("mode %d: VMM supported %d, granularity %zu\n", mode, vmm, gran);
    const size_t bytes = ((size_t)7 << 20) / gran * gran;
    const size_t n = bytes / 4;
    float* src = nullptr;
    int* d_bad = nullptr;
    TRY(hipMalloc(&src, bytes));
    TRY(hipMalloc(&d_bad, 4));
    void* kept_va = nullptr;
    char* hint = nullptr;
    char* arena = nullptr;
    if (mode >= 5) TRY(hipMemAddressReserve((void**)&arena, 64 * bytes, gran, nullptr, 0));
    int fails = 0;
    for (int it = 0; it < 6; it++) {
        hipMemGenericAllocationHandle_t h;
        TRY(hipMemCreate(&h, bytes, &prop, 0));
        void* va = kept_va;
        if (mode >= 5) va = arena + (size_t)it * 2 * bytes;
        if (!va) {
            TRY(hipMemAddressReserve(&va, 2 * bytes, gran, hint, 0));
            if (mode == 4) kept_va = va;
        }
        if (mode == 3) hint = (char*)va + 4 * bytes;
        TRY(hipMemMap(va, bytes, 0, h, 0));
        TRY(hipMemMap((char*)va + bytes, bytes, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        TRY(hipMemSetAccess(va, 2 * bytes, &acc, 1));
        float* p = (float*)va;
        const float base = 1000.0f * (it + 1);
        // the source pattern by a kernel into plain memory, then a D2D copy into the ring ACROSS the seam (as the DM stage's feed does)
        hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, src, n, base);
        const size_t off = n - 1000;                      // start 1000 floats before the seam
        TRY(hipMemcpyAsync(p + off, src, bytes, hipMemcpyDeviceToDevice, 0));
        // a kernel reads the same window back through the double mapping
        TRY(hipMemsetAsync(d_bad, 0, 4, 0));
        hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, 0, p + off, n, base, d_bad);
        int bad = -1;
        TRY(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        // ... and the host through the FIRST mapping only: what was written past the seam sits at the ring's start
        std::vector<float> head(16);
        TRY(hipMemcpy(head.data(), p, 64, hipMemcpyDeviceToHost));
        const bool ok = bad == 0 && head[0] == base + 1000.0f;
        printf("  ring %d at %p: kernel sees %d wrong floats, head[0] = %.0f (want %.0f): %s\n", it, va, bad, head[0], base + 1000.0f, ok ? "ok" : "BROKEN");
        fails += !ok;
        TRY(hipDeviceSynchronize());
        TRY(hipMemUnmap(va, bytes));
        TRY(hipMemUnmap((char*)va + bytes, bytes));
        if (mode == 2) TRY(hipDeviceSynchronize());
        if (mode < 4) TRY(hipMemAddressFree(va, 2 * bytes));
        if (mode != 6) TRY(hipMemRelease(h));
        if (mode == 1) {
            void* t = nullptr;
            TRY(hipMalloc(&t, 1 << 20));
            TRY(hipFree(t));
        }
    }
    printf("mode %d: %s\n", mode, fails ? "BROKEN" : "ALL OK");
    return fails ? 1 : 0;
}
