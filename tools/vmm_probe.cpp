// vmm_probe.cpp -- does this HIP stack map ONE physical allocation at two consecutive virtual ranges (a ring without wrap-around
// for kernels that read a contiguous window)?  hipcc -O2 tools/vmm_probe.cpp -o tools/vmm_probe; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
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

int main()
{
    int dev = 0, vmm = 0;
    TRY(hipSetDevice(dev));
    TRY(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("hipDeviceAttributeVirtualMemoryManagementSupported = %d\n", vmm);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    const size_t bytes = ((size_t)512 << 20) / gran * gran;   // 512 MiB physical
    hipMemGenericAllocationHandle_t h;
    TRY(hipMemCreate(&h, bytes, &prop, 0));
    void* va = nullptr;
    TRY(hipMemAddressReserve(&va, 2 * bytes, gran, nullptr, 0));
    TRY(hipMemMap(va, bytes, 0, h, 0));
    TRY(hipMemMap((char*)va + bytes, bytes, 0, h, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    TRY(hipMemSetAccess(va, 2 * bytes, &acc, 1));
    float* p = (float*)va;
    const size_t n = bytes / 4;
    // write through the SECOND mapping across the wrap: [n - 1000, n + 1000) of the double range
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, p + n - 1000, (size_t)2000, 7.0f);
    TRY(hipDeviceSynchronize());
    std::vector<float> a(1000), b(1000);
    TRY(hipMemcpy(a.data(), p + n - 1000, 4000, hipMemcpyDeviceToHost));   // the tail through mapping 1
    TRY(hipMemcpy(b.data(), p, 4000, hipMemcpyDeviceToHost));              // the head through mapping 1 = what was written at n .. n + 1000
    bool ok = true;
    for (int i = 0; i < 1000; i++) ok = ok && a[i] == 7.0f + i && b[i] == 7.0f + 1000 + i;
    printf("wrap-around write visible through the first mapping: %s\n", ok ? "yes" : "NO");
    // a hipMemcpyAsync D2D and a memset across the seam
    TRY(hipMemsetAsync(p + n - 256, 0, 2048, 0));
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(b.data(), p, 1024, hipMemcpyDeviceToHost));
    bool ok2 = true;
    for (int i = 0; i < 256; i++) ok2 = ok2 && b[i] == 0.0f;
    printf("memset across the seam: %s\n", ok2 ? "yes" : "NO");
    TRY(hipMemUnmap(va, bytes));
    TRY(hipMemUnmap((char*)va + bytes, bytes));
    TRY(hipMemAddressFree(va, 2 * bytes));
    TRY(hipMemRelease(h));
    printf("%s\n", ok && ok2 ? "VMM RING OK" : "VMM RING BROKEN");
    return ok && ok2 ? 0 : 1;
}
