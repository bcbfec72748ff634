// alloc_probe3.hip -- one large buffer as a reserved address range backed by physical chunks (hipMemCreate / hipMemMap) against
// one hipMalloc, right after another process released its memory.  usage: alloc_probe3 <total_GB> <chunk_MB | 0 = hipMalloc>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(uint64_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i; }
int main(int argc, char **argv)
{
    const size_t total = (size_t)(atof(argv[1]) * (1ull << 30)); const size_t chunk = (size_t)atol(argv[2]) << 20;
    double t0 = now(); CK(hipSetDevice(0)); CK(hipFree(0)); double t1 = now();
    void *ptr = nullptr;
    if (!chunk) { CK(hipMalloc(&ptr, total)); }
    else {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        CK(hipMemAddressReserve(&ptr, total, gran, nullptr, 0));
        for (size_t off = 0; off < total; off += chunk) {
            const size_t n = total - off < chunk ? total - off : chunk;
            hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, n, &prop, 0)); CK(hipMemMap((char *)ptr + off, n, 0, h, 0)); CK(hipMemRelease(h));
        }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(ptr, total, &acc, 1));
        printf("granularity %zu; ", gran);
    }
    double t2 = now();
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (uint64_t *)ptr, total / 8); CK(hipDeviceSynchronize()); double t3 = now();
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (uint64_t *)ptr, total / 8); CK(hipDeviceSynchronize()); double t4 = now();
    printf("init %.3f s | %.0f GB as %s: alloc %.3f s, first fill %.3f s, second fill %.3f s (%.0f GB/s)\n", t1 - t0, total / 1073741824.0, chunk ? "mapped chunks" : "hipMalloc", t2 - t1, t3 - t2, t4 - t3, total / 1e9 / (t4 - t3));
    return 0;
}
