// host_mem_probe.hip -- cost of first-touching anonymous host memory in a process that holds a HIP context (1 GB each way)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(const char *what)
{
    const size_t N = 1ull << 30;
    double t0 = now();
    volatile char *p = (volatile char *)malloc(N); for (size_t i = 0; i < N; i += 4096) p[i] = 1; double t1 = now();
    free((void *)p); double t2 = now();
    char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(m, N, MADV_HUGEPAGE);
    for (size_t i = 0; i < N; i += 4096) ((volatile char *)m)[i] = 1; double t3 = now(); munmap(m, N); double t4 = now();
    char *q = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); double t5 = now(); munmap(q, N);
    // 16 threads first-touching disjoint parts of one 1 GB block
    char *r = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); double t6 = now();
    std::vector<std::thread> th; for (int t = 0; t < 16; t++) th.emplace_back([=]() { for (size_t i = (N / 16) * t; i < (N / 16) * (t + 1); i += 4096) ((volatile char *)r)[i] = 1; });
    for (auto &x : th) x.join(); double t7 = now(); munmap(r, N);
    printf("%-28s malloc+touch %.3f s, free %.3f s | MADV_HUGEPAGE touch %.3f s, munmap %.3f s | MAP_POPULATE %.3f s | 16 threads touch %.3f s\n", what, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t7 - t6);
}
int main()
{
    touch("before hip init");
    (void)hipSetDevice(0); (void)hipFree(0);
    touch("after hip init");
    void *d = nullptr; (void)hipMalloc(&d, 20ull << 30); (void)hipMemset(d, 0, 20ull << 30); (void)hipDeviceSynchronize();
    touch("after 20 GB hipMalloc");
    void *h = nullptr; (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    touch("after hipHostMalloc 64 MB");
    std::vector<char> v(200 << 20, 1); (void)hipMemcpy(d, v.data(), v.size(), hipMemcpyHostToDevice);
    touch("after pageable H2D 200 MB");
    return 0;
}
