// alloc_probe2.hip -- cost of device allocations right after another process released its memory.
// usage: alloc_probe2 <chunk_GB> <n_chunks> [touch]   allocates n chunks of chunk_GB each, timing every hipMalloc; exits without freeing
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 1.0; const int n = argc > 2 ? atoi(argv[2]) : 1; const bool touch = argc > 3;
    double t0 = now();
    (void)hipSetDevice(0); (void)hipFree(0);
    double t1 = now();
    size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot);
    double worst = 0, sum = 0;
    std::vector<void *> ps;
    for (int i = 0; i < n; i++) {
        void *p = nullptr; double a = now();
        if (hipMalloc(&p, (size_t)(gb * (1ull << 30))) != hipSuccess) { printf("  alloc %d failed\n", i); break; }
        double d = now() - a; sum += d; if (d > worst) worst = d;
        ps.push_back(p);
    }
    double t2 = now();
    if (touch) { for (auto p : ps) (void)hipMemsetAsync(p, 1, (size_t)(gb * (1ull << 30)), 0); (void)hipDeviceSynchronize(); }
    double t3 = now();
    printf("init %.3f s (free at start %.1f GB) | %d x %.2f GB: total %.3f s, worst %.3f s | touch %.3f s\n", t1 - t0, fr / 1073741824.0, n, gb, sum, worst, t3 - t2);
    return 0;
}
