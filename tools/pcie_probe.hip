// pcie_probe.hip -- H2D / D2H rate from pinned memory allocated on either NUMA node, one and two streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sched.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void bind(int lo, int hi) { cpu_set_t s; CPU_ZERO(&s); for (int c = lo; c <= hi; c++) CPU_SET(c, &s); sched_setaffinity(0, sizeof s, &s); }
int main()
{
    (void)hipSetDevice(0);
    const size_t N = 1ull << 30, P = 8u << 20;
    void *d = nullptr; (void)hipMalloc(&d, N);
    hipStream_t st[2]; (void)hipStreamCreate(&st[0]); (void)hipStreamCreate(&st[1]);
    for (int node = 0; node < 2; node++) {
        bind(node * 64, node * 64 + 63);
        char *h = nullptr; (void)hipHostMalloc((void **)&h, N, hipHostMallocDefault); memset(h, 1, N);
        for (int ns = 1; ns <= 2; ns++) for (int dir = 0; dir < 2; dir++) {
            double t0 = now();
            for (size_t o = 0, i = 0; o < N; o += P, i++) (void)hipMemcpyAsync(dir ? (char *)h + o : (char *)d + o, dir ? (char *)d + o : (char *)h + o, P, dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, st[i % ns]);
            (void)hipStreamSynchronize(st[0]); (void)hipStreamSynchronize(st[1]);
            printf("pinned on node %d, %d stream(s), %s: %.1f GB/s\n", node, ns, dir ? "D2H" : "H2D", N / 1e9 / (now() - t0));
        }
        (void)hipHostFree(h);
    }
    FILE *f = fopen("/sys/class/drm/card0/device/numa_node", "r"); if (f) { int n = -9; if (fscanf(f, "%d", &n) == 1) printf("card0 numa_node %d\n", n); fclose(f); }
    return 0;
}
