// vram_probe.hip -- what a large device allocation costs a fresh process on this box right now: hipMalloc of <GB> (default 56), one byte written
// at either end, hipFree; prints the three times.  Run several times with pauses (tools/vram_probe.sh) to see how long the driver takes to hand
// out memory another process has just released (released VRAM is wiped before it is handed out again).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/vram_probe tools/vram_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? (size_t)atoll(argv[1]) : 56;
    double t0 = now();
    (void)hipSetDevice(0); (void)hipFree(0);
    double t1 = now();
    char *p = nullptr;
    if (hipMalloc((void **)&p, gb << 30) != hipSuccess) { printf("hipMalloc %zu GB failed\n", gb); return 1; }
    double t2 = now();
    (void)hipMemset(p, 1, 1); (void)hipMemset(p + (gb << 30) - 1, 1, 1); (void)hipDeviceSynchronize();
    double t3 = now();
    (void)hipFree(p);
    double t4 = now();
    printf("init %.3f s, hipMalloc %zu GB %.3f s, touch %.3f s, free %.3f s\n", t1 - t0, gb, t2 - t1, t3 - t2, t4 - t3);
    return 0;
}
