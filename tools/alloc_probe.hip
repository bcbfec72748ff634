// alloc_probe.hip -- what device / pinned / page-cache memory costs on this box: hipMalloc + first touch + hipFree by size,
// hipHostMalloc, tmpfs file population through write(), mmap + parallel memcpy, fallocate.
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_probe tools/alloc_probe.hip -lpthread ; run: ./alloc_probe [dir]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    double t0 = now();
    hipSetDevice(0); hipFree(0);
    printf("hip init %.3f s\n", now() - t0);
    for (size_t gb : {1, 8, 48, 48, 22}) {
        void *p = nullptr;
        t0 = now(); hipError_t e = hipMalloc(&p, gb << 30); double t1 = now();
        if (e != hipSuccess) { printf("hipMalloc %zu GB failed\n", gb); continue; }
        hipMemset(p, 1, gb << 30); hipDeviceSynchronize(); double t2 = now();
        hipMemset(p, 2, gb << 30); hipDeviceSynchronize(); double t3 = now();
        hipFree(p); double t4 = now();
        printf("hipMalloc %2zu GB: alloc %.3f s, first memset %.3f s, second memset %.3f s, free %.3f s\n", gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    for (size_t mb : {8, 64, 512}) {
        void *p = nullptr;
        t0 = now(); hipHostMalloc(&p, mb << 20, hipHostMallocDefault); double t1 = now();
        memset(p, 1, mb << 20); double t2 = now();
        hipHostFree(p); double t3 = now();
        printf("hipHostMalloc %3zu MB: alloc %.4f s, touch %.4f s, free %.4f s\n", mb, t1 - t0, t2 - t1, t3 - t2);
    }
    const size_t N = 4ull << 30;
    std::vector<char> src(64 << 20, 'x');
    {   // write()
        std::string f = dir + "/probe_w"; int fd = open(f.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        t0 = now(); for (size_t o = 0; o < N; o += src.size()) if (write(fd, src.data(), src.size()) < 0) break;
        printf("write() 4 GB to %s: %.3f s\n", dir.c_str(), now() - t0); close(fd); unlink(f.c_str());
    }
    for (int nt : {1, 4, 16}) {   // mmap + memcpy by nt threads
        std::string f = dir + "/probe_m"; int fd = open(f.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        t0 = now(); if (ftruncate(fd, N)) return 1; char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { for (size_t o = (size_t)t * src.size(); o < N; o += nt * src.size()) memcpy(m + o, src.data(), src.size()); });
        for (auto &x : th) x.join();
        printf("mmap + memcpy, %2d threads, 4 GB: %.3f s\n", nt, now() - t0); munmap(m, N); close(fd); unlink(f.c_str());
    }
    {   // fallocate then mmap copy
        std::string f = dir + "/probe_f"; int fd = open(f.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        t0 = now(); int r = posix_fallocate(fd, 0, N); double t1 = now();
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::thread> th; const int nt = 16;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { for (size_t o = (size_t)t * src.size(); o < N; o += nt * src.size()) memcpy(m + o, src.data(), src.size()); });
        for (auto &x : th) x.join();
        printf("fallocate (%d) %.3f s, then mmap + memcpy 16 threads %.3f s\n", r, t1 - t0, now() - t1); munmap(m, N); close(fd); unlink(f.c_str());
    }
    {   // 16 files written in parallel
        t0 = now(); std::vector<std::thread> th;
        for (int t = 0; t < 16; t++) th.emplace_back([&, t]() { std::string f = dir + "/probe_p" + std::to_string(t); int fd = open(f.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            for (size_t o = 0; o < N / 16; o += src.size()) if (write(fd, src.data(), src.size()) < 0) break; close(fd); unlink(f.c_str()); });
        for (auto &x : th) x.join();
        printf("write() 16 files x 256 MB in parallel: %.3f s\n", now() - t0);
    }
    {   // anonymous memory: malloc + touch 1 GB
        t0 = now(); char *p = (char *)malloc(1ull << 30); memset(p, 1, 1ull << 30); printf("malloc + touch 1 GB: %.3f s\n", now() - t0); free(p);
    }
    return 0;
}
