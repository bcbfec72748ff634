// tools/slice_bw.hip -- what HBM delivers for the access pattern of union_kernel / assemble_kernel, without their table work.
// S "samples" hold nsub slices of W words each.  Workgroup j reads slice j of every sample (S pieces of W * 8 bytes); its
// waves take the samples in turn, G slices at a time (G lane groups of 64 / G lanes), every lane keeps U loads in flight.
//   layout 0: sample-major   (piece (s, j) at (s * nsub + j) * W: the pieces of one workgroup are a whole sample apart)
//   layout 1: bucket-major   (piece (s, j) at (((j >> 4) * S + s) * 16 + (j & 15)) * W: they are 16 * W words apart)
//   probe  1: one 8-byte LDS read at a data-dependent address per word (the hit path of the union table)
//   hipcc --offload-arch=gfx950 -O3 -o slice_bw tools/slice_bw.hip && ./slice_bw [S] [W] [lognsub] [layout] [probe] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef const uint64_t __attribute__((address_space(1))) *gw_t;

template <int G, int U, bool PROBE>
__global__ __launch_bounds__(1024) void slice_kernel(const uint64_t *words, int S, uint32_t W, uint32_t nsub, int layout, uint64_t *sink, unsigned char *matrix, uint32_t nout)
{
    __shared__ unsigned long long tab[4096];
    if (PROBE) { for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = i * 0x9E3779B97F4A7C15ull; __syncthreads(); }
    const uint64_t j = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int GL = 64 / G;
    const int g = lane / GL, gl = lane % GL;
    const int per = (S + nw - 1) / nw;
    const int wend = (wv + 1) * per < S ? (wv + 1) * per : S;
    uint64_t acc = 0;
    gw_t base = (gw_t)(uintptr_t)words;
    for (int s0 = wv * per; s0 < wend; s0 += G) {
        const int s = s0 + g;
        const bool on = s < wend;
        const uint64_t at = layout ? ((((j >> 4) * (uint64_t)S + (on ? s : s0)) << 4) + (j & 15)) * W : ((uint64_t)(on ? s : s0) * nsub + j) * W;
        for (uint32_t o = 0; o < W; o += GL * U) {
            uint64_t q[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t i = o + GL * u + gl; q[u] = (on && i < W) ? base[at + i] : 0ull; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (PROBE) { if (q[u]) acc ^= tab[(uint32_t)(q[u] >> 7) & 4095u]; }
                else acc ^= q[u];
            }
        }
        if (G == 1 && matrix) {                                // assemble_kernel's copy-out: nout bytes of sample s at column j * nout
            const uint64_t col = j * (uint64_t)nout;
            unsigned char *dst = matrix + (uint64_t)s * ((uint64_t)nsub * nout + 64) + col;
            const uint32_t head = (uint32_t)((16u - (col & 15u)) & 15u);
            if ((uint32_t)lane < head) dst[lane] = (unsigned char)acc;
            const uint32_t body = (nout - head) / 16u;
            const uint32_t x = (uint32_t)acc;
            for (uint32_t v = lane; v < body; v += 64) *reinterpret_cast<uint4 *>(dst + head + 16u * v) = make_uint4(x, x + v, x, x);
            const uint32_t done = head + body * 16u;
            if (done + lane < nout) dst[done + lane] = (unsigned char)acc;
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

template <int G, int U, bool PROBE>
static void run(const uint64_t *words, int S, uint32_t W, int lognsub, int layout, int threads, uint64_t *sink, unsigned char *matrix = nullptr, uint32_t nout = 0)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        if (rep == 1) (void)hipEventRecord(e0);
        hipLaunchKernelGGL((slice_kernel<G, U, PROBE>), dim3(1u << lognsub), dim3(threads), 0, 0, words, S, W, 1u << lognsub, layout, sink, matrix, nout);
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    const double gb = (double)S * (W * 8 + (matrix ? nout : 0)) * (double)(1u << lognsub) / 1e9;
    if (matrix) printf("+ %u-byte pieces written: ", nout);
    printf("G=%d U=%2d probe=%d threads=%4d: %7.2f ms  %6.2f TB/s\n", G, U, (int)PROBE, threads, ms, gb / ms);
}

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 1000;
    const uint32_t W = argc > 2 ? (uint32_t)atoi(argv[2]) : 300;
    const int lognsub = argc > 3 ? atoi(argv[3]) : 14;
    const int layout = argc > 4 ? atoi(argv[4]) : 0;
    const int threads = argc > 5 ? atoi(argv[5]) : 1024;
    const uint64_t n = (uint64_t)S * W << lognsub;
    uint64_t *words, *sink;
    if (hipMalloc((void **)&words, n * 8 + 4096) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(words, 0x5A, n * 8);
    printf("S=%d W=%u nsub=2^%d layout=%d : %.1f GB\n", S, W, lognsub, layout, n * 8 / 1e9);
    run<1, 5, false>(words, S, W, lognsub, layout, threads, sink);
    run<2, 10, false>(words, S, W, lognsub, layout, threads, sink);
    run<4, 10, false>(words, S, W, lognsub, layout, threads, sink);
    run<8, 10, false>(words, S, W, lognsub, layout, threads, sink);
    run<16, 10, false>(words, S, W, lognsub, layout, threads, sink);
    run<2, 10, true>(words, S, W, lognsub, layout, threads, sink);
    run<8, 10, true>(words, S, W, lognsub, layout, threads, sink);
    const uint32_t nout = argc > 6 ? (uint32_t)atoi(argv[6]) : 1380;
    unsigned char *matrix;
    if (hipMalloc((void **)&matrix, (uint64_t)S * (((uint64_t)nout << lognsub) + 64)) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    run<1, 5, false>(words, S, W, lognsub, layout, threads, sink, matrix, nout);
    run<1, 5, true>(words, S, W, lognsub, layout, threads, sink, matrix, nout);
    return 0;
}
