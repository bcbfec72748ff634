// tools/scatter_bw.hip -- what HBM delivers for the extraction kernel's memory pattern, without any of its arithmetic.
// One 512-thread workgroup = one tile of 8 192 positions of one sample: it reads the tile's 8 KB of bases (streaming) and
// writes 8 192 eight-byte words as CHUNK-word pieces, one piece into each of 8192/CHUNK bucket regions of that sample
// (regions of `cap` words, `cap * 8` bytes apart) at the tile's place in the region -- the copy-out of extract_kernel with
// the cursors replaced by arithmetic.  CHUNK = 8192 is a plain streaming write.  mode bit 0: regions start at odd word offsets
// (chunks straddle 64-B / 128-B lines, as the real cursors make them); bit 1: extract_kernel's block -> (sample, tile) order.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_bw tools/scatter_bw.hip && ./scatter_bw [samples] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int CHUNK>
__global__ __launch_bounds__(512) void scatter_kernel(const uint8_t *bases, uint64_t *words, int tiles, uint32_t cap, int skew)
{
    const uint64_t L = blockIdx.x;
    int sample = (int)(L / tiles), tile = (int)(L % tiles);
    if (skew & 2) {                                     // extract_kernel's order: block b -> XCD b % 8 -> sample b % 8 of its group of eight, tiles in turn
        const uint64_t per_group = 8ull * tiles;
        sample = (int)((L / per_group) * 8 + (L % per_group) % 8); tile = (int)((L % per_group) / 8);
    }
    const uint4 v = ((const uint4 *)(bases + ((uint64_t)sample * tiles + tile) * 8192))[threadIdx.x];      // 512 x 16 B
    const uint64_t seed = ((uint64_t)v.x << 32 | v.y) ^ ((uint64_t)v.z << 32 | v.w);
    constexpr int NB = 8192 / CHUNK;
    uint64_t *span = words + (uint64_t)sample * NB * cap;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t i = threadIdx.x + 512u * j;
        const uint32_t b = i / CHUNK, within = i % CHUNK;
        uint64_t *dst = span + (uint64_t)b * cap + (uint64_t)tile * CHUNK + within + ((skew & 1) ? (b * 3u) & 7u : 0u);
        if (skew & 4) __builtin_nontemporal_store(seed + i, dst); else *dst = seed + i;
    }
}

template <int CHUNK>
static double run(const uint8_t *bases, uint64_t *words, int samples, int tiles, int skew)
{
    const uint32_t cap = (uint32_t)tiles * CHUNK + 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        if (rep == 1) (void)hipEventRecord(e0);
        hipLaunchKernelGGL(scatter_kernel<CHUNK>, dim3((unsigned)samples * tiles), dim3(512), 0, 0, bases, words, tiles, cap, skew);
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv)
{
    const int samples = argc > 1 ? atoi(argv[1]) : 200, skew = argc > 2 ? atoi(argv[2]) : 0, tiles = 611;      // 5 Mbp / 8 192
    const uint64_t nb = (uint64_t)samples * tiles * 8192, nw = (uint64_t)samples * (8192ull * tiles + 1024 * 64 + 8192) + 4096;
    uint8_t *bases; uint64_t *words;
    if (hipMalloc((void **)&bases, nb) != hipSuccess || hipMalloc((void **)&words, nw * 8) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(bases, 1, nb);
    const double gb = (double)samples * tiles * (8192.0 * 8 + 8192) / 1e9;
    double ms;
    ms = run<8>(bases, words, samples, tiles, skew);    printf("chunk   8 words ( 64 B, 1024 regions per tile): %7.2f ms  %6.2f TB/s\n", ms, gb / ms);
    ms = run<16>(bases, words, samples, tiles, skew);   printf("chunk  16 words (128 B,  512 regions per tile): %7.2f ms  %6.2f TB/s\n", ms, gb / ms);
    ms = run<32>(bases, words, samples, tiles, skew);   printf("chunk  32 words (256 B,  256 regions per tile): %7.2f ms  %6.2f TB/s\n", ms, gb / ms);
    ms = run<128>(bases, words, samples, tiles, skew);  printf("chunk 128 words (  1 KB,  64 regions per tile): %7.2f ms  %6.2f TB/s\n", ms, gb / ms);
    ms = run<8192>(bases, words, samples, tiles, 0);    printf("streaming write (one 64 KB piece per tile)    : %7.2f ms  %6.2f TB/s\n", ms, gb / ms);
    return 0;
}
