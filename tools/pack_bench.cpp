// tools/pack_bench.cpp -- host side of the read-set pipeline alone: T threads each stream their own FASTQ file through stream_fastq_file with
// (0) a sink that counts, (1) the byte-stream copy the pipeline made before round 4, (2) the bit-plane packing it makes now.
// build: g++ -O2 -pthread -o /tmp/pack_bench tools/pack_bench.cpp -L ska.rust_amd -lskx -Wl,-rpath,$PWD/ska.rust_amd ; usage: pack_bench <mode> <file>...
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
namespace skx { void pack_bases_planes(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad); void pack_qual_plane(const uint8_t *q, size_t n, int min_qual, uint64_t *qb);
int stream_fastq_file(const char *path, const std::function<int(int which, const uint8_t *p, size_t n)> &emit); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void one(const char *path, int mode, size_t *bytes)
{
    std::vector<uint8_t> sink(16u << 20);
    size_t used = 0, total = 0; uint64_t cur[5] = {0}, pos = 0; std::vector<uint64_t> pl(4 * 8 + 64);
    std::function<int(int, const uint8_t *, size_t)> emit;
    if (mode == 0) emit = [&](int, const uint8_t *, size_t nb) { total += nb; return 0; };
    else if (mode == 1) emit = [&](int which, const uint8_t *p, size_t nb) { if (used + nb + 1 > (8u << 20)) used = 0; memcpy(sink.data() + (which ? 8u << 20 : 0) + used, p, nb); used += nb + 1; total += nb; return 0; };
    else emit = [&](int which, const uint8_t *p, size_t nb) -> int {
        const size_t words = (nb + 1 + 63) / 64; total += nb;
        if (which == 0) { if (pl.size() < 4 * words) pl.resize(4 * words + 64); for (int q = 0; q < 4; q++) pl[q * words + words - 1] = 0; skx::pack_bases_planes(p, nb, &pl[0], &pl[words], &pl[2 * words]); return 0; }
        skx::pack_qual_plane(p, nb, 20, &pl[3 * words]);
        const uint64_t *lo = &pl[0], *hi = &pl[words], *bd = &pl[2 * words], *qb = &pl[3 * words];
        for (size_t w = 0; w < words; w++) {
            const unsigned take = (unsigned)std::min<size_t>(64, nb + 1 - 64 * w), off = (unsigned)(pos & 63);
            const uint64_t v[5] = {lo[w], hi[w], bd[w], nb / 64 == w ? 1ull << (nb & 63) : 0ull, qb[w]};
            for (int q = 0; q < 5; q++) cur[q] |= v[q] << off;
            pos += take;
            if (off + take >= 64) { if (used + 40 > (16u << 20)) used = 0; memcpy(sink.data() + used, cur, 40); used += 40; for (int q = 0; q < 5; q++) cur[q] = off ? v[q] >> (64 - off) : 0ull; }
        }
        return 0; };
    if (skx::stream_fastq_file(path, emit) != 0) { fprintf(stderr, "failed on %s\n", path); exit(1); }
    *bytes = total;
}
int main(int argc, char **argv)
{
    const int mode = atoi(argv[1]); const int T = argc - 2;
    std::vector<size_t> bytes(T, 0); std::vector<std::thread> th;
    const double t = now();
    for (int i = 0; i < T; i++) th.emplace_back(one, argv[2 + i], mode, &bytes[i]);
    for (auto &x : th) x.join();
    const double dt = now() - t; size_t tot = 0; for (auto b : bytes) tot += b;
    printf("mode %d, %d threads: %.3f s, %.1f GB of sequence + quality bytes per second\n", mode, T, dt, tot / dt / 1e9);
}
