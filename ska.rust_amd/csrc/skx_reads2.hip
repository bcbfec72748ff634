// skx_reads2.hip -- KmerFilter (bloom_filter.rs:35-148) for a whole FASTQ sample with the engine's own kernels: the windows of
// the sample are partitioned by bloom word and every partition is evaluated in LDS.  (skx_reads.hip holds the window pass and the
// first form, a chain of sorts / scans / selections on skx_prims.hip; that form remains for the samples this one does not take.)
//
// What the reference computes per k-mer hash h, with occurrences t_1 < t_2 < ... in stream order:
//   FP(h)  <=>  fp(h) is a subset of OR{ fp(g) : loc(g) == loc(h), first(g) < first(h) }          (the blocked bloom's false positive)
//   min_count == 2 : every occurrence passes except t_1, which passes iff FP(h)
//   min_count >= 3 : exactly the (min_count - FP(h))-th occurrence passes
// loc(h) = (mix(h) * 3 145 728) >> 64 with mix a bijection (bloom_filter.rs:50-58,72-74), so ordering the windows by m = mix(h)
// groups equal hashes AND equal bloom words: one partition by loc serves both.  3 145 728 = 65 536 x 48, so a final partition is
// exactly 48 consecutive bloom words (~1 800 windows of a 50x isolate; 24 or 12 words for larger samples) -- no bloom word straddles two.
//   reads_windows_kernel<false>   : (skx_reads.hip) per tile of 4 096 positions the gated windows as (ntHash, position in the tile), compacted
//   rs_scatter_kernel<12288,256,1>: two such tiles a workgroup -> 256 coarse partitions x 8 slices (one per XCD)
//   rs_scatter_kernel<48,256,0>   : each (slice, coarse partition) -> the coarse partition's 256 final ones, its sources on one XCD in sequence
//   rs_groups_kernel              : a final partition in LDS: counting sort by m into micro-buckets, rank by position (by (m, position) in
//                                   the few micro-buckets that hold two hashes), hash groups, bloom-word groups, FP, verdict per occurrence
//                                   -> positions that pass
//   words_* kernels               : passing positions -> packed words -> (sample, bucket) regions of a dictset, which the
//                                   assemblies' dedupe_mb_kernel sorts / folds (so reads get sub-indexed regions too)
#include "skx_internal.h"
#include <algorithm>

namespace skx {

constexpr uint64_t BLOOM_WORDS = 3145728ull;                           // bloom_filter.rs:93-97
constexpr uint64_t MIX_C = 0x85D059AA333121CFull, MIX_CINV = 0x756de8dbb3c2452full;      // MIX_C * MIX_CINV == 1 (mod 2^64)
__device__ static inline uint64_t mixh(uint64_t h) { return (h ^ (h >> 31)) * MIX_C; }               // cheap_mix
__device__ static inline uint64_t unmixh(uint64_t m) { const uint64_t x = m * MIX_CINV; return x ^ (x >> 31) ^ (x >> 62); }
__device__ static inline uint32_t loc_of(uint64_t m) { return (uint32_t)__umul64hi(m, BLOOM_WORDS); }
__device__ static inline uint64_t bloom_fp5(uint64_t h)                  // bloom_filter.rs:62-68
{
    return (1ull << (h & 63)) | (1ull << ((h >> 6) & 63)) | (1ull << ((h >> 12) & 63)) | (1ull << ((h >> 18) & 63)) | (1ull << ((h >> 24) & 63));
}

// ---------------------------------------------------------------------------------------------------------- partition
constexpr int RS_TILE = 4096, RS_NT = 512, RS_PER = RS_TILE / RS_NT;          // (256 threads: + 30 %, 1 024: the same)
constexpr uint32_t RS_GROUP = 2;                                               // tiles of the window pass a workgroup of the first pass takes
struct RsArgs {
    const uint64_t *src_h; const uint32_t *src_t; const uint16_t *src_t16;  // FROM_POS: the window pass's tiles (hash, position inside the tile; src_cnt a tile); else records of source partitions
    const uint32_t *src_cnt; uint64_t src_cap, n_pos;
    uint64_t *dst_h; uint32_t *dst_t; uint32_t *dst_cnt; uint64_t dst_cap;   // destination partition holds dst_cap records
    int *overflow;
    uint32_t src_mod, src_tiles;                                             // !FROM_POS: source s is slice s / src_mod of coarse partition s % src_mod; tiles a source
};
// Who writes a destination decides what its lines cost: workgroups go round the 8 XCDs by their linear index, each XCD has its own L2, and a
// run of ~11-16 records ends inside a line that the next run -- from whichever tile reserves next -- goes on with.  So the first pass keeps
// RS_SLICES areas per coarse partition, one per XCD (slice = blockIdx.x % 8: every line of a slice is put together in ONE L2: 1.35 -> 1.05 ms),
// and the second pass numbers its workgroups so that XCD x takes the sources x, x + 8, ... one after the other, all tiles of a source in a row
// (sources numbered slice-major, so x is also the coarse partition modulo 8): the 256 final partitions of a coarse one are written from one
// XCD, a few sources at a time, and their lines are complete before they leave its L2.
constexpr uint32_t RS_SLICES = 8;
template <int DIV, int FAN, bool FROM_POS>
__global__ __launch_bounds__(RS_NT) void rs_scatter_kernel(RsArgs a)
{
    __shared__ uint32_t s_hist[FAN], s_lbase[FAN + 1], s_gbase[FAN];
    __shared__ uint64_t s_h[RS_TILE];
    __shared__ uint32_t s_t[RS_TILE];
    const uint32_t xr = blockIdx.x / RS_SLICES;                                         // (second pass) my number on my XCD
    const uint32_t src = FROM_POS ? 0u : (xr / a.src_tiles) * RS_SLICES + blockIdx.x % RS_SLICES;
    const uint32_t dpre = FROM_POS ? blockIdx.x % RS_SLICES : src % a.src_mod;          // destination = dpre * FAN + sub
    // first pass: a workgroup takes RS_GROUP tiles of the window pass (RS_TILE positions each, their gated windows compacted at the tile's first
    // slots) -- together where they fit its RS_TILE record slots (a deep isolate gates 47 % of its positions: a destination's run is then ~22 records
    // instead of ~11), one after the other where they do not
    const uint32_t ntl = FROM_POS ? (uint32_t)((a.n_pos + RS_TILE - 1) / RS_TILE) : 0u, tl0 = blockIdx.x * RS_GROUP;
    const uint32_t c0 = FROM_POS ? a.src_cnt[tl0] : 0u, c1 = FROM_POS && tl0 + 1u < ntl ? a.src_cnt[tl0 + 1u] : 0u;
    const bool together = c0 + c1 <= (uint32_t)RS_TILE;
    const uint32_t rounds = FROM_POS ? (together ? 1u : 2u) : 1u;
    for (uint32_t rd = 0; rd < rounds; rd++) {
    const uint32_t nA = !FROM_POS ? 0u : (together || rd == 0) ? c0 : c1, nB = FROM_POS && together ? c1 : 0u;      // records of the round's first / second tile
    const uint64_t baseA = (uint64_t)(tl0 + (together ? 0u : rd)) * RS_TILE, baseB = (uint64_t)(tl0 + 1u) * RS_TILE;
    const uint64_t n = FROM_POS ? (uint64_t)(nA + nB) : (uint64_t)a.src_cnt[src];
    const uint64_t i0 = FROM_POS ? 0ull : (uint64_t)(xr % a.src_tiles) * RS_TILE;
    if (i0 >= n) { if (FROM_POS) continue; return; }
    const uint64_t sbase = (uint64_t)src * a.src_cap;
    if (rd) __syncthreads();                                                // (the round before is done with the tables)
    for (int i = threadIdx.x; i < FAN; i += RS_NT) s_hist[i] = 0;
    __syncthreads();
    uint64_t h[RS_PER]; uint32_t t[RS_PER], rk[RS_PER];
#pragma unroll
    for (int j = 0; j < RS_PER; j++) {
        const uint64_t i = i0 + threadIdx.x + (uint64_t)RS_NT * j;
        const bool valid = i < n;
        const uint64_t at = !FROM_POS ? sbase + i : i < nA ? baseA + i : baseB + (i - nA);
        h[j] = valid ? a.src_h[at] : 0ull;
        t[j] = !valid ? 0u : FROM_POS ? (uint32_t)(at & ~(uint64_t)(RS_TILE - 1)) + (uint32_t)a.src_t16[at] : a.src_t[at];
        rk[j] = 0xFFFFFFFFu;
        if (valid) { const uint32_t sub = (loc_of(mixh(h[j])) / (uint32_t)DIV) % (uint32_t)FAN; rk[j] = (sub << 16) | atomicAdd(&s_hist[sub], 1u); }
    }
    __syncthreads();
    // one cursor reservation per non-empty destination; local starts by a serial walk of the (<= 256) counters in one wave
    for (int b = threadIdx.x; b < FAN; b += RS_NT) {
        const uint32_t nb = s_hist[b];
        uint32_t g = 0;
        if (nb) { g = atomicAdd(&a.dst_cnt[(uint64_t)dpre * FAN + b], nb); if ((uint64_t)g + nb > a.dst_cap) *a.overflow = 1; }
        s_gbase[b] = g;
    }
    if (threadIdx.x < 64) {
        constexpr int PERL = (FAN + 63) / 64;
        uint32_t c[PERL], sum = 0;
#pragma unroll
        for (int u = 0; u < PERL; u++) { const int b = threadIdx.x * PERL + u; c[u] = b < FAN ? s_hist[b] : 0u; sum += c[u]; }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += y; }
        uint32_t run = inc - sum;
#pragma unroll
        for (int u = 0; u < PERL; u++) { const int b = threadIdx.x * PERL + u; if (b < FAN) s_lbase[b] = run; run += c[u]; }
        if (threadIdx.x == 63) s_lbase[FAN] = run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_PER; j++)
        if (rk[j] != 0xFFFFFFFFu) { const uint32_t o = s_lbase[rk[j] >> 16] + (rk[j] & 0xFFFFu); s_h[o] = h[j]; s_t[o] = t[j]; }
    __syncthreads();
    const uint32_t total = s_lbase[FAN];
    for (uint32_t i = threadIdx.x; i < total; i += RS_NT) {
        const uint64_t hh = s_h[i];
        const uint32_t sub = (loc_of(mixh(hh)) / (uint32_t)DIV) % (uint32_t)FAN;
        const uint64_t r = (uint64_t)s_gbase[sub] + (i - s_lbase[sub]);
        if (r < a.dst_cap) { const uint64_t o = ((uint64_t)dpre * FAN + sub) * a.dst_cap + r; a.dst_h[o] = hh; a.dst_t[o] = s_t[i]; }
    }
    }
}

// ---------------------------------------------------------------------------------------------------------- one partition in LDS
// WPP = bloom words per final partition: 48 (65 536 partitions; samples of up to ~160 M windows), 24 (131 072) or 12 (262 144 partitions).
// 1 024 threads x 4 records, 78 KB of LDS: two workgroups per CU.
constexpr int RG_NT = 1024, RG_ITEMS = 4, RG_CAP = RG_NT * RG_ITEMS, RG_MB = 48 * 64;       // micro-bucket = (bloom word, top bits of the fraction)
constexpr int RG_PAD = 8;                                                                     // records behind the last one that the rank step may read
struct RgArgs { const uint64_t *h; const uint32_t *t; const uint32_t *cnt; uint64_t cap; int min_count; uint32_t *out_t; unsigned long long *out_n; int *overflow; unsigned long long *dbg; };
#define RG_MARK(i) do { if (a.dbg && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&a.dbg[i], now_ - t_mark); t_mark = now_; } } while (0)
template <int WPP>
__device__ static inline uint32_t micro_of(uint64_t m, uint32_t loc0, uint32_t &loc_rel)
{
    constexpr int FB = WPP == 48 ? 6 : WPP == 24 ? 7 : 8;        // 48 x 64 = 24 x 128 = 12 x 256 = 3 072 micro-buckets
    const uint64_t frac = m * BLOOM_WORDS;                       // low 64 bits of the 128-bit product whose high part is the bloom word
    loc_rel = loc_of(m) - loc0;
    return loc_rel * (1u << FB) + (uint32_t)(frac >> (64 - FB));
}
// less += (mq, tq) < (m, tt) as 96-bit numbers: a subtract-with-borrow chain whose last borrow is added (4 vector instructions; the
// compiler's form of the same comparison -- two 64-bit compares, an equality and their selects -- came to 16)
__device__ static inline void add_if_below(uint32_t &less, uint64_t mq, uint32_t tq, uint64_t m, uint32_t tt)
{
    uint32_t scratch;
    asm("v_sub_co_u32 %1, vcc, %2, %5\n\tv_subb_co_u32 %1, vcc, %3, %6, vcc\n\tv_subb_co_u32 %1, vcc, %4, %7, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(less), "=&v"(scratch)
        : "v"(tq), "v"((uint32_t)mq), "v"((uint32_t)(mq >> 32)), "v"(tt), "v"((uint32_t)m), "v"((uint32_t)(m >> 32))
        : "vcc");
}
template <int WPP>
#ifndef SKX_RG_STEP
#define SKX_RG_STEP 4
#endif
__global__ __launch_bounds__(RG_NT, 8) void rs_groups_kernel(RgArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    uint64_t *s_m = reinterpret_cast<uint64_t *>(s_mem);                  // [RG_CAP + pad] mix(hash) sorted; then, at group heads, the hash's bloom fingerprint
    uint32_t *s_t = reinterpret_cast<uint32_t *>(s_m + RG_CAP + RG_PAD);   // [RG_CAP + pad] stream position
    uint32_t *s_cnt = s_t + RG_CAP + RG_PAD + 1;                           // [-1] = 0 | [RG_MB] counts -> cursors (= micro-bucket ends; bit 31: holds more than one hash)
    uint16_t *s_mb = reinterpret_cast<uint16_t *>(s_cnt + RG_MB + 1);      // [RG_CAP] micro-bucket of the record at a sorted position
    uint8_t *s_loc = reinterpret_cast<uint8_t *>(s_mb + RG_CAP);           // [RG_CAP] bloom word inside the partition (0..47) | [RG_CAP] the same of the group heads, dense
    __shared__ uint32_t s_hc[64];
    __shared__ uint32_t s_tmp[17];
    __shared__ unsigned long long s_gb;
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u), wv = (int)(tid >> 6);
    const uint64_t region = blockIdx.x;
    const uint32_t n = a.cnt[region];
    if (n == 0) return;
    if (n > (uint32_t)RG_CAP || n > a.cap) { if (tid == 0) *a.overflow = 1; return; }
    const uint64_t base = region * a.cap;
    const uint32_t loc0 = (uint32_t)region * (uint32_t)WPP;
    unsigned long long t_mark = __builtin_readcyclecounter();
    uint64_t e_m[RG_ITEMS]; uint32_t e_t[RG_ITEMS], e_mb[RG_ITEMS];
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        const uint32_t p = tid + (uint32_t)RG_NT * j;
        const uint32_t pc = p < n ? p : 0u;                     // (every load asked for before the first is used: a load under its condition is waited for on the spot)
        e_m[j] = a.h[base + pc]; e_t[j] = a.t[base + pc];
    }
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        const uint32_t p = tid + (uint32_t)RG_NT * j;
        e_mb[j] = 0xFFFFFFFFu;
        if (p < n) { e_m[j] = mixh(e_m[j]); uint32_t lr; e_mb[j] = micro_of<WPP>(e_m[j], loc0, lr); }
    }
    for (uint32_t i = tid; i < (uint32_t)RG_MB; i += RG_NT) s_cnt[i] = 0;
    if (tid == 0) s_cnt[-1] = 0;
    __syncthreads();
    RG_MARK(0);
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) if (e_mb[j] != 0xFFFFFFFFu) atomicAdd(&s_cnt[e_mb[j]], 1u);
    __syncthreads();
    auto block_excl_scan = [&](uint32_t v, uint32_t &total) -> uint32_t {       // over the 1 024 threads
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        __syncthreads();
        if (lane == 63) s_tmp[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RG_NT / 64; w++) { const uint32_t c = s_tmp[w]; if (w < wv) pre += c; tot += c; }
        total = tot;
        return pre + inc - v;
    };
    {   // exclusive scan of the counts: 3 consecutive micro-buckets per thread
        constexpr uint32_t R = RG_MB / RG_NT;
        const uint32_t m0 = tid * R;
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t u = 0; u < R; u++) sum += s_cnt[m0 + u];
        uint32_t tot;
        uint32_t run = block_excl_scan(sum, tot);
#pragma unroll
        for (uint32_t u = 0; u < R; u++) { const uint32_t c = s_cnt[m0 + u]; s_cnt[m0 + u] = run; run += c; }      // start; the scatter turns it into the end
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++)
        if (e_mb[j] != 0xFFFFFFFFu) { const uint32_t o = atomicAdd(&s_cnt[e_mb[j]], 1u); s_m[o] = e_m[j]; s_t[o] = e_t[j]; s_mb[o] = (uint16_t)e_mb[j]; }
    if (tid < (uint32_t)RG_PAD) { s_m[n + tid] = ~0ull; s_t[n + tid] = ~0u; }      // (2^64 - 1, 2^32 - 1): below no record
    __syncthreads();
    // a micro-bucket nearly always holds the occurrences of ONE hash (3 072 micro-buckets for the ~100 hashes of a deep isolate's partition): the
    // ones that hold more are marked, and only those rank by (m, position) -- the others by position alone
    constexpr int FB = WPP == 48 ? 6 : WPP == 24 ? 7 : 8;
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        const uint32_t p = tid + (uint32_t)RG_NT * j;
        e_mb[j] = 0xFFFFFFFFu;
        if (p >= n) continue;
        e_m[j] = s_m[p]; e_t[j] = s_t[p]; e_mb[j] = s_mb[p];
        if (s_m[s_cnt[(int)e_mb[j] - 1] & 0x7FFFFFFFu] != e_m[j]) atomicOr(&s_cnt[e_mb[j]], 0x80000000u);       // (the cursors below are final: only bit 31 changes)
    }
    __syncthreads();
    RG_MARK(1);
    // rank inside the micro-bucket: positions are unique, so there are no ties
    uint32_t npos[RG_ITEMS];
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        npos[j] = 0xFFFFFFFFu;
        if (e_mb[j] == 0xFFFFFFFFu) continue;
        const uint64_t m = e_m[j]; const uint32_t tt = e_t[j], mb = e_mb[j];
        const uint32_t cb = s_cnt[(int)mb - 1], ce = s_cnt[mb];
        const uint32_t b = cb & 0x7FFFFFFFu, e = ce & 0x7FFFFFFFu;
        uint32_t less = 0, q = b;
        constexpr uint32_t RS = SKX_RG_STEP;                                     // independent LDS reads in flight per step
        static_assert(RS <= (uint32_t)RG_PAD, "the pad covers one step");
        if (!(ce >> 31)) {                                                       // one hash: by position alone (2 instructions a record, a third of the LDS bytes)
            for (; q + RS <= e; q += RS) {
                uint32_t tq[RS];
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) tq[u] = s_t[q + u];
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) less += tq[u] < tt ? 1u : 0u;
            }
            if (q < e) {                                                         // the last one to three: what lies behind the end is read and not counted
                uint32_t tq[RS];
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) tq[u] = s_t[q + u];
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) less += (q + u < e && tq[u] < tt) ? 1u : 0u;
            }
        } else {
            // (m, position) as one 96-bit number.  A step reads past the bucket's end without a clamp: what lies there are the records of the
            // following micro-buckets, whose m is larger (the micro-bucket index is monotone in m), and behind the last record the pad
            for (; q < e; q += RS) {
                uint64_t mq[RS]; uint32_t tq[RS];
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) { mq[u] = s_m[q + u]; tq[u] = s_t[q + u]; }
#pragma unroll
                for (uint32_t u = 0; u < RS; u++) add_if_below(less, mq[u], tq[u], m, tt);
            }
        }
        npos[j] = b + less;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) if (npos[j] != 0xFFFFFFFFu) { s_m[npos[j]] = e_m[j]; s_t[npos[j]] = e_t[j]; s_loc[npos[j]] = (uint8_t)(e_mb[j] >> FB); }
    __syncthreads();
    // group heads (first = earliest occurrence of a hash): their positions go into a compact list, in order, and their s_m entry becomes the
    // hash's bloom fingerprint, so that everything below runs one lane per GROUP with short, similar loops (scanning the records themselves
    // cost ~100 cycles of LDS latency per step in a few divergent lanes: 75 k of 110 k cycles)
    uint16_t *s_hidx = reinterpret_cast<uint16_t *>(s_cnt - 1);             // [heads + 1] positions of the group heads, in order (the cursors are dead)
    bool is_head[RG_ITEMS]; uint64_t bal[RG_ITEMS];
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        const uint32_t p = tid + (uint32_t)RG_NT * j;
        is_head[j] = p < n && (p == 0 || s_m[p - 1] != s_m[p]);
        bal[j] = __ballot(is_head[j]);
        if (lane == 0) s_hc[j * (RG_NT / 64) + wv] = (uint32_t)__popcll(bal[j]);
    }
    static_assert(RG_ITEMS * (RG_NT / 64) == 64, "one wave scans the head counts");
    __syncthreads();                                                        // (also: every s_cnt cursor has been read)
    if (wv == 0) {
        const uint32_t v = s_hc[lane];
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        s_hc[lane] = inc - v;
        if (lane == 63) s_tmp[16] = inc;
    }
    __syncthreads();
    const uint32_t n_heads = s_tmp[16];
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++)
        if (is_head[j]) s_hidx[s_hc[j * (RG_NT / 64) + wv] + (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1ull))] = (uint16_t)(tid + (uint32_t)RG_NT * j);
    if (tid == 0) s_hidx[n_heads] = (uint16_t)n;
    __syncthreads();
    // what the groups need of their heads, dense and in group order (fingerprint over the dead s_m, position over the dead cursors' tail + s_mb +
    // s_loc, bloom word behind them): the neighbour loops below read consecutive entries instead of following s_hidx into three arrays
    uint64_t *s_hfp = s_m; uint32_t *s_ht = (s_cnt - 1) + (RG_CAP + 2) / 2; uint8_t *s_hloc = s_loc + RG_CAP;
    static_assert(((RG_MB + 2) * 4 + RG_CAP * 3) / 4 - (RG_CAP + 2) / 2 >= RG_CAP, "the positions of RG_CAP heads fit behind the head list");
    {
        uint64_t r_fp[RG_ITEMS]; uint32_t r_t[RG_ITEMS]; uint8_t r_loc[RG_ITEMS];
#pragma unroll
        for (int j = 0; j < RG_ITEMS; j++) {
            const uint32_t kh = tid + (uint32_t)RG_NT * j;
            r_fp[j] = 0; r_t[j] = 0; r_loc[j] = 0;
            if (kh < n_heads) { const uint32_t p = s_hidx[kh]; r_fp[j] = bloom_fp5(unmixh(s_m[p])); r_t[j] = s_t[p]; r_loc[j] = s_loc[p]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RG_ITEMS; j++) {
            const uint32_t kh = tid + (uint32_t)RG_NT * j;
            if (kh < n_heads) { s_hfp[kh] = r_fp[j]; s_ht[kh] = r_t[j]; s_hloc[kh] = r_loc[j]; }
        }
        __syncthreads();
    }
    RG_MARK(2);
    // per group: size, FP where it can change the verdict (bloom fingerprints of the hashes of the same bloom word first seen
    // earlier), and what the group emits
    const uint32_t mc = (uint32_t)a.min_count;
    uint32_t g_p[RG_ITEMS], g_occ[RG_ITEMS], g_emit[RG_ITEMS], g_fp = 0, mine = 0;
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        const uint32_t kh = tid + (uint32_t)RG_NT * j;
        g_p[j] = 0; g_occ[j] = 0; g_emit[j] = 0;
        if (kh >= n_heads) continue;
        const uint32_t p = s_hidx[kh], occ = (uint32_t)s_hidx[kh + 1] - p;
        uint32_t fpos = 0;
        if (mc == 2 || occ + 1 >= mc) {               // a smaller group passes nothing whatever its FP (occurrence number <= occ < min_count - 1)
            const uint8_t lc = s_hloc[kh]; const uint32_t tp = s_ht[kh];
            uint64_t seen = 0;
            for (int kk = (int)kh - 1; kk >= 0 && s_hloc[kk] == lc; kk--) if (s_ht[kk] < tp) seen |= s_hfp[kk];
            for (uint32_t kk = kh + 1; kk < n_heads && s_hloc[kk] == lc; kk++) if (s_ht[kk] < tp) seen |= s_hfp[kk];
            fpos = (s_hfp[kh] & ~seen) == 0;
        }
        g_p[j] = p; g_occ[j] = occ; g_fp |= fpos << j;
        g_emit[j] = mc == 2 ? occ - 1 + fpos : (occ >= mc - fpos ? 1u : 0u);
        mine += g_emit[j];
    }
    RG_MARK(3);
    uint32_t total;
    const uint32_t ex = block_excl_scan(mine, total);
    if (tid == 0) s_gb = total ? atomicAdd(a.out_n, (unsigned long long)total) : 0ull;
    __syncthreads();
    RG_MARK(4);
    unsigned long long o = s_gb + ex;
#pragma unroll
    for (int j = 0; j < RG_ITEMS; j++) {
        if (!g_emit[j]) continue;
        const uint32_t p = g_p[j], fpos = (g_fp >> j) & 1u;
        if (mc == 2) { for (uint32_t u = fpos ? 0u : 1u; u < g_occ[j]; u++) a.out_t[o++] = s_t[p + u]; }      // every occurrence but the first; the first iff FP
        else a.out_t[o++] = s_t[p + (mc - fpos) - 1];                                                          // exactly the (min_count - FP)-th
    }
    RG_MARK(5);
}

// ---------------------------------------------------------------------------------------------------------- small kernels
// flagged positions -> their packed words (no count filter: min_count <= 1, FASTA samples, oversize assemblies)
__global__ __launch_bounds__(256) void words_from_flags_kernel(const uint8_t *flag, const uint64_t *wlo, const uint64_t *whi, uint64_t n, uint64_t *out_lo, uint64_t *out_hi,
                                                               unsigned long long *out_n)
{
    __shared__ uint32_t s_w[4]; __shared__ unsigned long long s_b;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && flag[i];
    const unsigned long long bal = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_w[wv] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 4; w++) { const uint32_t c = s_w[w]; s_w[w] = t; t += c; } s_b = t ? atomicAdd(out_n, (unsigned long long)t) : 0ull; }
    __syncthreads();
    if (f) {
        const unsigned long long o = s_b + s_w[wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        out_lo[o] = wlo[i]; if (whi) out_hi[o] = whi[i];
    }
}
__global__ void words_gather_kernel(const uint32_t *pos, uint64_t n, const uint64_t *wlo, const uint64_t *whi, uint64_t *out_lo, uint64_t *out_hi)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = pos[i];
        out_lo[i] = wlo[p]; if (whi) out_hi[i] = whi[p];
    }
}
// bucket of a packed word: top logB bits of its hash H (word = H << 4 | mask, H of `bits` bits; wide: hi:lo is the 128-bit word)
__device__ static inline uint32_t word_bucket(uint64_t lo, uint64_t hi, int bits, int logB, bool wide)
{
    if (logB == 0) return 0u;
    const int sh = bits + 4 - logB;                               // bucket = word >> sh
    if (!wide) return (uint32_t)(lo >> sh);
    return sh >= 64 ? (uint32_t)(hi >> (sh - 64)) : (uint32_t)((hi << (64 - sh)) | (lo >> sh));
}
// COUNT: raw[region] += words of the region; else: words into their regions through the cursors (exact offsets, so nothing can
// overflow).  A workgroup takes 4 096 words, ranks them per bucket in an LDS histogram and touches every global counter once.
constexpr int WR_NT = 1024, WR_PER = 4;
template <bool COUNT>
__global__ __launch_bounds__(WR_NT) void words_regions_kernel(const uint64_t *in_lo, const uint64_t *in_hi, uint64_t n, int bits, int logB, uint64_t region0,
                                                              uint32_t *raw, const uint64_t *off, uint32_t *cursor, uint64_t *words)
{
    extern __shared__ uint32_t s_bin[];                              // [2^logB] counts, then the workgroup's first slot in each region
    const bool wide = in_hi != nullptr;
    const uint32_t B = 1u << logB;
    for (uint32_t i = threadIdx.x; i < B; i += WR_NT) s_bin[i] = 0;
    __syncthreads();
    uint64_t lo[WR_PER], hi[WR_PER]; uint32_t rk[WR_PER], bk[WR_PER];
#pragma unroll
    for (int j = 0; j < WR_PER; j++) {
        const uint64_t i = (uint64_t)blockIdx.x * (WR_NT * WR_PER) + threadIdx.x + (uint64_t)WR_NT * j;
        bk[j] = 0xFFFFFFFFu; lo[j] = hi[j] = 0; rk[j] = 0;
        if (i < n) { lo[j] = in_lo[i]; hi[j] = wide ? in_hi[i] : 0ull; bk[j] = word_bucket(lo[j], hi[j], bits, logB, wide); rk[j] = atomicAdd(&s_bin[bk[j]], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += WR_NT) {
        const uint32_t c = s_bin[b];
        if (c) s_bin[b] = COUNT ? atomicAdd(&raw[region0 + b], c) : atomicAdd(&cursor[region0 + b], c);
    }
    if (COUNT) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WR_PER; j++) {
        if (bk[j] == 0xFFFFFFFFu) continue;
        const uint64_t o = off[region0 + bk[j]] + s_bin[bk[j]] + rk[j];
        if (wide) { words[2 * o] = lo[j]; words[2 * o + 1] = hi[j]; } else words[o] = lo[j];
    }
}


void launch_words_regions(bool count, const uint64_t *in_lo, const uint64_t *in_hi, uint64_t n, int bits, int logB, uint64_t region0, uint32_t *raw,
                          const uint64_t *off, uint32_t *cursor, uint64_t *words, hipStream_t st)
{
    if (!n) return;
    const unsigned g = (unsigned)((n + WR_NT * WR_PER - 1) / (WR_NT * WR_PER));
    const size_t lds = (size_t)4 << logB;
    if (count) hipLaunchKernelGGL(words_regions_kernel<true>, dim3(g), dim3(WR_NT), lds, st, in_lo, in_hi, n, bits, logB, region0, raw, off, cursor, words);
    else hipLaunchKernelGGL(words_regions_kernel<false>, dim3(g), dim3(WR_NT), lds, st, in_lo, in_hi, n, bits, logB, region0, raw, off, cursor, words);
}

// One FASTQ / oversize sample -> the packed words of the windows that enter its dictionary (unsorted, duplicates included).
// SKF_NOT_TAKEN: the sample is left to the first form (partition overflow: a hash far more frequent than a partition holds).
int reads_sample_words(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q,
                       DevBuf<uint64_t> &out_lo, DevBuf<uint64_t> &out_hi, uint64_t *n_out, const uint64_t *planes)
{
    hipStream_t st = ctx->stream;
    const bool wide = k > 31;
    *n_out = 0;
    if (len == 0) return SKX_OK;
    if (len > 0xFFFFFFF0ull) { set_error("FASTQ sample longer than 4 G bases"); return SKX_EUNSUP; }
    DevBuf<uint64_t> wlo, whi, hash; DevBuf<uint8_t> flag; DevBuf<uint16_t> rec_t; DevBuf<uint32_t> tile_cnt;
    DevBuf<unsigned long long> d_n; DevBuf<int> d_over;            // d_n[0]: positions / words that leave; d_n[1 .. 257): gated windows (spread counters)
    SKX_TRY(d_n.alloc(257)); SKX_TRY(d_n.zero(st)); SKX_TRY(d_over.alloc(1)); SKX_TRY(d_over.zero(st));
    const bool all_words = q.min_count <= 1;                       // every gated window enters: the window pass writes the words itself
    SKX_TRY(reads_windows(ctx, d_seq, d_qual, len, k, rc, q, hash, wlo, whi, flag, d_n.p + 1, all_words, planes, rec_t, tile_cnt));      // (planes: the sample packed, d_seq / d_qual unused)
    unsigned long long n_acc = 0;
    if (q.min_count <= 1) {                                        // KmerFilter: 0 | 1 => every gated window enters
        SKX_TRY(out_lo.alloc(len)); if (wide) SKX_TRY(out_hi.alloc(len));
        hipLaunchKernelGGL(words_from_flags_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, (const uint8_t *)flag.p, (const uint64_t *)wlo.p,
                           wide ? (const uint64_t *)whi.p : (const uint64_t *)nullptr, len, out_lo.p, wide ? out_hi.p : (uint64_t *)nullptr, d_n.p);
        SKX_HIP(hipMemcpyAsync(&n_acc, d_n.p, 8, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        SKX_HIP(hipGetLastError());
        *n_out = n_acc;
        return SKX_OK;
    }
    // two partition passes (coarse, then 256 final partitions each); capacities from the number of gated windows, hashes being uniform
    unsigned long long n_win = 0, parts[256];                      // counted by the window kernel
    SKX_HIP(hipMemcpyAsync(parts, d_n.p + 1, sizeof parts, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    for (auto c : parts) n_win += c;
    if (n_win == 0) return SKX_OK;
    // partition sizes vary more than a Poisson count: the ~35 occurrences of a k-mer of a 50x isolate move together
    // (sigma ~ sqrt(35 x mean)); half the mean + 1 024 records of head-room covers that
    // 65 536 partitions of 48 bloom words while a partition's records fit the group kernel with a quarter of the mean + 1 024 records of
    // head-room (the mean + 5.7 sigma at 2 000 records; half the mean where that fits as well); beyond ~161 M windows 262 144 partitions of
    // 12 bloom words, which the group kernel fills to an eighth only (6.6 ms against 2.7: k = 31 on a 50x isolate used to land there)
    const uint64_t mean48 = n_win / 65536;
    // (SKX_KNOBS=reads_layout=2 | 3: the 24- / 12-word layouts whatever the size -- the tests run small samples through every instantiation)
    const long forced = knob("reads_layout");
    const bool coarse = forced ? forced == 1 : mean48 + mean48 / 4 + 1024 <= (uint64_t)RG_CAP;
    const bool mid = forced ? forced == 2 : !coarse && (mean48 / 2) * 3 / 2 + 1024 <= (uint64_t)RG_CAP;         // 131 072 partitions of 24 bloom words (k = 17 on a 50x isolate: ~170 M windows)
    const bool fine = !coarse && !mid;
    const bool roomy = mean48 * 3 / 2 + 1024 <= (uint64_t)RG_CAP;
    const uint64_t n_part = fine ? 262144 : mid ? 131072 : 65536, fan1 = n_part / 256;
    const uint64_t nsrc = fan1 * RS_SLICES;                                                  // areas of the first pass: (slice, coarse partition)
    const uint64_t cap1 = n_win / nsrc + n_win / (nsrc * 16) + 4096,
                   cap2 = (!coarse || roomy) ? (n_win / n_part) * 3 / 2 + 1024 : mean48 + mean48 / 4 + 1024;
    if (cap2 > (uint64_t)RG_CAP) return SKF_NOT_TAKEN;                                     // beyond ~530 M windows
    DevBuf<uint64_t> h1, h2; DevBuf<uint32_t> t1, t2, c1, c2, acc_t;
    SKX_TRY(h1.alloc(nsrc * cap1)); SKX_TRY(t1.alloc(nsrc * cap1)); SKX_TRY(c1.alloc(nsrc)); SKX_TRY(c1.zero(st));
    SKX_TRY(h2.alloc(n_part * cap2)); SKX_TRY(t2.alloc(n_part * cap2)); SKX_TRY(c2.alloc(n_part)); SKX_TRY(c2.zero(st));
    RsArgs a1{hash.p, nullptr, rec_t.p, tile_cnt.p, 0, len, h1.p, t1.p, c1.p, cap1, d_over.p, 0u, 1u};
    RsArgs a2{h1.p, t1.p, nullptr, c1.p, cap1, 0, h2.p, t2.p, c2.p, cap2, d_over.p, (uint32_t)fan1, (uint32_t)((cap1 + RS_TILE - 1) / RS_TILE)};
    if (reads_tile() != RS_TILE) { set_error("the window pass and the partition pass disagree on the tile"); return SKX_EUNSUP; }
    const dim3 g1((unsigned)(((len + RS_TILE - 1) / RS_TILE + RS_GROUP - 1) / RS_GROUP)), g2((unsigned)(nsrc * ((cap1 + RS_TILE - 1) / RS_TILE)));
    SKX_TRY(acc_t.alloc(q.min_count == 2 ? n_win : n_win / 2 + 1024));                     // min_count >= 3: one position per group of >= 2... at most n_win / 2
    DevBuf<unsigned long long> d_dbg;
    if (getenv("SKX_DEBUG")) { SKX_TRY(d_dbg.alloc(8)); SKX_TRY(d_dbg.zero(st)); }
    RgArgs ag{h2.p, t2.p, c2.p, cap2, (int)q.min_count, acc_t.p, d_n.p, d_over.p, d_dbg.p};
    const size_t lds = (size_t)(RG_CAP + RG_PAD) * 12 + ((size_t)RG_MB + 2) * 4 + (size_t)RG_CAP * 4 + 64;
    if (mid) {
        hipLaunchKernelGGL((rs_scatter_kernel<24 * 256, 512, true>), g1, dim3(RS_NT), 0, st, a1);
        hipLaunchKernelGGL((rs_scatter_kernel<24, 256, false>), g2, dim3(RS_NT), 0, st, a2);
        (void)hipFuncSetAttribute((const void *)rs_groups_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(rs_groups_kernel<24>, dim3((unsigned)n_part), dim3(RG_NT), lds, st, ag);
    } else if (!fine) {
        hipLaunchKernelGGL((rs_scatter_kernel<48 * 256, 256, true>), g1, dim3(RS_NT), 0, st, a1);
        hipLaunchKernelGGL((rs_scatter_kernel<48, 256, false>), g2, dim3(RS_NT), 0, st, a2);
        (void)hipFuncSetAttribute((const void *)rs_groups_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(rs_groups_kernel<48>, dim3((unsigned)n_part), dim3(RG_NT), lds, st, ag);
    } else {
        hipLaunchKernelGGL((rs_scatter_kernel<12 * 256, 1024, true>), g1, dim3(RS_NT), 0, st, a1);
        hipLaunchKernelGGL((rs_scatter_kernel<12, 256, false>), g2, dim3(RS_NT), 0, st, a2);
        (void)hipFuncSetAttribute((const void *)rs_groups_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(rs_groups_kernel<12>, dim3((unsigned)n_part), dim3(RG_NT), lds, st, ag);
    }
    int over = 0;
    SKX_HIP(hipMemcpyAsync(&n_acc, d_n.p, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(&over, d_over.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    if (d_dbg.p) { unsigned long long hd[8]; SKX_HIP(hipMemcpy(hd, d_dbg.p, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[skx] rs_groups cycles per partition: load %.0f, count+scan+scatter %.0f, rank+write+heads %.0f, groups/FP (lane 0) %.0f, wait+scan %.0f, output %.0f (windows %llu, partitions %llu, passing %llu)\n",
                hd[0] / (double)n_part, hd[1] / (double)n_part, hd[2] / (double)n_part, hd[3] / (double)n_part, hd[4] / (double)n_part, hd[5] / (double)n_part, n_win, (unsigned long long)n_part, n_acc); }
    if (over) { if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] reads: a partition overflowed, sample left to the sort-based form\n"); return SKF_NOT_TAKEN; }
    if (n_acc == 0) return SKX_OK;
    SKX_TRY(out_lo.alloc(n_acc)); if (wide) SKX_TRY(out_hi.alloc(n_acc));
    if (planes) launch_words_rebuild_planes(acc_t.p, n_acc, planes, k, rc, out_lo.p, wide ? out_hi.p : nullptr, st);
    else launch_words_rebuild(acc_t.p, n_acc, d_seq, len, k, rc, out_lo.p, wide ? out_hi.p : nullptr, st);
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    *n_out = n_acc;
    return SKX_OK;
}

}  // namespace skx
