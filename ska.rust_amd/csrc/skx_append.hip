// skx_append.hip -- MergeSkaDict::append on the device, straight from the extraction kernel's regions (gfx950, wave64).
//
// The reference appends a sample to the merged dictionary in ONE pass over its unsorted hash map (merge_ska_dict.rs:77-109: look the
// split k-mer up, insert a row of '-' if it is new, write the sample's column), and a sample's own dictionary folds repeated split
// k-mers by OR-ing their middle bases into an IUPAC code (ska_dict.rs:76-113).  Rounds 1-3 sorted and folded every sample's regions
// first (dedupe_mb_kernel: 80 GB of traffic, a third of the step) so that the union could read them in order.  This file does what the
// reference does: one workgroup owns a *row block* -- the split k-mers whose top logQ hash bits are j -- and its sixteen waves append
// samples independently of one another (wave w takes samples w, w + 16, ...: no barrier until the last sample is in), each reading its
// sample's raw region (words as extract_kernel scattered them: unsorted, with repeats):
//
//   * the block's rows live in an order-preserving LDS hash table (slot = monotone function of H, linear probing without wrap); an entry is
//     {top 32 of the low hash bits ; (the rest << 14) | first-seen rank + 1}: the rank a row got when it was first inserted is its column in
//     everything the pass writes while the final row order (the order of H) is not known yet;
//   * a region holds the words of A = 2^(logQ - logB) row blocks.  The launch is persistent (one workgroup per CU, a row block per round); the A
//     blocks of a region are taken by A workgroups of the same XCD (blockIdx -> XCD is b % 8) in the same round, which start it together: each
//     reads the whole region and keeps its share, and a line is fetched from HBM once if its readers come by while it is in that XCD's L2.  A
//     wave reads 1 024 words at a time (the next chunk on its way meanwhile), keeps its block's, and compacts them into a queue of its own so
//     that the look-ups run with full waves, 128 words at a time (two per lane): a byte per slot tells where around home the key may be, then
//     that one entry is read and compared in full; the ~1 % that are not found this way go to a second queue and through the insert loop 64
//     at a time, four slots per step (looked up in place, they would make every batch wait for its unluckiest lane);
//   * a sample's cells of the block are OR-ed into the wave's row buffer of 4-bit base sets indexed by rank (one LDS atomic without a return
//     value per word: a repeated split k-mer folds here, which is where ska_dict.rs:92-101 folds);
//   * the row buffer leaves as the sample's *piece* of the block: plen ranks, two per byte, at a fixed place (pieces[(j * S + s) * cap / 2]).
//     Ranks beyond plen were handed out later: the cell is '-'.  The rows x samples matrix in the order of H is produced from the pieces
//     by pieces_rows_kernel (all rows, a window of row blocks, or only the rows a filter keeps) -- 1 byte per cell there, 4 bits and no cell
//     for unseen rows here;
//   * what the cells add up to per row (present, unambiguous, code set: merge_ska_array.rs:139-186) is counted from the finished pieces by
//     pieces_stats_kernel, nibble-parallel and without atomics; a sample's k-mer count, when asked for, by pieces_cells_kernel;
//   * when the last sample of a round is in, the table is emitted in key order: the block's row keys and perm (rank -> row of the block).
// Measurement builds: -DSKX_AP_PROF (cycle stamps and counters per phase), -DAP_X_NOBATCH / _NOFILTER / _NORECORD / _NOPROBE / _FAKE4 (the kernel with
// one part left out: what showed that loads + filter + queues are 6.8 of its 24 ms; results are wrong by construction) -- tools/mkvariant.sh.
#include "skx_device.h"
#include <cstdlib>
#include <type_traits>

namespace skx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *g4_t;

// Measurement only (-DSKX_AP_PROF): cycles of every wave's lane 0 in the phases of append_kernel, summed over waves, workgroups and launches:
// 0 waiting for a chunk's words, 1 filter into the queue, 2 look-ups of full batches, 3 the insert loop's batches, 4 end of a sample (drain + piece), 5 the tail (emit)
#ifdef SKX_AP_PROF
__device__ unsigned long long g_ap_prof[16];
#define AP_PROF_START unsigned long long tprof = __builtin_readcyclecounter(), tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define AP_PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[(i)] += t_ - tprof; tprof = t_; } while (0)
#define AP_PROF_FLUSH() do { for (int i_ = 0; i_ < 12; i_++) if (tacc[i_] && (i_ >= 6 || (threadIdx.x & 63) == 0)) atomicAdd(&g_ap_prof[i_], tacc[i_]); } while (0)
extern "C" void skx_debug_phase_prof(unsigned long long *out, int reset)
{
    unsigned long long h[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ap_prof), sizeof(h));
    for (int i = 0; i < 16; i++) out[i] = h[i];
    if (reset) { for (auto &x : h) x = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ap_prof), h, sizeof(h)); }
}
#else
#define AP_PROF_START do { } while (0)
#define AP_PROF(i) do { } while (0)
#define AP_PROF_FLUSH() do { } while (0)
#endif
#ifdef SKX_AP_PROF
#define AP_COUNT(i, v) do { tacc[(i)] += (v); } while (0)
#else
#define AP_COUNT(i, v) do { } while (0)
#endif

// Wave priorities (s_setprio): a wave that is taking words off its loads -- filter, queue, the next loads' issue -- goes before one that is
// inside a look-up batch, and that before one in the insert loop: 23.7 -> 21.5 ms per 1 000 x 5 Mbp (the reverse order -- look-ups first --
// 22.9; any order beats none: profiles/r04zx_ab_append_priorities.log).  Measured, not derived; the likely reason: the sixteen waves of a
// workgroup run independently, a look-up batch is a chain of LDS round trips with little to issue in between, and the arbiter's round robin
// gives such a wave its turn as often as a wave with a hundred instructions ready -- with the order, the short dense phase gets through and
// its wave reaches its own LDS chain sooner, so more chains overlap.
#ifndef AP_PRIO_L
#define AP_PRIO_L 1
#endif
#ifndef AP_PRIO_I
#define AP_PRIO_I 0
#endif
constexpr int AP_PRIO_STREAM = 3, AP_PRIO_LOOKUP = AP_PRIO_L, AP_PRIO_INSERT = AP_PRIO_I;
constexpr int AP_THREADS = 1024, AP_WAVES = 16;
constexpr uint32_t AP_PAD = 64;             // slots behind the table's last home slot (probing does not wrap)
constexpr uint32_t AP_Q = 256, AP_SQ = 64;  // a wave's queue of kept words: the 127 a batch may leave + the 128 of one load -- it cannot overflow, whatever the sample (no test, no
                                            // way out: round 5); its insert queue (one batch of the insert loop: emptied first when a look-up batch's misses would not fit)
#ifndef AP_CH_N
#define AP_CH_N 8
#endif
#ifndef AP_RESYNC
#define AP_RESYNC 16
#endif
constexpr int AP_CH = AP_CH_N;              // 16-byte loads per lane and chunk: a wave reads 128 x AP_CH words at a time
constexpr int AP_RANK_BITS = 14;
constexpr uint32_t AP_RANK_MASK = (1u << AP_RANK_BITS) - 1;

__device__ static inline uint32_t ap_wave_excl(uint32_t v, uint32_t *total)
{
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += t; }
    *total = __shfl(inc, 63, 64);
    return inc - v;
}
__device__ static inline uint32_t ap_block_excl(uint32_t v, uint32_t *s_tmp /*[17]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t wt;
    const uint32_t ex = ap_wave_excl(v, &wt);
    if (lane == 63) s_tmp[wv] = wt;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < nw; i++) { const uint32_t t = s_tmp[i]; s_tmp[i] = run; run += t; } s_tmp[16] = run; }
    __syncthreads();
    const uint32_t r = s_tmp[wv] + ex;
    *total = s_tmp[16];
    __syncthreads();
    return r;
}
__device__ static inline uint32_t ap_mbcnt(unsigned long long b)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// a volatile view of an LDS location AS an LDS location: through a generic pointer the compiler issues flat instructions and waits for the
// wave's vector-memory counter -- that is, for the chunk of words on its way -- before and after each
typedef volatile uint32_t __attribute__((address_space(3))) *lds_vol_t;
__device__ static inline lds_vol_t lds_vol(void *p) { return (lds_vol_t)(__attribute__((address_space(3))) void *)p; }
// ctl words: 0 ranks handed out, 1 failure, 8..24 scan scratch
constexpr int CTL_NROWS = 0, CTL_FAIL = 1, CTL_TMP = 8, CTL_WORDS = 32;

static inline size_t append_lds_bytes(uint32_t nslots, uint32_t cap, bool count_only)
{
    size_t b = (size_t)(nslots + AP_PAD) * 8 + (size_t)(nslots + AP_PAD) + (size_t)AP_WAVES * (AP_Q + AP_SQ) * 8 + CTL_WORDS * 4;      // (AP_PAD is a multiple of 16)
    if (!count_only) b += (size_t)AP_WAVES * (cap / 8) * 4;
    return b;
}

// HI: at least 32 hash bits below the block bits (rem >= 32): the table is addressed by their top 32 and the part bits of a word lie in its
// upper half -- 32-bit operations throughout.  !HI (short k-mers, small inputs): the same steps on left-aligned fields.
template <bool COUNT_ONLY, bool HI>
__global__ __launch_bounds__(AP_THREADS) void append_kernel(AppendArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t total_slots = a.nslots + AP_PAD;
    const uint32_t cap = a.cap, rbw = cap / 8;                       // ranks per block; dwords of one row buffer
    unsigned char *s_fp = s_raw;                                     // [total_slots] a byte of every slot's key (0: empty), what a look-up reads first -- at LDS address 0: a slot's byte is at the slot's number
    unsigned long long *s_tab = reinterpret_cast<unsigned long long *>(s_raw + total_slots);      // [total_slots] (total_slots is a multiple of 16)
    unsigned long long *s_q = s_tab + total_slots;                   // [AP_WAVES][AP_Q kept words + AP_SQ words for the insert loop]
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_q + (size_t)AP_WAVES * (AP_Q + AP_SQ));
    uint32_t *s_rb = s_ctl + CTL_WORDS;                              // [AP_WAVES][rbw] 4-bit base sets by rank, one buffer per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.n_samples;
    const int shA = a.logQ - a.logB;
    const uint32_t A = 1u << shA;
    // The launch is persistent: as many workgroups as the chip holds at once (one per CU), each taking a row block per round.  The A blocks of
    // a region are taken by A workgroups of the same XCD (blockIdx -> XCD is b % 8) in the same round, and these start a round together (a
    // counter per region in global memory): all of them read the region's words, sample after sample, and only the first reader of a line
    // fetches it from HBM if the others come by while it is still in that XCD's L2.  (Launched as one workgroup per block, the four
    // readers of a region started whenever a CU came free: 103 GB fetched for 40 GB of words, and the kernel ran at the speed of that traffic.)
    const uint32_t G = gridDim.x;
    const bool xmap = a.logB >= 3 && G % (8u * A) == 0u;
    const int rem = a.bits - a.logQ;                                 // hash bits below the block bits
    const int psh = rem + 4;                                         // the part bits of a word start here (>= 36 when HI)
    const int lowb = HI ? rem - 32 : 0;                              // hash bits below the 32 that address the table
    const uint32_t fmask = HI ? (A - 1u) << (psh - 32) : 0u;         // the part bits in a word's upper half
    uint32_t region = 0, part = 0, pshift = 0;
    uint64_t j = 0;
    uint8_t *piece0 = nullptr;
    unsigned long long *q = s_q + (size_t)wv * (AP_Q + AP_SQ), *sq = q + AP_Q;      // kept words ; words for the insert loop
    uint32_t *rb = s_rb + (size_t)wv * rbw;
    for (uint32_t rnd = 0; rnd < a.rounds; rnd++) {
    if (xmap) {
        const uint32_t xcd = blockIdx.x & 7u, ls = blockIdx.x >> 3, gpr = (G >> 3) >> shA;      // groups of A workgroups per XCD
        region = ((rnd * gpr + (ls >> shA)) << 3) + xcd; part = ls & (A - 1u);
    } else { const uint32_t blk = rnd * G + blockIdx.x; region = blk >> shA; part = blk & (A - 1u); }
    j = ((uint64_t)region << shA) + part;                            // the block's place in the order of H
    pshift = HI ? part << (psh - 32) : 0u;
    piece0 = a.pieces + j * (uint64_t)S * (cap / 2);

    for (uint32_t i = tid; i < total_slots; i += AP_THREADS) s_tab[i] = 0ull;
    for (uint32_t i = tid; i < total_slots / 4u; i += AP_THREADS) reinterpret_cast<uint32_t *>(s_fp)[i] = 0u;
    if (!COUNT_ONLY) for (uint32_t i = tid; i < AP_WAVES * rbw; i += AP_THREADS) s_rb[i] = 0u;
    if (tid < CTL_WORDS) s_ctl[tid] = 0u;
    if (tid == 0 && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_fp != 0u) atomicOr(a.overflow, 4);      // (a look-up takes a slot's number for the address of its byte)
    if (xmap && A > 1u && a.bar && tid == 0) {                       // the region's readers start together (bounded: late ones are not waited for for ever)
        atomicAdd(&a.bar[region], 1);
        for (int it = 0; it < (1 << 18) && __hip_atomic_load(&a.bar[region], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)A; it++) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();

    // region offsets and fills are read through the constant address space: scalar loads
    typedef const uint64_t __attribute__((address_space(4))) *cq_t;
    typedef const uint32_t __attribute__((address_space(4))) *c1_t;
    cq_t c_off = (cq_t)(uintptr_t)a.off;
    c1_t c_raw = (c1_t)(uintptr_t)a.raw;
    const uint64_t rstride = 1ull << a.logB;                         // regions per sample
    // wave-uniform: the fills of the two queues -- the queue of kept words as the LDS byte address of its end (what the filter step adds to: the
    // step's two write addresses and the new end are two scalar shift-adds)
    const uint32_t qbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long *)q, qfull = qbase + 128u * 8u;
    uint32_t qw = qbase, nsq = 0;
    AP_PROF_START;
    __builtin_amdgcn_s_setprio(AP_PRIO_STREAM);

    // a word's key as the table holds it: hi = the top 32 of its low hash bits (= what the home slot is computed from), lo = the rest << 14
    auto key_hi = [&](uint32_t wlo, uint32_t whi) -> uint32_t {
        if (HI) return __builtin_amdgcn_alignbit(whi, wlo, (uint32_t)(rem - 28));
        return rem == 0 ? 0u : (uint32_t)(((wlo >> 4) | (whi << 28)) << (32 - rem));
    };
    // home slot: monotone in the key.  The table's size is a power of two (append_ok): a shift -- the multiply-high it stands for runs at a
    // quarter of the rate, and the look-ups are bound by their vector instructions
    const uint32_t hshift = (uint32_t)__builtin_clz(a.nslots) + 1u;
    auto home = [&](uint32_t kh) -> uint32_t { return kh >> hshift; };
    auto key_lo = [&](uint32_t wlo) -> uint32_t { return HI ? __builtin_amdgcn_ubfe(wlo, 4u, (uint32_t)lowb) << AP_RANK_BITS : 0u; };
    auto key_fp = [&](uint32_t kh, uint32_t kl) -> uint32_t {          // a byte of the key, 1..255 (0 = an empty slot)
        const uint32_t v = HI ? kh & 0xFFu : ((kh ^ (kl >> AP_RANK_BITS)) * 0x9E3779B1u) >> 24;      // (the home slot comes from kh's top bits: its low byte is independent of it)
        return v | 1u;
    };
    // the cell of (current sample, rank1 - 1) takes base set m4: one LDS atomic without a return value -- what the cells add up to per row (present,
    // unambiguous, code set) is counted from the finished pieces by pieces_stats_kernel, 4 bits per cell and no atomics, instead of here
    auto record = [&](uint32_t rank1, uint32_t m4) {
        const uint32_t rank = rank1 - 1u;
        atomicOr(&rb[rank >> 3], m4 << ((rank & 7u) * 4u));
    };
    // up to 64 words of the insert queue: the full probe sequence, new keys inserted
    auto slow_batch = [&]() {
        __builtin_amdgcn_s_setprio(AP_PRIO_INSERT);
        const uint32_t take = nsq < 64u ? nsq : 64u;
        nsq -= take;
        if (lane == 0) { AP_COUNT(8, 1); AP_COUNT(9, take); }
        if ((uint32_t)lane < take) {
            const unsigned long long w = sq[nsq + lane];
            const uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
            const uint32_t kh = key_hi(wlo, whi), kl = key_lo(wlo);
            const unsigned long long keyE = ((unsigned long long)kh << 32) | kl;
            // A row gets its rank from the lane whose CAS put its key into the table -- AFTER the CAS (sixteen waves meet the same new key in
            // sixteen related samples: a rank taken before the CAS would be wasted fifteen times).  The key goes in with the rank field all ones;
            // who finds it so waits for the rank (bounded).  Two statements, in this order: the lanes that won write their ranks before any lane
            // of the same wave waits for one.
            // four slots per step of the probe sequence (at 5 500 rows in 8 192 slots a few keys in a hundred sit eight slots and more from home,
            // the unluckiest of 64 some thirty: one slot per step made this loop two thirds of the kernel's time)
            uint32_t rank1 = 0;
            for (uint32_t t = home(kh); t + 3u < total_slots;) {
                AP_COUNT(10, 1); if (ap_mbcnt(__ballot(true)) == 0u) AP_COUNT(11, 1);
                const unsigned long long e0 = s_tab[t], e1 = s_tab[t + 1], e2 = s_tab[t + 2], e3 = s_tab[t + 3];
                const bool m0 = (e0 ^ keyE) <= AP_RANK_MASK, m1 = (e1 ^ keyE) <= AP_RANK_MASK, m2 = (e2 ^ keyE) <= AP_RANK_MASK, m3 = (e3 ^ keyE) <= AP_RANK_MASK;
                bool won = false;
                uint32_t at = 0;
                if (!(m0 | m1 | m2 | m3)) {
                    // not in these four: into the first empty one, if there is one (a lost race for it: the same four again)
                    at = e0 == 0ull ? 0u : e1 == 0ull ? 1u : e2 == 0ull ? 2u : e3 == 0ull ? 3u : 4u;
                    if (at < 4u) won = atomicCAS(&s_tab[t + at], 0ull, keyE | AP_RANK_MASK) == 0ull;
                }
                if (won) {
                    const uint32_t mine = atomicAdd(&s_ctl[CTL_NROWS], 1u) + 1u;
                    if (mine > (COUNT_ONLY ? AP_RANK_MASK - 1u : cap)) s_ctl[CTL_FAIL] = 1u;
                    lds_vol(&s_tab[t + at])[0] = kl | (mine < AP_RANK_MASK ? mine : AP_RANK_MASK - 1u);
                    s_fp[t + at] = (unsigned char)key_fp(kh, kl);
                    rank1 = mine;
                }
                __builtin_amdgcn_wave_barrier();
                if (m0 | m1 | m2 | m3) {
                    const uint32_t mt = t + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u);
                    uint32_t r = (uint32_t)(m0 ? e0 : m1 ? e1 : m2 ? e2 : e3) & AP_RANK_MASK;
                    for (int it = 0; it < (1 << 16) && r == AP_RANK_MASK; it++) r = lds_vol(&s_tab[mt])[0] & AP_RANK_MASK;
                    if (r == AP_RANK_MASK) { s_ctl[CTL_FAIL] = 1u; r = 0; }
                    rank1 = r;
                    break;
                }
                if (won) break;
                if (at == 4u) t += 4u;
            }
            if (rank1 > cap && !COUNT_ONLY) rank1 = 0;
            if (rank1 == 0) s_ctl[CTL_FAIL] = 1u;
            else if (!COUNT_ONLY) record(rank1, wlo & 15u);
        }
        __builtin_amdgcn_s_setprio(AP_PRIO_STREAM);
    };
    // up to 128 queued words, two per lane, against their home slots and the slots behind them: the two look-ups are independent, so every
    // LDS round trip of the chain queue -> table -> row buffer serves two words (the kernel waits for these round trips, it does not compute:
    // VALU 47 % busy with one word per lane).  What is not found there goes to the insert queue.
    // (full_tag: a batch of 128 -- every batch but a sample's last: no lane is idle, nothing is masked)
    auto batch = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        __builtin_amdgcn_s_setprio(AP_PRIO_LOOKUP);
        uint32_t nq = (qw - qbase) >> 3;
        const uint32_t take = FULL ? 128u : nq < 128u ? nq : 128u;
        nq -= take;
        qw -= take * 8u;
        if (lane == 0) { AP_COUNT(6, 1); AP_COUNT(7, take); }
        const bool va = FULL || (uint32_t)lane < take, vb = FULL || (uint32_t)lane + 64u < take;
        unsigned long long wa = 0, wb = 0;
        if (va) wa = q[nq + lane];
        if (vb) wb = q[nq + 64u + lane];
        const uint32_t alo = (uint32_t)wa, ahi = (uint32_t)(wa >> 32), blo = (uint32_t)wb, bhi = (uint32_t)(wb >> 32);
        const uint32_t kha = key_hi(alo, ahi), kla = key_lo(alo), khb = key_hi(blo, bhi), klb = key_lo(blo);
        const uint32_t hsa = home(kha), hsb = home(khb);      // (an idle lane probes the slot of key 0: harmless)
        // Two small reads instead of a scan of the slots: the bytes of the eight slots around home (one byte of every key, odd; 0 = empty) say
        // where the key may be, then that one entry is read and compared in full.  (Reading the entries of 2 x 8 slots per word: 33 ms; 2 x 4:
        // 29 ms.)  A key lies up to seven slots behind its home for all but two words in a hundred, and those -- with the one in 64 whose byte
        // matches a wrong key first -- take the insert loop.  In as few vector operations as it takes (the kernel answers to every one of
        // them: profiles/r04zzh_ab_append_lean.log): (x - 0x01..01) & ~x & 0x80..80 flags every zero byte of x and, above a zero byte only, a
        // byte that is 1 -- the lowest flag is always a true match, and a false one behind a match in front of home just sends the word to the
        // insert loop; v_ffbl gives all ones for "no flag", which survives the OR with 32 and the minimum; no match among the bytes read: the
        // slot behind them is read instead (in the table's slack at worst), and that entry is either the key -- a hit all the same -- or
        // not.  Returns the entry's byte offset in the table.
        auto where8 = [&](uint32_t hs, uint32_t fp) -> uint32_t {
            typedef const uint32_t __attribute__((address_space(3))) *lds_c32_t;
            const lds_c32_t f = (lds_c32_t)(uintptr_t)(hs & ~3u);                         // (s_fp is LDS address 0 -- checked at the kernel's start: the slot's number is the address)
            const uint32_t fp4 = __builtin_amdgcn_perm(fp, fp, 0u);
            const uint32_t x0 = f[0] ^ fp4, x1 = f[1] ^ fp4;
            const uint32_t hs8 = hs << 3;
            const uint32_t z0 = (x0 - 0x01010101u) & ~x0 & (0x80808080u << (hs8 & 31u)), z1 = (x1 - 0x01010101u) & ~x1 & 0x80808080u;
            uint32_t p0, p1;
            asm("v_ffbl_b32 %0, %1" : "=v"(p0) : "v"(z0));                  // (all ones for 0: the compiler's own count-trailing-zeros guards that case with two more instructions)
            asm("v_ffbl_b32 %0, %1" : "=v"(p1) : "v"(z1));
            p1 |= 32u;
            const uint32_t p = (p0 < p1 ? p0 : p1) & ~7u;                   // bit 8 i + 7 -> the entry's byte offset 8 i
            return (hs8 & ~31u) + (p < 64u ? p : 64u);
        };
        // (entry ^ key)'s low half is the rank field (rank + 1) when the low key bits agree and something above it when they do not: less one,
        // a single unsigned compare tells a rank from "not the key", from an empty slot (0) and from a rank still to come (all ones)
        const uint32_t oa = where8(hsa, key_fp(kha, kla)), ob = where8(hsb, key_fp(khb, klb));
        const unsigned long long ea = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const unsigned char *>(s_tab) + oa),
                                 eb = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const unsigned char *>(s_tab) + ob);
        const uint32_t ra0 = ((uint32_t)ea ^ kla) - 1u, rb0 = ((uint32_t)eb ^ klb) - 1u;
        const bool hita = va & ((uint32_t)(ea >> 32) == kha) & (ra0 < AP_RANK_MASK - 1u), hitb = vb & ((uint32_t)(eb >> 32) == khb) & (rb0 < AP_RANK_MASK - 1u);
        const unsigned long long xa = __builtin_amdgcn_ballot_w64(va & !hita), xb = __builtin_amdgcn_ballot_w64(vb & !hitb);      // (taken before the branches below: after them the compiler carries the conditions through registers)
        if (!COUNT_ONLY) {
            if (hita) atomicOr(rb + __builtin_amdgcn_ubfe(ra0, 3u, 11u), (alo & 15u) << ((ra0 << 2) & 31u));
            if (hitb) atomicOr(rb + __builtin_amdgcn_ubfe(rb0, 3u, 11u), (blo & 15u) << ((rb0 << 2) & 31u));
        }
        if (xa | xb) {
            // (the misses of the lanes' first words, then of their second words: either lot fits the emptied insert queue)
            const uint32_t na = (uint32_t)__popcll(xa), nb = (uint32_t)__popcll(xb);
            if (nsq + na > AP_SQ) slow_batch();
            if (va & !hita) sq[nsq + ap_mbcnt(xa)] = wa;
            nsq += na;
            if (nsq + nb > AP_SQ) slow_batch();
            if (vb & !hitb) sq[nsq + ap_mbcnt(xb)] = wb;
            nsq += nb;
        }
        __builtin_amdgcn_s_setprio(AP_PRIO_STREAM);
    };

    // the wave's stream of chunks: (sample, chunk) for its samples in turn, eight 16-byte loads per lane and chunk.  The loads are inline
    // assembly and waited for by hand (nothing else in the loop is a vector-memory load); the words are consumed from the registers they
    // arrive in, by volatile assembly placed behind the wait, and each register quad is asked for again -- the same load of the next chunk --
    // as soon as it has been consumed: no copies (32 moves per chunk before), no second set of registers.
    u32x4 nxt[AP_CH];
    // one 16-byte load per lane of a chunk (load r of the chunk at b), into the registers the same load of the chunk before was consumed from
    // a moment ago.  (No clamping of addresses: a chunk may reach past its region's fill -- the words there fail the fill test -- and past its
    // capacity into the next region; the buffer ends in slack.)
    auto issue_one = [&](int r, uint64_t b) {
        const uint32_t vo = (uint32_t)lane * 16u;
        const uint64_t bq = b + (uint64_t)(r >> 2) * 4096u;
        switch (r & 3) {
        case 0: asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(nxt[r]) : "v"(vo), "s"(bq) : "memory"); break;
        case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(nxt[r]) : "v"(vo), "s"(bq) : "memory"); break;
        case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(nxt[r]) : "v"(vo), "s"(bq) : "memory"); break;
        default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(nxt[r]) : "v"(vo), "s"(bq) : "memory"); break;
        }
    };
    auto region_of = [&](int smp, uint64_t &off, uint32_t &cnt) { const uint64_t rr = (uint64_t)smp * rstride + region; off = c_off[rr]; cnt = c_raw[rr]; };
    if (wv < S) {
        int s = wv;                                                   // the sample of the chunk in hand
        uint64_t off_c, off_n = 0; uint32_t cnt_c, cnt_n = 0, c = 0;
        region_of(s, off_c, cnt_c);
        {
            const uint64_t b0 = (uint64_t)(uintptr_t)(a.words + off_c);
#pragma unroll
            for (int r = 0; r < AP_CH; r++) issue_one(r, b0);
        }
        for (;;) {
            // (the wait and the moves are volatile assembly, in this order: a plain copy may be placed in front of the wait by the compiler,
            // which knows nothing of loads still on their way into these registers)
            AP_PROF(1);
            const uint32_t nch = (cnt_c + 64u * AP_CH * 2u - 1u) / (64u * AP_CH * 2u);
            const bool last_chunk = c + 1u >= nch;                    // (an empty region has its one, empty, chunk)
            // the chunk after this one: the sample's next, or the first of the wave's next sample
            // (ONE place where loads are issued: two would be given registers of their own and be joined by copies at the loop's end -- of
            // registers whose loads are still on their way)
            const bool more = !last_chunk || s + AP_WAVES < S;
            if (last_chunk && more) region_of(s + AP_WAVES, off_n, cnt_n);
            // No copies: load r of the next chunk is issued into nxt[r] as soon as load r of this chunk has been taken from it, so the loads
            // younger than the one about to be consumed are always seven (7 - r of this chunk, r of the next; they return in order; a
            // sample's piece stores in between only make the wait longer) -- s_waitcnt vmcnt(7) in front of every load's words.  After the
            // wave's last chunk the same chunk is asked for again (never looked at: no branch around the loads) and waited for at the end.
            const uint64_t nb1 = (uint64_t)(uintptr_t)(a.words + (more && last_chunk ? off_n : off_c)) + (uint64_t)(more ? (last_chunk ? 0u : c + 1u) : c) * (64u * AP_CH * 16u);
            // keep this block's words -- the two words of a 16-byte load at a time: about 32 of 128 stay --, look full batches up as they come
            // together (straight-line code: a loop over the loads with the batch code in one place spent three quarters of the kernel's time on
            // its own control flow).  The queue holds the 63 words a batch may leave plus 96 of a load; a load that keeps more (a sample that
            // is one repeat) fails the launch and the host takes the sorted path.
            const uint32_t w0 = 2u * (c * (64u * AP_CH) + (uint32_t)lane);
            const uint32_t w0l = 2u * (uint32_t)lane, w1l = w0l + 1u;
            const int fill_left = (int)(cnt_c - c * (64u * AP_CH * 2u));           // words of the region's fill from this chunk's start on (>= 0: c < nch)
            const bool full_chunk = __builtin_amdgcn_readfirstlane((uint32_t)((c + 1u) * (64u * AP_CH * 2u) <= cnt_c)) != 0u;      // every word of the chunk lies inside the region's fill: no fill test
#pragma unroll
            for (int r = 0; r < AP_CH; r++) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AP_CH - 1) : "memory");
                AP_PROF(0);
                uint32_t alo, ahi, blo, bhi;
                if (HI) { alo = nxt[r].x; ahi = nxt[r].y; blo = nxt[r].z; bhi = nxt[r].w; }      // (read by volatile assembly only, below the wait)
                else {
                    asm volatile("v_mov_b32 %0, %1" : "=v"(alo) : "v"(nxt[r].x));
                    asm volatile("v_mov_b32 %0, %1" : "=v"(ahi) : "v"(nxt[r].y));
                    asm volatile("v_mov_b32 %0, %1" : "=v"(blo) : "v"(nxt[r].z));
                    asm volatile("v_mov_b32 %0, %1" : "=v"(bhi) : "v"(nxt[r].w));
                }
                const uint32_t widx = w0 + 128u * r;
#if defined(AP_X_NOFILTER)
                if (false) {
#else
                if (HI) {
#endif
                    // which of the two words are this block's (part bits in the upper half, the region's fill), as lane masks; then the kept
                    // words side by side into the queue.  Written out: the compiler's version of the same spends three times the instructions
                    // on turning conditions into lane masks and back (ten vector instructions here per 128 words, eight where the fill test is skipped).
                    // (the fill test only where a chunk reaches past the region's fill -- a sample's last chunk: one scalar compare otherwise;
                    // word index against fill as w0 against the fill less the load's offset, so no vector add per load either.)  The queue
                    // holds whatever a load keeps on top of what a batch left (127 + 128 <= AP_Q): no test.  All 64 lanes are active here:
                    // the execution mask is set, not saved -- 14 scalar instructions a step where round 4's kernel had 21 (the CU's one
                    // scalar unit is the busiest part of this kernel: profiles/r04zzo_pmc_append_lean.log)
                    unsigned long long ma, mb; uint32_t na, nb, ta, tb, qb, cA;      // (cA: signed -- a load past the fill compares as less than any lane's 2 * lane)
                    // (the load's offset in the fill test is a literal of the instruction: one assembly text per load of the chunk)
#define AP_FILTER_STEP(OFF) asm volatile("v_and_b32 %4, %9, %11\n\t" \
                                 "v_cmp_eq_u32_e64 %0, %10, %4\n\t" \
                                 "v_and_b32 %5, %9, %12\n\t" \
                                 "v_cmp_eq_u32_e64 %1, %10, %5\n\t" \
                                 "s_cmp_lg_u32 %16, 0\n\t" \
                                 "s_cbranch_scc1 .Lapf%=\n\t" \
                                 "s_sub_i32 %7, %13, " #OFF "\n\t" \
                                 "v_cmp_gt_i32_e32 vcc, %7, %14\n\t" \
                                 "s_and_b64 %0, %0, vcc\n\t" \
                                 "v_cmp_gt_i32_e32 vcc, %7, %15\n\t" \
                                 "s_and_b64 %1, %1, vcc\n" \
                                 ".Lapf%=:\n\t" \
                                 "s_bcnt1_i32_b64 %2, %0\n\t" \
                                 "s_bcnt1_i32_b64 %3, %1\n\t" \
                                 "s_lshl3_add_u32 %6, %2, %8\n\t" \
                                 "s_mov_b64 exec, %0\n\t" \
                                 "v_mbcnt_lo_u32_b32 %4, exec_lo, 0\n\t" \
                                 "v_mbcnt_hi_u32_b32 %4, exec_hi, %4\n\t" \
                                 "v_lshl_add_u32 %4, %4, 3, %8\n\t" \
                                 "ds_write2_b32 %4, %17, %11 offset1:1\n\t" \
                                 "s_mov_b64 exec, %1\n\t" \
                                 "v_mbcnt_lo_u32_b32 %5, exec_lo, 0\n\t" \
                                 "v_mbcnt_hi_u32_b32 %5, exec_hi, %5\n\t" \
                                 "v_lshl_add_u32 %5, %5, 3, %6\n\t" \
                                 "ds_write2_b32 %5, %18, %12 offset1:1\n\t" \
                                 "s_mov_b64 exec, -1\n\t" \
                                 "s_lshl3_add_u32 %8, %3, %6" \
                                 : "=&s"(ma), "=&s"(mb), "=&s"(na), "=&s"(nb), "=&v"(ta), "=&v"(tb), "=&s"(qb), "=&s"(cA), "+s"(qw) \
                                 : "s"(fmask), "s"(pshift), "v"(ahi), "v"(bhi), "s"(fill_left), "v"(w0l), "v"(w1l), "s"((uint32_t)full_chunk), "v"(alo), "v"(blo) \
                                 : "vcc", "scc", "memory")
                    static_assert(AP_CH <= 12, "one assembly text per load of a chunk");
                    switch (r) {
                    case 0: AP_FILTER_STEP(0); break; case 1: AP_FILTER_STEP(128); break; case 2: AP_FILTER_STEP(256); break; case 3: AP_FILTER_STEP(384); break;
                    case 4: AP_FILTER_STEP(512); break; case 5: AP_FILTER_STEP(640); break; case 6: AP_FILTER_STEP(768); break; case 7: AP_FILTER_STEP(896); break;
                    case 8: AP_FILTER_STEP(1024); break; case 9: AP_FILTER_STEP(1152); break; case 10: AP_FILTER_STEP(1280); break; default: AP_FILTER_STEP(1408); break;
                    }
#undef AP_FILTER_STEP
#if defined(AP_X_NOFILTER)
                } else if (alo == 0x12345u && bhi == 0x54321u) {
#else
                } else {
#endif
                    const uint32_t pa = (uint32_t)((((uint64_t)ahi << 32) | alo) >> psh), pb = (uint32_t)((((uint64_t)bhi << 32) | blo) >> psh);
                    const bool ka = widx < cnt_c && (pa & (A - 1u)) == part, kbb = widx + 1u < cnt_c && (pb & (A - 1u)) == part;
                    const unsigned long long ba = __ballot(ka), bb = __ballot(kbb);
                    const uint32_t na = (uint32_t)__popcll(ba), nb = (uint32_t)__popcll(bb);
                    const uint32_t nq = (qw - qbase) >> 3;
                    if (ka) q[nq + ap_mbcnt(ba)] = ((unsigned long long)ahi << 32) | alo;      // (127 + 128 <= AP_Q: it fits)
                    if (kbb) q[nq + na + ap_mbcnt(bb)] = ((unsigned long long)bhi << 32) | blo;
                    qw += (na + nb) * 8u;
                }
                issue_one(r, nb1);
                AP_PROF(1);
#if defined(AP_X_NOBATCH)
                if (qw >= qfull) qw -= 128u * 8u;
#else
                while (qw >= qfull) { batch(std::true_type{}); AP_PROF(2);
#if defined(AP_X_NOSLOW)
                    if (nsq >= 64u) nsq -= 64u;
#else
                    while (nsq >= 64u) { slow_batch(); AP_PROF(3); }
#endif
                }
#endif
            }
            if (last_chunk) {
                while (qw != qbase) { batch(std::false_type{}); AP_PROF(2); while (nsq >= 64u) { slow_batch(); AP_PROF(3); } }
                while (nsq) { slow_batch(); AP_PROF(3); }
            }
            AP_PROF(1);
            if (last_chunk) {
                if (!COUNT_ONLY) {
                    uint32_t n = *lds_vol(&s_ctl[CTL_NROWS]);
                    n = __builtin_amdgcn_readfirstlane(n);
                    if (n > cap) n = cap;
                    uint8_t *dst = piece0 + (uint64_t)s * (cap / 2);
                    for (uint32_t v = lane; v * 32u < n; v += 64u) {
                        uint4 *src = reinterpret_cast<uint4 *>(rb) + v;
                        const uint4 x = *src;
                        *reinterpret_cast<uint4 *>(dst + v * 16u) = x;
                        *src = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (lane == 0) a.plen[j * (uint64_t)S + s] = (uint16_t)n;
                }
                AP_PROF(4);
                if (!more) break;
                s += AP_WAVES; c = 0; off_c = off_n; cnt_c = cnt_n;
                // The readers of a region meet again every AP_RESYNC samples of a wave: the workgroup's waves at a barrier, then the region's A
                // workgroups through a counter (bounded).  Readers that drift apart fetch a region's lines up to A times: without the meetings
                // FETCH_SIZE x 2 is 54.5 GB for the 40 GB of words of 1 000 x 5 Mbp, with one every 16 samples 43.8 GB -- for 0.15 ms of the pass's
                // 15.7 (profiles/r05n, r05o; every 4: 42.3 GB and + 0.9 ms; per-wave meetings without the barrier cost more: the waiting waves
                // take issue slots, r05p).  0 = none.
                if (AP_RESYNC > 0) {
                    const int it = (s - wv) / AP_WAVES;                       // the wave's sample number (the same count in every wave up to S / 16)
                    if (xmap && A > 1u && a.bar && it % (AP_RESYNC > 0 ? AP_RESYNC : 1) == 0 && it <= (S - AP_WAVES) / AP_WAVES) {
                        __syncthreads();
                        if (tid == 0) {
                            int *ctr = a.bar + (1 << a.logB) + region;
                            const int want = (int)A * (it / (AP_RESYNC > 0 ? AP_RESYNC : 1));
                            atomicAdd(ctr, 1);
                            for (int spin = 0; spin < (1 << 16) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; spin++) __builtin_amdgcn_s_sleep(4);
                        }
                        __syncthreads();
                    }
                }
            } else c++;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the chunk asked for after the last one)
    }
    AP_PROF_FLUSH();
    __syncthreads();
    const uint32_t nr = s_ctl[CTL_NROWS];                             // ranks handed out (a few may belong to no row: a lost insertion race)
    if (s_ctl[CTL_FAIL]) { if (tid == 0) atomicOr(a.overflow, (int)s_ctl[CTL_FAIL]); __syncthreads(); continue; }
    uint32_t *s_tmp = s_ctl + CTL_TMP;
    if (COUNT_ONLY) {
        uint32_t c = 0;
        for (uint32_t i = tid; i < total_slots; i += AP_THREADS) c += s_tab[i] != 0ull;
        uint32_t tot; (void)ap_block_excl(c, s_tmp, &tot);
        if (tid == 0) { atomicAdd(&a.probe[0], (unsigned long long)tot); atomicMax(&a.probe[1], (unsigned long long)tot); }
        continue;
    }
    // emit in key order: row keys, rank -> row
    uint16_t *s_perm = reinterpret_cast<uint16_t *>(s_q);            // [cap] (the queues are done)
    for (uint32_t i = tid; i < cap; i += AP_THREADS) s_perm[i] = 0xFFFFu;
    __syncthreads();
    const uint32_t per = (total_slots + AP_THREADS - 1) / AP_THREADS;
    const uint32_t lo = tid * per, hi = lo + per < total_slots ? lo + per : total_slots;
    uint32_t c = 0;
    for (uint32_t i = lo; i < hi; i++) c += s_tab[i] != 0ull;
    uint32_t total;
    uint32_t pos = ap_block_excl(c, s_tmp, &total);
    uint64_t *slab = a.stage + j * (uint64_t)a.stride;
    for (uint32_t i = lo; i < hi; i++) {
        const unsigned long long e = s_tab[i];
        if (!e) continue;
        uint32_t gl = 0, lr = 0;                                       // a run of occupied slots is sorted on its own
        for (int64_t t = (int64_t)i - 1; t >= 0; t--) { const unsigned long long o = s_tab[t]; if (!o) break; gl += o > e; }
        for (uint32_t t = i + 1; t < total_slots; t++) { const unsigned long long o = s_tab[t]; if (!o) break; lr += o < e; }
        const uint32_t idx = pos - gl + lr, rank = ((uint32_t)e & AP_RANK_MASK) - 1u;
        pos++;
        if (idx < a.stride && rank < cap) {
            // the low hash bits back from {top 32 ; rest << 14 | rank}
            uint64_t hl;
            if (HI) hl = ((uint64_t)(uint32_t)(e >> 32) << lowb) | (uint64_t)((uint32_t)e >> AP_RANK_BITS);
            else hl = rem == 0 ? 0ull : (uint64_t)((uint32_t)(e >> 32) >> (32 - rem));
            slab[idx] = ((((uint64_t)j << rem) | hl) << 4) | 1ull;
            s_perm[rank] = (uint16_t)idx;
        }
    }
    __syncthreads();
    uint16_t *pj = a.perm + j * (uint64_t)cap;
    for (uint32_t i = tid; i < cap; i += AP_THREADS) pj[i] = s_perm[i];
    if (tid == 0) { a.ncnt[j] = total; a.nrank[j] = nr < cap ? nr : cap; if (total > a.stride) atomicOr(a.overflow, 1); }
    __syncthreads();
    }   // rounds
}

template <bool COUNT_ONLY>
static void launch_append_t(const AppendArgs &a, unsigned blocks, hipStream_t st)
{
    const size_t lds = append_lds_bytes(a.nslots, a.cap, COUNT_ONLY);
    const bool hi = a.bits - a.logQ >= 32;
    if (hi) {
        (void)hipFuncSetAttribute((const void *)append_kernel<COUNT_ONLY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((append_kernel<COUNT_ONLY, true>), dim3(blocks), dim3(AP_THREADS), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)append_kernel<COUNT_ONLY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((append_kernel<COUNT_ONLY, false>), dim3(blocks), dim3(AP_THREADS), lds, st, a);
    }
}
// what the pass takes: hash bits below the block bits that leave room for a rank in a table entry; a table, sixteen row buffers and the
// queues that fit the LDS
bool append_ok(int bits, int logB, int logQ, uint32_t region_cap, uint32_t nslots, uint32_t cap)
{
    const int rem = bits - logQ;
    (void)region_cap;
    return logQ >= logB && rem >= 0 && rem <= 50 && cap % 128u == 0 && cap >= 128u && cap <= APPEND_MAX_CAP && nslots >= cap && (nslots & (nslots - 1u)) == 0u &&
           append_lds_bytes(nslots, cap, false) <= 160u * 1024u - 256u;
}
void launch_append(const AppendArgs &a, uint32_t region_cap, hipStream_t st) { (void)region_cap; launch_append_t<false>(a, (1u << a.logQ) / a.rounds, st); }
void launch_append_probe(const AppendArgs &a, uint32_t region_cap, unsigned blocks, hipStream_t st) { (void)region_cap; launch_append_t<true>(a, blocks, st); }

// ------------------------------------------------------------------------------------------------
// Row statistics from the finished pieces: per first-seen rank of a row block the number of samples with a cell (present), with an
// unambiguous one, and the set of IUPAC codes that occur (merge_ska_array.rs:139-186 counts the same from the rows).  A thread owns one
// 16-byte column of the block's pieces -- 32 ranks, 4 bits each -- and walks the samples: nibble-parallel counters (a byte per rank, folded
// into 16 bits every 255 samples), no atomics, the pieces read once, coalesced.  Cells are single bases almost everywhere: the code set of such
// a row follows from the OR of its cells; rows with an ambiguous cell (a palindrome's W / S, a sample that folded two bases) are found
// afterwards (unambiguous < present) and their code sets taken cell by cell.
// The results belong to ROWS (perm: rank -> row of the block) and rows are what the output arrays are indexed by: a block's three results are
// staged in LDS by row (16 bits each: <= 65 535 samples, code sets are 16 bits) and leave as whole lines -- four arrays x 4 bytes x rows is all
// this kernel writes (round 5 stored them from the ranks' threads, 4 bytes at a time through perm: 3.5 GB of write traffic for 0.36 GB of
// results, profiles/r05_final_pmc_traffic.txt).  A block with more rows than the stage holds (global rows of a sharded job over unrelated
// samples) or split over several workgroups (never at the append pass's capacities) stores per rank as before.
// ------------------------------------------------------------------------------------------------
// (the stage is as large as the launch's piece capacity -- a pass's own rows: nrows <= nrank <= cap -- so that 3 072-rank blocks of 128-bit keys
// keep their occupancy: a fixed 36 KB cost the k = 41 form more than the stores had, profiles/r05v_ab_stats_staged.log)
__global__ __launch_bounds__(256) void pieces_stats_kernel(const uint8_t *pieces, const uint16_t *plen, const uint16_t *perm, const uint32_t *nrank, const uint32_t *ncnt,
                                                           const uint64_t *roff, uint32_t cap, int S, uint32_t *o_present, uint32_t *o_unambig, uint32_t *o_mask,
                                                           uint32_t *o_vcount, int force_direct, uint32_t stage)
{
    constexpr int W = 4;                                              // dwords per thread: one 16-byte load per sample (append_kernel writes the pieces 16 bytes -- 32 ranks -- at a time,
                                                                      // so such a piece of a sample's piece is either written whole or not at all)
    extern __shared__ __attribute__((aligned(16))) uint16_t s_stage[];  // [3][stage]: present, unambiguous, code set -- by row
    const uint32_t PS_STAGE = stage;
    uint16_t *const s_pr = s_stage, *const s_un = s_stage + stage, *const s_mk = s_stage + 2 * (size_t)stage;
    const uint64_t j = blockIdx.x;
    const uint32_t nr = nrank[j], nrows = ncnt[j];
    const uint64_t r0 = roff[j];
    const uint32_t d = blockIdx.y * 256u + threadIdx.x;              // 16-byte column: ranks 32 d .. 32 d + 31
    if (blockIdx.y * 256u * 8u * W >= nr && !(blockIdx.y == 0 && nrows)) return;
    const bool staged = gridDim.y == 1 && nrows <= PS_STAGE && !force_direct;
    if (staged) {
        for (uint32_t p = threadIdx.x; p < nrows; p += 256) { s_pr[p] = 0; s_un[p] = 0; s_mk[p] = 0; }      // (a row without a rank here: no cell in these samples)
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint16_t *pl = plen + j * (uint64_t)S;
    const uint4 *col = reinterpret_cast<const uint4 *>(pieces + j * (uint64_t)S * (cap / 2)) + d;
    const uint32_t step = cap / 32;                                   // 16-byte pieces from one sample's piece to the next
    const bool mine = d * 32u < nr;
    uint32_t pN[W], aN[W];                                            // nibble counters (a rank each): present / ambiguous cells, folded into the byte counters every 12 samples
    uint32_t pE[W], pO[W], aE[W], aO[W], uni[W];                      // byte counters: present / ambiguous cells of the even and odd ranks; OR of the cells
    uint32_t PE0[W], PE1[W], PO0[W], PO1[W], AE0[W], AE1[W], AO0[W], AO1[W];      // the same, 16 bits per rank
#pragma unroll
    for (int w = 0; w < W; w++) { pN[w] = aN[w] = 0; pE[w] = pO[w] = aE[w] = aO[w] = uni[w] = 0; PE0[w] = PE1[w] = PO0[w] = PO1[w] = AE0[w] = AE1[w] = AO0[w] = AO1[w] = 0; }
    auto nibfold = [&]() {
#pragma unroll
        for (int w = 0; w < W; w++) {
            pE[w] += pN[w] & 0x0F0F0F0Fu; pO[w] += (pN[w] >> 4) & 0x0F0F0F0Fu; aE[w] += aN[w] & 0x0F0F0F0Fu; aO[w] += (aN[w] >> 4) & 0x0F0F0F0Fu;
            pN[w] = aN[w] = 0;
        }
    };
    auto fold = [&]() {
        nibfold();
#pragma unroll
        for (int w = 0; w < W; w++) {
            PE0[w] += pE[w] & 0x00FF00FFu; PE1[w] += (pE[w] >> 8) & 0x00FF00FFu; PO0[w] += pO[w] & 0x00FF00FFu; PO1[w] += (pO[w] >> 8) & 0x00FF00FFu;
            AE0[w] += aE[w] & 0x00FF00FFu; AE1[w] += (aE[w] >> 8) & 0x00FF00FFu; AO0[w] += aO[w] & 0x00FF00FFu; AO1[w] += (aO[w] >> 8) & 0x00FF00FFu;
            pE[w] = pO[w] = aE[w] = aO[w] = 0;
        }
    };
    auto take = [&](int w, uint32_t x) {
        uint32_t t = x | (x >> 1); t |= t >> 2;
        const uint32_t nz = t & 0x11111111u;                          // one bit per nibble that is not 0
        const uint32_t y = x & (x - nz);                              // every nibble without its lowest bit
        uint32_t u = y | (y >> 1); u |= u >> 2;
        const uint32_t am = u & 0x11111111u;                          // one bit per nibble with two bases or more
        pN[w] += nz; aN[w] += am;                                     // (a nibble holds 12 samples' worth before it is folded)
        uni[w] |= x;
    };
    // 64 samples at a time: their piece lengths arrive as one vector load (a lane each) and are handed round with v_readlane; four 16-byte loads in
    // flight per lane (a dword per lane and load made 27 M small requests of this pass: 2.7 ms); the byte counters are folded every 192 samples
    for (int s0 = 0, since = 0, nib = 0; s0 < S; s0 += 64) {
        const uint32_t plv = s0 + lane < S ? (uint32_t)pl[s0 + lane] : 0u;
        const int n = S - s0 < 64 ? S - s0 : 64;
#pragma unroll 1
        for (int i0 = 0; i0 < n; i0 += 4) {
            if (nib == 12) { nibfold(); nib = 0; }
            nib += 4;
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)plv, (i0 + u) & 63);      // (past n: length 0)
                x[u] = make_uint4(0u, 0u, 0u, 0u);
                if (mine && i0 + u < n && d * 32u < l) x[u] = col[(uint64_t)(s0 + i0 + u) * step];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { take(0, x[u].x); take(1, x[u].y); take(2, x[u].z); take(3, x[u].w); }
        }
        since += 64;
        if (since >= 192) { fold(); since = 0; nib = 0; }
    }
    fold();
    const uint16_t *pj = perm + j * (uint64_t)cap;
    if (mine) {
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint32_t pres[8], amb[8];
            pres[0] = PE0[w] & 0xFFFFu; pres[4] = PE0[w] >> 16; pres[2] = PE1[w] & 0xFFFFu; pres[6] = PE1[w] >> 16;
            pres[1] = PO0[w] & 0xFFFFu; pres[5] = PO0[w] >> 16; pres[3] = PO1[w] & 0xFFFFu; pres[7] = PO1[w] >> 16;
            amb[0] = AE0[w] & 0xFFFFu; amb[4] = AE0[w] >> 16; amb[2] = AE1[w] & 0xFFFFu; amb[6] = AE1[w] >> 16;
            amb[1] = AO0[w] & 0xFFFFu; amb[5] = AO0[w] >> 16; amb[3] = AO1[w] & 0xFFFFu; amb[7] = AO1[w] >> 16;
            // the eight ranks' rows: one 16-byte load (perm is 2 bytes a rank, the column starts at a multiple of 32 ranks)
            const uint4 pv = *reinterpret_cast<const uint4 *>(pj + d * 32u + (uint32_t)w * 8u);
            const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t r = d * 32u + (uint32_t)w * 8u + i;
                if (r >= nr) break;
                const uint32_t p = (pw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;      // the rank's row of the block (0xFFFF: none)
                if (p >= nrows) continue;
                const uint32_t u = (uni[w] >> (4 * i)) & 15u;
                const uint32_t mk = ((u & 1u) << 1) | ((u & 2u) << 1) | ((u & 4u) << 2) | ((u & 8u) << 5);
                if (staged) { s_pr[p] = (uint16_t)pres[i]; s_un[p] = (uint16_t)(pres[i] - amb[i]); s_mk[p] = (uint16_t)mk; }
                else {
                    o_vcount[r0 + p] = pres[i];                                  // (variant_count: merge_ska_array.rs:172)
                    __hip_atomic_store(&o_present[r0 + p], pres[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&o_unambig[r0 + p], pres[i] - amb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o_mask[r0 + p] = mk;
                }
            }
        }
    }
    if (!staged) __threadfence();                                      // (the walk below reads what the ranks' threads stored)
    __syncthreads();
    // rows with an ambiguous cell: their code sets cell by cell, a wave per row, found by a walk over the workgroup's ranks (64 at a time)
    const uint32_t rk0 = blockIdx.y * 256u * 8u * W, rk1 = rk0 + 256u * 8u * W < nr ? rk0 + 256u * 8u * W : nr;
    for (uint32_t rb = rk0 + (uint32_t)wv * 64u; rb < rk1; rb += 256u) {
        const uint32_t rr = rb + (uint32_t)lane;
        uint32_t pp = 0xFFFFu; bool hit = false;
        if (rr < rk1) {
            pp = pj[rr];
            if (pp < nrows) {
                if (staged) hit = s_un[pp] != s_pr[pp];
                else hit = __hip_atomic_load(&o_unambig[r0 + pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != __hip_atomic_load(&o_present[r0 + pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t r = rb + (uint32_t)src, prow = (uint32_t)__shfl((int)pp, src, 64);
            uint32_t m = 0;
            for (int t = lane; t < S; t += 64) {
                if ((uint32_t)pl[t] <= r) continue;
                const uint32_t x = *(reinterpret_cast<const uint32_t *>(pieces + (j * (uint64_t)S + t) * (cap / 2)) + (r >> 3));
                const uint32_t nib = (x >> ((r & 7u) * 4u)) & 15u;
                if (nib) m |= 1u << nib;
            }
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) m |= __shfl_xor(m, dd, 64);
            if (lane == 0) { if (staged) s_mk[prow] = (uint16_t)m; else o_mask[r0 + prow] = m; }
        }
    }
    if (!staged) return;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < nrows; p += 256) {
        const uint32_t pr = s_pr[p];
        o_present[r0 + p] = pr; o_vcount[r0 + p] = pr;                // (variant_count: merge_ska_array.rs:172)
        o_unambig[r0 + p] = s_un[p]; o_mask[r0 + p] = s_mk[p];
    }
}
void launch_pieces_stats(const uint8_t *pieces, const uint16_t *plen, const uint16_t *perm, const uint32_t *nrank, const uint32_t *ncnt, const uint64_t *roff, uint32_t cap,
                         int n_samples, int n_blocks, uint32_t *present, uint32_t *unambig, uint32_t *mask, uint32_t *vcount, hipStream_t st)
{
    if (n_blocks <= 0) return;
    const unsigned gy = (cap / 32 + 255u) / 256u;
    static const int direct = knob("stats_direct") ? 1 : 0;           // (tests: the per-rank stores of blocks the stage does not hold)
    const uint32_t stage = (cap + 63u) / 64u * 64u;                    // <= 6 016 (12 032 with gy > 1, where nothing is staged): <= 36 KB
    hipLaunchKernelGGL(pieces_stats_kernel, dim3((unsigned)n_blocks, gy), dim3(256), (size_t)stage * 6, st, pieces, plen, perm, nrank, ncnt, roff, cap, n_samples, present, unambig, mask, vcount,
                       direct, stage);
}
// split k-mers per sample (SkaDict::ksize): the cells of its pieces that are not empty; a wave per piece
__global__ __launch_bounds__(256) void pieces_cells_kernel(const uint8_t *pieces, const uint16_t *plen, uint32_t cap, int S, unsigned long long *out)
{
    const uint64_t j = blockIdx.x;
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.y * 4 + (threadIdx.x >> 6); s < S; s += gridDim.y * 4) {
        const uint32_t pl = plen[j * (uint64_t)S + s];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(pieces + (j * (uint64_t)S + s) * (cap / 2));
        uint32_t c = 0;
        for (uint32_t i = lane; i < (pl + 7u) / 8u; i += 64) { const uint32_t x = src[i]; uint32_t t = x | (x >> 1); t |= t >> 2; c += (uint32_t)__popc(t & 0x11111111u); }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) c += __shfl_xor(c, dd, 64);
        if (lane == 0 && c) atomicAdd(&out[s], (unsigned long long)c);
    }
}
void launch_pieces_cells(const uint8_t *pieces, const uint16_t *plen, uint32_t cap, int n_samples, int n_blocks, unsigned long long *out, hipStream_t st)
{
    if (n_blocks <= 0 || n_samples <= 0) return;
    unsigned gy = (unsigned)((n_samples + 3) / 4); if (gy > 16) gy = 16;
    hipLaunchKernelGGL(pieces_cells_kernel, dim3((unsigned)n_blocks, gy), dim3(256), 0, st, pieces, plen, cap, n_samples, out);
}
// windows per sample (the sum of its regions' fills): a sample without any has no valid sequence
__global__ __launch_bounds__(256) void region_totals_kernel(const uint32_t *raw, int logB, unsigned long long *out)
{
    __shared__ unsigned long long s_sum[256];
    const uint64_t B = 1ull << logB;
    unsigned long long t = 0;
    for (uint64_t b = threadIdx.x; b < B; b += 256) t += raw[(uint64_t)blockIdx.x * B + b];
    s_sum[threadIdx.x] = t;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s_sum[threadIdx.x] += s_sum[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = s_sum[0];
}
void launch_region_totals(const uint32_t *raw, int n_samples, int logB, unsigned long long *out, hipStream_t st)
{
    if (n_samples > 0) hipLaunchKernelGGL(region_totals_kernel, dim3((unsigned)n_samples), dim3(256), 0, st, raw, logB, out);
}

// ------------------------------------------------------------------------------------------------
// pieces -> rows x samples cells (sample-major, ASCII), rows in the order of H.
// One workgroup = one row block x a range of samples.  The block's output columns are listed once per workgroup -- src[c] = the first-seen
// rank whose row lands in column c: every row of the block (c = its place in the block), or the rows a filter keeps (c = kpos[row] - kpos[first
// row of the block]) -- then every wave takes samples in turn: the sample's piece into LDS, one dword of four cells per lane from there
// (IUPAC letter of the 4-bit base set; '-' where the rank lies beyond the piece), stored at its own alignment so that a wave writes whole lines.
// ------------------------------------------------------------------------------------------------
constexpr int PR_WAVES = 8;
// 4 base sets (a byte each) -> their IUPAC letters: two byte permutes over "-ACMTWYH" / "GRSVKDBN" and a select on bit 3
__device__ static inline uint32_t ap_iupac4(uint32_t m)
{
    const uint32_t idx = m & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(0x48595754u, 0x4D43412Du, idx);      // bytes 0-3 from "-ACM", 4-7 from "TWYH"
    const uint32_t hi = __builtin_amdgcn_perm(0x4E42444Bu, 0x56535247u, idx);      // "GRSV", "KDBN"
    const uint32_t sel = ((m >> 3) & 0x01010101u) * 0xFFu;
    return (hi & sel) | (lo & ~sel);
}
template <bool KEPT>
__global__ __launch_bounds__(64 * PR_WAVES) void pieces_rows_kernel(PiecesRowsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t cap = a.cap;
    // [cap + 16] which rank an output byte takes its cell from, laid out like the aligned dwords the block's cells are stored in (so that a
    // lane reads the four ranks of its dword at once); `cap` = none: that "rank" lies in a word of the piece buffer that is always 0
    uint16_t *s_src = reinterpret_cast<uint16_t *>(s_raw);
    uint32_t *s_piece = reinterpret_cast<uint32_t *>(s_raw + (((size_t)cap + 16) * 2 + 15) / 16 * 16);      // [PR_WAVES][cap / 8 + 4]
    const uint64_t j = (uint64_t)blockIdx.x + a.j_base;
    const uint32_t n = a.ncnt[j];
    if (n == 0) return;
    const uint64_t r0 = a.roff[j];
    const uint64_t k0 = KEPT ? a.kpos[r0] : 0;
    const uint32_t nout = KEPT ? (uint32_t)(a.kpos[r0 + n] - k0) : n;
    if (nout == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t ocol = KEPT ? k0 : r0 - a.col_base;                        // first output column of the block
    const uint32_t shift = (uint32_t)(ocol & 3u);
    const uint32_t ndw = (nout + shift + 3u) / 4u;                            // aligned dwords that hold the block's cells
    const uint16_t *pj = a.perm + j * (uint64_t)cap;
    const uint32_t nr = a.nrank[j];
    const int S = a.n_samples;
    const int s_lo = blockIdx.y * a.samples_per_wg, s_hi = s_lo + a.samples_per_wg < S ? s_lo + a.samples_per_wg : S;
    uint32_t *pc = s_piece + (size_t)wv * (cap / 8 + 4);
    const unsigned char *pcb = reinterpret_cast<const unsigned char *>(pc);
    const uint32_t nrw = (nr + 7u) / 8u;                                      // piece words that ranks of this block can name
    if (lane < 4) pc[cap / 8 + lane] = 0u;                                    // (rank `cap`: no cell)
    // The block's output columns are taken `cap` at a time (one pass when they are the block's own rows; the rows of a sharded job's hash
    // range -- all ranks' rows, most of them without a cell here -- may be several times as many).  sc = shift + column: the index into the
    // aligned dwords.
    for (uint32_t t0 = 0; t0 < nout + shift; t0 += cap) {
        for (uint32_t i = tid; i < cap + 16u; i += blockDim.x) s_src[i] = (uint16_t)cap;
        __syncthreads();
        for (uint32_t r = tid; r < nr; r += blockDim.x) {
            const uint32_t p = pj[r];
            if (p == 0xFFFFu || p >= n) continue;
            uint32_t sc;
            if (!KEPT) sc = shift + p;
            else if (a.keep[r0 + p] == 1) sc = shift + (uint32_t)(a.kpos[r0 + p] - k0);
            else continue;
            if (sc >= t0 && sc < t0 + cap) s_src[sc - t0] = (uint16_t)r;
        }
        __syncthreads();
        const uint32_t v_lo = t0 / 4u, v_hi = (t0 + cap) / 4u < ndw ? (t0 + cap) / 4u : ndw;      // this pass's dwords
        for (int s = s_lo + wv; s < s_hi; s += PR_WAVES) {
            const uint32_t pl = a.plen[j * (uint64_t)S + s];
            const uint32_t plw = (pl + 7u) / 8u;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.pieces + (j * (uint64_t)S + s) * (cap / 2));
            {   // sixteen bytes per lane and step (the pieces and the wave's buffer are 16-byte aligned: cap is a multiple of 128); a dword per
                // lane made this copy a quarter of the kernel's time (3.95 -> 3.10 ms for the kept rows of 1 000 x 5 Mbp)
                const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
                uint4 *pc4 = reinterpret_cast<uint4 *>(pc);
                for (uint32_t i = lane; i < (nrw + 3u) / 4u; i += 64) {
                    uint4 x = 4u * i < plw ? src4[i] : make_uint4(0u, 0u, 0u, 0u);      // ranks handed out after this sample: no cell
                    if (4u * i + 1u >= plw) x.y = 0u;
                    if (4u * i + 2u >= plw) x.z = 0u;
                    if (4u * i + 3u >= plw) x.w = 0u;
                    pc4[i] = x;
                }
            }
            __builtin_amdgcn_wave_barrier();
            unsigned char *dst = a.out + (uint64_t)s * a.pitch + (ocol - shift);
            for (uint32_t v = v_lo + lane; v < v_hi; v += 64) {
                const uint2 rr = *reinterpret_cast<const uint2 *>(s_src + 4u * (v - v_lo));      // four ranks
                const uint32_t ra = rr.x & 0xFFFFu, rb = rr.x >> 16, rc = rr.y & 0xFFFFu, rd = rr.y >> 16;
                const uint32_t na = ((uint32_t)pcb[ra >> 1] >> ((ra & 1u) * 4u)) & 15u, nb = ((uint32_t)pcb[rb >> 1] >> ((rb & 1u) * 4u)) & 15u;
                const uint32_t nc = ((uint32_t)pcb[rc >> 1] >> ((rc & 1u) * 4u)) & 15u, nd = ((uint32_t)pcb[rd >> 1] >> ((rd & 1u) * 4u)) & 15u;
                uint32_t m = na | (nb << 8) | (nc << 16) | (nd << 24);
                uint32_t word = ap_iupac4(m);
                if (a.mask_ambig) {                                                 // cells of two bases and more are written as 'N'
                    const uint32_t y = m & (m - ((m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x01010101u));      // every byte without its lowest bit
                    const uint32_t amb = ((y | (y >> 1) | (y >> 2) | (y >> 3)) & 0x01010101u) * 0xFFu;
                    word = (word & ~amb) | (0x4E4E4E4Eu & amb);
                }
                const bool head = v == 0 && shift != 0, tail = v == ndw - 1 && ((nout + shift) & 3u) != 0;
                if (!head && !tail) *reinterpret_cast<uint32_t *>(dst + 4u * v) = word;
                else {
#pragma unroll
                    for (int b = 0; b < 4; b++) { const int c = (int)(4u * v + b) - (int)shift; if (c >= 0 && (uint32_t)c < nout) dst[4u * v + b] = (unsigned char)(word >> (8 * b)); }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
}
void launch_pieces_rows(const PiecesRowsArgs &a, uint32_t n_blocks, hipStream_t st)
{
    if (!n_blocks || a.n_samples <= 0) return;
    PiecesRowsArgs b = a;
    // enough workgroups to fill the chip several times over, few enough that listing a block's columns is shared by many samples
    int spw = a.n_samples;
    while (spw > PR_WAVES * 4 && (uint64_t)n_blocks * ((a.n_samples + spw - 1) / spw) < 8192) spw = (spw + 1) / 2;
    b.samples_per_wg = spw;
    const unsigned gy = (unsigned)((a.n_samples + spw - 1) / spw);
    const size_t lds = (((size_t)a.cap + 16) * 2 + 15) / 16 * 16 + (size_t)PR_WAVES * (a.cap / 8 + 4) * 4;
    if (a.keep) {
        (void)hipFuncSetAttribute((const void *)pieces_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pieces_rows_kernel<true>, dim3(n_blocks, gy), dim3(64 * PR_WAVES), lds, st, b);
    } else {
        (void)hipFuncSetAttribute((const void *)pieces_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pieces_rows_kernel<false>, dim3(n_blocks, gy), dim3(64 * PR_WAVES), lds, st, b);
    }
}

}  // namespace skx

#include "skx_append_wide.inc"
