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
//   * a region holds the words of A = 2^(logQ - logB) row blocks.  The A workgroups of a region run on the same XCD at the same time
//     (blockIdx -> XCD is b % 8), each reads the whole region and keeps its share: the first reader takes a line from HBM, the others find it
//     in that XCD's L2 (or the memory-side cache).  A wave reads 512 words at a time (the next 512 are on their way meanwhile), keeps its
//     block's, and compacts them into a queue of its own so that the look-ups run with full waves: 64 words at a time against the home slot
//     and its successor; the few words that sit further from home go to a second queue and through the insert loop 64 at a time as well
//     (looked up in place, they would make every batch wait for its unluckiest lane);
//   * a sample's cells of the block are OR-ed into the wave's row buffer of 4-bit base sets indexed by rank (LDS atomics; the returned old value
//     tells a first sighting -- counted in the row's statistics and the sample's k-mer count -- from a repeat, which is where ska_dict.rs folds);
//   * the row buffer leaves as the sample's *piece* of the block: plen ranks, two per byte, at a fixed place (pieces[(j * S + s) * cap / 2]).
//     Ranks beyond plen were handed out later: the cell is '-'.  The rows x samples matrix in the order of H is produced from the pieces
//     by pieces_rows_kernel (all rows, a window of row blocks, or only the rows a filter keeps) -- 1 byte per cell there, 4 bits and no cell
//     for unseen rows here;
//   * per row the pass keeps a present count and the union of the bases seen (two LDS atomics per first sighting); rows where a cell is
//     ambiguous (a palindrome's W / S, or a sample that folded two bases) are marked and get their statistics from the finished pieces;
//   * when the last sample is in, the table is emitted in key order: the block's row keys, perm (rank -> row of the block) and the row statistics.
#include "skx_device.h"
#include <cstdlib>

namespace skx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *g4_t;

// Measurement only (-DSKX_AP_PROF): cycles of every wave's lane 0 in the phases of append_kernel, summed over waves, workgroups and launches:
// 0 waiting for a chunk's words, 1 filter into the queue, 2 look-ups of full batches, 3 the insert loop's batches, 4 end of a sample (drain + piece), 5 the tail (emit)
#ifdef SKX_AP_PROF
__device__ unsigned long long g_ap_prof[16];
#define AP_PROF_START unsigned long long tprof = __builtin_readcyclecounter(), tacc[6] = {0, 0, 0, 0, 0, 0}
#define AP_PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[(i)] += t_ - tprof; tprof = t_; } while (0)
#define AP_PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 6; i_++) atomicAdd(&g_ap_prof[i_], tacc[i_]); } while (0)
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

constexpr int AP_THREADS = 1024, AP_WAVES = 16;
constexpr uint32_t AP_PAD = 64;             // slots behind the table's last home slot (probing does not wrap)
constexpr uint32_t AP_Q = 160;              // entries of a wave's queue of kept words: 63 left over + the ~32 +- 5 of one load (more: a batch goes first); the insert loop's queue holds 128
constexpr int AP_CH = 4;                    // 16-byte loads per lane and chunk: a wave reads 512 words at a time
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

// ctl words: 0 ranks handed out, 1 failure, 2 a row is dirty, 8..24 scan scratch
constexpr int CTL_NROWS = 0, CTL_FAIL = 1, CTL_DIRTY = 2, CTL_TMP = 8, CTL_WORDS = 32;

static inline size_t append_lds_bytes(uint32_t nslots, uint32_t cap, bool count_only)
{
    size_t b = (size_t)(nslots + AP_PAD) * 8 + (size_t)AP_WAVES * (AP_Q + 128u) * 8 + CTL_WORDS * 4;
    if (!count_only) b += (size_t)(cap / 2) * 4 + (size_t)(cap / 8) * 4 + (size_t)(cap / 32) * 4 + (size_t)AP_WAVES * (cap / 8) * 4;
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
    unsigned long long *s_tab = reinterpret_cast<unsigned long long *>(s_raw);
    unsigned long long *s_q = s_tab + total_slots;                   // [AP_WAVES][AP_Q kept words + 128 words for the insert loop]
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_q + (size_t)AP_WAVES * (AP_Q + 128u));
    uint32_t *s_cnt = s_ctl + CTL_WORDS;                             // [cap / 2] present counts, two ranks per word
    uint32_t *s_uni = s_cnt + cap / 2;                               // [cap / 8] union of the bases seen, eight ranks per word
    uint32_t *s_dirty = s_uni + cap / 8;                             // [cap / 32] ranks whose statistics are taken from the finished cells
    uint32_t *s_rb = s_dirty + cap / 32;                             // [AP_WAVES][rbw] 4-bit base sets by rank, one buffer per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.n_samples;
    const int shA = a.logQ - a.logB;
    const uint32_t A = 1u << shA;
    uint32_t region, part;
    if (a.logB >= 3) { const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; region = ((slot >> shA) << 3) + xcd; part = slot & (A - 1u); }
    else { region = blockIdx.x >> shA; part = blockIdx.x & (A - 1u); }
    const uint64_t j = ((uint64_t)region << shA) + part;             // the block's place in the order of H
    const int rem = a.bits - a.logQ;                                 // hash bits below the block bits
    const int psh = rem + 4;                                         // the part bits of a word start here (>= 36 when HI)
    const int lowb = HI ? rem - 32 : 0;                              // hash bits below the 32 that address the table
    const uint32_t fmask = HI ? (A - 1u) << (psh - 32) : 0u, pshift = HI ? part << (psh - 32) : 0u;      // the part bits in a word's upper half

    for (uint32_t i = tid; i < total_slots; i += AP_THREADS) s_tab[i] = 0ull;
    if (!COUNT_ONLY) {
        for (uint32_t i = tid; i < cap / 2; i += AP_THREADS) s_cnt[i] = 0u;
        for (uint32_t i = tid; i < cap / 8; i += AP_THREADS) s_uni[i] = 0u;
        for (uint32_t i = tid; i < cap / 32; i += AP_THREADS) s_dirty[i] = 0u;
        for (uint32_t i = tid; i < AP_WAVES * rbw; i += AP_THREADS) s_rb[i] = 0u;
    }
    if (tid < CTL_WORDS) s_ctl[tid] = 0u;
    __syncthreads();

    // region offsets and fills are read through the constant address space: scalar loads
    typedef const uint64_t __attribute__((address_space(4))) *cq_t;
    typedef const uint32_t __attribute__((address_space(4))) *c1_t;
    cq_t c_off = (cq_t)(uintptr_t)a.off;
    c1_t c_raw = (c1_t)(uintptr_t)a.raw;
    const uint64_t rstride = 1ull << a.logB;                         // regions per sample
    unsigned long long *q = s_q + (size_t)wv * (AP_Q + 128u), *sq = q + AP_Q;      // kept words ; words for the insert loop
    uint32_t *rb = s_rb + (size_t)wv * rbw;
    uint8_t *piece0 = a.pieces + j * (uint64_t)S * (cap / 2);
    uint32_t nq = 0, nsq = 0, firsts = 0;                            // wave-uniform: queue fills, first sightings of the current sample
    AP_PROF_START;

    // a word's key as the table holds it: hi = the top 32 of its low hash bits (= what the home slot is computed from), lo = the rest << 14
    auto key_hi = [&](uint32_t wlo, uint32_t whi) -> uint32_t {
        if (HI) return __builtin_amdgcn_alignbit(whi, wlo, (uint32_t)(rem - 28));
        return rem == 0 ? 0u : (uint32_t)(((wlo >> 4) | (whi << 28)) << (32 - rem));
    };
    auto key_lo = [&](uint32_t wlo) -> uint32_t { return HI ? __builtin_amdgcn_ubfe(wlo, 4u, (uint32_t)lowb) << AP_RANK_BITS : 0u; };
    // the cell of (current sample, rank1 - 1) takes base set m4; first sightings are counted
    auto record = [&](uint32_t rank1, uint32_t m4) -> bool {
        const uint32_t rank = rank1 - 1u, sh = (rank & 7u) * 4u, di = rank >> 3, val = m4 << sh;
        const uint32_t old = atomicOr(&rb[di], val);
        const uint32_t on = __builtin_amdgcn_ubfe(old, sh, 4u);
        if (on == 0u) {
            atomicOr(&s_uni[di], val);
            atomicAdd(&s_cnt[rank >> 1], 1u << (16u * (rank & 1u)));
            if (m4 & (m4 - 1u)) { atomicOr(&s_dirty[rank >> 5], 1u << (rank & 31u)); s_ctl[CTL_DIRTY] = 1u; }      // a palindrome's two bases: an ambiguous cell
            return true;
        }
        if ((on | m4) != on) {                                         // the sample has this split k-mer again with another middle base (ska_dict.rs:92-101)
            atomicOr(&s_dirty[rank >> 5], 1u << (rank & 31u)); s_ctl[CTL_DIRTY] = 1u;
        }
        return false;
    };
    // up to 64 queued words against their home slot and its successor; what is not there goes to the insert queue
    auto slow_batch = [&]() {
        const uint32_t take = nsq < 64u ? nsq : 64u;
        nsq -= take;
        bool first = false;
        if ((uint32_t)lane < take) {
            const unsigned long long w = sq[nsq + lane];
            const uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
            const uint32_t kh = key_hi(wlo, whi), kl = key_lo(wlo);
            const unsigned long long keyE = ((unsigned long long)kh << 32) | kl;
            // A row gets its rank from the lane whose CAS put its key into the table -- AFTER the CAS (sixteen waves meet the same new key in
            // sixteen related samples: a rank taken before the CAS would be wasted fifteen times).  The key goes in with the rank field all ones;
            // who finds it so waits for the rank (bounded).  Two statements, in this order: the lanes that won write their ranks before any lane
            // of the same wave waits for one.
            uint32_t rank1 = 0;
            for (uint32_t t = __umulhi(kh, a.nslots); t < total_slots; t++) {
                unsigned long long e = s_tab[t];
                bool won = false;
                if (e == 0ull) { e = atomicCAS(&s_tab[t], 0ull, keyE | AP_RANK_MASK); won = e == 0ull; }
                if (won) {
                    const uint32_t mine = atomicAdd(&s_ctl[CTL_NROWS], 1u) + 1u;
                    if (mine > (COUNT_ONLY ? AP_RANK_MASK - 1u : cap)) s_ctl[CTL_FAIL] = 1u;
                    reinterpret_cast<volatile uint32_t *>(&s_tab[t])[0] = kl | (mine < AP_RANK_MASK ? mine : AP_RANK_MASK - 1u);
                    rank1 = mine;
                }
                __builtin_amdgcn_wave_barrier();
                if (!won && (e ^ keyE) <= AP_RANK_MASK) {
                    uint32_t r = (uint32_t)e & AP_RANK_MASK;
                    for (int it = 0; it < (1 << 16) && r == AP_RANK_MASK; it++) r = reinterpret_cast<volatile uint32_t *>(&s_tab[t])[0] & AP_RANK_MASK;
                    if (r == AP_RANK_MASK) { s_ctl[CTL_FAIL] = 1u; r = 0; }
                    rank1 = r;
                    break;
                }
                if (won) break;
            }
            if (rank1 > cap && !COUNT_ONLY) rank1 = 0;
            if (rank1 == 0) s_ctl[CTL_FAIL] = 1u;
            else if (!COUNT_ONLY) first = record(rank1, wlo & 15u);
        }
        if (!COUNT_ONLY) firsts += (uint32_t)__popcll(__ballot(first));
    };
    auto batch = [&]() {
        const uint32_t take = nq < 64u ? nq : 64u;
        nq -= take;
        bool miss = false, first = false;
        unsigned long long w = 0;
        if ((uint32_t)lane < take) {
            w = q[nq + lane];
            const uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
            const uint32_t kh = key_hi(wlo, whi), kl = key_lo(wlo);
            const uint32_t hs = __umulhi(kh, a.nslots);
            const unsigned long long e0 = s_tab[hs], e1 = s_tab[hs + 1];
            const uint32_t e0l = (uint32_t)e0, e1l = (uint32_t)e1;
            // (both slots are read at once and compared without short cuts: one LDS round trip per batch)
            const bool m0 = ((uint32_t)(e0 >> 32) == kh) & ((e0l ^ kl) <= AP_RANK_MASK);
            const bool m1 = (e0l != 0u) & ((uint32_t)(e1 >> 32) == kh) & ((e1l ^ kl) <= AP_RANK_MASK);
            const uint32_t rank1 = (m0 ? e0l : m1 ? e1l : 0u) & AP_RANK_MASK;
            if (rank1 == 0 || rank1 == AP_RANK_MASK) miss = true;          // not there, further from home -- or there, its rank still being written
            else if (!COUNT_ONLY) first = record(rank1, wlo & 15u);
        }
        if (!COUNT_ONLY) firsts += (uint32_t)__popcll(__ballot(first));
        const unsigned long long mb = __ballot(miss);
        if (mb) {
            if (miss) sq[nsq + ap_mbcnt(mb)] = w;
            nsq += (uint32_t)__popcll(mb);
        }
    };

    // the wave's stream of chunks: (sample, chunk) for its samples in turn.  The next chunk is on its way (nxt) while one is looked at (cur: copied
    // out of nxt once it has arrived -- sixteen moves per 512 words, and one instance of the code below instead of two).  The loads are inline
    // assembly and waited for by hand: nothing else in the loop is a vector-memory load.
    u32x4 nxt[AP_CH], cur[AP_CH];
    auto issue = [&](uint64_t off, uint32_t cnt, uint32_t c) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.words + off);
        const uint32_t last = cnt ? (cnt - 1u) >> 1 : 0u;
#pragma unroll
        for (int r = 0; r < AP_CH; r++) {
            const uint32_t p = c * (64u * AP_CH) + 64u * r + (uint32_t)lane;      // pair index
            const uint32_t vo = (p < last ? p : last) * 16u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(nxt[r]) : "v"(vo), "s"(base) : "memory");
        }
    };
    auto region_of = [&](int smp, uint64_t &off, uint32_t &cnt) { const uint64_t rr = (uint64_t)smp * rstride + region; off = c_off[rr]; cnt = c_raw[rr]; };
    if (wv < S) {
        int s = wv;                                                   // the sample of the chunk in hand
        uint64_t off_c, off_n = 0; uint32_t cnt_c, cnt_n = 0, c = 0;
        region_of(s, off_c, cnt_c);
        issue(off_c, cnt_c, 0u);
        for (;;) {
            // (the wait and the moves are volatile assembly, in this order: a plain copy may be placed in front of the wait by the compiler,
            // which knows nothing of loads still on their way into these registers)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < AP_CH; r++) {
                uint32_t x, y, z, w;
                asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(nxt[r].x));
                asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(nxt[r].y));
                asm volatile("v_mov_b32 %0, %1" : "=v"(z) : "v"(nxt[r].z));
                asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(nxt[r].w));
                cur[r].x = x; cur[r].y = y; cur[r].z = z; cur[r].w = w;
            }
            AP_PROF(0);
            const uint32_t nch = (cnt_c + 64u * AP_CH * 2u - 1u) / (64u * AP_CH * 2u);
            const bool last_chunk = c + 1u >= nch;                    // (an empty region has its one, empty, chunk)
            // the chunk after this one: the sample's next, or the first of the wave's next sample
            // (ONE place where loads are issued: two would be given registers of their own and be joined by copies at the loop's end -- of
            // registers whose loads are still on their way)
            const bool more = !last_chunk || s + AP_WAVES < S;
            if (last_chunk && more) region_of(s + AP_WAVES, off_n, cnt_n);
            if (more) issue(last_chunk ? off_n : off_c, last_chunk ? cnt_n : cnt_c, last_chunk ? 0u : c + 1u);
            // keep this block's words -- the two words of a 16-byte load at a time: about 32 of 128 stay --, look full batches up as they come
            // together (straight-line code: a loop over the loads with the batch code in one place spent three quarters of the kernel's time on
            // its own control flow).  The queue holds the 63 words a batch may leave plus 96 of a load; a load that keeps more (a sample that
            // is one repeat) fails the launch and the host takes the sorted path.
            const uint32_t w0 = 2u * (c * (64u * AP_CH) + (uint32_t)lane);
            const uint32_t qbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long *)q;      // the queue's LDS byte address
#pragma unroll
            for (int r = 0; r < AP_CH; r++) {
                const uint32_t alo = cur[r].x, ahi = cur[r].y, blo = cur[r].z, bhi = cur[r].w;
                const uint32_t widx = w0 + 128u * r;
                if (HI) {
                    // which of the two words are this block's (part bits in the upper half, the region's fill), as lane masks; then the kept
                    // words side by side into the queue.  Written out: the compiler's version of the same spends three times the instructions
                    // on turning conditions into lane masks and back (twelve vector instructions here per 128 words).
                    unsigned long long ma, mb; uint32_t na, nb, ta, tb;
                    asm volatile("v_and_b32 %4, %6, %8\n\t"
                                 "v_cmp_eq_u32_e32 vcc, %7, %4\n\t"
                                 "s_mov_b64 %0, vcc\n\t"
                                 "v_cmp_gt_u32_e32 vcc, %10, %11\n\t"
                                 "s_and_b64 %0, %0, vcc\n\t"
                                 "v_and_b32 %5, %6, %9\n\t"
                                 "v_cmp_eq_u32_e32 vcc, %7, %5\n\t"
                                 "s_mov_b64 %1, vcc\n\t"
                                 "v_cmp_gt_u32_e32 vcc, %10, %12\n\t"
                                 "s_and_b64 %1, %1, vcc\n\t"
                                 "s_bcnt1_i32_b64 %2, %0\n\t"
                                 "s_bcnt1_i32_b64 %3, %1"
                                 : "=&s"(ma), "=&s"(mb), "=&s"(na), "=&s"(nb), "=&v"(ta), "=&v"(tb)
                                 : "s"(fmask), "s"(pshift), "v"(ahi), "v"(bhi), "s"(cnt_c), "v"(widx), "v"(widx + 1u)
                                 : "vcc", "scc");
                    if (nq + na + nb > AP_Q) { if (lane == 0) s_ctl[CTL_FAIL] = 2u; }
                    else {
                        const uint32_t qa = qbase + nq * 8u, qb = qa + na * 8u;
                        unsigned long long sv;
                        asm volatile("s_mov_b64 vcc, %3\n\t"
                                     "v_mbcnt_lo_u32_b32 %0, vcc_lo, 0\n\t"
                                     "v_mbcnt_hi_u32_b32 %0, vcc_hi, %0\n\t"
                                     "v_lshl_add_u32 %0, %0, 3, %5\n\t"
                                     "s_mov_b64 vcc, %4\n\t"
                                     "v_mbcnt_lo_u32_b32 %1, vcc_lo, 0\n\t"
                                     "v_mbcnt_hi_u32_b32 %1, vcc_hi, %1\n\t"
                                     "v_lshl_add_u32 %1, %1, 3, %6\n\t"
                                     "s_and_saveexec_b64 %2, %3\n\t"
                                     "ds_write2_b32 %0, %7, %8 offset1:1\n\t"
                                     "s_mov_b64 exec, %2\n\t"
                                     "s_and_saveexec_b64 %2, %4\n\t"
                                     "ds_write2_b32 %1, %9, %10 offset1:1\n\t"
                                     "s_mov_b64 exec, %2"
                                     : "=&v"(ta), "=&v"(tb), "=&s"(sv)
                                     : "s"(ma), "s"(mb), "s"(qa), "s"(qb), "v"(alo), "v"(ahi), "v"(blo), "v"(bhi)
                                     : "vcc", "scc", "memory");
                        nq += na + nb;
                    }
                } else {
                    const uint32_t pa = (uint32_t)((((uint64_t)ahi << 32) | alo) >> psh), pb = (uint32_t)((((uint64_t)bhi << 32) | blo) >> psh);
                    const bool ka = widx < cnt_c && (pa & (A - 1u)) == part, kbb = widx + 1u < cnt_c && (pb & (A - 1u)) == part;
                    const unsigned long long ba = __ballot(ka), bb = __ballot(kbb);
                    const uint32_t na = (uint32_t)__popcll(ba), nb = (uint32_t)__popcll(bb);
                    if (nq + na + nb > AP_Q) { if (lane == 0) s_ctl[CTL_FAIL] = 2u; }
                    else {
                        if (ka) q[nq + ap_mbcnt(ba)] = ((unsigned long long)ahi << 32) | alo;
                        if (kbb) q[nq + na + ap_mbcnt(bb)] = ((unsigned long long)bhi << 32) | blo;
                        nq += na + nb;
                    }
                }
                AP_PROF(1);
                while (nq >= 64u) { batch(); AP_PROF(2); if (nsq >= 64u) { slow_batch(); AP_PROF(3); } }
            }
            if (last_chunk) {
                while (nq) { batch(); AP_PROF(2); if (nsq >= 64u) { slow_batch(); AP_PROF(3); } }
                while (nsq) { slow_batch(); AP_PROF(3); }
            }
            AP_PROF(1);
            if (last_chunk) {
                if (!COUNT_ONLY) {
                    uint32_t n = *reinterpret_cast<volatile uint32_t *>(&s_ctl[CTL_NROWS]);
                    n = __builtin_amdgcn_readfirstlane(n);
                    if (n > cap) n = cap;
                    uint8_t *dst = piece0 + (uint64_t)s * (cap / 2);
                    for (uint32_t v = lane; v * 32u < n; v += 64u) {
                        uint4 *src = reinterpret_cast<uint4 *>(rb) + v;
                        const uint4 x = *src;
                        *reinterpret_cast<uint4 *>(dst + v * 16u) = x;
                        *src = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (lane == 0) {
                        a.plen[j * (uint64_t)S + s] = (uint16_t)n;
                        if (firsts) atomicAdd(&a.sample_cells[s], (unsigned long long)firsts);
                    }
                    firsts = 0;
                }
                AP_PROF(4);
                if (!more) break;
                s += AP_WAVES; c = 0; off_c = off_n; cnt_c = cnt_n;
            } else c++;
        }
    }
    AP_PROF_FLUSH();
    __syncthreads();
    const uint32_t nr = s_ctl[CTL_NROWS];                             // ranks handed out (a few may belong to no row: a lost insertion race)
    if (s_ctl[CTL_FAIL]) { if (tid == 0) atomicOr(a.overflow, (int)s_ctl[CTL_FAIL]); return; }
    uint32_t *s_tmp = s_ctl + CTL_TMP;
    if (COUNT_ONLY) {
        uint32_t c = 0;
        for (uint32_t i = tid; i < total_slots; i += AP_THREADS) c += s_tab[i] != 0ull;
        uint32_t tot; (void)ap_block_excl(c, s_tmp, &tot);
        if (tid == 0) { atomicAdd(&a.probe[0], (unsigned long long)tot); atomicMax(&a.probe[1], (unsigned long long)tot); }
        return;
    }
    // statistics by rank: present count, unambiguous count, code set.  Clean rows hold single bases only: every present cell is unambiguous and
    // the code set follows from the union of the bases; dirty rows are counted from the finished cells of all samples.
    uint32_t *s_un = s_rb;                                            // [cap] unambiguous | code set << 16, over the row buffers (done as well)
    __syncthreads();
    for (uint32_t r = tid; r < cap; r += AP_THREADS) {
        const uint32_t p = (s_cnt[r >> 1] >> (16u * (r & 1u))) & 0xFFFFu;
        const uint32_t u = (s_uni[r >> 3] >> ((r & 7u) * 4u)) & 15u;
        const uint32_t m16 = ((u & 1u) << 1) | ((u & 2u) << 1) | ((u & 4u) << 2) | ((u & 8u) << 5);
        s_un[r] = p | (m16 << 16);
    }
    __syncthreads();
    if (s_ctl[CTL_DIRTY]) {
        __threadfence();
        __syncthreads();
        uint32_t seen = 0;
        for (uint32_t i = 0; i < cap / 32; i++) {
            uint32_t bits = s_dirty[i];
            while (bits) {
                const uint32_t r = i * 32u + (uint32_t)__ffs(bits) - 1u; bits &= bits - 1u;
                if ((int)(seen++ & 15u) != wv) continue;              // dirty rows are dealt to the sixteen waves in turn
                uint32_t m = 0, pc = 0, uc = 0;
                for (int s = lane; s < S; s += 64) {
                    const uint64_t pi = j * (uint64_t)S + s;
                    const uint32_t pw2 = __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.plen) + (pi >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t pl = (pw2 >> (16u * (uint32_t)(pi & 1u))) & 0xFFFFu;
                    if (pl <= r) continue;
                    const uint32_t *pw = reinterpret_cast<const uint32_t *>(piece0 + (uint64_t)s * (cap / 2)) + (r >> 3);
                    const uint32_t x = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t nib = (x >> ((r & 7u) * 4u)) & 15u;
                    if (nib) { m |= 1u << nib; pc++; uc += (nib & (nib - 1u)) == 0u; }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { m |= __shfl_xor(m, d, 64); pc += __shfl_xor(pc, d, 64); uc += __shfl_xor(uc, d, 64); }
                if (lane == 0) { s_un[r] = uc | (m << 16); const uint32_t sh16 = 16u * (r & 1u); atomicAnd(&s_cnt[r >> 1], ~(0xFFFFu << sh16)); atomicOr(&s_cnt[r >> 1], pc << sh16); }
            }
        }
        __syncthreads();
    }
    // emit in key order: row keys, rank -> row, statistics
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
    uint16_t *o_p = a.st_present + j * (uint64_t)a.stride, *o_u = a.st_unambig + j * (uint64_t)a.stride, *o_m = a.st_mask + j * (uint64_t)a.stride;
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
            const uint32_t un = s_un[rank];
            o_p[idx] = (uint16_t)((s_cnt[rank >> 1] >> (16u * (rank & 1u))) & 0xFFFFu);
            o_u[idx] = (uint16_t)(un & 0xFFFFu);
            o_m[idx] = (uint16_t)(un >> 16);
        }
    }
    __syncthreads();
    uint16_t *pj = a.perm + j * (uint64_t)cap;
    for (uint32_t i = tid; i < cap; i += AP_THREADS) pj[i] = s_perm[i];
    if (tid == 0) { a.ncnt[j] = total; a.nrank[j] = nr < cap ? nr : cap; if (total > a.stride) atomicOr(a.overflow, 1); }
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
    return logQ >= logB && rem >= 0 && rem <= 50 && cap % 128u == 0 && cap >= 128u && cap <= APPEND_MAX_CAP && nslots >= cap &&
           append_lds_bytes(nslots, cap, false) <= 160u * 1024u - 256u;
}
void launch_append(const AppendArgs &a, uint32_t region_cap, hipStream_t st) { (void)region_cap; launch_append_t<false>(a, 1u << a.logQ, st); }
void launch_append_probe(const AppendArgs &a, uint32_t region_cap, unsigned blocks, hipStream_t st) { (void)region_cap; launch_append_t<true>(a, blocks, st); }

// the statistics of the row blocks (16-bit, one slab per block) as the array holds them: one 32-bit value per row, rows in the order of H
__global__ __launch_bounds__(256) void append_stats_kernel(const uint16_t *sp, const uint16_t *su, const uint16_t *sm, uint32_t stride, const uint32_t *ncnt,
                                                           const uint64_t *roff, uint32_t *present, uint32_t *unambig, uint32_t *mask, uint32_t *vcount)
{
    const uint64_t j = blockIdx.x;
    const uint32_t n = ncnt[j];
    const uint64_t r0 = roff[j], b = j * (uint64_t)stride;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t p = sp[b + i];
        present[r0 + i] = p; unambig[r0 + i] = su[b + i]; mask[r0 + i] = sm[b + i]; vcount[r0 + i] = p;      // (variant_count: merge_ska_array.rs:172)
    }
}
void launch_append_stats(const uint16_t *sp, const uint16_t *su, const uint16_t *sm, uint32_t stride, const uint32_t *ncnt, const uint64_t *roff, int n_blocks,
                         uint32_t *present, uint32_t *unambig, uint32_t *mask, uint32_t *vcount, hipStream_t st)
{
    if (n_blocks > 0) hipLaunchKernelGGL(append_stats_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, sp, su, sm, stride, ncnt, roff, present, unambig, mask, vcount);
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
__device__ static inline uint32_t ap_iupac(uint32_t m4)
{
    const uint64_t lo = 0x485957544D43412Dull;      // "-ACMTWYH"
    const uint64_t hi = 0x4E42444B56535247ull;      // "GRSVKDBN"
    return (uint32_t)(((m4 & 8u) ? hi : lo) >> (8u * (m4 & 7u))) & 0xFFu;
}
constexpr int PR_WAVES = 8;
template <bool KEPT>
__global__ __launch_bounds__(64 * PR_WAVES) void pieces_rows_kernel(PiecesRowsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t cap = a.cap;
    uint16_t *s_src = reinterpret_cast<uint16_t *>(s_raw);                    // [cap + 8] output column -> rank (0xFFFF: none)
    uint32_t *s_piece = reinterpret_cast<uint32_t *>(s_raw + (((size_t)cap + 8) * 2 + 15) / 16 * 16);      // [PR_WAVES][cap / 8]
    const uint64_t j = (uint64_t)blockIdx.x + a.j_base;
    const uint32_t n = a.ncnt[j];
    if (n == 0) return;
    const uint64_t r0 = a.roff[j];
    const uint64_t k0 = KEPT ? a.kpos[r0] : 0;
    const uint32_t nout = KEPT ? (uint32_t)(a.kpos[r0 + n] - k0) : n;
    if (nout == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint32_t i = tid; i < cap + 8; i += blockDim.x) s_src[i] = 0xFFFFu;
    __syncthreads();
    const uint16_t *pj = a.perm + j * (uint64_t)cap;
    const uint32_t nr = a.nrank[j];
    for (uint32_t r = tid; r < nr; r += blockDim.x) {
        const uint32_t p = pj[r];
        if (p == 0xFFFFu || p >= n) continue;
        if (!KEPT) s_src[p] = (uint16_t)r;
        else if (a.keep[r0 + p] == 1) s_src[(uint32_t)(a.kpos[r0 + p] - k0)] = (uint16_t)r;
    }
    __syncthreads();
    const uint64_t ocol = KEPT ? k0 : r0 - a.col_base;                        // first output column of the block
    const uint32_t shift = (uint32_t)(ocol & 3u);
    const uint32_t ndw = (nout + shift + 3u) / 4u;                            // aligned dwords that hold the block's cells
    const int S = a.n_samples;
    const int s_lo = blockIdx.y * a.samples_per_wg, s_hi = s_lo + a.samples_per_wg < S ? s_lo + a.samples_per_wg : S;
    uint32_t *pc = s_piece + (size_t)wv * (cap / 8);
    const unsigned char *pcb = reinterpret_cast<const unsigned char *>(pc);
    for (int s = s_lo + wv; s < s_hi; s += PR_WAVES) {
        const uint32_t pl = a.plen[j * (uint64_t)S + s];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.pieces + (j * (uint64_t)S + s) * (cap / 2));
        for (uint32_t i = lane; i < (pl + 7u) / 8u; i += 64) pc[i] = src[i];
        __builtin_amdgcn_wave_barrier();
        unsigned char *dst = a.out + (uint64_t)s * a.pitch + (ocol - shift);
        for (uint32_t v = lane; v < ndw; v += 64) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = (int)(4u * v + b) - (int)shift;                  // column of the block
                uint32_t ch = 0;
                if (c >= 0 && (uint32_t)c < nout) {
                    const uint32_t r = s_src[c];
                    uint32_t m4 = 0;
                    if (r < pl) m4 = ((uint32_t)pcb[r >> 1] >> ((r & 1u) * 4u)) & 15u;
                    ch = (a.mask_ambig && (m4 & (m4 - 1u))) ? (uint32_t)'N' : ap_iupac(m4);
                }
                word |= ch << (8 * b);
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
    const size_t lds = (((size_t)a.cap + 8) * 2 + 15) / 16 * 16 + (size_t)PR_WAVES * (a.cap / 8) * 4;
    if (a.keep) {
        (void)hipFuncSetAttribute((const void *)pieces_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pieces_rows_kernel<true>, dim3(n_blocks, gy), dim3(64 * PR_WAVES), lds, st, b);
    } else {
        (void)hipFuncSetAttribute((const void *)pieces_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pieces_rows_kernel<false>, dim3(n_blocks, gy), dim3(64 * PR_WAVES), lds, st, b);
    }
}

}  // namespace skx
