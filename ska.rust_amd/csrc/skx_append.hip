// skx_append.hip -- MergeSkaDict::append on the device, straight from the extraction kernel's regions (gfx950, wave64).
//
// The reference appends a sample to the merged dictionary in ONE pass over its unsorted hash map (merge_ska_dict.rs:77-109: look the
// split k-mer up, insert a row of '-' if it is new, write the sample's column), and a sample's own dictionary folds repeated split
// k-mers by OR-ing their middle bases into an IUPAC code (ska_dict.rs:76-113).  Rounds 1-3 sorted and folded every sample's regions
// first (dedupe_mb_kernel: 80 GB of traffic, a third of the step) so that the union could read them in order.  This file does what the
// reference does: one workgroup owns a *row block* -- the split k-mers whose top logQ hash bits are j -- and appends the samples one after
// the other, in sample order, reading each sample's raw region (words as extract_kernel scattered them: unsorted, with repeats):
//
//   * the block's rows live in an order-preserving LDS hash table (slot = monotone function of H, linear probing without wrap); an entry is
//     (low hash bits << 14) | (first-seen rank + 1): the rank a row got when it was first inserted is its column in everything the pass writes
//     while the final row order (the order of H) is not known yet;
//   * a region holds the words of A = 2^(logQ - logB) row blocks.  The A workgroups of a region run on the same XCD at the same time
//     (blockIdx -> XCD is b % 8), each reads the whole region and keeps its share: the first reader takes a line from HBM, the others find it in
//     that XCD's L2.  Kept words are compacted through an LDS queue so that the look-ups run with full waves;
//   * a sample's cells of the block are OR-ed into a row buffer of 4-bit base sets indexed by rank (LDS atomics; the returned old value tells a first
//     sighting -- counted in the row's present / unambiguous statistics and the sample's k-mer count -- from a repeat, which is where ska_dict.rs folds);
//   * the row buffer leaves as the sample's *piece* of the block: plen ranks, two per byte, at a fixed place (pieces[(j * S + s) * cap / 2]).
//     Ranks beyond plen were first seen by later samples: the cell is '-'.  The rows x samples matrix in the order of H is produced from
//     the pieces by pieces_rows_kernel (all rows, a window of row blocks, or only the rows a filter keeps) -- 1 byte per cell there, 4 bits
//     and no cell for unseen rows here;
//   * when the last sample is in, the table is emitted in key order: the block's row keys, perm (rank -> row of the block) and the row statistics.
//
// Three activities overlap in one pass of the sample loop, each on its own buffers, with ONE barrier per sample: filter sample s into queue
// s & 1, look up / insert the queue of sample s - 1 into row buffer (s - 1) % 3, write out the piece of sample s - 3.
#include "skx_device.h"
#include <cstdlib>

namespace skx {

typedef const uint64_t __attribute__((address_space(1))) *gq_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *g4_t;
typedef const uint32_t __attribute__((address_space(1))) *g1_t;

// Measurement only (-DSKX_AP_PROF): cycles of every wave's lane 0 between the phases of append_kernel's sample loop, summed over waves, workgroups and
// launches: 0 waiting for the sample's words, 1 filter into the queue, 2 look-ups / inserts, 3 piece written, 4 at the barrier, 5 the tail (emit)
#ifdef SKX_AP_PROF
__device__ unsigned long long g_ap_prof[16];
#define AP_PROF_START unsigned long long tprof = __builtin_readcyclecounter()
#define AP_PROF(i) do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_ap_prof[(i)], t_ - tprof); tprof = t_; } } while (0)
#define AP_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
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
#define AP_WAIT() do { } while (0)
#endif

constexpr int AP_THREADS = 1024;
constexpr uint32_t AP_PAD = 128;            // slots behind the table's last home slot (probing does not wrap)
constexpr uint32_t AP_QCAP = 2048;          // kept words of one sample and block the queue holds
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

// top 32 of the word's rem local hash bits (word bits [4, rem + 4)), left-aligned
template <bool HI>
__device__ static inline uint32_t ap_l32(uint32_t lo, uint32_t hi, int rem)
{
    if (HI) return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(rem - 28));      // 32 <= rem
    return rem == 0 ? 0u : (uint32_t)(((lo >> 4) | (hi << 28)) << (32 - rem));   // rem < 32: the bits above fall off the left end
}

// ctl words: 0..3 queue fill, 4..6 rank snapshots, 7 ranks handed out, 8 failure, 9..12 first sightings per sample, 13..29 scan scratch, 30 a row is dirty
constexpr int CTL_QN = 0, CTL_SNAP = 4, CTL_NROWS = 7, CTL_FAIL = 8, CTL_CELLS = 9, CTL_TMP = 13, CTL_DIRTY = 30, CTL_WORDS = 32;

static inline size_t append_lds_bytes(uint32_t nslots, uint32_t cap, bool count_only)
{
    size_t b = (size_t)(nslots + AP_PAD) * 8 + (size_t)AP_QCAP * 2 * 8 + CTL_WORDS * 4;
    if (!count_only) b += (size_t)cap * 4 + (size_t)cap * 2 + (size_t)(cap / 8) * 4 * 3 + (size_t)(cap / 32) * 4 + 128 * 4;
    return b;
}

template <int ROUNDS, bool COUNT_ONLY, bool HI>
__global__ __launch_bounds__(AP_THREADS) void append_kernel(AppendArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t total_slots = a.nslots + AP_PAD;
    const uint32_t cap = a.cap, rbw = cap / 8;                       // ranks per block; dwords of one row buffer
    unsigned long long *s_tab = reinterpret_cast<unsigned long long *>(s_raw);
    unsigned long long *s_q = s_tab + total_slots;                   // [2][AP_QCAP]
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_q + 2 * AP_QCAP);
    uint32_t *s_cnt = s_ctl + CTL_WORDS;                             // [cap] present | unambiguous << 16
    uint32_t *s_msk = s_cnt + cap;                                   // [cap / 2] 16-bit code sets, two ranks per word
    uint32_t *s_rb = s_msk + cap / 2;                                // [3][rbw] 4-bit base sets by rank
    uint32_t *s_dirty = s_rb + 3 * rbw;                              // [cap / 32] ranks whose code set must be taken from the finished cells
    uint32_t *s_park = s_dirty + cap / 32;                           // [128] piece length | first sightings << 16 of the last samples
    const int tid = threadIdx.x, lane = tid & 63;
    const int S = a.n_samples;
    const int shA = a.logQ - a.logB;
    const uint32_t A = 1u << shA;
    uint32_t region, part;
    if (a.logB >= 3) { const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; region = ((slot >> shA) << 3) + xcd; part = slot & (A - 1u); }
    else { region = blockIdx.x >> shA; part = blockIdx.x & (A - 1u); }
    const uint64_t j = ((uint64_t)region << shA) + part;             // the block's place in the order of H
    const int rem = a.bits - a.logQ;                                 // hash bits below the block bits
    const int psh = rem + 4;                                         // the part bits of a word start here (>= 32 when HI)

    for (uint32_t i = tid; i < total_slots; i += AP_THREADS) s_tab[i] = 0ull;
    if (!COUNT_ONLY) {
        for (uint32_t i = tid; i < cap; i += AP_THREADS) s_cnt[i] = 0u;
        for (uint32_t i = tid; i < cap / 2; i += AP_THREADS) s_msk[i] = 0u;
        for (uint32_t i = tid; i < 3 * rbw; i += AP_THREADS) s_rb[i] = 0u;
        for (uint32_t i = tid; i < cap / 32; i += AP_THREADS) s_dirty[i] = 0u;
    }
    if (tid < CTL_WORDS) s_ctl[tid] = 0u;
    __syncthreads();

    // region offsets and fills are read through the constant address space: scalar loads, no vector-memory counter involved
    typedef const uint64_t __attribute__((address_space(4))) *cq_t;
    typedef const uint32_t __attribute__((address_space(4))) *c1_t;
    cq_t c_off = (cq_t)(uintptr_t)a.off;
    c1_t c_raw = (c1_t)(uintptr_t)a.raw;
    const uint64_t rstride = 1ull << a.logB;                         // regions per sample
    // Three samples' words are on their way at any time, each set in registers of its own (the loop below is unrolled three times so that no
    // set is ever copied: a copy would wait for the loads).  A load is issued by every lane, whatever the region's fill -- lanes past the
    // fill re-read its last pair -- so that the compiler can count the loads behind the one it waits for (a conditional load makes that
    // wait a wait for everything, the stores of the pieces included).
    u32x4 bufA[ROUNDS], bufB[ROUNDS], bufC[ROUNDS];
    uint32_t cntA = 0, cntB = 0, cntC = 0;
    // The loads are written as inline assembly and waited for by hand (ap_wait below): the compiler's own count at the head of the unrolled
    // loop comes out as "everything" (vmcnt(0): the loads just issued and the piece stores included), which would put the memory latency
    // back into every step.  Nothing else in the loop is a vector-memory load, so the hand count is exact up to the stores in between.
    auto issue = [&](u32x4 (&buf)[ROUNDS], uint64_t off, uint32_t cnt) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.words + off);
        const uint32_t last = cnt ? (cnt - 1u) >> 1 : 0u;
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const uint32_t w = (uint32_t)tid + (uint32_t)AP_THREADS * r;      // pair index
            const uint32_t vo = (w < last ? w : last) * 16u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[r]) : "v"(vo), "s"(base) : "memory");
        }
    };
    // the words of the set whose loads are followed by those of the two other sets: 2 * ROUNDS younger loads may stay in flight
    auto ap_wait = [&](u32x4 (&buf)[ROUNDS]) {
        if (ROUNDS == 1) asm volatile("s_waitcnt vmcnt(2)" : "+v"(buf[0]));
        if (ROUNDS == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(buf[0]), "+v"(buf[ROUNDS > 1 ? 1 : 0]));
        if (ROUNDS == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(buf[0]), "+v"(buf[ROUNDS > 1 ? 1 : 0]), "+v"(buf[ROUNDS > 2 ? 2 : 0]));
        if (ROUNDS == 4) asm volatile("s_waitcnt vmcnt(8)" : "+v"(buf[0]), "+v"(buf[ROUNDS > 1 ? 1 : 0]), "+v"(buf[ROUNDS > 2 ? 2 : 0]), "+v"(buf[ROUNDS > 3 ? 3 : 0]));
    };
    auto region_of = [&](int smp, uint64_t &off, uint32_t &cnt) {
        const uint64_t rr = (uint64_t)smp * rstride + region;
        off = c_off[rr]; cnt = c_raw[rr];
    };
    uint64_t off_n = 0; uint32_t cnt_n = 0;                           // of the sample whose words are requested next
    region_of(0, off_n, cnt_n); issue(bufA, off_n, cnt_n); cntA = cnt_n;
    region_of(S > 1 ? 1 : S - 1, off_n, cnt_n); issue(bufB, off_n, cnt_n); cntB = cnt_n;
    region_of(S > 2 ? 2 : S - 1, off_n, cnt_n); issue(bufC, off_n, cnt_n); cntC = cnt_n;
    region_of(S > 3 ? 3 : S - 1, off_n, cnt_n);
    uint8_t *piece0 = a.pieces + j * (uint64_t)S * (cap / 2);
    AP_PROF_START;
    // one sample step; M3 = s % 3 (static: the step is instantiated three times)
    auto step = [&](const int s, u32x4 (&buf)[ROUNDS], uint32_t &cnt_c, const int M3) {
        ap_wait(buf);
        AP_PROF(0);
        if (tid == 0) {
            s_ctl[CTL_QN + ((s + 1) & 3)] = 0u;
            if (s >= 2) s_ctl[CTL_SNAP + (M3 == 2 ? 0 : M3 + 1)] = s_ctl[CTL_NROWS];      // (s - 2) % 3 == (s + 1) % 3: the ranks handed out once sample s - 2 is in
        }
        // ---- F(s): this block's share of sample s goes into queue s & 1
        if (s < S) {
            unsigned long long *q = s_q + (size_t)(s & 1) * AP_QCAP;
            uint32_t kf = 0, before[2 * ROUNDS], tot = 0;
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
                const uint32_t w = 2u * ((uint32_t)tid + (uint32_t)AP_THREADS * r);
                const uint32_t h0 = HI ? buf[r].y >> (psh - 32) : (uint32_t)((((uint64_t)buf[r].y << 32) | buf[r].x) >> psh);
                const uint32_t h1 = HI ? buf[r].w >> (psh - 32) : (uint32_t)((((uint64_t)buf[r].w << 32) | buf[r].z) >> psh);
                const bool k0 = w < cnt_c && (h0 & (A - 1u)) == part, k1 = w + 1u < cnt_c && (h1 & (A - 1u)) == part;
                const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
                before[2 * r] = tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
                tot += (uint32_t)__popcll(b0);
                before[2 * r + 1] = tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                tot += (uint32_t)__popcll(b1);
                kf |= (uint32_t)k0 << (2 * r) | (uint32_t)k1 << (2 * r + 1);
            }
            uint32_t base = 0;
            if (lane == 0 && tot) base = atomicAdd(&s_ctl[CTL_QN + (s & 3)], tot);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + tot > AP_QCAP) { if (lane == 0) s_ctl[CTL_FAIL] = 2u; }
            else {
#pragma unroll
                for (int r = 0; r < ROUNDS; r++) {
                    if ((kf >> (2 * r)) & 1u) q[base + before[2 * r]] = ((unsigned long long)buf[r].y << 32) | buf[r].x;
                    if ((kf >> (2 * r + 1)) & 1u) q[base + before[2 * r + 1]] = ((unsigned long long)buf[r].w << 32) | buf[r].z;
                }
            }
        }
        AP_PROF(1);
        // ---- W(s - 3): the piece of sample s - 3 leaves (its rank count was snapshot a step ago); stores before the loads below, which are the
        // ones the next steps wait for
        if (!COUNT_ONLY && s >= 3 && s - 3 < S) {
            const int sw = s - 3;
            uint32_t n = s_ctl[CTL_SNAP + M3];                       // (s - 3) % 3 == s % 3
            if (n > cap) n = cap;
            uint32_t *rb = s_rb + (size_t)M3 * rbw;
            if ((uint32_t)tid * 32u < n) {
                uint4 *src = reinterpret_cast<uint4 *>(rb) + tid;
                const uint4 v = *src;
                *reinterpret_cast<uint4 *>(piece0 + (uint64_t)sw * (cap / 2) + (uint32_t)tid * 16u) = v;
                *src = make_uint4(0u, 0u, 0u, 0u);
            }
            // the piece's length and the sample's first sightings are parked and leave 64 samples at a time (wave 3: it stores no piece)
            if (tid == 0) {
                s_park[sw & 127] = n | (s_ctl[CTL_CELLS + (sw & 3)] << 16);
                s_ctl[CTL_CELLS + (sw & 3)] = 0u;
            }
        }
        if (!COUNT_ONLY && s >= 4 && (tid >> 6) == 3 && (((s - 4) & 63) == 63 || s - 4 == S - 1)) {      // samples up to s - 4 are parked (the step before this one)
            const int last = s - 4, first = last & ~63, sm = first + lane;
            if (sm <= last) {
                const uint32_t v = s_park[(first & 127) + lane];
                a.plen[j * (uint64_t)S + sm] = (uint16_t)(v & 0xFFFFu);
                if (v >> 16) atomicAdd(&a.sample_cells[sm], (unsigned long long)(v >> 16));
            }
        }
        AP_PROF(3);
        // the words of sample s + 3 are requested into the registers F(s) has just read
        // (always, past the last sample too -- it is read again: a conditional load would not be counted, see above)
        issue(buf, off_n, cnt_n); cnt_c = cnt_n;
        region_of(s + 4 < S ? s + 4 : S - 1, off_n, cnt_n);
        // ---- I(s - 1): the queue of sample s - 1 against the table; its cells into row buffer (s - 1) % 3
        if (s >= 1 && s - 1 < S) {
            const unsigned long long *q = s_q + (size_t)((s - 1) & 1) * AP_QCAP;
            uint32_t qn = s_ctl[CTL_QN + ((s - 1) & 3)];
            if (qn > AP_QCAP) qn = AP_QCAP;
            uint32_t *rb = s_rb + (size_t)(M3 == 0 ? 2 : M3 - 1) * rbw;
            uint32_t firsts = 0;
            for (uint32_t i = tid; i < qn; i += AP_THREADS) {
                const unsigned long long w = q[i];
                const uint32_t hs = __umulhi(ap_l32<HI>((uint32_t)w, (uint32_t)(w >> 32), rem), a.nslots);
                const unsigned long long e0 = s_tab[hs], e1 = s_tab[hs + 1];
                const unsigned long long keyE = ((w << (60 - rem)) >> (50 - rem)) & ~(unsigned long long)AP_RANK_MASK;      // (low hash bits) << 14
                uint32_t rank1 = 0;
                if ((e0 ^ keyE) <= AP_RANK_MASK) rank1 = (uint32_t)e0 & AP_RANK_MASK;
                else if (e0 != 0ull && (e1 ^ keyE) <= AP_RANK_MASK) rank1 = (uint32_t)e1 & AP_RANK_MASK;
                if (rank1 == 0) {                                      // first sighting, or a key displaced further: the insert loop
                    uint32_t mine = 0;
                    for (uint32_t t = hs; t < total_slots; t++) {
                        unsigned long long e = s_tab[t];
                        if (e == 0ull) {
                            if (!mine) { mine = atomicAdd(&s_ctl[CTL_NROWS], 1u) + 1u; if (mine > (COUNT_ONLY ? AP_RANK_MASK : cap)) { s_ctl[CTL_FAIL] = 1u; break; } }
                            e = atomicCAS(&s_tab[t], 0ull, keyE | mine);
                            if (e == 0ull) { rank1 = mine; break; }
                        }
                        if ((e ^ keyE) <= AP_RANK_MASK && (e & AP_RANK_MASK)) { rank1 = (uint32_t)e & AP_RANK_MASK; break; }
                    }
                    if (rank1 == 0) { s_ctl[CTL_FAIL] = 1u; continue; }
                }
                if (COUNT_ONLY) continue;
                const uint32_t rank = rank1 - 1u, m4 = (uint32_t)w & 15u, sh = (rank & 7u) * 4u;
                const uint32_t old = atomicOr(&rb[rank >> 3], m4 << sh);
                const uint32_t on = (old >> sh) & 15u, nn = on | m4;
                if (on == 0u) {
                    const uint32_t single = (m4 & (m4 - 1u)) == 0u;
                    atomicAdd(&s_cnt[rank], 1u | (single << 16));
                    atomicOr(&s_msk[rank >> 1], (1u << m4) << (16u * (rank & 1u)));
                    firsts++;
                } else if (nn != on) {                                 // the sample has this split k-mer again with another middle base (ska_dict.rs:92-101)
                    if ((on & (on - 1u)) == 0u) atomicSub(&s_cnt[rank], 1u << 16);
                    atomicOr(&s_dirty[rank >> 5], 1u << (rank & 31u)); s_ctl[CTL_DIRTY] = 1u;
                }
            }
            if (!COUNT_ONLY) {
                const unsigned long long anyf = __ballot(firsts != 0u);
                if (anyf) {
                    uint32_t wt; (void)ap_wave_excl(firsts, &wt);
                    if (lane == 0) atomicAdd(&s_ctl[CTL_CELLS + ((s - 1) & 3)], wt);
                }
            }
        }
        AP_PROF(2);
        __syncthreads();
        AP_PROF(4);
    };
    for (int s = 0; s < S + 4; s += 3) {
        step(s, bufA, cntA, 0);
        step(s + 1, bufB, cntB, 1);
        step(s + 2, bufC, cntC, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bufA[0]), "+v"(bufB[0]), "+v"(bufC[0]));      // (the loads past the last sample: nothing may still be on its way into these registers)
    if (ROUNDS > 1) asm volatile("" : "+v"(bufA[ROUNDS > 1 ? 1 : 0]), "+v"(bufB[ROUNDS > 1 ? 1 : 0]), "+v"(bufC[ROUNDS > 1 ? 1 : 0]));
    if (ROUNDS > 2) asm volatile("" : "+v"(bufA[ROUNDS > 2 ? 2 : 0]), "+v"(bufB[ROUNDS > 2 ? 2 : 0]), "+v"(bufC[ROUNDS > 2 ? 2 : 0]));
    if (ROUNDS > 3) asm volatile("" : "+v"(bufA[ROUNDS > 3 ? 3 : 0]), "+v"(bufB[ROUNDS > 3 ? 3 : 0]), "+v"(bufC[ROUNDS > 3 ? 3 : 0]));
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
    // code sets of the rows where a sample folded two middle bases: from the finished cells (the intermediate codes never were a cell)
    if (s_ctl[CTL_DIRTY]) {
        __threadfence();
        __syncthreads();
        const int wv = tid >> 6;
        uint32_t seen = 0;
        for (uint32_t i = 0; i < cap / 32; i++) {
            uint32_t bits = s_dirty[i];
            while (bits) {
                const uint32_t r = i * 32u + (uint32_t)__ffs(bits) - 1u; bits &= bits - 1u;
                if ((int)(seen++ & 15u) != wv) continue;              // dirty rows are dealt to the sixteen waves in turn
                uint32_t m = 0;
                for (int s = lane; s < S; s += 64) {
                    const uint64_t pi = j * (uint64_t)S + s;
                    const uint32_t pw2 = __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.plen) + (pi >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t pl = (pw2 >> (16u * (uint32_t)(pi & 1u))) & 0xFFFFu;
                    if (pl <= r) continue;
                    const uint32_t *pw = reinterpret_cast<const uint32_t *>(piece0 + (uint64_t)s * (cap / 2)) + (r >> 3);
                    const uint32_t x = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t nib = (x >> ((r & 7u) * 4u)) & 15u;
                    if (nib) m |= 1u << nib;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
                if (lane == 0) {                                       // (the word's other half is rank r ^ 1's, possibly another wave's: atomics)
                    const uint32_t shm = 16u * (r & 1u);
                    atomicAnd(&s_msk[r >> 1], ~(0xFFFFu << shm));
                    atomicOr(&s_msk[r >> 1], m << shm);
                }
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
            slab[idx] = ((((uint64_t)j << rem) | (uint64_t)(e >> AP_RANK_BITS)) << 4) | 1ull;
            s_perm[rank] = (uint16_t)idx;
            const uint32_t cw = s_cnt[rank];
            o_p[idx] = (uint16_t)(cw & 0xFFFFu); o_u[idx] = (uint16_t)(cw >> 16);
            o_m[idx] = (uint16_t)((s_msk[rank >> 1] >> (16u * (rank & 1u))) & 0xFFFFu);
        }
    }
    __syncthreads();
    uint16_t *pj = a.perm + j * (uint64_t)cap;
    for (uint32_t i = tid; i < cap; i += AP_THREADS) pj[i] = s_perm[i];
    if (tid == 0) { a.ncnt[j] = total; a.nrank[j] = nr < cap ? nr : cap; if (total > a.stride) atomicOr(a.overflow, 1); }
    AP_PROF(5);
}

template <int ROUNDS, bool COUNT_ONLY>
static void launch_append_t(const AppendArgs &a, unsigned blocks, hipStream_t st)
{
    const size_t lds = append_lds_bytes(a.nslots, a.cap, COUNT_ONLY);
    const bool hi = a.bits - a.logQ >= 32;
    if (hi) {
        (void)hipFuncSetAttribute((const void *)append_kernel<ROUNDS, COUNT_ONLY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((append_kernel<ROUNDS, COUNT_ONLY, true>), dim3(blocks), dim3(AP_THREADS), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)append_kernel<ROUNDS, COUNT_ONLY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((append_kernel<ROUNDS, COUNT_ONLY, false>), dim3(blocks), dim3(AP_THREADS), lds, st, a);
    }
}
template <bool COUNT_ONLY>
static void launch_append_r(const AppendArgs &a, uint32_t region_cap, unsigned blocks, hipStream_t st)
{
    const uint32_t rounds = (region_cap + 2 * AP_THREADS - 1) / (2 * AP_THREADS);
    if (rounds <= 1) launch_append_t<1, COUNT_ONLY>(a, blocks, st);
    else if (rounds == 2) launch_append_t<2, COUNT_ONLY>(a, blocks, st);
    else if (rounds == 3) launch_append_t<3, COUNT_ONLY>(a, blocks, st);
    else launch_append_t<4, COUNT_ONLY>(a, blocks, st);
}
// what the pass takes: hash bits below the block bits that leave room for a rank in a table entry, regions of at most four load rounds,
// a table and row buffers that fit the LDS
bool append_ok(int bits, int logB, int logQ, uint32_t region_cap, uint32_t nslots, uint32_t cap)
{
    const int rem = bits - logQ;
    return logQ >= logB && rem >= 0 && rem <= 50 && region_cap <= 8u * AP_THREADS && cap % 32u == 0 && cap >= 32u && cap <= APPEND_MAX_CAP &&
           nslots >= cap && append_lds_bytes(nslots, cap, false) <= 160u * 1024u;
}
void launch_append(const AppendArgs &a, uint32_t region_cap, hipStream_t st) { launch_append_r<false>(a, region_cap, 1u << a.logQ, st); }
void launch_append_probe(const AppendArgs &a, uint32_t region_cap, unsigned blocks, hipStream_t st) { launch_append_r<true>(a, region_cap, blocks, st); }

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
