// skx_device.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the split-k-mer engine.
//
// Everything here is HBM-bound integer/byte work (no MFMA): coalesced 16-B loads of the record
// stream, LDS-staged 2-bit windows, LDS atomics for bucket ranks, order-preserving LDS hash tables
// for dedupe / set union, popcount bit planes for the pairwise distance.
//
// Reference semantics implemented (bacpop/ska.rust v0.5.2, paths relative to its src/):
//   ska_dict/bit_encoding.rs:30-54   encode_base / valid_base / rc_base
//   ska_dict/split_kmer.rs:78-217    build + roll_fwd window enumeration, incl. the `idx + k >= len` end rule
//   ska_dict/split_kmer.rs:281-295   canonical (min of fwd / rc) split k-mer, :144-146 self_palindrome
//   ska_dict.rs:76-113               per-sample value = IUPAC code of the union of observed middle bases
//   merge_ska_dict.rs:77-151         samples -> columns, 0/'-' where absent
//   merge_ska_array.rs:139-186,289-402,416-438,587-632   counts, filter, distance
#include "skx_device.h"
#include <cstdlib>

namespace skx {

// ------------------------------------------------------------------------------------------------
// hashing: a bijection on `bits`-bit integers so that buckets (top bits) are uniform whatever the
// genome's composition; "engine order" of keys is the order of H(key).
// ------------------------------------------------------------------------------------------------
HashParams make_hash_params(int k)
{
    HashParams p;
    p.bits = 2 * (k - 1);
    p.hb = k - 1;
    p.hmask = p.hb >= 32 ? ~0u : ((1u << p.hb) - 1);
    p.c[0] = 0x9E3779B1u; p.c[1] = 0x85EBCA6Bu; p.c[2] = 0xC2B2AE35u; p.c[3] = 0x27D4EB2Fu;
    return p;
}

// reverse complement of an arm of n (<= 16) 2-bit symbols (cf. bit_encoding.rs:182-195)
__device__ static inline uint32_t revcomp_arm(uint32_t x, int n)
{
    x = __builtin_bitreverse32(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x ^= 0xAAAAAAAAu;
    return n ? x >> (32 - 2 * n) : 0;
}

// 16 ASCII bytes -> 2-bit codes (first base most significant), bad-base mask, newline mask (bit j = byte j)
__device__ static inline void pack16(const uint32_t w[4], uint32_t &code, uint32_t &bad, uint32_t &nl)
{
    code = 0; bad = 0; nl = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = w[i];
        uint32_t c = (x >> 1) & 0x03030303u;                       // encode_base per byte
        code = (code << 8) | ((c * 0x40100401u) >> 24);
        uint32_t t = (x & 0x0F0F0F0Fu) ^ 0x0E0E0E0Eu;              // low nibble == 14  <=>  !valid_base
        uint32_t nzt = (t + 0x7F7F7F7Fu) & 0x80808080u;
        uint32_t u = x ^ 0x0A0A0A0Au;                              // '\n' record terminator
        uint32_t znl = ~(((u & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | u | 0x7F7F7F7Fu);
        uint32_t fb = (((~nzt) & 0x80808080u) | znl) >> 7;
        uint32_t fn = znl >> 7;
        bad |= (((fb * 0x01020408u) >> 24) & 0xFu) << (4 * i);
        nl |= (((fn * 0x01020408u) >> 24) & 0xFu) << (4 * i);
    }
}
// the same without the newline mask (extract_kernel finds the few positions where it matters from the text itself)
__device__ static inline void pack16_nonl(const uint32_t w[4], uint32_t &code, uint32_t &bad)
{
    code = 0; bad = 0;
    uint32_t c0f = 0x0F0F0F0Fu, c7f = 0x7F7F7F7Fu;
    asm volatile("" : "+v"(c0f), "+v"(c7f));                      // built here, for this call (as plain constants they are hoisted out of the tile loop into registers that stay live through it)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = w[i];
        uint32_t c = (x >> 1) & 0x03030303u;
        code = (code << 8) | ((c * 0x40100401u) >> 24);
        uint32_t t = (x & c0f) ^ 0x0E0E0E0Eu;                      // low nibble == 14  <=>  !valid_base ('\n' = 0x0A is valid_base by this test ...)
        uint32_t nzt = (t + c7f) & 0x80808080u;
        uint32_t u = x ^ 0x0A0A0A0Au;                              // ... so the record terminator is tested for on its own
        uint32_t znl = ~(((u & c7f) + c7f) | u | c7f);
        uint32_t fb = (((~nzt) & 0x80808080u) | znl) >> 7;
        bad |= (((fb * 0x01020408u) >> 24) & 0xFu) << (4 * i);
    }
}
__device__ static inline uint32_t qualbad16(const uint32_t w[4], int min_qual)
{
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t q = (w[i] >> (8 * b)) & 0xFF;
            m |= (uint32_t)(((q - 33u) & 0xFFu) <= (uint32_t)min_qual) << (4 * i + b);   // !((q-33) > min_qual), u8 wrap
        }
    return m;
}

// ------------------------------------------------------------------------------------------------
// K1/K2: split k-mer extraction (+ bucket histogram | bucket scatter).  One workgroup = one tile of
// TILE window-end positions of one sample; 8 consecutive samples run concurrently, one per XCD
// (block b -> XCD b%8), so a sample's bucket cursors and partially written lines stay in one L2.
// ------------------------------------------------------------------------------------------------
// forward declaration (defined with the block-wide helpers below)
__device__ static inline uint32_t block_excl_scan(uint32_t v, uint32_t *s_tmp, uint32_t *total);
__device__ static inline uint32_t block_excl_scan_row(uint32_t v, uint32_t *s_tmp, uint32_t *total, int tid);

// LDS carve (dynamic, 16-B aligned): [B+4] hist -> local starts (+ a dummy counter for invalid windows) | [B] chunk
// bases | scan scratch | codes+masks, later aliased by the staging buffer.  16 window-end positions per thread.  The tile's
// words are staged in bucket order (TILE words, or half of them at a time: SPLIT) so that the copy-out writes every
// (tile, bucket) chunk with adjacent lanes (few, wide L2 write requests).  HI: the bucket bits of a packed word lie in its
// upper half.  SPLIT: the staging buffer holds half a tile at a time (the lower half of the bucket space, then the upper
// half), so a 12 288-position tile needs 57 KB of LDS instead of 105 KB and two 768-thread workgroups share a CU (85-VGPR
// cap), with the same (tile, bucket) chunks in the word buffer as single-pass staging would write.
//
// What bounds it (round 3): not its arithmetic.  A build with k as a compile-time constant issues 12 % fewer
// VALU instructions (SQ_INSTS_VALU 1.24e9 -> 1.09e9 per 200 genomes) and takes the same time; cycle stamps between the barriers
// (-DSKX_EXT_PROF, tools/kbench.py prints them) show a workgroup spending 24 % of its 37 800 cycles before its FIRST barrier --
// every thread issues one 16-byte load and waits for it behind the chip's store traffic -- against 33 % in the window loop.
// Walking several tiles per workgroup with the next tile's text requested early does not hide that wait: loads, atomics and
// stores share one counter (vmcnt) on gfx9, so the wait for the text is a wait for the previous tile's stores, which take
// longer than a tile to be acknowledged (16.0-16.8 ms against 13.2; profiles/r03s_ab_tpw.log).
// Measurement only (-DSKX_PHASE_PROF=1: extract_kernel, =2: dedupe_mb_kernel): cycles of wave 0 between a kernel's barriers, summed over
// the workgroups; tools/kbench.py prints the shares.  This is what showed where these kernels wait.
#ifdef SKX_PHASE_PROF
__device__ unsigned long long g_phase_prof[1024 * 16];
#define PHASE_PROF_(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_prof[(blockIdx.x & 1023) * 16 + (i)], t_ - tprof); tprof = t_; } } while (0)
#define PHASE_PROF_START unsigned long long tprof = __builtin_readcyclecounter()
extern "C" void skx_debug_phase_prof(unsigned long long *out, int reset)
{
    static unsigned long long h[1024 * 16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_prof), sizeof(h));
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int j = 0; j < 1024; j++) out[i] += h[j * 16 + i]; }
    if (reset) { for (auto &x : h) x = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_prof), h, sizeof(h)); }
}
#else
#define PHASE_PROF_(i) do { } while (0)
#define PHASE_PROF_START do { } while (0)
#endif
#if defined(SKX_PHASE_PROF) && SKX_PHASE_PROF == 1
#define EXT_PROF(i) do { if (SCATTER) PHASE_PROF_(i); } while (0)
#else
#define EXT_PROF(i) do { } while (0)
#endif
#if defined(SKX_PHASE_PROF) && SKX_PHASE_PROF == 4      // union_kernel: per wave (lane 0), 0 slice look-up, 1 waiting for a slice's words, 2 looking them up, 3 emit
#define UN_PROF(i) do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_prof[(blockIdx.x & 1023) * 16 + (i)], t_ - tprof); tprof = t_; } } while (0)
#define UN_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define UN_PROF(i) do { } while (0)
#define UN_WAIT() do { } while (0)
#endif
#if defined(SKX_PHASE_PROF) && SKX_PHASE_PROF == 2
#define DD_PROF(i) PHASE_PROF_(i)
#else
#define DD_PROF(i) do { } while (0)
#endif
template <bool SCATTER, int TILE, bool HI, int PPT, bool SPLIT, int RMAX>
__global__ __launch_bounds__(TILE / PPT, SPLIT ? 6 : 1) void extract_kernel(ExtractArgs a)
{
    constexpr int NT = TILE / PPT;
    constexpr int NCH = PPT / 16;                                      // 16-base chunks per thread
    constexpr int NCHUNK = TILE / 16 + 5;                              // 4 halo chunks before, 1 after
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int B = 1 << a.logB;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_raw);            // [B] + dummy at [B]
    uint32_t *s_base = s_hist + B + 4;
    uint32_t *s_tmp = s_base + B;                                      // [20] block-scan scratch
    unsigned char *s_rest = reinterpret_cast<unsigned char *>(s_tmp + 20);
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_rest);
    uint16_t *s_bad = reinterpret_cast<uint16_t *>(s_code + NCHUNK + 3);
    uint16_t *s_qbad = s_bad + NCHUNK + 3;
    uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_rest);           // aliases the code arrays once they are dead

    const int tid = threadIdx.x;
    PHASE_PROF_START;
    const uint64_t per_group = 8ull * (uint64_t)a.tiles_max;
    const uint64_t L = blockIdx.x;
    // workgroup L runs on XCD L % 8: a sample's tiles stay on one XCD, one after the other (its regions' lines are put together in ONE L2).  With
    // fewer than eight samples that would leave XCDs idle (one 100 Mbp sample: an eighth of the chip): then a sample's tiles are dealt to
    // `parts` XCDs in contiguous ranges
    int sample; uint64_t tile;
    if (a.parts > 1) {
        const int v = (int)(L & 7u);
        if (v >= a.n_samples * a.parts) return;
        sample = v / a.parts; tile = (uint64_t)(v % a.parts) * (uint64_t)a.tiles_part + (L >> 3);
        if (tile >= (uint64_t)a.tiles_max || (L >> 3) >= (uint64_t)a.tiles_part) return;
    } else { sample = (int)((L / per_group) * 8 + (L % per_group) % 8); tile = (L % per_group) / 8; }
    if (sample >= a.n_samples) return;
    const uint64_t len = a.lens[sample];
    const uint64_t T0 = tile * TILE;
    if (T0 >= len) return;
    // record streams live in HBM: explicit global address space (a pointer fetched from memory is generic -> flat_load)
    typedef const uint8_t __attribute__((address_space(1))) *gbytes_t;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) *gvec_t;
    gbytes_t seq = (gbytes_t)(uintptr_t)a.seqs[sample];
    gbytes_t qual = a.quals ? (gbytes_t)(uintptr_t)a.quals[sample] : (gbytes_t)0;

    const int k = a.k, h = (k - 1) / 2;
    const HashParams hp = a.hp;
    const int hb = hp.hb;
    const int bshift = hp.bits - a.logB;                    // (word >> 4) >> bshift == bucket
    const int bsh_hi = hp.bits + 4 - 32 - a.logB;           // HI: bucket = upper half >> bsh_hi
    const uint32_t rcflag = a.rc ? ~0u : 0u;
    const bool top_from_hl = a.logB <= hb;                  // bucket = top logB bits of H = top bits of the hashed upper arm
    const int top_shift = top_from_hl ? hb - a.logB : 0;
    const uint32_t am = (1u << (2 * h)) - 1;                // arm mask (h <= 15)

    for (int i = tid; i < B + 4; i += NT) s_hist[i] = 0;
    for (int c = tid; c < NCHUNK; c += NT) {
        const int64_t p = (int64_t)T0 - 64 + 16 * (int64_t)c;
        uint32_t w[4], q[4] = {0, 0, 0, 0};
        if (p >= 0 && (uint64_t)p + 16 <= len) {
            const u32x4 v = *(gvec_t)(seq + p);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            if (qual) { const u32x4 u = *(gvec_t)(qual + p); q[0] = u.x; q[1] = u.y; q[2] = u.z; q[3] = u.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t x = 0, y = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int64_t pos = p + 4 * i + b;
                    bool in = pos >= 0 && (uint64_t)pos < len;
                    x |= (uint32_t)(in ? seq[pos] : (uint8_t)'\n') << (8 * b);
                    if (qual) y |= (uint32_t)(in ? qual[pos] : (uint8_t)'~') << (8 * b);
                }
                w[i] = x; q[i] = y;
            }
        }
        uint32_t code, bad;
        pack16_nonl(w, code, bad);
        uint32_t qb = 0;
        if (qual) {
            uint32_t c2, b2, nl;
            pack16(w, c2, b2, nl);
            qb = qualbad16(q, a.min_qual) & ~nl;
            if (a.qual_filter == 2) bad |= qb;          // QualFilter::Strict (split_kmer.rs:98-101,170-172)
        }
        s_code[c] = code; s_bad[c] = (uint16_t)bad; if (qual) s_qbad[c] = (uint16_t)qb;
    }
    __syncthreads();
    EXT_PROF(0);
    // The text of a tile this XCD will take up shortly (a sample's tiles stay on one XCD; PF_AHEAD tiles = ~4 us of its dispatch order) is
    // pulled into this L2 now: one dword of each of its 128-byte lines, by the first TILE / 128 threads, kept in a register until the wait
    // before the copy-out so that nothing waits for it.  The workgroup that processes that tile then finds its one load per thread --
    // which it can do nothing but wait for -- in the L2 instead of behind the chip's store traffic: 13.2 -> 12.3 ms per 1 000 x 5 Mbp
    // (profiles/r03v_ab_prefetch*.log; 8 tiles ahead is too late, 16 to 24 are alike, 64 fades).  Past the sample's end the head of
    // the sample this XCD takes up next is fetched instead.
    constexpr uint64_t PF_AHEAD = 20;
    uint32_t pfw = 0;
    if (SCATTER && tid < TILE / 128) {
        uint64_t pos = T0 + PF_AHEAD * TILE + (uint64_t)tid * 128;
        gbytes_t ps = seq;
        uint64_t plen = len;
        if (T0 + PF_AHEAD * TILE >= len && sample + 8 < a.n_samples) {
            pos = (tile + PF_AHEAD - (len + TILE - 1) / TILE) * TILE + (uint64_t)tid * 128;
            ps = (gbytes_t)(uintptr_t)a.seqs[sample + 8]; plen = a.lens[sample + 8];
        }
        if (pos + 4 <= plen) pfw = *(const uint32_t __attribute__((address_space(1))) *)(ps + pos);       // (never a byte beyond the stream)
    }

    uint64_t wv[PPT];
    uint32_t rk[PPT];                                       // (bucket << 16) | rank within the tile's bucket; bucket B (the dummy counter) = no window
    {
    const int c0 = tid * NCH + 4;
    // ---- rolling split k-mer over my PPT positions; arms are <= 30 bits, all 32-bit arithmetic ----
    const uint64_t prev = ((uint64_t)s_code[c0 - 2] << 32) | s_code[c0 - 1];  // 32 bases before p0, first base most significant
    // window ending at p0-1: upper arm | middle | lower arm  (split_kmer.rs:104-116)
    uint32_t lower = (uint32_t)prev & am;
    uint32_t mid = (uint32_t)(prev >> (2 * h)) & 3u;
    uint32_t upper = (uint32_t)(prev >> (2 * h + 2)) & am;
    uint32_t rc_upper = revcomp_arm(lower, h), rc_lower = revcomp_arm(upper, h), rc_mid = mid ^ 2u;   // :149-153
#pragma unroll
    for (int half = 0; half < NCH; half++) {
        const int c = c0 + half;
        // which of these 16 windows exist: bit i of the 48-bit fields <-> position (chunk start) - 32 + i.
        // G = good positions; W[i] = all of the k positions ending at i are good (doubling + binary decomposition of k);
        // split_kmer.rs:89,121: a (re)start at idx is abandoned when idx + k >= len  <=>  the clean run is exactly k long
        // (position i-k is bad) and ends at the record's last base (position i+1 is the terminator)
        const uint64_t G = ~((uint64_t)s_bad[c - 2] | ((uint64_t)s_bad[c - 1] << 16) | ((uint64_t)s_bad[c] << 32)) & 0xFFFFFFFFFFFFull;
        uint64_t Wk = ~0ull;
        {
            uint64_t A = G; int offset = 0, span = 1;
            for (int bit = 0; bit < 6; bit++) {
                if ((k >> bit) & 1) { Wk &= A << offset; offset += span; }
                A &= A << span; span <<= 1;
            }
        }
        // the end rule needs "position i + 1 is the record terminator".  A terminator is a bad base, and a clean run of exactly k
        // bases in front of a bad base is rare: only then is the text looked at again
        uint64_t V = Wk;
        const uint64_t cand = Wk & ~(G << k) & ((~G >> 1) | ((uint64_t)(s_bad[c + 1] & 1u) << 47)) & (0xFFFFull << 32);
        if (cand) {
            const int64_t pc = (int64_t)T0 - 64 + 16 * (int64_t)c;                  // the position of bit 32
            uint32_t m = (uint32_t)(cand >> 32);
            while (m) {
                const int j = __builtin_ctz(m); m &= m - 1;
                const uint64_t pos = (uint64_t)(pc + j + 1);
                if (pos >= len || seq[pos] == (uint8_t)'\n') V &= ~(1ull << (32 + j));
            }
        }
        if (qual && a.qual_filter != 0)                                          // middle_base_qual, split_kmer.rs:328-339
            V &= ~((((uint64_t)s_qbad[c - 2] | ((uint64_t)s_qbad[c - 1] << 16) | ((uint64_t)s_qbad[c] << 32))) << h);
        const uint32_t vm = (uint32_t)(V >> 32) & 0xFFFFu;
        const uint32_t cw = s_code[c];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t code = (cw >> (30 - 2 * j)) & 3u;
            // roll_fwd (split_kmer.rs:199-213)
            upper = ((upper << 2) | mid) & am;
            mid = lower >> (2 * h - 2);
            lower = ((lower << 2) | code) & am;
            rc_lower = (rc_lower >> 2) | (rc_mid << (2 * h - 2));
            rc_mid = mid ^ 2u;
            rc_upper = (rc_upper >> 2) | ((code ^ 2u) << (2 * h - 2));
            // canonical = min(fwd, rc) (split_kmer.rs:281-295); equal arms = self-palindrome -> both middles (ska_dict.rs:85-113)
            const uint64_t kf = ((uint64_t)upper << 32) | lower, kr = ((uint64_t)rc_upper << 32) | rc_lower;   // one 64-bit compare each
            const bool gt = kf > kr && rcflag;
            const bool eq = kf == kr && rcflag;
            uint32_t hl = gt ? rc_upper : upper, hr = gt ? rc_lower : lower;
            const uint32_t m4 = (1u << (gt ? rc_mid : mid)) | (eq ? (1u << rc_mid) : 0u);
            hmix_halves(hl, hr, hp);
            const uint64_t w = ((uint64_t)hl << (hb + 4)) | ((uint64_t)hr << 4) | m4;
            wv[16 * half + j] = w;
            const bool valid = (vm >> j) & 1u;
            const uint32_t bk = valid ? (top_from_hl ? hl >> top_shift : (uint32_t)((w >> 4) >> bshift)) : (uint32_t)B;
            const uint32_t r = atomicAdd(&s_hist[bk], 1u);
            rk[16 * half + j] = (bk << 16) | r;             // r <= TILE < 65 536
        }
    }
    }
    __syncthreads();                                        // also: the code arrays are dead from here on
    EXT_PROF(1);
    uint32_t *ghist = a.hist + ((uint64_t)sample << a.logB);
    if (!SCATTER) {
        for (int i = tid; i < B; i += NT) { uint32_t n = s_hist[i]; if (n) atomicAdd(&ghist[i], n); }
        return;
    }
    // reserve one chunk per non-empty bucket in the sample's region (global cursor); the returned bases are not needed
    // until the copy-out, so the atomics stay in flight behind the block scan and the staging pass
    const int R = (B + NT - 1) / NT;                        // buckets per thread (<= RMAX: the launcher picks RMAX from logB)
    const int b0 = tid * R < B ? tid * R : B, b1 = b0 + R < B ? b0 + R : B;
    uint32_t gb[RMAX], lsum = 0;
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        gb[r] = 0;
        if (b0 + r < b1) { const uint32_t n = s_hist[b0 + r]; if (n) gb[r] = atomicAdd(&ghist[b0 + r], n); lsum += n; }
    }
    uint32_t total;
    uint32_t lrun = block_excl_scan_row(lsum, s_tmp, &total, tid);
    EXT_PROF(2);
    // s_hist: counts -> local starts (position of the bucket's first word in the staging buffer)
#pragma unroll
    for (int r = 0; r < RMAX; r++)
        if (b0 + r < b1) { const uint32_t n = s_hist[b0 + r]; s_hist[b0 + r] = lrun; lrun += n; }
    if (tid == 0) s_hist[B] = total;                        // sentinel: start of the (non-existent) bucket B
    __syncthreads();
    EXT_PROF(3);
    constexpr uint32_t HALF = SPLIT ? TILE / 2 : TILE;      // staged words per round
    uint32_t sidx_[SPLIT ? PPT : 1];                        // SPLIT: the staged index of my words, kept for round two
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        // an invalid window gets an index at or behind the tile's word count (s_hist[B] = total; total + invalid windows = TILE): a slot
        // the copy-out never reads
        const uint32_t idx = s_hist[rk[j] >> 16] + (rk[j] & 0xFFFFu);
        if (SPLIT) sidx_[j] = idx;
        if (idx < HALF) s_stage[idx] = wv[j];
    }
    // first use of the cursor atomics' results.  A staged word at index i goes to word s_base[bucket] + i of the sample's
    // span of the word buffer: s_base = region offset inside the span + chunk base - local start (all 32-bit; the span
    // starts at a wave-uniform 64-bit base, so the store is base + 32-bit offset)
    const uint64_t *off = a.off + ((uint64_t)sample << a.logB);
    const bool fixed = a.capacity != 0xFFFFFFFFu;           // fixed-capacity regions: offsets are arithmetic
    const uint64_t span0 = fixed ? ((uint64_t)sample << a.logB) * a.capacity : off[0];
    bool dropped = false;
#pragma unroll
    for (int r = 0; r < RMAX; r++)
        if (b0 + r < b1) {
            const uint32_t b = (uint32_t)(b0 + r), start = s_hist[b], n = s_hist[b + 1] - start;
            // a full region: the host falls back to exact offsets and builds every dictionary again, so the chunk only has to stay inside
            // the word buffer -- it goes to the head of the sample's span (n <= this sample's windows <= the span)
            const bool over = n && gb[r] + n > a.capacity;
            if (over) dropped = true;
            s_base[b] = over ? 0u - start : (fixed ? b * a.capacity : (uint32_t)(off[b] - off[0])) + gb[r] - start;
        }
    asm volatile("" :: "v"(pfw));                           // (the prefetch's register is free from here: the wait for the cursors above covered it)
    __syncthreads();
    EXT_PROF(4);
    typedef uint64_t __attribute__((address_space(1))) *gout_t;
    gout_t out = (gout_t)(uintptr_t)(a.words + span0);
    const uint32_t end0 = total < HALF ? total : HALF;
    for (uint32_t i = tid; i < end0; i += NT) {
        const uint64_t w = s_stage[i];
        const uint32_t b = HI ? (uint32_t)(w >> 32) >> bsh_hi : (uint32_t)((w >> 4) >> bshift);
        out[s_base[b] + i] = w;
    }
    EXT_PROF(5);
    if (SPLIT && total > HALF) {
        __syncthreads();
        EXT_PROF(6);
#pragma unroll
        for (int j = 0; j < PPT; j++) if (sidx_[j] - HALF < HALF) s_stage[sidx_[j] - HALF] = wv[j];
        __syncthreads();
        EXT_PROF(7);
        for (uint32_t i = HALF + tid; i < total; i += NT) {
            const uint64_t w = s_stage[i - HALF];
            const uint32_t b = HI ? (uint32_t)(w >> 32) >> bsh_hi : (uint32_t)((w >> 4) >> bshift);
            out[s_base[b] + i] = w;
        }
    }
    EXT_PROF(8);
    if (dropped) *a.overflow = 1;
}
template <int TILE, bool SPLIT>
static inline size_t extract_lds(const ExtractArgs &a, bool scatter)
{
    size_t codes = (size_t)(TILE / 16 + 8) * 8 + 64;
    size_t stage = scatter ? (size_t)TILE * (SPLIT ? 4 : 8) : 0;
    return ((size_t)8 << a.logB) + 16 + 80 + (stage > codes ? stage : codes);
}
// fewer than eight samples with enough tiles each: 8 / n ranges of tiles per sample
static inline void extract_parts(ExtractArgs &a)
{
    a.parts = 0; a.tiles_part = 0;
    if (a.n_samples >= 1 && a.n_samples < 8 && a.tiles_max >= 64) { a.parts = 8 / a.n_samples; a.tiles_part = (a.tiles_max + a.parts - 1) / a.parts; if (a.parts < 2) a.parts = 0; }
}
template <bool SCATTER, int TILE, bool HI, int PPT, bool SPLIT, int RMAX>
static void launch_extract_t(const ExtractArgs &a_, hipStream_t st)
{
    ExtractArgs a = a_;
    extract_parts(a);
    const uint64_t g = a.parts > 1 ? 8ull * (uint64_t)a.tiles_part : (((uint64_t)a.n_samples + 7) / 8) * 8ull * (uint64_t)a.tiles_max;
    if (!g) return;
    const size_t lds = extract_lds<TILE, SPLIT>(a, SCATTER);
    (void)hipFuncSetAttribute((const void *)extract_kernel<SCATTER, TILE, HI, PPT, SPLIT, RMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((extract_kernel<SCATTER, TILE, HI, PPT, SPLIT, RMAX>), dim3((unsigned)g), dim3(TILE / PPT), lds, st, a);
}
template <bool SCATTER>
static void launch_extract(const ExtractArgs &a, hipStream_t st)
{
    const bool hi = a.hp.bits + 4 - 32 - a.logB >= 0;
    if (a.logB <= 10) {          // 12 288 positions x 768 threads, half-tile staging: two workgroups per CU, 12-word chunks
        if (hi) launch_extract_t<SCATTER, 12288, true, 16, SCATTER, 2>(a, st); else launch_extract_t<SCATTER, 12288, false, 16, SCATTER, 2>(a, st);
    } else if (a.logB == 11) {   // the same with three buckets per thread (6-word chunks; the wider cursor array costs 3 VGPRs and a small spill)
        if (hi) launch_extract_t<SCATTER, 12288, true, 16, SCATTER, 4>(a, st); else launch_extract_t<SCATTER, 12288, false, 16, SCATTER, 4>(a, st);
    } else {                     // 8 192 x 512, half-tile staging (4 096 / 8 192 buckets: the histogram takes 32 / 64 KB of the LDS; 2-word
                                 // chunks -- 300 x 15 Mbp: 49 ms against 63 ms unsplit, either way 4 x the time per base of a 5 Mbp sample)
        if (hi) launch_extract_t<SCATTER, 8192, true, 16, SCATTER, 16>(a, st); else launch_extract_t<SCATTER, 8192, false, 16, SCATTER, 16>(a, st);
    }
}
// Tile sizes.  What HBM delivers for this kernel's writes depends on the size of the (tile, bucket) chunk (tools/scatter_bw.hip:
// 3.5 TB/s for 64-byte pieces, 5.2 TB/s from 128 bytes up, 5.7 TB/s streaming), so the tile is as large as two resident
// workgroups allow: 1 024-bucket samples get 12 288 positions (12-word chunks) with the staging buffer holding half a tile
// at a time (57 KB of LDS, 77 VGPRs x 768 threads), and so do 2 048-bucket samples (6-word chunks, but two workgroups: 15.6 ms
// against 18.2 ms with 16 384-position tiles and one workgroup, 800 x 6 Mbp).
int extract_tile_bases(int logB) { return logB <= 11 ? 12288 : 8192; }
void launch_hist(const ExtractArgs &a, hipStream_t st) { launch_extract<false>(a, st); }
void launch_scatter(const ExtractArgs &a, hipStream_t st)
{
    launch_extract<true>(a, st);
}
// ------------------------------------------------------------------------------------------------
// block-wide helpers
// ------------------------------------------------------------------------------------------------
__device__ static inline uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
    return v;
}
// exclusive scan over the block (<= 1024 threads); total returned through *total
__device__ static inline uint32_t block_excl_scan(uint32_t v, uint32_t *s_tmp /*[17]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < nw; i++) { uint32_t t = s_tmp[i]; s_tmp[i] = run; run += t; } s_tmp[16] = run; }
    __syncthreads();
    uint32_t r = s_tmp[wv] + inc - v;
    if (total) *total = s_tmp[16];
    __syncthreads();
    return r;
}

// inclusive scan over a wave on DPP moves (no LDS permutes)
__device__ static inline uint32_t wave_incl_scan_dpp(uint32_t v)
{
    uint32_t inc = v;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);      // row_shr:1
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);      // row_shr:2
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);      // row_shr:4
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);      // row_shr:8
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return inc;
}
// exclusive scan over the block without thread 0's walk of the waves' sums: those are scanned by sixteen lanes of every wave on DPP moves
// and picked with v_readlane -- two registers, ONE barrier (the caller separates two uses of s_tmp by barriers of its own); tid = the thread
// index as the caller holds it.  (A form that had every thread add up the sixteen sums from LDS kept sixteen registers busy and spilled
// extract_kernel at its 80-VGPR cap.)
__device__ static inline uint32_t block_excl_scan_row(uint32_t v, uint32_t *s_tmp /*[16]*/, uint32_t *total, int tid)
{
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)((blockDim.x + 63) >> 6);
    const uint32_t inc = wave_incl_scan_dpp(v);
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    uint32_t t = lane < nw ? s_tmp[lane & 15] : 0u;                                        // <= 16 waves: one row
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xf, 0xf, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xf, 0xf, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xf, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xf, 0xf, true);
    const uint32_t before = wv ? (uint32_t)__builtin_amdgcn_readlane((int)t, wv - 1) : 0u;
    if (total) *total = (uint32_t)__builtin_amdgcn_readlane((int)t, nw - 1);
    return before + inc - v;
}

// single-workgroup exclusive scan u32 -> u64 (n is at most a few million; launch-bound otherwise)
__global__ __launch_bounds__(1024) void scan_u32_kernel(const uint32_t *in, uint64_t *out, uint64_t n, uint32_t *max_out)
{
    __shared__ uint32_t s_tmp[17];
    __shared__ uint32_t s_max;
    uint64_t carry = 0;
    uint32_t mx = 0;
    if (threadIdx.x == 0) s_max = 0;
    for (uint64_t base = 0; base < n; base += 4096) {
        uint32_t v[4]; uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { uint64_t idx = base + (uint64_t)threadIdx.x * 4 + i; v[i] = idx < n ? in[idx] : 0; sum += v[i]; mx = v[i] > mx ? v[i] : mx; }
        uint32_t tot;
        uint32_t ex = block_excl_scan(sum, s_tmp, &tot);
        uint64_t run = carry + ex;
#pragma unroll
        for (int i = 0; i < 4; i++) { uint64_t idx = base + (uint64_t)threadIdx.x * 4 + i; if (idx < n) out[idx] = run; run += v[i]; }
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
    if (max_out) { atomicMax(&s_max, mx); __syncthreads(); if (threadIdx.x == 0) *max_out = s_max; }
}
__global__ void fill_offsets_kernel(uint64_t *off, uint64_t n, uint32_t capacity)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) off[i] = i * capacity;
}
void launch_fill_offsets(uint64_t *off, uint64_t n, uint32_t capacity, hipStream_t st)
{
    hipLaunchKernelGGL(fill_offsets_kernel, dim3(1024), dim3(256), 0, st, off, n, capacity);
}
void launch_scan_u32(const uint32_t *in, uint64_t *out, uint64_t n, uint32_t *max_out, hipStream_t st)
{
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, in, out, n, max_out);
}

// multi-block exclusive scan of (flag == 1): per-block sums -> single-block scan of the sums -> apply
constexpr int SCAN8_PER_BLOCK = 8192;
__global__ __launch_bounds__(1024) void scan_u8_sums_kernel(const uint8_t *in, uint64_t n, uint32_t *sums)
{
    __shared__ uint32_t s_tmp[17];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN8_PER_BLOCK + (uint64_t)threadIdx.x * 8;
    uint32_t sum = 0;
    if (base + 8 <= n) { uint64_t v = *reinterpret_cast<const uint64_t *>(in + base); for (int i = 0; i < 8; i++) sum += ((v >> (8 * i)) & 0xFF) == 1; }
    else for (int i = 0; i < 8; i++) sum += (base + i < n) && in[base + i] == 1;
    uint32_t tot; block_excl_scan(sum, s_tmp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void scan_u8_apply_kernel(const uint8_t *in, uint64_t n, const uint64_t *block_off, uint64_t *out)
{
    __shared__ uint32_t s_tmp[17];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN8_PER_BLOCK + (uint64_t)threadIdx.x * 8;
    uint32_t v[8]; uint32_t sum = 0;
    for (int i = 0; i < 8; i++) { v[i] = (base + i < n) && in[base + i] == 1; sum += v[i]; }
    uint32_t ex = block_excl_scan(sum, s_tmp, nullptr);
    uint64_t run = block_off[blockIdx.x] + ex;
    for (int i = 0; i < 8; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == blockDim.x - 1) out[n] = run;
}
uint64_t scan_u8_blocks(uint64_t n) { return (n + SCAN8_PER_BLOCK - 1) / SCAN8_PER_BLOCK; }
// scratch (caller-owned, on the stream's device): sums[scan_u8_blocks(n)] u32, offs[scan_u8_blocks(n) + 1] u64
void launch_scan_u8(const uint8_t *flags, uint64_t *pos, uint64_t n, uint32_t *sums, uint64_t *offs, hipStream_t st)
{
    const uint64_t nb = scan_u8_blocks(n);
    if (nb == 0) { (void)hipMemsetAsync(pos, 0, 8, st); return; }
    hipLaunchKernelGGL(scan_u8_sums_kernel, dim3((unsigned)nb), dim3(1024), 0, st, flags, n, sums);
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)sums, offs, nb, (uint32_t *)nullptr);
    hipLaunchKernelGGL(scan_u8_apply_kernel, dim3((unsigned)nb), dim3(1024), 0, st, flags, n, (const uint64_t *)offs, pos);
}

// ------------------------------------------------------------------------------------------------
// order-preserving LDS table: slot = monotone function of the (already uniform) hashed key, linear
// probing without wrap-around.  Maximal runs of occupied slots are sorted independently and the
// runs themselves are in key order, so a sorted, duplicate-free emit is O(n).  Slot value 0 == empty
// (a stored word always has a non-zero base mask in its low 4 bits).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TABLE_PAD = 128;

// bits [sh, sh + popcount(mask)) of a packed word.  HI: the caller guarantees sh >= 32 (field in the upper half): two
// 32-bit operations instead of a 64-bit shift and an AND (on gfx950 the 64-bit shift itself issues at the full rate: tools/valu_rate.hip).  (A run-time test of the uniform sh inside the unrolled
// loops makes the compiler unswitch them and triples the register count.)
template <bool HI>
__device__ static inline uint32_t word_field(uint64_t w, int sh, uint32_t mask)
{
    return HI ? ((uint32_t)(w >> 32) >> (sh - 32)) & mask : (uint32_t)(w >> sh) & mask;
}

__device__ static inline uint32_t home_slot(uint64_t w, int rem_bits, uint32_t nslots)
{
    // l32 = the top 32 of the word's rem_bits local hash bits (word bits [rem_bits - 28, rem_bits + 4)), left-aligned when fewer
    uint32_t l32;
    if (rem_bits >= 32) {
        const int sh = rem_bits - 28;                          // 4..32, wave-uniform: one v_alignbit instead of a 64-bit shift
        l32 = sh >= 32 ? (uint32_t)(w >> 32) : __builtin_amdgcn_alignbit((uint32_t)(w >> 32), (uint32_t)w, (uint32_t)sh);
    } else {
        l32 = rem_bits == 0 ? 0u : (uint32_t)(((w >> 4) & ((1ull << rem_bits) - 1)) << (32 - rem_bits));
    }
    return __umulhi(l32, nslots);
}
// FAST: the caller guarantees 32 <= rem_bits <= 59 (one v_alignbit, and no uniform branch for the compiler to unswitch the
// unrolled probe loops on: that tripled the register count of union_kernel)
template <bool FAST>
__device__ static inline uint32_t home_slot_t(uint64_t w, int rem_bits, uint32_t nslots)
{
    if (FAST) return __umulhi(__builtin_amdgcn_alignbit((uint32_t)(w >> 32), (uint32_t)w, (uint32_t)(rem_bits - 28)), nslots);
    return home_slot(w, rem_bits, nslots);
}
__device__ static inline bool table_insert(unsigned long long *tab, uint32_t total_slots, uint32_t home, uint64_t w)
{
    const uint64_t key = w >> 4;
    for (uint32_t i = home; i < total_slots; ++i) {
        unsigned long long old = tab[i];
        if (old == 0ull) old = atomicCAS(&tab[i], 0ull, (unsigned long long)w);
        if (old == 0ull) return true;
        if ((old >> 4) == key) {
            if ((old | w) != old) atomicOr(&tab[i], (unsigned long long)(w & 15ull));
            return true;
        }
    }
    return false;
}
// as table_insert, reporting where the key lives: 0 = table full, 1 = the key was there, 2 = this call put it there
__device__ static inline int table_insert_slot(unsigned long long *tab, uint32_t total_slots, uint32_t home, uint64_t w, uint32_t &slot)
{
    const uint64_t key = w >> 4;
    for (uint32_t i = home; i < total_slots; ++i) {
        unsigned long long old = tab[i];
        bool mine = false;
        if (old == 0ull) { old = atomicCAS(&tab[i], 0ull, (unsigned long long)w); mine = old == 0ull; }
        if (mine) { slot = i; return 2; }
        if ((old >> 4) == key) {
            if ((old | w) != old) atomicOr(&tab[i], (unsigned long long)(w & 15ull));
            slot = i; return 1;
        }
    }
    return 0;
}
// sorted emit of the table; f(idx, word) is called once per distinct key with its rank; returns the count
template <typename F>
__device__ static inline uint32_t table_emit_sorted(const unsigned long long *tab, uint32_t total_slots, uint32_t *s_tmp, F f)
{
    const uint32_t per = (total_slots + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < total_slots ? lo + per : total_slots;
    uint32_t cnt = 0;
    for (uint32_t i = lo; i < hi; i++) cnt += tab[i] != 0ull;
    uint32_t total;
    uint32_t c = block_excl_scan(cnt, s_tmp, &total);
    for (uint32_t i = lo; i < hi; i++) {
        const unsigned long long w = tab[i];
        if (!w) continue;
        uint32_t gl = 0, lr = 0;
        for (int64_t j = (int64_t)i - 1; j >= 0; j--) { unsigned long long o = tab[j]; if (!o) break; gl += o > w; }
        for (uint32_t j = i + 1; j < total_slots; j++) { unsigned long long o = tab[j]; if (!o) break; lr += o < w; }
        f(c - gl + lr, (uint64_t)w);
        c++;
    }
    return total;
}
// the same, f(idx, word, slot)
template <typename F>
__device__ static inline uint32_t table_emit_sorted_slot(const unsigned long long *tab, uint32_t total_slots, uint32_t *s_tmp, F f)
{
    const uint32_t per = (total_slots + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < total_slots ? lo + per : total_slots;
    uint32_t cnt = 0;
    for (uint32_t i = lo; i < hi; i++) cnt += tab[i] != 0ull;
    uint32_t total;
    uint32_t c = block_excl_scan(cnt, s_tmp, &total);
    for (uint32_t i = lo; i < hi; i++) {
        const unsigned long long w = tab[i];
        if (!w) continue;
        uint32_t gl = 0, lr = 0;
        for (int64_t j = (int64_t)i - 1; j >= 0; j--) { unsigned long long o = tab[j]; if (!o) break; gl += o > w; }
        for (uint32_t j = i + 1; j < total_slots; j++) { unsigned long long o = tab[j]; if (!o) break; lr += o < w; }
        f(c - gl + lr, (uint64_t)w, i);
        c++;
    }
    return total;
}

// pointers that reach a kernel inside an argument struct are generic (flat_load: slower, and it also ticks the LDS
// counter); dictionary words only ever live in HBM, so read them through an explicit global-address-space pointer
typedef const uint64_t __attribute__((address_space(1))) *gwords_t;
__device__ static inline gwords_t as_global(const uint64_t *p) { return (gwords_t)(uintptr_t)p; }

// K3: per (sample,bucket) region: dedupe (OR of base masks) + sort, in place
// K3 (fallback for regions larger than the counting sort's LDS capacity, i.e. heavy repeat content: tandem repeats,
// homopolymers): duplicates collapse on insertion, so only the number of DISTINCT keys has to fit.
__global__ __launch_bounds__(256) void dedupe_kernel(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt,
                                                     uint32_t nslots, int rem_bits, int *overflow, uint32_t min_n, uint16_t *sidx, int sb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_tab[];
    __shared__ uint32_t s_tmp[17];
    __shared__ uint32_t s_sub[SUBIDX];
    __shared__ int s_fail;
    const uint64_t region = blockIdx.x;
    const uint32_t n = raw[region];
    if (n <= min_n) return;                                  // handled by the counting-sort kernel
    if (n == 0) { if (threadIdx.x == 0) ucnt[region] = 0; if (threadIdx.x < SUBIDX) sidx[region * SUBIDX + threadIdx.x] = 0; return; }
    if (threadIdx.x < SUBIDX) s_sub[threadIdx.x] = 0xFFFFFFFFu;
    const uint32_t total_slots = nslots + TABLE_PAD;
    for (uint32_t i = threadIdx.x; i < total_slots; i += blockDim.x) s_tab[i] = 0ull;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    uint64_t *reg = words + off[region];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t w = reg[i];
        if (!table_insert(s_tab, total_slots, home_slot(w, rem_bits, nslots), w)) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) { if (threadIdx.x == 0) { *overflow = 1; ucnt[region] = 0; } if (threadIdx.x < SUBIDX) sidx[region * SUBIDX + threadIdx.x] = 0; return; }
    const uint64_t lmask = rem_bits >= 60 ? ~0ull : ((1ull << rem_bits) - 1);
    uint32_t total = table_emit_sorted(s_tab, total_slots, s_tmp, [&](uint32_t idx, uint64_t w) {
        reg[idx] = w;
        atomicMin(&s_sub[sb ? (uint32_t)(((w >> 4) & lmask) >> (rem_bits - sb)) : 0u], idx);
    });
    if (threadIdx.x == 0) ucnt[region] = total;
    __syncthreads();
    if (threadIdx.x < SUBIDX) {                              // first word of sub-range s = first set entry at or above s
        uint32_t v = total;
        for (int s2 = SUBIDX - 1; s2 >= (int)threadIdx.x; s2--) if (s_sub[s2] != 0xFFFFFFFFu) v = s_sub[s2];
        sidx[region * SUBIDX + threadIdx.x] = (uint16_t)v;
    }
}
void launch_dedupe(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_regions,
                   uint32_t table_slots, int rem_bits, int *overflow, uint32_t min_n, uint16_t *sidx, int sb, hipStream_t st)
{
    if (!n_regions) return;
    size_t lds = (size_t)(table_slots + TABLE_PAD) * 8;
    (void)hipFuncSetAttribute((const void *)dedupe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dedupe_kernel, dim3((unsigned)n_regions), dim3(256), lds, st, words, off, raw, ucnt, table_slots, rem_bits, overflow, min_n, sidx, sb);
}

// (Leaving a region grouped by micro-bucket but unordered inside -- all its consumers need, and 5 % cheaper here: 20.05 -> 19.05 ms -- was
// measured and dropped: union and assemble then run at less than half their rate (8.4 -> 19.9 ms, 14.9 -> 28.4 ms), because with sorted
// slices neighbouring lanes look up neighbouring table slots / rows in LDS; profiles/r03j_ab_grouped_full.log.)
// K3 (fast path): counting sort of a region into micro-buckets of ~2-4 words by the next hash bits, then a tiny
// per-thread insertion sort with duplicate folding (OR of base masks).  No CAS loops, no data-dependent probe
// chains: cost is O(n) LDS operations per region whatever the duplication level.
template <int ITEMS, bool HI, int NT>
__global__ __launch_bounds__(NT) void dedupe_mb_kernel(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt,
                                                        uint32_t cap, int rem_bits, int *overflow, uint16_t *sidx, int sb,
                                                        const uint32_t *big, uint32_t big_from, int big_mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    __shared__ uint32_t s_tmp[17];
    // big_mode 0: every region; 1: every region, those above this launch's capacity are on the list `big` ([0] = count) and
    // left to a launch of mode 2; 2: the listed regions from entry big_from on, one per workgroup (bit 4 of *overflow: the list is
    // longer than this grid, the host launches the rest)
    PHASE_PROF_START;
    uint64_t region = blockIdx.x;
    if (big_mode == 2) {
        const uint32_t cnt = big[0], idx = big_from + blockIdx.x;
        if (blockIdx.x == 0 && threadIdx.x == 0 && cnt > big_from + gridDim.x) atomicOr(overflow, 4);
        if (idx >= cnt) return;
        region = big[1 + idx];
    }
    const uint32_t n = raw[region];
    if (n == 0) { if (threadIdx.x == 0) ucnt[region] = 0; return; }
    if (big_mode == 1 && n > (uint32_t)NT * ITEMS) return;
    if (n > cap || n > (uint32_t)NT * ITEMS) { if (threadIdx.x == 0) atomicOr(overflow, 2); return; }      // left to dedupe_kernel
    uint64_t *s_elem = reinterpret_cast<uint64_t *>(s_mem);                 // [cap]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_mem + (size_t)cap * 8 + 32) + 1;   // (four sentinel words first) [-1] = 0 | [M] counts -> cursors (= bucket ends) -> leader counts
    int logM = 31 - __clz(n);                                                // ~1..2 words per micro-bucket
    if (logM < 0) logM = 0;
    if (logM > rem_bits) logM = rem_bits;
    while ((1u << logM) > cap) logM--;
    const uint32_t M = 1u << logM;
    const int mshift = rem_bits - logM + 4;                                  // micro-bucket = word_field<HI>(w, mshift, M - 1)
    uint64_t *reg = words + off[region];
    // the whole region goes into registers with all loads in flight at once (word 0 never occurs: base masks are non-zero).
    // (Requesting the words before the region's count is known -- fixed-capacity regions have arithmetic addresses -- was measured: 19.0 -> 19.7 ms,
    // profiles/r03y_ab_dedupe_specload.log.)
    uint64_t e[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; t++) { const uint32_t i = threadIdx.x + (uint32_t)NT * t; e[t] = i < n ? reg[i] : 0ull; }
    for (uint32_t i = threadIdx.x; i < M; i += NT) s_cnt[i] = 0;
    if (threadIdx.x == 0) s_cnt[-1] = 0;                                     // end of the bucket before the first
    if (threadIdx.x < 4) s_elem[n + threadIdx.x] = ~0ull;                    // what the ranking reads behind the last word
    __syncthreads();
    DD_PROF(0);
#pragma unroll
    for (int t = 0; t < ITEMS; t++) if (e[t]) atomicAdd(&s_cnt[word_field<HI>(e[t], mshift, M - 1)], 1u);
    __syncthreads();
    DD_PROF(1);
    // exclusive scan of the counts (each thread owns R consecutive micro-buckets)
    const uint32_t R = (M + NT - 1) / NT;
    const uint32_t m0 = threadIdx.x * R < M ? threadIdx.x * R : M, m1 = m0 + R < M ? m0 + R : M;
    uint32_t sum = 0;
    for (uint32_t m = m0; m < m1; m++) sum += s_cnt[m];
    uint32_t run = block_excl_scan_row(sum, s_tmp, nullptr, (int)threadIdx.x);
    for (uint32_t m = m0; m < m1; m++) { uint32_t c = s_cnt[m]; s_cnt[m] = run; run += c; }      // start of m; the scatter below turns it into its end = start of m + 1
    __syncthreads();
    DD_PROF(2);
    uint32_t pos[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; t++) {
        pos[t] = 0;
        if (e[t]) { pos[t] = atomicAdd(&s_cnt[word_field<HI>(e[t], mshift, M - 1)], 1u); s_elem[pos[t]] = e[t]; }
    }
    __syncthreads();
    DD_PROF(3);
    // rank every word inside its micro-bucket (all words in parallel, 4 LDS reads in flight per step): sorted position =
    // start + #smaller keys + #equal keys at lower positions; equal keys also fold their base masks together
    // position-ordered from here on (p = tid + NT t): neighbouring lanes touch neighbouring LDS words, so the random-bank
    // conflicts of the scatter above do not come back.  Rank of a word inside its micro-bucket = #smaller keys + #equal
    // keys at lower positions; equal keys also fold their base masks together.
    // The three dependent LDS reads per word (the word, its micro-bucket's bounds, the bucket's first slots) are issued
    // for all of the thread's words at once, stage by stage, so a thread waits for three round trips instead of 3 x ITEMS.
    uint32_t npos[ITEMS], bb[ITEMS], ee[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; t++) { const uint32_t p = threadIdx.x + (uint32_t)NT * t; e[t] = s_elem[p < n ? p : n - 1]; }
#pragma unroll
    for (int t = 0; t < ITEMS; t++) asm volatile("" : "+v"(e[t]));
#pragma unroll
    for (int t = 0; t < ITEMS; t++) { const uint32_t m = word_field<HI>(e[t], mshift, M - 1); bb[t] = s_cnt[(int)m - 1]; ee[t] = s_cnt[m]; }
#pragma unroll
    for (int t = 0; t < ITEMS; t++) asm volatile("" : "+v"(bb[t]), "+v"(ee[t]));
#pragma unroll
    for (int t = 0; t < ITEMS; t++) {
        const uint32_t p = threadIdx.x + (uint32_t)NT * t;
        npos[t] = 0;
        if (p >= n) { e[t] = 0; continue; }
        const uint64_t w0 = e[t], w0lo = w0 & ~15ull;
        const uint32_t b = bb[t], eend = ee[t];
        // key_j < key  <=>  w_j < (w0 & ~15);  key_j == key  <=>  (w_j ^ w0) < 16: no 64-bit shifts in the loop.
        // Fast path: count the smaller keys and notice whether the key occurs again; only then (rare) order the equal
        // keys by position and fold their base masks.
        uint32_t less = 0, eqb = 0, mor = (uint32_t)w0 & 15u;
        // Branch-free over the bucket's first NS slots: a slot past the bucket's end holds a word of a later bucket (larger, and
        // never this key) or one of the four all-ones sentinels behind the region's last word, so it needs no validity test;
        // `same` counts the slots holding this key -- the word itself is one of them when it lies in the first NS.
        uint32_t same = 0;
        auto step = [&](uint64_t w) {                                       // (same key <=> equal once the base mask is cleared: one 32-bit AND and
            less += w < w0lo ? 1u : 0u;                                     // one 64-bit compare instead of two XORs, a shift, an OR and a compare)
            same += (w & ~15ull) == w0lo ? 1u : 0u;
        };
        constexpr uint32_t NS = 4;                                             // micro-buckets hold ~1.2 words on average
        uint64_t wq[NS];
#pragma unroll
        for (uint32_t u = 0; u < NS; u++) wq[u] = s_elem[b + u];             // reads in flight whatever the bucket's size
#pragma unroll
        for (uint32_t u = 0; u < NS; u++) asm volatile("" : "+v"(wq[u]));     // (keeps the compiler from sinking each read into its own branch)
#pragma unroll
        for (uint32_t u = 0; u < NS; u++) step(wq[u]);
        bool dup = same > (p - b < NS ? 1u : 0u);
        for (uint32_t j = b + NS; j < eend; j++) {                           // longer buckets finish in a loop
            const uint64_t w = s_elem[j];
            less += w < w0lo ? 1u : 0u;
            dup |= j != p && (w & ~15ull) == w0lo;
        }
        if (dup)
            for (uint32_t j = b; j < eend; j++) {
                const uint64_t w = s_elem[j];
                const bool iseq = (w & ~15ull) == w0lo;
                eqb += iseq && j < p;
                mor |= iseq ? (uint32_t)w & 15u : 0u;
            }
        npos[t] = b + less + eqb;
        e[t] = (w0 & ~15ull) | mor;
    }
    __syncthreads();
    DD_PROF(4);
#pragma unroll
    for (int t = 0; t < ITEMS; t++) if (e[t]) s_elem[npos[t]] = e[t];
    __syncthreads();
    DD_PROF(5);
    // keep the first word of every run of equal keys; compaction index from wave ballots + a tiny per-row table
    constexpr int NW = NT / 64;
    uint32_t *s_rows = s_cnt;                               // [ITEMS][NW] leaders per (row, wave); the cursors are dead now
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t flags = 0, sflags = 0, below[ITEMS];
    const int subshift = rem_bits - sb + 4;                 // sub-range of a word = its next sb hash bits
    const uint32_t submask = (1u << sb) - 1;
#pragma unroll
    for (int t = 0; t < ITEMS; t++) {
        const uint32_t p = threadIdx.x + (uint32_t)NT * t;
        bool lead = false;
        if (p < n) {
            e[t] = s_elem[p];
            // one XOR with the predecessor answers both questions: keys differ <=> a bit above the base mask differs;
            // new sub-range <=> a bit at or above subshift differs (all words of a region agree above rem_bits + 4)
            const uint64_t x = p ? s_elem[p - 1] ^ e[t] : ~0ull;
            lead = x > 15ull;
            if (HI ? (uint32_t)(x >> 32) >= (1u << (subshift - 32)) : x >= (1ull << subshift)) sflags |= 1u << t;
        }
        const unsigned long long bal = __ballot(lead);
        below[t] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (lead) flags |= 1u << t;
        if (lane == 0) s_rows[t * NW + wv] = __popcll(bal);
    }
    __syncthreads();
    DD_PROF(6);
    // counts per (row, wave) -> exclusive prefix in output order (one wave, entries i and i + 64), total at [ITEMS * 4]
    if (wv == 0) {
        constexpr int NE = ITEMS * NW;
        const uint32_t a = lane < NE ? s_rows[lane] : 0u, b2 = lane + 64 < NE ? s_rows[lane + 64] : 0u;
        const uint32_t ia = wave_incl_scan_dpp(a), ta = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
        const uint32_t ib = NE > 64 ? wave_incl_scan_dpp(b2) : 0u;
        if (lane < NE) s_rows[lane] = ia - a;
        if (NE > 64 && lane + 64 < NE) s_rows[lane + 64] = ta + ib - b2;
        if (lane == 63) s_rows[NE] = ta + (NE > 64 ? ib : 0u);
    }
    __syncthreads();
    DD_PROF(7);
#pragma unroll
    for (int t = 0; t < ITEMS; t++) {
        if (!((flags >> t) & 1u)) continue;
        const uint32_t o = s_rows[t * NW + wv] + below[t];
        reg[o] = e[t];
        // first word of a new sub-range: record where it starts (sub-ranges without words keep the 0xFFFF the host
        // pre-filled; readers take the next recorded start)
        if ((sflags >> t) & 1u) sidx[region * SUBIDX + word_field<false>(e[t], subshift, submask)] = (uint16_t)o;
    }
    if (threadIdx.x == 0) ucnt[region] = s_rows[ITEMS * NW];
    DD_PROF(8);
}
template <int ITEMS, int NT>
static void launch_dedupe_items(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_blocks, uint32_t cap,
                                int rem_bits, int *overflow, uint16_t *sidx, int sb, size_t lds, hipStream_t st,
                                const uint32_t *big, uint32_t big_from, int big_mode)
{
    // every field the kernel extracts (micro-bucket, sub-range) starts at bit rem_bits + 4 - (<= log2 cap) or higher
    int lc = 0; while ((1u << lc) < cap) lc++;
    if (rem_bits + 4 - lc >= 32 && rem_bits + 4 - sb >= 32) {
        (void)hipFuncSetAttribute((const void *)dedupe_mb_kernel<ITEMS, true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((dedupe_mb_kernel<ITEMS, true, NT>), dim3((unsigned)n_blocks), dim3(NT), lds, st, words, off, raw, ucnt, cap, rem_bits, overflow, sidx, sb, big, big_from, big_mode);
    } else {
        (void)hipFuncSetAttribute((const void *)dedupe_mb_kernel<ITEMS, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((dedupe_mb_kernel<ITEMS, false, NT>), dim3((unsigned)n_blocks), dim3(NT), lds, st, words, off, raw, ucnt, cap, rem_bits, overflow, sidx, sb, big, big_from, big_mode);
    }
}
// launch shapes of the counting sort: words per region = threads x words per thread.
// 512 threads while three regions (<= 4 096 words, 49 KB) share a CU; beyond that the LDS footprint fixes two regions per
// CU and 1 024 threads keep 32 waves on it (800 x 6 Mbp, 3 840-word regions: <6, 1024> 26.4 ms, <8, 512> 20.9 ms).
static uint32_t dedupe_shape_words(uint32_t cap)
{
    return cap <= 512u * 4 ? 512u * 4 : cap <= 512u * 7 ? 512u * 7 : cap <= 512u * 8 ? 512u * 8 : cap <= 1024u * 5 ? 1024u * 5 : 1024u * 6;
}
static void launch_dedupe_shape(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_blocks, uint32_t cap,
                                int rem_bits, int *overflow, uint16_t *sidx, int sb, hipStream_t st, const uint32_t *big, uint32_t big_from, int big_mode)
{
    size_t lds = (size_t)cap * 8 + 32 + (size_t)cap * 4 + 16;  // elements + four sentinels + two u32 arrays of cap/2 (+ sentinel)
    if (cap <= 512u * 4) launch_dedupe_items<4, 512>(words, off, raw, ucnt, n_blocks, cap, rem_bits, overflow, sidx, sb, lds, st, big, big_from, big_mode);
    else if (cap <= 512u * 7) launch_dedupe_items<7, 512>(words, off, raw, ucnt, n_blocks, cap, rem_bits, overflow, sidx, sb, lds, st, big, big_from, big_mode);
    else if (cap <= 512u * 8) launch_dedupe_items<8, 512>(words, off, raw, ucnt, n_blocks, cap, rem_bits, overflow, sidx, sb, lds, st, big, big_from, big_mode);
    else if (cap <= 1024u * 5) launch_dedupe_items<5, 1024>(words, off, raw, ucnt, n_blocks, cap, rem_bits, overflow, sidx, sb, lds, st, big, big_from, big_mode);
    else launch_dedupe_items<6, 1024>(words, off, raw, ucnt, n_blocks, cap, rem_bits, overflow, sidx, sb, lds, st, big, big_from, big_mode);    // host keeps regions <= 6144 words
}
uint32_t dedupe_spill_grid()
{
    const long v = knob("dedupe_spill_grid");
    return v > 0 ? (uint32_t)v : 16384u;
}
// regions with more than `above` words -> list[1..], count at list[0] (zeroed by the caller)
__global__ void list_big_regions_kernel(const uint32_t *raw, uint64_t n, uint32_t above, uint32_t *list)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n && raw[i] > above) list[1 + atomicAdd(&list[0], 1u)] = (uint32_t)i;
}
// cap = the largest region the launch must hold (the regions' capacity); typical (0: unknown) = a size all but a few in ten
// thousand regions stay below (mean + 3.3 sigma of the Poisson counts).  When typical needs a smaller launch shape than cap, the
// regions are sorted in that shape -- a 4 900-word region fills 95 % of <5, 1024> but 80 % of <6, 1024>, and the kernel is bound
// by its instruction count -- and the few above it, listed by a first small kernel, in a second launch of the full shape.
// big_list: [n_regions + 1] words of scratch; big_from > 0 relaunches the second stage from that list entry (overflow bit 4).
void launch_dedupe_mb(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_regions, uint32_t cap,
                      int rem_bits, int *overflow, uint16_t *sidx, int sb, hipStream_t st, uint32_t typical, uint32_t *big_list, uint32_t big_from)
{
    if (!n_regions) return;
    const uint32_t shape_all = dedupe_shape_words(cap), shape_typ = typical ? dedupe_shape_words(typical) : shape_all;
    if (!big_list || shape_typ >= shape_all || n_regions > 0xFFFFFFFFull) {
        launch_dedupe_shape(words, off, raw, ucnt, n_regions, cap, rem_bits, overflow, sidx, sb, st, nullptr, 0u, 0);
        return;
    }
    const unsigned spill_grid = dedupe_spill_grid();
    if (big_from == 0) {
        (void)hipMemsetAsync(big_list, 0, 4, st);
        hipLaunchKernelGGL(list_big_regions_kernel, dim3((unsigned)((n_regions + 255) / 256)), dim3(256), 0, st, raw, n_regions, shape_typ, big_list);
        launch_dedupe_shape(words, off, raw, ucnt, n_regions, shape_typ, rem_bits, overflow, sidx, sb, st, big_list, 0u, 1);
    }
    launch_dedupe_shape(words, off, raw, ucnt, spill_grid, cap, rem_bits, overflow, sidx, sb, st, big_list, big_from, 2);
}


// first index in [lo,hi) whose hashed key (word >> 4) is >= x
__device__ static inline uint32_t lower_bound_words(const uint64_t *reg_, uint32_t lo, uint32_t hi, uint64_t x)
{
    gwords_t reg = as_global(reg_);
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if ((reg[mid] >> 4) < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// slice [lo,hi) of sample's region that belongs to sub-bucket j of a 2^logN split (logN >= logB)
__device__ static inline void sub_slice(const DictView &d, int sample, uint64_t j, int logN, const uint64_t *&reg, uint32_t &lo, uint32_t &hi)
{
    const int sh = logN - d.logB;
    const uint64_t b = j >> sh;
    const uint64_t region = ((uint64_t)sample << d.logB) + b;
    reg = d.words + d.off[region];
    const uint32_t n = d.ucnt[region];
    if (sh == 0) { lo = 0; hi = n; return; }
    const uint32_t jj = (uint32_t)(j & ((1ull << sh) - 1)), last = (1u << sh) - 1;
    const int rb = d.bits - logN;
    const uint64_t xlo = j << rb, xhi = (j + 1) << rb;
    if (d.sidx) {
        // the dedupe kernels left the starts of the region's 2^sb sub-ranges: direct look-up, or a short search inside one
        typedef const uint16_t __attribute__((address_space(1))) *gidx_t;
        gidx_t ix = (gidx_t)(uintptr_t)(d.sidx + region * SUBIDX);
        // start of sub-range a = first recorded start at or above a (0xFFFF: no word there), else the region's end
        auto start_of = [&](uint32_t a) -> uint32_t {
            for (; a < (1u << d.sb); a++) { const uint32_t v = ix[a]; if (v != 0xFFFFu) return v; }
            return n;
        };
        if (sh <= d.sb) {
            lo = start_of(jj << (d.sb - sh));
            hi = jj == last ? n : start_of((jj + 1) << (d.sb - sh));
            return;
        }
        if (d.sb > 0) {
            const uint32_t c = jj >> (sh - d.sb);
            const uint32_t a0 = start_of(c), a1 = start_of(c + 1);
            lo = lower_bound_words(reg, a0, a1, xlo);
            hi = ((jj + 1) & ((1u << (sh - d.sb)) - 1)) == 0 ? a1 : lower_bound_words(reg, lo, a1, xhi);
            return;
        }
    }
    lo = lower_bound_words(reg, 0, n, xlo);
    hi = jj == last ? n : lower_bound_words(reg, lo, n, xhi);
}

// K4: distinct keys of sub-bucket j over all samples -> sorted slab
#ifndef SKX_UNION_U
#define SKX_UNION_U 5
#endif
#ifndef SKX_UNION_PREFETCH
#define SKX_UNION_PREFETCH 0
#endif
// SIDE: the pass also leaves, for every word of every dictionary, WHERE its key went -- side[word index] = (first-seen rank of the key
// in its sub-bucket << 4) | base set, 16 bits -- and per sub-bucket the map first-seen rank -> sorted row (perm).  assemble_side_kernel
// then fills the matrix from those 2 bytes per word instead of reading the 8-byte words again and looking every key up a second time
// (merge_ska_dict.rs:77-109 appends a sample in one pass over its dictionary: so does this pair of kernels, up to the 2-byte note).
// A key gets its rank from the lane whose CAS put it into the table; a lane of another wave that meets the key before the rank is
// written (0xFFFF) takes the slow path and waits for it there.
template <bool COUNT_ONLY, bool FAST, bool SIDE>
__global__ __launch_bounds__(1024, 8) void union_kernel(DictView d, int logN, uint64_t *stage, uint32_t stride, uint32_t *ncnt,
                                                    uint32_t nslots, int *overflow, uint16_t *side, uint16_t *perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_tab[];
    __shared__ uint32_t s_tmp[17];
    __shared__ int s_fail;
    __shared__ uint32_t s_nrows;
    PHASE_PROF_START;
    const uint64_t j = blockIdx.x;
    const uint32_t total_slots = nslots + TABLE_PAD;
    uint16_t *s_rank = reinterpret_cast<uint16_t *>(s_tab + total_slots);      // SIDE: [total_slots] first-seen rank of the key in each slot
    for (uint32_t i = threadIdx.x; i < total_slots; i += blockDim.x) { s_tab[i] = 0ull; if (SIDE) s_rank[i] = 0xFFFFu; }
    if (threadIdx.x == 0) { s_fail = 0; s_nrows = 0; }
    __syncthreads();
    typedef uint16_t __attribute__((address_space(1))) *gside_t;
    gside_t gside = (gside_t)(uintptr_t)side;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rem_bits = d.bits - logN;
    // every lane binary-searches the slice of one sample; the wave then streams the 64 slices one after another
    // samples are dealt evenly to the waves (a few hundred samples would otherwise leave most waves idle)
    const int per = (d.n_samples + nw - 1) / nw;
    const int wend = (wv + 1) * per < d.n_samples ? (wv + 1) * per : d.n_samples;
    for (int s0 = wv * per; s0 < wend; s0 += 64) {
        const uint64_t *my_reg = nullptr; uint32_t my_lo = 0, my_hi = 0;
        const int cnt = wend - s0 < 64 ? wend - s0 : 64;
        if (lane < cnt) sub_slice(d, s0 + lane, j, logN, my_reg, my_lo, my_hi);
        // One slice at a time, UU words per lane (a slice of the 1 000 x 5 Mbp case is ~300 words: one batch).  >99 % of the
        // words are already in the table with their base in the stored mask, so the home slots of a whole batch are probed
        // with independent LDS reads (one latency per batch, not one per word) and only the rest takes the insert loop;
        // the next slice's loads are issued before the current batch is looked at.
        constexpr int UU = SKX_UNION_U;
        auto fetch = [&](int t, uint64_t (&q)[UU], uint32_t &lo, uint32_t &hi, gwords_t &reg) {
            reg = as_global(reinterpret_cast<const uint64_t *>(__shfl((unsigned long long)my_reg, t, 64)));
            lo = __shfl(my_lo, t, 64); hi = __shfl(my_hi, t, 64);
#pragma unroll
            for (int u = 0; u < UU; u++) { const uint32_t i = lo + 64u * u + lane; q[u] = i < hi ? reg[i] : 0ull; }
        };
        auto absorb = [&](const uint64_t (&q)[UU], gwords_t reg, uint32_t first) {
            uint32_t miss = 0;
            const uint64_t sbase = SIDE ? (uint64_t)((const uint64_t *)reg - d.words) + first + lane : 0;      // this lane's first word in the side array
#pragma unroll
            for (int g = 0; g < UU; g += 3) {
                unsigned long long s0[3], s1[3]; uint32_t hs[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {                            // home slot and its successor (one ds_read2_b64): a key displaced by
                    if (g + u >= UU) continue;                           // one slot is still a hit
                    const uint32_t hslot = home_slot_t<FAST>(q[g + u], rem_bits, nslots);
                    s0[u] = s_tab[hslot]; s1[u] = s_tab[hslot + 1]; hs[u] = hslot;
                }
                uint32_t rkq[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    if (g + u >= UU) continue;
                    const uint64_t w = q[g + u];
                    const bool at0 = ((s0[u] ^ w) >> 4) == 0ull;
                    const unsigned long long at = at0 ? s0[u] : (s0[u] != 0ull && ((s1[u] ^ w) >> 4) == 0ull) ? s1[u] : 0ull;
                    if (w && (w & 15ull & ~at) != 0ull) miss |= 1u << (g + u);      // not found, or found without this base in its mask
                    if (SIDE) rkq[u] = s_rank[hs[u] + (at0 ? 0u : 1u)];
                }
                if (SIDE) {
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        if (g + u >= UU) continue;
                        const uint64_t w = q[g + u];
                        if (!w || ((miss >> (g + u)) & 1u)) continue;
                        if (rkq[u] == 0xFFFFu) { miss |= 1u << (g + u); continue; }       // the key is there, its rank is still being written
                        gside[sbase + 64u * (g + u)] = (uint16_t)((rkq[u] << 4) | ((uint32_t)w & 15u));
                    }
                }
            }
            while (miss) {                                               // first sightings, new bases, keys displaced further: the word is
                const int u = __ffs(miss) - 1; miss &= miss - 1;         // read again (L2 / L1 hit) rather than selected from q[] by a
                const uint64_t w = reg[first + 64u * u + lane];          // run-time index, which would put the batch into scratch memory
                if (!SIDE) { if (!table_insert(s_tab, total_slots, home_slot_t<FAST>(w, rem_bits, nslots), w)) s_fail = 1; }
                else {
                    uint32_t sl = 0;
                    const int res = table_insert_slot(s_tab, total_slots, home_slot_t<FAST>(w, rem_bits, nslots), w, sl);
                    if (res == 0) { s_fail = 1; continue; }
                    // two statements, in this order: the lanes that put a key there write its rank before any lane of the wave waits for
                    // one (an if / else would let the compiler run the waiting side first, with the writers of the same wave masked off)
                    uint32_t r = 0xFFFFu;
                    if (res == 2) { r = atomicAdd(&s_nrows, 1u); reinterpret_cast<volatile uint16_t *>(s_rank)[sl] = (uint16_t)r; }
                    __builtin_amdgcn_wave_barrier();
                    if (res == 1) {                                       // (bounded: a rank that never arrives fails the launch instead of hanging it)
                        for (int it = 0; it < (1 << 16) && r == 0xFFFFu; it++) r = reinterpret_cast<volatile uint16_t *>(s_rank)[sl];
                        if (r == 0xFFFFu) s_fail = 1;
                    }
                    gside[sbase + 64u * u] = (uint16_t)((r << 4) | ((uint32_t)w & 15u));
                }
            }
        };
        uint64_t nx[UU]; uint32_t nlo, nhi; gwords_t nreg;
        UN_WAIT(); UN_PROF(0);
        fetch(0, nx, nlo, nhi, nreg);
        for (int t = 0; t < cnt; t++) {
            uint64_t cur[UU];
#pragma unroll
            for (int u = 0; u < UU; u++) cur[u] = nx[u];
            const uint32_t lo = nlo, hi = nhi; gwords_t reg = nreg;
#if SKX_UNION_PREFETCH
            if (t + 1 < cnt) fetch(t + 1, nx, nlo, nhi, nreg);
#endif
            UN_WAIT(); UN_PROF(1);
            absorb(cur, reg, lo);
            for (uint32_t o = lo + 64u * UU; o < hi; o += 64u * UU) {          // longer slices: the rest, batch by batch
#pragma unroll
                for (int u = 0; u < UU; u++) { const uint32_t i = o + 64u * u + lane; cur[u] = i < hi ? reg[i] : 0ull; }
                absorb(cur, reg, o);
            }
            UN_PROF(2);
#if !SKX_UNION_PREFETCH
            if (t + 1 < cnt) fetch(t + 1, nx, nlo, nhi, nreg);
#endif
        }
    }
    __syncthreads();
    UN_PROF(2);
    if (s_fail) { if (threadIdx.x == 0) *overflow = 1; return; }
    if (COUNT_ONLY) {
        uint32_t cnt = 0;
        for (uint32_t i = threadIdx.x; i < total_slots; i += blockDim.x) cnt += s_tab[i] != 0ull;
        uint32_t tot; block_excl_scan(cnt, s_tmp, &tot);
        if (threadIdx.x == 0) atomicAdd(ncnt, tot);
        return;
    }
    uint64_t *slab = stage + j * (uint64_t)stride;
    uint32_t total;
    if (!SIDE) total = table_emit_sorted(s_tab, total_slots, s_tmp, [&](uint32_t idx, uint64_t w) { if (idx < stride) slab[idx] = (w & ~15ull) | 1ull; });
    else {
        uint16_t *pj = perm + j * (uint64_t)stride;                    // first-seen rank -> sorted row of the sub-bucket
        total = table_emit_sorted_slot(s_tab, total_slots, s_tmp, [&](uint32_t idx, uint64_t w, uint32_t slot) {
            if (idx < stride) { slab[idx] = (w & ~15ull) | 1ull; const uint32_t r = s_rank[slot]; if (r < stride) pj[r] = (uint16_t)idx; }
        });
    }
    if (threadIdx.x == 0) { ncnt[j] = total; if (total > stride || (SIDE && total > 4095u)) *overflow = 1; }
    UN_PROF(3);
}
template <bool COUNT_ONLY>
static void launch_union_t(const DictView &d, int logN, unsigned blocks, uint64_t *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots,
                           int *overflow, hipStream_t st)
{
    size_t lds = (size_t)(table_slots + TABLE_PAD) * 8;
    const int rem = d.bits - logN;
    if (rem >= 32 && rem <= 59) {
        (void)hipFuncSetAttribute((const void *)union_kernel<COUNT_ONLY, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((union_kernel<COUNT_ONLY, true, false>), dim3(blocks), dim3(1024), lds, st, d, logN, stage, stride, ncnt, table_slots, overflow, (uint16_t *)nullptr, (uint16_t *)nullptr);
    } else {
        (void)hipFuncSetAttribute((const void *)union_kernel<COUNT_ONLY, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((union_kernel<COUNT_ONLY, false, false>), dim3(blocks), dim3(1024), lds, st, d, logN, stage, stride, ncnt, table_slots, overflow, (uint16_t *)nullptr, (uint16_t *)nullptr);
    }
}
// the union that also notes where every word went (side: one u16 per word of d.words; perm: [2^logN][stride] u16); the caller checks
// union_side_ok first
bool union_side_ok(const DictView &d, int logN, uint32_t stride) { const int rem = d.bits - logN; return rem >= 32 && rem <= 59 && stride <= 4096u; }
void launch_union_side(const DictView &d, int logN, uint64_t *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots, int *overflow,
                       uint16_t *side, uint16_t *perm, hipStream_t st)
{
    size_t lds = (size_t)(table_slots + TABLE_PAD) * 10;
    (void)hipFuncSetAttribute((const void *)union_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((union_kernel<false, true, true>), dim3(1u << logN), dim3(1024), lds, st, d, logN, stage, stride, ncnt, table_slots, overflow, side, perm);
}
void launch_union(const DictView &d, int logN, uint64_t *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots,
                  int *overflow, hipStream_t st)
{
    launch_union_t<false>(d, logN, 1u << logN, stage, stride, ncnt, table_slots, overflow, st);
}
void launch_union_probe(const DictView &d, int logP, int probe, uint32_t *cnt, uint32_t table_slots, int *overflow, hipStream_t st)
{
    launch_union_t<true>(d, logP, (unsigned)probe, nullptr, 0u, cnt, table_slots, overflow, st);
}

// IUPAC letter of a base set; bit i of the set == 2-bit code i (A0 C1 T2 G3), cf. bit_encoding.rs:337-368
// IUPAC letter of a base set as two 64-bit immediates (pure ALU: a __device__ array lookup would be a global gather per cell)
__device__ static inline unsigned char mask2iupac(uint32_t m4)
{
    const uint64_t lo = 0x485957544D43412Dull;      // "-ACMTWYH"
    const uint64_t hi = 0x4E42444B56535247ull;      // "GRSVKDBN"
    return (unsigned char)(((m4 & 8u) ? hi : lo) >> (8u * (m4 & 7u)));
}

#ifndef SKX_ASM_U
#define SKX_ASM_U 6
#endif
#ifndef SKX_ASM_GRP
#define SKX_ASM_GRP 3
#endif
#ifndef SKX_ASM_WPS
#define SKX_ASM_WPS 6
#endif
#ifndef SKX_ASM_PREFETCH
#define SKX_ASM_PREFETCH 0
#endif
// K5: rows = key slabs, columns = samples: fill the sample-major matrix + per-row statistics.
// MODE 0: matrix + statistics (columns land at roff[j] - col_base: a window of sub-buckets can be assembled into a small
//         buffer, which is how a lazily held array is streamed into a .skf without ever existing in full);
// MODE 1: statistics only (the filter of a lazily held array decides from these);
// MODE 2: kept rows only -- keep[] / kpos[] (flags and their exclusive scan over all rows) select the rows, which land at
//         column kpos[row]: the filtered array is written directly, the unfiltered one never is.
template <int MODE>
__global__ __launch_bounds__(512, SKX_ASM_WPS) void assemble_kernel(AssembleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint64_t j = (uint64_t)blockIdx.x + a.j_base;
    const uint32_t n = a.ncnt[j];
    if (n == 0) return;
    const uint32_t maxr = (a.max_rows + 15u) & ~15u;
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(s_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_raw + (size_t)maxr * 8);
    uint16_t *s_map = reinterpret_cast<uint16_t *>(s_cnt);                // MODE 2: row -> kept slot (0xFFFF: dropped); no counts in that mode
    uint32_t *s_msk = s_cnt + maxr;                                   // [maxr/2] 16-bit code sets, two rows per word
    uint32_t *s_idx = s_msk + maxr / 2;                                   // [maxr/2 + 2] row index by the next hash bits
    unsigned char *s_rows = reinterpret_cast<unsigned char *>(s_idx + maxr / 2 + 4);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t *slab = a.stage + j * (uint64_t)a.stride;
    const uint64_t r0 = a.roff[j];
    const uint64_t k0 = MODE == 2 ? a.kpos[r0] : 0;
    const uint32_t nout = MODE == 2 ? (uint32_t)(a.kpos[r0 + n] - k0) : n;      // cells every sample writes for this sub-bucket
    if (MODE == 2 && nout == 0) return;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        s_keys[i] = slab[i] >> 4;
        if (MODE == 2) s_map[i] = a.keep[r0 + i] == 1 ? (uint16_t)(a.kpos[r0 + i] - k0) : (uint16_t)0xFFFF;
        else { s_cnt[i] = 0; if (!(i & 1u)) s_msk[i >> 1] = 0; }
    }
    __syncthreads();
    // direct index: rows are uniform in the sub-bucket's hash range, so the next logI bits of the key select ~2 rows
    const int rem = a.d.bits - a.logN;
    int logI = 31 - __clz(n | 1u);                                    // ~n/2..n index cells
    if (logI > rem) logI = rem;
    while ((1u << logI) > maxr / 2) logI--;
    const uint32_t nidx = 1u << logI;
    const uint64_t lmask = rem >= 60 ? ~0ull : ((1ull << rem) - 1);
    for (uint32_t b = threadIdx.x; b <= nidx; b += blockDim.x) {
        uint32_t l = 0, r = n;
        if (b == nidx) l = n;
        else { const uint64_t x = ((uint64_t)b) << (rem - logI); while (l < r) { uint32_t m = (l + r) >> 1; if ((s_keys[m] & lmask) < x) l = m + 1; else r = m; } }
        s_idx[b] = l;
    }
    __syncthreads();
    const uint64_t ocol = MODE == 2 ? k0 : r0 - a.col_base;           // first output column of this sub-bucket
    const uint32_t shift = (uint32_t)(ocol & 15u);
    unsigned char *row = s_rows + (size_t)wv * (maxr + 32u);      // 16-B aligned; cell i lives at row[shift + i]
    // every lane binary-searches the slice of one sample (64 searches in flight); the wave then handles the slices in turn.
    // Slices are addressed as offsets from the kernel-argument pointer (global address space, no flat loads) and read
    // with clamped, branch-free loads so that 8 words per lane are in flight per memory round trip.
    const uint64_t *wbase = a.d.words;
    const int per = (a.d.n_samples + nw - 1) / nw;                 // samples are dealt evenly to the waves
    const int wend = (wv + 1) * per < a.d.n_samples ? (wv + 1) * per : a.d.n_samples;
    for (int sbase = wv * per; sbase < wend; sbase += 64) {
        uint64_t my_off = 0; uint32_t my_lo = 0, my_hi = 0;
        const int cnt = wend - sbase < 64 ? wend - sbase : 64;
        if (lane < cnt) {
            const uint64_t *my_reg = nullptr;
            sub_slice(a.d, sbase + lane, j, a.logN, my_reg, my_lo, my_hi);
            my_off = (uint64_t)(my_reg - wbase);
        }
        constexpr int AU = SKX_ASM_U, GRP = SKX_ASM_GRP;
        auto fetch = [&](int t, uint64_t (&q)[AU], uint32_t &lo, uint32_t &hi, gwords_t &reg) {
            reg = as_global(wbase + __shfl((unsigned long long)my_off, t, 64));
            lo = __shfl(my_lo, t, 64); hi = __shfl(my_hi, t, 64);
#pragma unroll
            for (int u = 0; u < AU; u++) { const uint32_t i = lo + 64u * u + lane; q[u] = i < hi ? reg[i] : 0ull; }
        };
        // look-ups of GRP words at a time: index cells, then first keys (independent LDS reads: two latencies per group instead
        // of two or more per word); the statistics / cell writes need no return value.  A word is never 0 (non-empty base set).
        auto place = [&](const uint64_t (&q)[AU]) {
#pragma unroll
            for (int h = 0; h < AU; h += GRP) {
                uint32_t l[GRP], le[GRP]; uint64_t kk[GRP];
#pragma unroll
                for (int u = 0; u < GRP; u++) {
                    if (h + u >= AU) continue;
                    const uint32_t ib = word_field<false>(q[h + u], rem - logI + 4, nidx - 1);
                    l[u] = s_idx[ib]; le[u] = s_idx[ib + 1];
                }
#pragma unroll
                for (int u = 0; u < GRP; u++) { if (h + u >= AU) continue; kk[u] = s_keys[l[u]]; }     // (l <= n <= maxr: inside the LDS block)
#pragma unroll
                for (int u = 0; u < GRP; u++) {
                    if (h + u >= AU) continue;
                    const uint64_t w = q[h + u];
                    if (!w) continue;
                    const uint64_t key = w >> 4;
                    uint32_t ll = l[u]; uint64_t k = kk[u];
                    while (ll < le[u] && k < key) { ll++; k = s_keys[ll]; }
                    if (ll < le[u] && k == key) {
                        const uint32_t m4 = (uint32_t)(w & 15u);
                        if (MODE == 0) row[shift + ll] = mask2iupac(m4);
                        if (MODE == 2) { const uint32_t o = s_map[ll]; if (o != 0xFFFFu) row[shift + o] = (a.mask_ambig && (m4 & (m4 - 1))) ? (unsigned char)'N' : mask2iupac(m4); }
                        if (MODE != 2) {
                            const uint32_t single = (m4 & (m4 - 1)) == 0;
                            atomicAdd(&s_cnt[ll], 1u | (single << 16));
                            atomicOr(&s_msk[ll >> 1], (1u << m4) << (16u * (ll & 1u)));
                        }
                    } else {
                        *a.missing = 1;
                    }
                }
            }
        };
        uint64_t nx[AU]; uint32_t nlo, nhi; gwords_t nreg;
        fetch(0, nx, nlo, nhi, nreg);
        for (int t = 0; t < cnt; t++) {
            const int s = sbase + t;
            uint64_t cur[AU];
#pragma unroll
            for (int u = 0; u < AU; u++) cur[u] = nx[u];
            const uint32_t lo = nlo, hi = nhi; gwords_t reg = nreg;
#if SKX_ASM_PREFETCH
            if (t + 1 < cnt) fetch(t + 1, nx, nlo, nhi, nreg);             // the next slice is on its way while this one is placed
#endif
            if (MODE != 1) {
                // fill with '-'
                for (uint32_t i = lane * 4; i < nout + shift + 3; i += 256) *reinterpret_cast<uint32_t *>(row + i) = 0x2D2D2D2Du;
                __builtin_amdgcn_wave_barrier();
            }
            place(cur);
            for (uint32_t o = lo + 64u * AU; o < hi; o += 64u * AU) {          // longer slices: the rest, batch by batch
#pragma unroll
                for (int u = 0; u < AU; u++) { const uint32_t i = o + 64u * u + lane; cur[u] = i < hi ? reg[i] : 0ull; }
                place(cur);
            }
#if !SKX_ASM_PREFETCH
            if (t + 1 < cnt) fetch(t + 1, nx, nlo, nhi, nreg);
#endif
            if (MODE != 1) {
                __builtin_amdgcn_wave_barrier();
                // copy out: output column ocol + i  <-  row[shift + i]; 16-B body, byte head/tail
                unsigned char *dst = a.matrix + (uint64_t)s * a.pitch + ocol;
                const uint32_t head = (16u - shift) & 15u;
                const uint32_t h = head < nout ? head : nout;
                if ((uint32_t)lane < h) dst[lane] = row[shift + lane];
                const uint32_t body = (nout - h) / 16u;
                for (uint32_t v = lane; v < body; v += 64) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(row + shift + h + 16u * v);
                    *reinterpret_cast<uint4 *>(dst + h + 16u * v) = x;
                }
                const uint32_t done = h + body * 16u;
                if (done + lane < nout) dst[done + lane] = row[shift + done + lane];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (MODE == 2) return;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        a.col_present[r0 + i] = s_cnt[i] & 0xFFFFu;
        a.col_unambig[r0 + i] = s_cnt[i] >> 16;
        a.col_mask[r0 + i] = (s_msk[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;
    }
}

// (skx_device_wide.inc) slice [lo, hi) of a sample's region of 16-byte words that belongs to sub-bucket j; returns the region's first word index
__device__ static inline uint64_t sub_slice_off_wide(const DictView &d, int sample, uint64_t j, int logN, uint32_t &lo, uint32_t &hi);
// K5': the matrix from the union's notes (union_kernel<.., SIDE>): per word 2 bytes -- (first-seen rank of its key in the sub-bucket << 4) |
// base set -- instead of the 8-byte word, and no key look-up: perm[rank] is the row.  Same output as assemble_kernel<0>.
template <bool WIDE>                 // WIDE: the dictionaries hold 16-byte words (the notes are the same: 2 bytes per word, indexed by word)
__global__ __launch_bounds__(512, SKX_ASM_WPS) void assemble_side_kernel(AssembleArgs a, const uint16_t *side, const uint16_t *perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint64_t j = (uint64_t)blockIdx.x + a.j_base;
    const uint32_t n = a.ncnt[j];
    if (n == 0) return;
    const uint32_t maxr = (a.max_rows + 15u) & ~15u;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_raw);               // [maxr] present | unambiguous << 16
    uint32_t *s_msk = s_cnt + maxr;                                       // [maxr/2] 16-bit code sets, two rows per word
    uint16_t *s_perm = reinterpret_cast<uint16_t *>(s_msk + maxr / 2);    // [maxr] first-seen rank -> row
    unsigned char *s_rows = reinterpret_cast<unsigned char *>(s_perm + maxr);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint16_t *pj = perm + j * (uint64_t)a.stride;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { s_cnt[i] = 0; if (!(i & 1u)) s_msk[i >> 1] = 0; s_perm[i] = i < a.stride ? pj[i] : (uint16_t)0; }      // (composed notes: n = the global rows of the range may exceed the ranks a perm row holds)
    __syncthreads();
    const uint64_t r0 = a.roff[j];
    const uint64_t ocol = r0 - a.col_base;
    const uint32_t shift = (uint32_t)(ocol & 15u);
    unsigned char *row = s_rows + (size_t)wv * (maxr + 32u);
    typedef const uint16_t __attribute__((address_space(1))) *gside_t;
    gside_t gs = (gside_t)(uintptr_t)side;
    const uint64_t *wbase = a.d.words;
    const int per = (a.d.n_samples + nw - 1) / nw;
    const int wend = (wv + 1) * per < a.d.n_samples ? (wv + 1) * per : a.d.n_samples;
    constexpr int AU = SKX_ASM_U;
    for (int sbase = wv * per; sbase < wend; sbase += 64) {
        uint64_t my_off = 0; uint32_t my_lo = 0, my_hi = 0;
        const int cnt = wend - sbase < 64 ? wend - sbase : 64;
        if (lane < cnt) {
            if (!WIDE) {
                const uint64_t *my_reg = nullptr;
                sub_slice(a.d, sbase + lane, j, a.logN, my_reg, my_lo, my_hi);
                my_off = (uint64_t)(my_reg - wbase);
            } else my_off = sub_slice_off_wide(a.d, sbase + lane, j, a.logN, my_lo, my_hi);
        }
        auto fetch = [&](int t, uint32_t (&q)[AU], uint32_t &lo, uint32_t &hi) {
            const uint64_t off = __shfl((unsigned long long)my_off, t, 64);
            lo = __shfl(my_lo, t, 64); hi = __shfl(my_hi, t, 64);
#pragma unroll
            for (int u = 0; u < AU; u++) { const uint32_t i = lo + 64u * u + lane; q[u] = i < hi ? (uint32_t)gs[off + i] : 0xFFFFFFFFu; }
        };
        auto place = [&](const uint32_t (&q)[AU]) {
            uint32_t rr[AU];
#pragma unroll
            for (int u = 0; u < AU; u++) rr[u] = q[u] != 0xFFFFFFFFu ? (uint32_t)s_perm[(q[u] >> 4) < n ? (q[u] >> 4) : 0u] : 0u;      // independent LDS reads
#pragma unroll
            for (int u = 0; u < AU; u++) {
                if (q[u] == 0xFFFFFFFFu) continue;
                if ((q[u] >> 4) >= n) { *a.missing = 1; continue; }
                const uint32_t ll = rr[u], m4 = q[u] & 15u;
                row[shift + ll] = mask2iupac(m4);
                const uint32_t single = (m4 & (m4 - 1)) == 0;
                atomicAdd(&s_cnt[ll], 1u | (single << 16));
                atomicOr(&s_msk[ll >> 1], (1u << m4) << (16u * (ll & 1u)));
            }
        };
        uint32_t nx[AU]; uint32_t nlo, nhi;
        fetch(0, nx, nlo, nhi);
        for (int t = 0; t < cnt; t++) {
            const int s = sbase + t;
            uint32_t cur[AU];
#pragma unroll
            for (int u = 0; u < AU; u++) cur[u] = nx[u];
            const uint32_t lo = nlo, hi = nhi;
            const uint64_t off = __shfl((unsigned long long)my_off, t, 64);
            for (uint32_t i = lane * 4; i < n + shift + 3; i += 256) *reinterpret_cast<uint32_t *>(row + i) = 0x2D2D2D2Du;
            __builtin_amdgcn_wave_barrier();
            place(cur);
            for (uint32_t o = lo + 64u * AU; o < hi; o += 64u * AU) {          // longer slices: the rest, batch by batch
#pragma unroll
                for (int u = 0; u < AU; u++) { const uint32_t i = o + 64u * u + lane; cur[u] = i < hi ? (uint32_t)gs[off + i] : 0xFFFFFFFFu; }
                place(cur);
            }
            if (t + 1 < cnt) fetch(t + 1, nx, nlo, nhi);
            __builtin_amdgcn_wave_barrier();
            unsigned char *dst = a.matrix + (uint64_t)s * a.pitch + ocol;
            const uint32_t head = (16u - shift) & 15u;
            const uint32_t h = head < n ? head : n;
            if ((uint32_t)lane < h) dst[lane] = row[shift + lane];
            const uint32_t body = (n - h) / 16u;
            for (uint32_t v = lane; v < body; v += 64) {
                const uint4 x = *reinterpret_cast<const uint4 *>(row + shift + h + 16u * v);
                *reinterpret_cast<uint4 *>(dst + h + 16u * v) = x;
            }
            const uint32_t done = h + body * 16u;
            if (done + lane < n) dst[done + lane] = row[shift + done + lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        a.col_present[r0 + i] = s_cnt[i] & 0xFFFFu;
        a.col_unambig[r0 + i] = s_cnt[i] >> 16;
        a.col_mask[r0 + i] = (s_msk[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;
    }
}
void launch_assemble_side(const AssembleArgs &a, const uint16_t *side, const uint16_t *perm, hipStream_t st, bool wide)
{
    const uint32_t maxr = (a.max_rows + 15u) & ~15u;
    const int nw = 8;
    size_t lds = (size_t)maxr * 4 + (size_t)maxr * 2 + (size_t)maxr * 2 + 64 + (size_t)nw * (maxr + 32u);
    if (wide) {
        (void)hipFuncSetAttribute((const void *)assemble_side_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(assemble_side_kernel<true>, dim3((1u << a.logN) - a.j_base), dim3(64 * nw), lds, st, a, side, perm);
        return;
    }
    (void)hipFuncSetAttribute((const void *)assemble_side_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(assemble_side_kernel<false>, dim3((1u << a.logN) - a.j_base), dim3(64 * nw), lds, st, a, side, perm);
}
template <int MODE>
static void launch_assemble_t(const AssembleArgs &a, uint32_t n_blocks, hipStream_t st)
{
    const uint32_t maxr = (a.max_rows + 15u) & ~15u;
    const int nw = 8;
    size_t lds = (size_t)maxr * 14 + ((size_t)maxr / 2 + 4) * 4 + (MODE == 1 ? 0 : (size_t)nw * (maxr + 32u));
    (void)hipFuncSetAttribute((const void *)assemble_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(assemble_kernel<MODE>, dim3(n_blocks), dim3(64 * nw), lds, st, a);
}
// sub-buckets [a.j_base, a.j_base + n_blocks); n_blocks = 0: all of them
void launch_assemble(const AssembleArgs &a, hipStream_t st, int mode, uint32_t n_blocks)
{
    if (!n_blocks) n_blocks = (1u << a.logN) - a.j_base;
    if (mode == 0) launch_assemble_t<0>(a, n_blocks, st); else if (mode == 1) launch_assemble_t<1>(a, n_blocks, st); else launch_assemble_t<2>(a, n_blocks, st);
}

// ------------------------------------------------------------------------------------------------
// key conversions
// ------------------------------------------------------------------------------------------------
__global__ void gather_keys_kernel(const uint64_t *stage, uint32_t stride, const uint32_t *ncnt, const uint64_t *roff, uint64_t *out,
                                   int unhash, HashParams hp)
{
    const uint64_t j = blockIdx.x;
    const uint32_t n = ncnt[j];
    const uint64_t *slab = stage + j * (uint64_t)stride;
    uint64_t *dst = out + roff[j];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = unhash ? hunmix(slab[i] >> 4, hp) : slab[i];
}
void launch_gather_keys(const uint64_t *stage, uint32_t stride, const uint32_t *ncnt, const uint64_t *roff, int n_sub, uint64_t *out,
                        int unhash, HashParams hp, hipStream_t st)
{
    hipLaunchKernelGGL(gather_keys_kernel, dim3((unsigned)n_sub), dim3(256), 0, st, stage, stride, ncnt, roff, out, unhash, hp);
}
// A rank's notes (union_kernel<.., SIDE> over its own dictionaries) against the rows of the whole job: workgroup j = own sub-bucket j (2^l_logN
// of them), whose hash range is covered by 2^(g_logN - l_logN) consecutive global sub-buckets.  Every own row is looked up in the global slab it
// falls into (binary search; the slabs hold the same packed words, base nibble 1); g_perm[j][rank] = that row's place among the range's global
// rows, g_base[j] / g_n[j] = the range's first global row and row count.  *bad: a key missing from the global rows, or a range beyond 65 535 rows.
__global__ __launch_bounds__(256) void compose_perm_kernel(const uint64_t *l_stage, uint32_t l_stride, const uint32_t *l_ncnt, const uint16_t *l_perm, int l_logN,
                                                           const uint64_t *g_stage, uint32_t g_stride, const uint32_t *g_ncnt, const uint64_t *g_roff, int g_logN,
                                                           int bits, uint16_t *g_perm, uint32_t *g_n, uint64_t *g_base, uint32_t *g_max, int *bad)
{
    extern __shared__ uint16_t s_rel[];                                   // [l_stride] own row -> place in the range
    const uint64_t j = blockIdx.x;
    const int d = g_logN - l_logN;
    const uint64_t base = g_roff[j << d], end = g_roff[(j + 1) << d];
    const uint32_t nl = l_ncnt[j];
    if (threadIdx.x == 0) {
        g_base[j] = base; g_n[j] = (uint32_t)(end - base);
        if (end - base > 65535ull) *bad = 1;
        atomicMax(g_max, (uint32_t)(end - base));
    }
    const uint64_t *ls = l_stage + j * (uint64_t)l_stride;
    for (uint32_t r = threadIdx.x; r < nl; r += blockDim.x) {
        const uint64_t key = ls[r] >> 4;
        const uint64_t jg = key >> (bits - g_logN);
        const uint64_t *gs = g_stage + jg * (uint64_t)g_stride;
        uint32_t lo = 0, hi = g_ncnt[jg];
        const uint32_t n = hi;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((gs[mid] >> 4) < key) lo = mid + 1; else hi = mid; }
        if (lo >= n || (gs[lo] >> 4) != key) *bad = 1;
        s_rel[r] = (uint16_t)(g_roff[jg] + lo - base);
    }
    __syncthreads();
    const uint16_t *lp = l_perm + j * (uint64_t)l_stride;
    uint16_t *gp = g_perm + j * (uint64_t)l_stride;
    for (uint32_t r = threadIdx.x; r < nl; r += blockDim.x) { const uint32_t row = lp[r]; gp[r] = row < nl ? s_rel[row] : (uint16_t)0; }
}
void launch_compose_perm(const uint64_t *l_stage, uint32_t l_stride, const uint32_t *l_ncnt, const uint16_t *l_perm, int l_logN, const uint64_t *g_stage,
                         uint32_t g_stride, const uint32_t *g_ncnt, const uint64_t *g_roff, int g_logN, int bits, uint16_t *g_perm, uint32_t *g_n,
                         uint64_t *g_base, uint32_t *g_max, int *bad, hipStream_t st)
{
    hipLaunchKernelGGL(compose_perm_kernel, dim3(1u << l_logN), dim3(256), (size_t)l_stride * 2, st, l_stage, l_stride, l_ncnt, l_perm, l_logN, g_stage, g_stride,
                       g_ncnt, g_roff, g_logN, bits, g_perm, g_n, g_base, g_max, bad);
}
// split k-mers as the .skf stores them: CBOR uints, 0x1b + 8 big-endian bytes each (a key below 2^32 has a shorter minimal form:
// *short_key is set and the host encodes the list instead).  256 keys per workgroup through LDS, written as dwords.
__global__ __launch_bounds__(256) void keys_cbor_kernel(const uint64_t *words, uint64_t n, HashParams hp, uint8_t *out, int *short_key)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_b[256 * 9 + 12];
    const uint64_t i0 = (uint64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
    if (i < n) {
        const uint64_t key = hunmix(words[i] >> 4, hp);
        if (key <= 0xFFFFFFFFull) *short_key = 1;
        uint8_t *p = s_b + 9 * threadIdx.x;
        p[0] = 0x1b;
#pragma unroll
        for (int b = 0; b < 8; b++) p[1 + b] = (uint8_t)(key >> (56 - 8 * b));
    }
    __syncthreads();
    const uint32_t nb = (uint32_t)((n - i0 < 256 ? n - i0 : 256) * 9);
    uint8_t *dst = out + i0 * 9;                                   // 2304 i0: dword aligned
    for (uint32_t v = threadIdx.x; v < (nb + 3) / 4; v += 256) reinterpret_cast<uint32_t *>(dst)[v] = reinterpret_cast<const uint32_t *>(s_b)[v];   // the buffer is padded past 9 n
}
void launch_keys_cbor(const uint64_t *words, uint64_t n, HashParams hp, uint8_t *out, int *short_key, hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(keys_cbor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, n, hp, out, short_key);
}
__global__ void hash_keys_kernel(const uint64_t *keys, uint64_t *words, uint64_t n, HashParams hp)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        words[i] = (hmix(keys[i], hp) << 4) | 1ull;
}
void launch_hash_keys(const uint64_t *keys, uint64_t *words, uint64_t n, HashParams hp, hipStream_t st)
{
    if (!n) return;
    unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(hash_keys_kernel, dim3(g), dim3(256), 0, st, keys, words, n, hp);
}
__global__ void unhash_dict_kernel(const uint64_t *words, uint64_t n, uint64_t *keys, uint8_t *bases, HashParams hp)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        keys[i] = hunmix(words[i] >> 4, hp);
        bases[i] = mask2iupac((uint32_t)words[i] & 15u);
    }
}
void launch_unhash_dict(const uint64_t *words, uint64_t n, uint64_t *keys, uint8_t *bases, HashParams hp, hipStream_t st)
{
    if (!n) return;
    unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(unhash_dict_kernel, dim3(g), dim3(256), 0, st, words, n, keys, bases, hp);
}

}  // namespace skx
#include "skx_device2.inc"
#include "skx_device_wide.inc"
