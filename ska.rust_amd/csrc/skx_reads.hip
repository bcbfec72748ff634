// skx_reads.hip -- the FASTQ side of SkaDict::new (ska_dict.rs:118-180 with is_reads): quality gates of
// split_kmer.rs:66-71,328-339, ntHash of the whole k-mer (nthash.rs:35-76) and an exact, order-free evaluation of
// KmerFilter::filter (bloom_filter.rs:62-148) -- blocked bloom (5-bit fingerprint in one of 3 145 728 u64 words) in front
// of a HashMap<u64,u16> that passes a k-mer exactly when its count reaches min_count.
//
// The reference's filter is sequential and order dependent only through bloom false positives.  Per distinct hash h
// (SURVEY.md A.6), with occurrences t_1 < t_2 < ... in stream order (file 1 then file 2):
//   FP(h)  <=>  fp(h) is a subset of OR{ fp(g) : loc(g) == loc(h), first(g) < first(h) }
//   min_count == 2 : every occurrence passes except t_1, which passes iff FP(h)
//   min_count >= 3 : exactly the (min_count - FP(h))-th occurrence passes
// so the device evaluates it with a stable sort by hash, a sort of the distinct hashes by (bloom word, first
// occurrence) and a segmented prefix-OR.  Sorts, scans and selections are the engine's own primitives (skx_prims.hip; rocPRIM calls
// until round 5); everything specific to the path is below.
#include <cstring>
#include "skx_internal.h"

namespace skx {

__device__ static const uint64_t NT_H[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x295549f54be24456ull, 0x20323ed082572324ull};
__device__ static const uint64_t NT_RC[4] = {0x295549f54be24456ull, 0x20323ed082572324ull, 0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull};
__device__ static inline uint64_t rotl64d(uint64_t x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }

struct ReadsArgs {
    const uint8_t *seq, *qual; uint64_t len;
    int k, rc, min_qual, qual_filter;
    HashParams hp; WideHash wh;
    uint64_t *hash; uint64_t *wlo; uint64_t *whi; uint8_t *flag;
    unsigned long long *n_valid;      // optional: [256] counters, slot blockIdx % 256 += windows that pass the gates (one hot counter costs 2 ms)
    const uint64_t *planes;           // optional: the sample as packed bit planes (skx_device.h planes_bytes16) instead of seq / qual
    uint16_t *rec_t; uint32_t *tile_cnt;      // WORDS = false: the tile's gated windows leave compacted -- hash[tile * RW_TILE + r], rec_t[..] = position inside the tile, tile_cnt[tile] of them
};

// Sixteen consecutive window-end positions per thread (two batches of eight).
//   * the tile's text never lies in LDS as bytes: whoever loads 16 positions (16 + 16 bytes, or a slice of the packed planes) boils them down
//     to 32 code bits and three 16-bit masks -- rejected base, line end, quality below the threshold -- and stores those 10 bytes.  Everything a
//     thread needs afterwards is a few words of these arrays at uniform offsets from its own index: the 64 positions before its first (window
//     state, clean run), its sixteen, the bases that leave its windows (k back), the middle bases' verdicts (h back).  (The byte form cost four
//     dependent LDS round trips per position: base, leaving base, their table words, quality -- half the kernel's time was waiting for them.)
//   * the state of the window before the first position is built from its k codes with a table of pre-rotated ntHash words; the rest roll, one
//     16-byte table entry per position: [leaving base][entering base] -> what the two strands' hashes are XORed with.
//   * the workgroup's 2 048 hashes per batch go through one LDS buffer position-major (row stride 264: conflict-free both ways) and leave as
//     contiguous 16 KB pieces -- a per-thread store of its own 64 bytes would cost one L2 request per 8 bytes; the flags leave as one
//     16-byte store per thread.
#ifndef SKX_RW_NB
#define SKX_RW_NB 2
#endif
constexpr int RW_PPT = 8, RW_NB = SKX_RW_NB, RW_NT = 256, RW_TILE = RW_PPT * RW_NB * RW_NT, RW_STRIDE = RW_NT + 8;
constexpr int RW_CH = (RW_TILE + 80) / 16;                  // chunks of 16 positions of a tile: [p0 - 64, p0 + RW_TILE + 16)
static_assert(RW_PPT * RW_NB == 16 && RW_CH >= RW_NT + 5, "a thread's positions are one chunk; it reads chunks tid .. tid + 5");
__device__ static inline uint32_t zero_bytes(uint32_t x)    // 0x80 in every byte of x that is zero (exact: no borrow between bytes)
{
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}
__device__ static inline uint32_t flags4(uint32_t f)        // bits 7, 15, 23, 31 -> bits 0..3
{
    return (((f >> 7) * 0x00204081u) >> 21) & 0xFu;
}
__device__ static inline uint32_t codes4(uint32_t d)        // bytes b0..b3 -> their codes ((b >> 1) & 3) at bits 0-1 .. 6-7
{
    uint32_t c = (d >> 1) & 0x03030303u;
    c |= c >> 6;
    return (c | (c >> 12)) & 0xFFu;
}
__device__ static inline uint32_t spread16(uint32_t x)      // bit j -> bit 2 j
{
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u;
    return (x | (x << 1)) & 0x55555555u;
}
// 16 sequence bytes + 16 quality bytes -> codes, N mask ((b & 15) == 14: split_kmer.rs:66-71's rejects as the engine's text holds them), line
// ends, quality verdicts ((q - 33) as a byte <= min_qual, i.e. NOT (q - 33 > min_qual): split_kmer.rs:328-339)
__device__ static inline void pack_bytes16(const uint32_t sw[4], const uint32_t qw[4], bool has_q, uint32_t mq4, uint32_t &code, uint32_t &nb, uint32_t &nl, uint32_t &qb)
{
    constexpr uint32_t H = 0x80808080u;
    code = 0; nb = 0; nl = 0; qb = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t w = sw[i], q = qw[i];
        code |= codes4(w) << (8 * i);
        nb |= flags4(zero_bytes((w & 0x0F0F0F0Fu) ^ 0x0E0E0E0Eu)) << (4 * i);
        nl |= flags4(zero_bytes(w ^ 0x0A0A0A0Au)) << (4 * i);
        if (has_q) {
            const uint32_t r = (q | H) - 0x21212121u;                          // per byte, no borrow: low 7 bits of q - 33; bit 7 = (low 7 of q >= 33)
            const uint32_t y = (r & ~H) | (~(q ^ r) & H);                       // q - 33 as a byte
            const uint32_t t = (mq4 | H) - (y & ~H);                            // bit 7 = (low 7 of min_qual >= low 7 of y)
            qb |= flags4(((~y & mq4) | (~(y ^ mq4) & t)) & H) << (4 * i);       // y <= min_qual
        }
    }
}
// WORDS = false: ntHash and the gates only (the count filter passes ~2 % of a deep read set's windows: their packed words are put together
// afterwards from the text, words_rebuild_kernel) -- 9 instead of 25 bytes written per window, no canonical compare, no key hash
template <bool WORDS>
__global__ __launch_bounds__(RW_NT) void reads_windows_kernel(ReadsArgs a)
{
    __shared__ uint32_t s_code[RW_CH + 3];
    __shared__ uint16_t s_bad[RW_CH + 3], s_nl[RW_CH + 3], s_qb[RW_CH + 3];
    __shared__ __attribute__((aligned(16))) uint64_t s_out[RW_PPT * RW_STRIDE];
    __shared__ uint16_t s_off[WORDS ? 1 : RW_PPT * RW_NT];               // (compact form) the position inside the tile of a batch's gated windows
    __shared__ uint32_t s_ws[RW_NT / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint64_t s_T[32];            // [2 (4 leaving + entering)] = rotl(H[leaving], k) ^ H[entering], [.. + 1] = rotr(R[leaving], 1) ^ rotl(R[entering], k - 1)
    const uint32_t tid = threadIdx.x;
    const int k = a.k, h = (k - 1) / 2;
    if (tid < 16) {
        const int cout = (int)tid >> 2, c = (int)tid & 3;
        s_T[2 * tid] = rotl64d(NT_H[cout], (unsigned)k) ^ NT_H[c];
        s_T[2 * tid + 1] = rotl64d(NT_RC[cout], 63u) ^ rotl64d(NT_RC[c], (unsigned)(k - 1));
    }
    // Before the first put() the output buffer holds ntHash's words as the FIRST window needs them: [4 i + c] = rotl(H[c], k - 1 - i) and
    // [256 + 4 i + c] = rotl(R[c], i) for the window's i-th base
    uint64_t *s_rot = s_out;
    static_assert(RW_PPT * RW_STRIDE >= 1024, "the start-up tables fit the output buffer");
    if constexpr (WORDS) {
        for (int i = (int)tid; i < 4 * k; i += RW_NT) {
            s_rot[i] = rotl64d(NT_H[i & 3], (unsigned)(k - 1 - (i >> 2)));
            s_rot[256 + i] = rotl64d(NT_RC[i & 3], (unsigned)(i >> 2));
        }
    } else {
        // hashes only: two bases a look-up -- [16 ip + c0 + 4 c1] = what bases 2 ip (c0) and 2 ip + 1 (c1) add to the forward hash, [512 + ..] to the
        // reverse strand's; an odd k's last base stands alone (every c1 gives the same)
        for (int i = (int)tid; i < 16 * ((k + 1) / 2); i += RW_NT) {
            const int ip = i >> 4, c0 = i & 3, c1 = (i >> 2) & 3, b0 = 2 * ip, b1 = 2 * ip + 1;
            uint64_t f = rotl64d(NT_H[c0], (unsigned)(k - 1 - b0)), r = rotl64d(NT_RC[c0], (unsigned)b0);
            if (b1 < k) { f ^= rotl64d(NT_H[c1], (unsigned)(k - 1 - b1)); r ^= rotl64d(NT_RC[c1], (unsigned)b1); }
            s_rot[i] = f; s_rot[512 + i] = r;
        }
    }
    uint64_t o_hash[RW_PPT], o_lo[WORDS ? RW_PPT : 1], o_hi[WORDS ? RW_PPT : 1];
    const uint64_t p0 = (uint64_t)blockIdx.x * RW_TILE;
    const bool has_q = a.qual != nullptr || a.planes != nullptr;
    const bool strict = a.qual_filter == 2, midq = a.qual_filter != 0;
    const uint32_t mq = (uint32_t)(uint8_t)a.min_qual, mq4 = mq * 0x01010101u;
    // the tile: chunk c = positions [p0 - 64 + 16 c, + 16)  (p0 - 64 is a multiple of 16 and the streams are 16-byte aligned)
    for (int c = (int)tid; c < RW_CH; c += RW_NT) {
        const int64_t pos = (int64_t)p0 - 64 + 16 * c;
        uint32_t code, nb, nl, qb;
        if (a.planes) {                                            // the packed planes hold these masks already (fastx.cpp pack_*_planes; the bytes planes_bytes16 stands for)
            if (pos >= 0 && (uint64_t)pos < a.len) {
                const uint64_t t = (uint64_t)pos / 16;
                const uint64_t *g = a.planes + (t >> 2) * 5;
                const int sh = (int)(t & 3) * 16;
                const uint32_t lo = (uint32_t)(g[0] >> sh) & 0xFFFFu, hi = (uint32_t)(g[1] >> sh) & 0xFFFFu, bd = (uint32_t)(g[2] >> sh) & 0xFFFFu,
                               qv = (uint32_t)(g[4] >> sh) & 0xFFFFu;
                const uint64_t room = a.len - (uint64_t)pos;
                nl = ((uint32_t)(g[3] >> sh) & 0xFFFFu) | (room < 16 ? (0xFFFFu << room) & 0xFFFFu : 0u);      // positions from `len` on are line ends
                nb = bd & ~nl;
                qb = ~nl & (qv | (mq == 255u ? 0xFFFFu : 0u));     // the verdict of the bytes ' ' / '!' that stand for a passing / failing quality
                code = spread16(nl | bd | lo) | (spread16(~nl & (bd | hi)) << 1);       // '\n' -> code 1, 'N' -> 3, as the bytes give
            } else { code = 0x55555555u; nb = 0; nl = 0xFFFFu; qb = 0; }
        } else {
            uint32_t sw[4], qw[4];
            if (pos >= 0 && (uint64_t)pos + 16 <= a.len) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(a.seq + pos);
                sw[0] = sv.x; sw[1] = sv.y; sw[2] = sv.z; sw[3] = sv.w;
                if (a.qual) { const uint4 qv = *reinterpret_cast<const uint4 *>(a.qual + pos); qw[0] = qv.x; qw[1] = qv.y; qw[2] = qv.z; qw[3] = qv.w; }
                else { qw[0] = qw[1] = qw[2] = qw[3] = 0x7E7E7E7Eu; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint32_t x = 0, y = 0;
                    for (int bb = 0; bb < 4; bb++) {
                        const int64_t q = pos + 4 * i + bb;
                        const bool in = q >= 0 && (uint64_t)q < a.len;
                        x |= (uint32_t)(in ? a.seq[q] : (uint8_t)'\n') << (8 * bb);
                        y |= (uint32_t)((in && a.qual) ? a.qual[q] : (uint8_t)'~') << (8 * bb);
                    }
                    sw[i] = x; qw[i] = y;
                }
            }
            pack_bytes16(sw, qw, has_q, mq4, code, nb, nl, qb);
        }
        s_code[c] = code; s_nl[c] = (uint16_t)nl; s_qb[c] = (uint16_t)(midq ? qb : 0u);
        s_bad[c] = (uint16_t)(nb | nl | (strict ? qb : 0u));
    }
    __syncthreads();
    const uint64_t pstart = p0 + (uint64_t)tid * (RW_PPT * RW_NB);
    const uint64_t left = a.len - p0 < (uint64_t)RW_TILE ? a.len - p0 : (uint64_t)RW_TILE;
    uint64_t upper = 0, lower = 0, rc_upper = 0, rc_lower = 0, fh = 0, rh = 0;
    uint32_t mid = 0, rc_mid = 0, run = 0;
    const uint64_t am = (1ull << (2 * h)) - 1;        // arm mask (h <= 31)
    const bool userc = a.rc != 0;
    // k > 31: the hashed upper arm L starts at bit sh = hb + 4 of the 128-bit word (36..66): its part of the low half = (L << wa1) & wm1, of
    // the high half = (L >> wa2) << wa3 (wave-uniform counts, one formula on both sides of 64)
    const int wsh = a.wh.hb + 4;
    const int wa1 = wsh < 64 ? wsh : 0, wa2 = wsh < 64 ? 64 - wsh : 0, wa3 = wsh < 64 ? 0 : wsh - 64;
    const uint64_t wm1 = wsh < 64 ? ~0ull : 0ull;
    // with r = position - (p0 - 64 + 16 tid): r = 0..63 is the history, 64..79 my positions.  Words at uniform offsets:
    uint32_t cw = s_code[tid + 4];                                             // my codes
    uint32_t coutw, badw = s_bad[tid + 4], nlw, qbw;
    {
        const int ro = 64 - k, wi = ro >> 4;                                   // the base that leaves the window ending at r is r - k
        coutw = __funnelshift_r(s_code[tid + wi], s_code[tid + wi + 1], 2 * (ro & 15));
        nlw = ((uint32_t)s_nl[tid + 4] >> 1) | (((uint32_t)s_nl[tid + 5] & 1u) << 15);      // is r + 1 a line end
        const int rq = 64 - h, wq = rq >> 4;                                   // the middle base of the window ending at r is r - h
        qbw = ((((uint32_t)s_qb[tid + wq]) | ((uint32_t)s_qb[tid + wq + 1] << 16)) >> (rq & 15)) & 0xFFFFu;
    }
    if (pstart < a.len) {
        // the window ending one position before my first (r = 63): its i-th base is r = 64 - k + i
        const uint64_t cl = (uint64_t)s_code[tid] | ((uint64_t)s_code[tid + 1] << 32), ch = (uint64_t)s_code[tid + 2] | ((uint64_t)s_code[tid + 3] << 32);
        const int sb = 2 * (64 - k);                                           // 2 .. 126
        const uint64_t wl = sb < 64 ? (cl >> sb) | (ch << (64 - sb)) : ch >> (sb - 64), wh = sb < 64 ? ch >> sb : 0ull;      // base i at bits 2 i, 2 i + 1
        auto wbase = [&](int i) -> uint32_t { return (uint32_t)((i < 32 ? wl >> (2 * i) : wh >> (2 * (i - 32))) & 3ull); };
        uint64_t w = wl;
        if constexpr (WORDS) {
            for (int i = 0; i < k; i++) {
                if (i == 32) w = wh;
                const uint32_t c = (uint32_t)w & 3u; w >>= 2;
                if (i < h) upper = (upper << 2) | c; else if (i == h) mid = c; else lower = (lower << 2) | c;
                fh ^= s_rot[4 * i + (int)c];
                rh ^= s_rot[256 + 4 * i + (int)c];
            }
        } else {
            for (int ip = 0; ip < (k + 1) / 2; ip++) {                          // (the bits above the window's last base are 0)
                if (ip == 16) w = wh;
                const int at = 16 * ip + (int)((uint32_t)w & 15u); w >>= 4;
                fh ^= s_rot[at];
                rh ^= s_rot[512 + at];
            }
        }
        if (WORDS) {
            for (int i = 0; i < h; i++) {
                rc_upper = (rc_upper << 2) | (wbase(k - 1 - i) ^ 2u);           // reverse complement of the lower arm
                rc_lower = (rc_lower << 2) | (wbase(h - 1 - i) ^ 2u);           // ... of the upper arm
            }
            rc_mid = mid ^ 2u;
        }
        // clean bases ending at r = 63, counted up to k + 1
        const uint64_t hb = (uint64_t)s_bad[tid] | ((uint64_t)s_bad[tid + 1] << 16) | ((uint64_t)s_bad[tid + 2] << 32) | ((uint64_t)s_bad[tid + 3] << 48);
        run = hb ? (uint32_t)__clzll((long long)hb) : 64u;
        run = min(run, (uint32_t)k + 1u);
    }
    __syncthreads();                                       // (the start-up table is done with: put() writes there)
    auto put = [&](const uint64_t (&v)[RW_PPT], uint64_t *dst, int b, bool mine) {       // my eight values -> LDS -> the array, in 64-byte runs
        if (mine) {
#pragma unroll
            for (int j = 0; j < RW_PPT; j++) s_out[j * RW_STRIDE + (int)tid] = v[j];
        }
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)(RW_PPT * RW_NT); i += RW_NT) {
            const uint32_t t = i / RW_PPT, j = i % RW_PPT, at = t * (RW_PPT * RW_NB) + (uint32_t)b * RW_PPT + j;
            if (at < left) dst[p0 + at] = s_out[(int)j * RW_STRIDE + (int)t];
        }
        __syncthreads();
    };
    (void)put;
    uint32_t flags = 0;                                    // bit j: my j-th window passes the gates
    uint32_t n_out = 0;                                    // (compact form) records of this tile written so far
#pragma unroll 1
    for (int b = 0; b < RW_NB; b++) {
    const bool mine = pstart + (uint64_t)b * RW_PPT < a.len;
    if (mine) {
        uint32_t fb = 0;
#pragma unroll
        for (int j = 0; j < RW_PPT; j++) {
            const uint32_t c = (cw >> (2 * j)) & 3u, cout = (coutw >> (2 * j)) & 3u;
            const uint4 tw = *reinterpret_cast<const uint4 *>(&s_T[2 * (4 * cout + c)]);
            // roll_fwd (split_kmer.rs:199-213)
            if (WORDS) {
                upper = ((upper << 2) | mid) & am;
                mid = (uint32_t)(lower >> (2 * h - 2));
                lower = ((lower << 2) | c) & am;
                rc_lower = (rc_lower >> 2) | ((uint64_t)rc_mid << (2 * h - 2));
                rc_mid = mid ^ 2u;
                rc_upper = (rc_upper >> 2) | ((uint64_t)(c ^ 2u) << (2 * h - 2));
            }
            // ntHash of the whole k-mer, both strands (nthash.rs:35-76)
            fh = ((fh << 1) | (fh >> 63)) ^ ((uint64_t)tw.x | ((uint64_t)tw.y << 32));
            rh = ((rh >> 1) | (rh << 63)) ^ ((uint64_t)tw.z | ((uint64_t)tw.w << 32));
            run = ((badw >> j) & 1u) ? 0u : min(run + 1u, (uint32_t)k + 1u);
            // split_kmer.rs:89,121: a clean run of exactly k ending at the record's last base is never started
            bool valid = run >= (uint32_t)k;
            if ((nlw >> j) & 1u) valid = valid && run > (uint32_t)k;
            // middle_base_qual (split_kmer.rs:328-339): Middle and Strict gate on the middle base
            const bool midq_ok = !((qbw >> j) & 1u);
            o_hash[j] = userc ? (fh < rh ? fh : rh) : fh;
            fb |= (valid && midq_ok ? 1u : 0u) << j;                                 // the `&&` of ska_dict.rs:155-157: the count filter is not touched otherwise
            if (!WORDS) continue;
            // canonical strand and base set without branches; the packed word on 64-bit halves (as extract_wide_kernel)
            const bool ueq = upper == rc_upper;
            const bool gt = userc & ((upper > rc_upper) | (ueq & (lower > rc_lower)));
            const bool eq = userc & ueq & (lower == rc_lower);
            const uint64_t hl = gt ? rc_upper : upper, hr = gt ? rc_lower : lower;
            const uint32_t m4 = (1u << (gt ? rc_mid : mid)) | (eq ? (1u << rc_mid) : 0u);
            uint64_t wlo, whi;
            if (k <= 31) { uint32_t L = (uint32_t)hl, R = (uint32_t)hr; hmix_halves(L, R, a.hp); wlo = ((uint64_t)L << (a.hp.hb + 4)) | ((uint64_t)R << 4) | m4; whi = 0; }
            else { uint64_t L = hl, R = hr; hmix_halves_w(L, R, a.wh); wlo = (R << 4) | m4 | ((L << wa1) & wm1); whi = (R >> 60) | ((L >> wa2) << wa3); }
            o_lo[WORDS ? j : 0] = wlo;
            o_hi[WORDS ? j : 0] = whi;
        }
        flags |= fb << (RW_PPT * b);
        cw >>= 2 * RW_PPT; coutw >>= 2 * RW_PPT; badw >>= RW_PPT; nlw >>= RW_PPT; qbw >>= RW_PPT;
    }
    if constexpr (WORDS) {
        put(o_hash, a.hash, b, mine);
        put(o_lo, a.wlo, b, mine);
        if (a.whi) put(o_hi, a.whi, b, mine);
    } else {
        // only the gated windows leave (47 % of a deep isolate's positions), as (hash, position in the tile) in the order of the threads: the partition
        // pass reads these instead of a hash and a flag per position
        const uint32_t at_b = tid * (RW_PPT * RW_NB) + (uint32_t)b * RW_PPT;
        uint32_t fb = mine ? (flags >> (RW_PPT * b)) & ((1u << RW_PPT) - 1u) : 0u;
        if (at_b + RW_PPT > (uint32_t)left) fb &= at_b < (uint32_t)left ? (1u << ((uint32_t)left - at_b)) - 1u : 0u;
        const uint32_t c = (uint32_t)__popc(fb);
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if ((int)(tid & 63u) >= d) inc += y; }
        if ((tid & 63u) == 63u) s_ws[tid >> 6] = inc;
        __syncthreads();
        uint32_t ex = inc - c, total = 0;
#pragma unroll
        for (int w = 0; w < RW_NT / 64; w++) { const uint32_t v = s_ws[w]; if (w < (int)(tid >> 6)) ex += v; total += v; }
#pragma unroll
        for (int j = 0; j < RW_PPT; j++)
            if ((fb >> j) & 1u) { s_out[ex] = o_hash[j]; s_off[ex] = (uint16_t)(at_b + j); ex++; }
        __syncthreads();
        const uint64_t tb = p0 + n_out;                                     // (a tile's slots are its RW_TILE positions)
        for (uint32_t i = tid; i < total; i += RW_NT) { a.hash[tb + i] = s_out[i]; a.rec_t[tb + i] = s_off[i]; }
        n_out += total;
        __syncthreads();
    }
    }
    if constexpr (!WORDS) {
        if (tid == 0) { a.tile_cnt[blockIdx.x] = n_out; if (a.n_valid && n_out) atomicAdd(a.n_valid + (blockIdx.x & 255u), (unsigned long long)n_out); }
        return;
    }
    // my sixteen flags: one byte each, 16 bytes at a multiple of 16
    const uint32_t at0 = tid * (RW_PPT * RW_NB);
    uint32_t nv = 0;
    if (at0 < left) {
        const uint32_t room = (uint32_t)left - at0;
        if (room < 16u) flags &= (1u << room) - 1u;
        nv = (uint32_t)__popc(flags);
        uint32_t fw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) fw[i] = (((flags >> (4 * i)) & 0xFu) * 0x00204081u) & 0x01010101u;
        if (room >= 16u) *reinterpret_cast<uint4 *>(a.flag + p0 + at0) = make_uint4(fw[0], fw[1], fw[2], fw[3]);
        else for (uint32_t i = 0; i < room; i++) a.flag[p0 + at0 + i] = (uint8_t)((flags >> i) & 1u);
    }
    if (a.n_valid) {
        for (int d = 32; d; d >>= 1) nv += __shfl_down(nv, d, 64);
        if ((tid & 63u) == 0 && nv) atomicAdd(a.n_valid + (blockIdx.x & 255u), (unsigned long long)nv);
    }
}

__global__ void gather_u64_kernel(const uint64_t *src, const uint32_t *idx, uint64_t *dst, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void iota_u32_kernel(uint32_t *v, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
// head[i] = first occurrence of its hash in the hash-sorted order; startpos[i] = i at heads else 0 (for a max-scan)
__global__ void heads_kernel(const uint64_t *hs, uint32_t *head, uint32_t *startpos, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool hd = i == 0 || hs[i] != hs[i - 1];
        head[i] = hd; startpos[i] = hd ? (uint32_t)i : 0u;
    }
}
// bloom word and fingerprint (bloom_filter.rs:50-74): loc = (cheap_mix(h) * buf_size) >> 64, fp = 5 bits of h
__device__ static inline uint64_t bloom_loc(uint64_t h) { const uint64_t m = (h ^ (h >> 31)) * 0x85D059AA333121CFull; return (uint64_t)(((u128)m * 3145728ull) >> 64); }
__device__ static inline uint64_t bloom_fp(uint64_t h)
{
    return (1ull << (h & 63)) | (1ull << ((h >> 6) & 63)) | (1ull << ((h >> 12) & 63)) | (1ull << ((h >> 18) & 63)) | (1ull << ((h >> 24) & 63));
}
// per distinct hash (group id = gsum[i]-1 at its head): composite key (loc << 32 | first occurrence) and fingerprint
__global__ void distinct_kernel(const uint64_t *hs, const uint32_t *ts, const uint32_t *head, const uint32_t *gsum, uint64_t *ckey, uint32_t *gidx,
                                uint64_t *dhash, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (head[i]) { const uint32_t g = gsum[i] - 1; ckey[g] = (bloom_loc(hs[i]) << 32) | ts[i]; gidx[g] = g; dhash[g] = hs[i]; }
}
__global__ void fp_prepare_kernel(const uint64_t *cks, const uint32_t *gs, const uint64_t *dhash, uint32_t *lockey, uint64_t *fpv, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        lockey[i] = (uint32_t)(cks[i] >> 32); fpv[i] = bloom_fp(dhash[gs[i]]);
    }
}
__global__ void fp_decide_kernel(const uint32_t *gs, const uint64_t *fpv, const uint64_t *prev, uint8_t *FP, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        FP[gs[i]] = (fpv[i] & ~prev[i]) == 0;
}
// KmerFilter::filter verdict per occurrence (occurrence number j is 1-based inside its hash group)
__global__ void accept_kernel(const uint32_t *gsum, const uint32_t *start, const uint8_t *FP, uint8_t *acc, int min_count, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)i - start[i] + 1, fpos = FP[gsum[i] - 1];
        acc[i] = min_count == 2 ? (j >= 2 || fpos) : (j == (uint32_t)min_count - fpos);
    }
}
// sorted words -> unique words with OR-ed masks
__global__ void word_heads_kernel(const uint64_t *lo, const uint64_t *hi, uint32_t *head, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        head[i] = i == 0 || (lo[i] >> 4) != (lo[i - 1] >> 4) || (hi && hi[i] != hi[i - 1]);
}
__global__ void word_fold_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *head, const uint32_t *hsum, uint64_t *out, int wpk, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        uint64_t w = lo[i];
        for (uint64_t j = i + 1; j < n && !head[j]; j++) w |= lo[j] & 15ull;
        const uint64_t o = hsum[i] - 1;
        out[o * wpk] = w;
        if (wpk == 2) out[o * 2 + 1] = hi[i];
    }
}

struct BitOr64 { __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a | b; } };

namespace {
inline unsigned grid_for(uint64_t n) { uint64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }
struct Temp { };          // (the primitives of skx_prims.hip hold their own scratch)
#define RP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

inline int sort_pairs(Temp &, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, hipStream_t st)
{
    return prim_sort_pairs_u64(kin, kout, vin, vout, n, 64, st);                  // stable: equal keys keep their order
}
}  // namespace

// The packed words of the windows at the given end positions (all of them clean runs of k bases: they passed the gates), put together from
// the text as reads_windows_kernel<true> rolls them: arms, canonical strand, base set (split_kmer.rs:141-217), then the key hash.
// The window's bytes arrive as dwords from the 4-byte boundary below its first base (up to 17 of them; byte loads -- 63 a window -- took 0.48 ms
// per 5 M windows), every dword's four 2-bit codes are squeezed into a byte, and the bytes form P: base j of the window at bits 2j, 2j + 1.
// Then: reverse complement's lower arm = P's first h pairs complemented, the upper arm = the same pairs in reverse order; likewise the other arm.
__device__ static inline uint64_t pairs_reversed(uint64_t x)             // the 32 two-bit fields of x in reverse order
{
    const uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}
__device__ static inline uint32_t codes_of_dword(uint32_t d)             // ASCII bases b0..b3 -> their codes ((b >> 1) & 3) at bits 0-1 .. 6-7
{
    uint32_t c = (d >> 1) & 0x03030303u;
    c |= c >> 6;
    return (c | (c >> 12)) & 0xFFu;
}
__global__ __launch_bounds__(256) void words_rebuild_kernel(const uint32_t *pos, uint64_t n, const uint8_t *seq, uint64_t len, int k, int rc, HashParams hp, WideHash wh,
                                                             uint64_t *out_lo, uint64_t *out_hi)
{
    const int h = (k - 1) / 2;
    const uint64_t am = (1ull << (2 * h)) - 1;
    const int wsh = wh.hb + 4;
    const int wa1 = wsh < 64 ? wsh : 0, wa2 = wsh < 64 ? 64 - wsh : 0, wa3 = wsh < 64 ? 0 : wsh - 64;
    const uint64_t wm1 = wsh < 64 ? ~0ull : 0ull;
    const bool more = k + 3 > 48;                                        // dwords 12..16 are needed
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t first = (uint64_t)pos[i] - (uint64_t)(k - 1);
        const uint64_t a0 = first - ((uint64_t)(uintptr_t)(seq + first) & 3ull);                     // (may be "-1..-3" for a stream that starts off a boundary: then the byte path)
        uint64_t P0 = 0, P1 = 0, P2 = 0;                                 // the codes from a0 on: 4 bases a byte, 136 bits
        if (a0 + 68 <= len && a0 <= first) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(seq + a0);
            uint32_t d[17];
#pragma unroll
            for (int j = 0; j < 12; j++) d[j] = w[j];
#pragma unroll
            for (int j = 12; j < 17; j++) d[j] = more ? w[j] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) P0 |= (uint64_t)codes_of_dword(d[j]) << (8 * j);
#pragma unroll
            for (int j = 8; j < 16; j++) P1 |= (uint64_t)codes_of_dword(d[j]) << (8 * (j - 8));
            P2 = codes_of_dword(d[16]);
        } else {                                                          // the stream's last bytes: one by one
            for (uint64_t q = first; q < first + (uint64_t)k; q++) {
                const uint64_t c = (seq[q] >> 1) & 3u; const unsigned at = 2u * (unsigned)(q - first + (first - a0));
                if (at < 64) P0 |= c << at; else if (at < 128) P1 |= c << (at - 64); else P2 |= c << (at - 128);
            }
        }
        const unsigned s0 = 2u * (unsigned)(first - a0);                 // 0, 2, 4 or 6: the window starts there
        if (s0) { P0 = (P0 >> s0) | (P1 << (64 - s0)); P1 = (P1 >> s0) | (P2 << (64 - s0)); }
        const unsigned sq = 2u * (unsigned)(h + 1);                      // the lower arm starts at bit 2 (h + 1) <= 64
        const uint64_t up = P0 & am, lo = (sq < 64 ? (P0 >> sq) | (P1 << (64 - sq)) : P1) & am;
        const uint32_t mid = (uint32_t)(2 * h < 64 ? (P0 >> (2 * h)) | (P1 << (64 - 2 * h)) : P1) & 3u, rc_mid = mid ^ 2u;
        const uint64_t comp = 0xAAAAAAAAAAAAAAAAull & am;
        const uint64_t upper = pairs_reversed(up) >> (64 - 2 * h), lower = pairs_reversed(lo) >> (64 - 2 * h);
        const uint64_t rc_lower = up ^ comp, rc_upper = lo ^ comp;
        const bool userc = rc != 0;
        const bool ueq = upper == rc_upper;
        const bool gt = userc & ((upper > rc_upper) | (ueq & (lower > rc_lower)));
        const bool eq = userc & ueq & (lower == rc_lower);
        const uint64_t hl = gt ? rc_upper : upper, hr = gt ? rc_lower : lower;
        const uint32_t m4 = (1u << (gt ? rc_mid : mid)) | (eq ? (1u << rc_mid) : 0u);
        if (k <= 31) { uint32_t L = (uint32_t)hl, R = (uint32_t)hr; hmix_halves(L, R, hp); out_lo[i] = ((uint64_t)L << (hp.hb + 4)) | ((uint64_t)R << 4) | m4; }
        else { uint64_t L = hl, R = hr; hmix_halves_w(L, R, wh); out_lo[i] = (R << 4) | m4 | ((L << wa1) & wm1); out_hi[i] = (R >> 60) | ((L >> wa2) << wa3); }
    }
}
// the same from the packed planes: a window's code bits are two 64-bit pieces of the lo / hi planes (the window may straddle two groups)
__global__ __launch_bounds__(256) void words_rebuild_planes_kernel(const uint32_t *pos, uint64_t n, const uint64_t *planes, int k, int rc, HashParams hp, WideHash wh,
                                                                    uint64_t *out_lo, uint64_t *out_hi)
{
    const int h = (k - 1) / 2;
    const int wsh = wh.hb + 4;
    const int wa1 = wsh < 64 ? wsh : 0, wa2 = wsh < 64 ? 64 - wsh : 0, wa3 = wsh < 64 ? 0 : wsh - 64;
    const uint64_t wm1 = wsh < 64 ? ~0ull : 0ull;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t last = pos[i], first = last - (uint64_t)(k - 1);
        const uint64_t g0 = first >> 6, g1 = last >> 6;
        const int s = (int)(first & 63);
        const uint64_t l0 = planes[g0 * 5], h0 = planes[g0 * 5 + 1], l1 = g1 != g0 ? planes[g1 * 5] : 0ull, h1 = g1 != g0 ? planes[g1 * 5 + 1] : 0ull;
        const uint64_t wl = s ? (l0 >> s) | (l1 << (64 - s)) : l0, wh_ = s ? (h0 >> s) | (h1 << (64 - s)) : h0;      // bit j = position first + j
        auto code = [&](int j) -> uint64_t { return ((wl >> j) & 1ull) | (((wh_ >> j) & 1ull) << 1); };
        uint64_t upper = 0, lower = 0, rc_upper = 0, rc_lower = 0;
#pragma unroll
        for (int j = 0; j < 31; j++) {
            if (j < h) {
                const uint64_t cu = code(j), cl = code(h + 1 + j);
                upper = (upper << 2) | cu; lower = (lower << 2) | cl;
                rc_lower |= (cu ^ 2u) << (2 * j);
                rc_upper |= (cl ^ 2u) << (2 * j);
            }
        }
        const uint32_t mid = (uint32_t)code(h), rc_mid = mid ^ 2u;
        const bool userc = rc != 0;
        const bool ueq = upper == rc_upper;
        const bool gt = userc & ((upper > rc_upper) | (ueq & (lower > rc_lower)));
        const bool eq = userc & ueq & (lower == rc_lower);
        const uint64_t hl = gt ? rc_upper : upper, hr = gt ? rc_lower : lower;
        const uint32_t m4 = (1u << (gt ? rc_mid : mid)) | (eq ? (1u << rc_mid) : 0u);
        if (k <= 31) { uint32_t L = (uint32_t)hl, R = (uint32_t)hr; hmix_halves(L, R, hp); out_lo[i] = ((uint64_t)L << (hp.hb + 4)) | ((uint64_t)R << 4) | m4; }
        else { uint64_t L = hl, R = hr; hmix_halves_w(L, R, wh); out_lo[i] = (R << 4) | m4 | ((L << wa1) & wm1); out_hi[i] = (R >> 60) | ((L >> wa2) << wa3); }
    }
}
void launch_words_rebuild_planes(const uint32_t *pos, uint64_t n, const uint64_t *planes, int k, int rc, uint64_t *out_lo, uint64_t *out_hi, hipStream_t st)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(words_rebuild_planes_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, st, pos, n, planes, k, rc, make_hash_params(k < 31 ? k : 31),
                       make_wide_hash(k), out_lo, out_hi);
}
void launch_words_rebuild(const uint32_t *pos, uint64_t n, const uint8_t *seq, uint64_t len, int k, int rc, uint64_t *out_lo, uint64_t *out_hi, hipStream_t st)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(words_rebuild_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, st, pos, n, seq, len, k, rc, make_hash_params(k < 31 ? k : 31), make_wide_hash(k),
                       out_lo, out_hi);
}

// the window pass alone: per window-end position its ntHash, whether it passes the quality gates and -- want_words -- its packed word
int reads_windows(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q, DevBuf<uint64_t> &hash,
                  DevBuf<uint64_t> &wlo, DevBuf<uint64_t> &whi, DevBuf<uint8_t> &flag, unsigned long long *d_n_valid, bool want_words, const uint64_t *planes,
                  DevBuf<uint16_t> &rec_t, DevBuf<uint32_t> &tile_cnt)
{
    const bool wide = k > 31;
    const uint64_t tiles = (len + RW_TILE - 1) / RW_TILE;
    SKX_TRY(hash.alloc(tiles * RW_TILE));
    if (want_words) { SKX_TRY(flag.alloc(len)); SKX_TRY(wlo.alloc(len)); if (wide) SKX_TRY(whi.alloc(len)); }
    else { SKX_TRY(rec_t.alloc(tiles * RW_TILE)); SKX_TRY(tile_cnt.alloc(tiles)); }          // the gated windows only, compacted per tile
    ReadsArgs ra{d_seq, d_qual, len, k, rc, q.min_qual, q.qual_filter, make_hash_params(k < 31 ? k : 31), make_wide_hash(k),
                 hash.p, want_words ? wlo.p : nullptr, want_words && wide ? whi.p : nullptr, flag.p, d_n_valid, planes, rec_t.p, tile_cnt.p};
    const dim3 g((unsigned)tiles);
    if (want_words) hipLaunchKernelGGL(reads_windows_kernel<true>, g, dim3(RW_NT), 0, ctx->stream, ra);
    else hipLaunchKernelGGL(reads_windows_kernel<false>, g, dim3(RW_NT), 0, ctx->stream, ra);
    return SKX_OK;
}
int reads_tile() { return RW_TILE; }

// One FASTQ sample -> its SkaDict as a sorted (engine order) list of unique packed words.
int reads_sample_dict(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q,
                      DevBuf<uint64_t> &out_words, uint64_t *n_out)
{
    hipStream_t st = ctx->stream;
    const bool wide = k > 31;
    *n_out = 0;
    if (len == 0) return SKX_OK;
    if (len > 0xFFFFFFF0ull) { set_error("FASTQ sample longer than 4 G bases"); return SKX_EUNSUP; }
    Temp tmp;
    DevBuf<uint64_t> hash, wlo, whi; DevBuf<uint8_t> flag;
    SKX_TRY(hash.alloc(len)); SKX_TRY(wlo.alloc(len)); SKX_TRY(flag.alloc(len));
    if (wide) SKX_TRY(whi.alloc(len));
    ReadsArgs ra{d_seq, d_qual, len, k, rc, q.min_qual, q.qual_filter, make_hash_params(k < 31 ? k : 31), make_wide_hash(k),
                 hash.p, wlo.p, wide ? whi.p : nullptr, flag.p};
    hipLaunchKernelGGL(reads_windows_kernel<true>, dim3((unsigned)((len + RW_TILE - 1) / RW_TILE)), dim3(RW_NT), 0, st, ra);

    // candidate windows in stream order
    DevBuf<uint32_t> idx; SKX_TRY(idx.alloc(len));
    uint64_t m = 0;
    SKX_TRY(prim_select_index_u8(flag.p, idx.p, len, &m, st));
    if (m == 0) return SKX_OK;

    DevBuf<uint32_t> acc_t;            // stream positions of the windows that enter the dictionary
    uint64_t m2 = 0;
    if (q.min_count <= 1) {            // KmerFilter: 0 | 1 => no filtering
        acc_t = std::move(idx); m2 = m;
    } else {
        DevBuf<uint64_t> hk, hs; DevBuf<uint32_t> ts;
        SKX_TRY(hk.alloc(m)); SKX_TRY(hs.alloc(m)); SKX_TRY(ts.alloc(m));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, hash.p, idx.p, hk.p, m);
        SKX_TRY(sort_pairs(tmp, hk.p, hs.p, idx.p, ts.p, m, st));                        // stable: occurrences stay in stream order
        DevBuf<uint32_t> head, startpos, gsum, start;
        SKX_TRY(head.alloc(m)); SKX_TRY(startpos.alloc(m)); SKX_TRY(gsum.alloc(m)); SKX_TRY(start.alloc(m));
        hipLaunchKernelGGL(heads_kernel, dim3(grid_for(m)), dim3(256), 0, st, hs.p, head.p, startpos.p, m);
        SKX_TRY(prim_scan_add_u32(head.p, gsum.p, m, st));
        SKX_TRY(prim_scan_max_u32(startpos.p, start.p, m, st));
        uint32_t nd = 0;
        RP(hipMemcpyAsync(&nd, gsum.p + (m - 1), 4, hipMemcpyDeviceToHost, st));
        RP(hipStreamSynchronize(st));
        DevBuf<uint64_t> ckey, cks, dhash, fpv, prev; DevBuf<uint32_t> gidx, gs, lockey; DevBuf<uint8_t> FP;
        SKX_TRY(ckey.alloc(nd)); SKX_TRY(cks.alloc(nd)); SKX_TRY(dhash.alloc(nd)); SKX_TRY(fpv.alloc(nd)); SKX_TRY(prev.alloc(nd));
        SKX_TRY(gidx.alloc(nd)); SKX_TRY(gs.alloc(nd)); SKX_TRY(lockey.alloc(nd)); SKX_TRY(FP.alloc(nd));
        hipLaunchKernelGGL(distinct_kernel, dim3(grid_for(m)), dim3(256), 0, st, hs.p, ts.p, head.p, gsum.p, ckey.p, gidx.p, dhash.p, m);
        SKX_TRY(sort_pairs(tmp, ckey.p, cks.p, gidx.p, gs.p, nd, st));
        hipLaunchKernelGGL(fp_prepare_kernel, dim3(grid_for(nd)), dim3(256), 0, st, cks.p, gs.p, dhash.p, lockey.p, fpv.p, (uint64_t)nd);
        SKX_TRY(prim_seg_exscan_or_u64(lockey.p, fpv.p, prev.p, nd, st));          // per bloom word: the OR of the fingerprints that came before
        hipLaunchKernelGGL(fp_decide_kernel, dim3(grid_for(nd)), dim3(256), 0, st, gs.p, fpv.p, prev.p, FP.p, (uint64_t)nd);
        DevBuf<uint8_t> acc; SKX_TRY(acc.alloc(m));
        hipLaunchKernelGGL(accept_kernel, dim3(grid_for(m)), dim3(256), 0, st, gsum.p, start.p, FP.p, acc.p, (int)q.min_count, m);
        SKX_TRY(acc_t.alloc(m));
        SKX_TRY(prim_select_u32(ts.p, acc.p, acc_t.p, m, &m2, st));
        if (m2 == 0) return SKX_OK;
    }

    // accepted windows -> sorted unique packed words
    DevBuf<uint64_t> alo, ahi;
    SKX_TRY(alo.alloc(m2));
    hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m2)), dim3(256), 0, st, wlo.p, acc_t.p, alo.p, m2);
    if (wide) {
        SKX_TRY(ahi.alloc(m2));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m2)), dim3(256), 0, st, whi.p, acc_t.p, ahi.p, m2);
    }
    return sort_fold_words(ctx, alo.p, wide ? ahi.p : nullptr, m2, out_words, n_out);
}

// m packed words (low halves, and high halves when the keys are 128 bits wide) -> sorted by key, equal keys folded into one word whose base
// set is the OR of theirs (ska_dict.rs:92-101): a sample's SkaDict as one sorted list.  The tail of the sort-based read-set form, and what
// skx::dictset_sort makes of an assembly whose regions are beyond the per-region LDS sort (samples above ~5 Mbp keep 1 024 regions that grow
// with them: round 6).  out_words: wpk words per key.
int sort_fold_words(skx_ctx *ctx, const uint64_t *alo, const uint64_t *ahi, uint64_t m2, DevBuf<uint64_t> &out_words, uint64_t *n_out)
{
    hipStream_t st = ctx->stream;
    const bool wide = ahi != nullptr;
    const int wpk = wide ? 2 : 1;
    *n_out = 0;
    if (m2 == 0) return SKX_OK;
    if (m2 > 0xFFFFFFF0ull) { set_error("sample longer than 4 G windows"); return SKX_EUNSUP; }
    Temp tmp;
    DevBuf<uint64_t> slo, shi;
    SKX_TRY(slo.alloc(m2));
    if (!wide) {
        SKX_TRY(prim_sort_keys_u64(alo, slo.p, m2, 64, st));
    } else {
        // 128-bit order = stable sort by the low word, then stable sort by the high word
        SKX_TRY(shi.alloc(m2));
        DevBuf<uint32_t> i0, i1, i2; DevBuf<uint64_t> t1, t2;
        SKX_TRY(i0.alloc(m2)); SKX_TRY(i1.alloc(m2)); SKX_TRY(i2.alloc(m2)); SKX_TRY(t1.alloc(m2)); SKX_TRY(t2.alloc(m2));
        hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(m2)), dim3(256), 0, st, i0.p, m2);
        SKX_TRY(sort_pairs(tmp, alo, t1.p, i0.p, i1.p, m2, st));                         // by low word
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m2)), dim3(256), 0, st, ahi, i1.p, t2.p, m2);
        SKX_TRY(sort_pairs(tmp, t2.p, shi.p, i1.p, i2.p, m2, st));                          // then by high word (stable)
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m2)), dim3(256), 0, st, alo, i2.p, slo.p, m2);
    }
    DevBuf<uint32_t> whead, whsum;
    SKX_TRY(whead.alloc(m2)); SKX_TRY(whsum.alloc(m2));
    hipLaunchKernelGGL(word_heads_kernel, dim3(grid_for(m2)), dim3(256), 0, st, slo.p, wide ? shi.p : nullptr, whead.p, m2);
    SKX_TRY(prim_scan_add_u32(whead.p, whsum.p, m2, st));
    uint32_t nu = 0;
    RP(hipMemcpyAsync(&nu, whsum.p + (m2 - 1), 4, hipMemcpyDeviceToHost, st));
    RP(hipStreamSynchronize(st));
    SKX_TRY(out_words.alloc((uint64_t)nu * wpk));
    hipLaunchKernelGGL(word_fold_kernel, dim3(grid_for(m2)), dim3(256), 0, st, slo.p, wide ? shi.p : nullptr, whead.p, whsum.p, out_words.p, wpk, m2);
    RP(hipStreamSynchronize(st));
    RP(hipGetLastError());
    *n_out = nu;
    return SKX_OK;
}

// the words of one sample's regions (as the extraction kernel left them: region r holds raw[r] words from off[r]) as one list: dst_off[r] =
// where region r's words go; 128-bit words are split into their halves
__global__ __launch_bounds__(256) void gather_regions_kernel(const uint64_t *words, const uint64_t *off, const uint32_t *raw, const uint64_t *dst_off, uint64_t r0, int wpk,
                                                             uint64_t *lo, uint64_t *hi)
{
    const uint64_t r = r0 + blockIdx.x;
    const uint64_t src = off[r], dst = dst_off[blockIdx.x];
    const uint32_t n = raw[r];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        if (wpk == 1) lo[dst + i] = words[src + i];
        else { lo[dst + i] = words[(src + i) * 2]; hi[dst + i] = words[(src + i) * 2 + 1]; }
    }
}
void launch_gather_regions(const uint64_t *words, const uint64_t *off, const uint32_t *raw, const uint64_t *dst_off, uint64_t r0, uint64_t n_regions, int wpk,
                           uint64_t *lo, uint64_t *hi, hipStream_t st)
{
    if (n_regions) hipLaunchKernelGGL(gather_regions_kernel, dim3((unsigned)n_regions), dim3(256), 0, st, words, off, raw, dst_off, r0, wpk, lo, hi);
}

// ------------------------------------------------------------------------------------------------
// `ska map`: RefSka::new + map + AlnWriter (ska_ref.rs:189-311,508-583, aln_writer.rs) on the device.
// The reference genome goes through the same per-position window kernel as reads (no quality stream): every window-end
// position p of the record stream gets its packed word and a validity flag, in stream (= position) order.
// ------------------------------------------------------------------------------------------------
int ref_windows(skx_ctx *ctx, const uint8_t *d_seq, uint64_t len, int k, int rc, DevBuf<uint64_t> &wlo, DevBuf<uint64_t> &whi, DevBuf<uint8_t> &flag)
{
    const bool wide = k > 31;
    DevBuf<uint64_t> hash;
    SKX_TRY(hash.alloc(len)); SKX_TRY(wlo.alloc(len)); SKX_TRY(flag.alloc(len));
    if (wide) SKX_TRY(whi.alloc(len));
    ReadsArgs ra{d_seq, nullptr, len, k, rc, 0, 0, make_hash_params(k < 31 ? k : 31), make_wide_hash(k), hash.p, wlo.p, wide ? whi.p : nullptr, flag.p};
    hipLaunchKernelGGL(reads_windows_kernel<true>, dim3((unsigned)((len + RW_TILE - 1) / RW_TILE)), dim3(RW_NT), 0, ctx->stream, ra);
    return SKX_OK;
}

// ------------------------------------------------------------------------------------------------
// `ska cov`: CoverageHistogram::new (coverage.rs:70-148) + the histogram step of fit_histogram (:158-163).
// Every window of both read files (qualities ignored), sorted by split k-mer; run lengths = occurrence counts;
// hist[c - 1] = number of split k-mers seen c times (c <= 1000).
// ------------------------------------------------------------------------------------------------
__global__ void and_not15_kernel(uint64_t *v, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] &= ~15ull;
}
__global__ void run_lengths_kernel(const uint32_t *starts, uint64_t nr, uint64_t n, uint32_t *len)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nr; i += (uint64_t)gridDim.x * blockDim.x)
        len[i] = (uint32_t)((i + 1 < nr ? (uint64_t)starts[i + 1] : n) - starts[i]);
}
__global__ void count_hist_kernel(const uint32_t *counts, uint64_t n, uint32_t *hist)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (counts[i] - 1u < 1000u) atomicAdd(&hist[counts[i] - 1u], 1u);
}
int cov_histogram(skx_ctx *ctx, const uint8_t *d_seq, uint64_t len, int k, int rc, uint32_t *d_hist)
{
    hipStream_t st = ctx->stream;
    const bool wide = k > 31;
    if (len == 0) return SKX_OK;
    if (len > 0xFFFFFFF0ull) { set_error("read set longer than 4 G bases"); return SKX_EUNSUP; }
    Temp tmp;
    DevBuf<uint64_t> wlo, whi; DevBuf<uint8_t> flag;
    SKX_TRY(ref_windows(ctx, d_seq, len, k, rc, wlo, whi, flag));
    DevBuf<uint32_t> idx; SKX_TRY(idx.alloc(len));
    uint64_t m = 0;
    SKX_TRY(prim_select_index_u8(flag.p, idx.p, len, &m, st));
    if (m == 0) return SKX_OK;
    DevBuf<uint64_t> lo, slo, hi, shi;
    SKX_TRY(lo.alloc(m)); SKX_TRY(slo.alloc(m));
    hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, wlo.p, idx.p, lo.p, m);
    hipLaunchKernelGGL(and_not15_kernel, dim3(grid_for(m)), dim3(256), 0, st, lo.p, m);       // the middle base does not count (kmer only)
    if (!wide) {
        SKX_TRY(prim_sort_keys_u64(lo.p, slo.p, m, 64, st));
    } else {
        // order by (hi, lo): stable LSD passes
        DevBuf<uint32_t> iota, p1, p2; DevBuf<uint64_t> h1;
        SKX_TRY(hi.alloc(m)); SKX_TRY(shi.alloc(m)); SKX_TRY(iota.alloc(m)); SKX_TRY(p1.alloc(m)); SKX_TRY(p2.alloc(m)); SKX_TRY(h1.alloc(m));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, whi.p, idx.p, hi.p, m);
        hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(m)), dim3(256), 0, st, iota.p, m);
        SKX_TRY(sort_pairs(tmp, lo.p, slo.p, iota.p, p1.p, m, st));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, hi.p, p1.p, h1.p, m);
        SKX_TRY(sort_pairs(tmp, h1.p, shi.p, p1.p, p2.p, m, st));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, lo.p, p2.p, slo.p, m);
    }
    // run lengths = occurrence counts: where the runs of equal split k-mers start, then the distances between the starts
    DevBuf<uint32_t> head, starts, rcnt;
    SKX_TRY(head.alloc(m)); SKX_TRY(starts.alloc(m)); SKX_TRY(rcnt.alloc(m));
    hipLaunchKernelGGL(word_heads_kernel, dim3(grid_for(m)), dim3(256), 0, st, slo.p, wide ? shi.p : nullptr, head.p, m);
    uint64_t nr = 0;
    SKX_TRY(prim_select_index_u32(head.p, starts.p, m, &nr, st));
    hipLaunchKernelGGL(run_lengths_kernel, dim3(grid_for(nr)), dim3(256), 0, st, starts.p, nr, m, rcnt.p);
    hipLaunchKernelGGL(count_hist_kernel, dim3(grid_for(nr)), dim3(256), 0, st, rcnt.p, (uint64_t)nr, d_hist);
    RP(hipStreamSynchronize(st));
    return SKX_OK;
}

// `ska map --repeat-mask` (ska_ref.rs:259-293): rep[p] = 1 for every window whose split k-mer (the middle base does not count) occurs
// more than once in the reference.  The windows' words sorted together with their stream positions (the engine's own radix sort), a window
// is a repeat when a neighbour in that order holds the same k-mer.
__global__ void mark_repeats_kernel(const uint64_t *slo, const uint64_t *shi, const uint32_t *spos, const uint32_t *via, uint64_t n, uint8_t *rep)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = slo[i], hi = shi ? shi[i] : 0;
        const bool prev = i > 0 && slo[i - 1] == lo && (!shi || shi[i - 1] == hi), next = i + 1 < n && slo[i + 1] == lo && (!shi || shi[i + 1] == hi);
        if (prev || next) rep[via ? via[spos[i]] : spos[i]] = 1;
    }
}
int ref_repeat_flags(skx_ctx *ctx, const uint64_t *wlo, const uint64_t *whi, const uint8_t *flag, uint64_t len, DevBuf<uint8_t> &rep)
{
    hipStream_t st = ctx->stream;
    const bool wide = whi != nullptr;
    SKX_TRY(rep.alloc(len + 1)); SKX_TRY(rep.zero(st));
    if (len == 0) return SKX_OK;
    Temp tmp;
    DevBuf<uint32_t> idx; SKX_TRY(idx.alloc(len));
    uint64_t m = 0;
    SKX_TRY(prim_select_index_u8(flag, idx.p, len, &m, st));
    if (m < 2) return SKX_OK;
    DevBuf<uint64_t> lo, slo, hi, shi; DevBuf<uint32_t> sidx;
    SKX_TRY(lo.alloc(m)); SKX_TRY(slo.alloc(m));
    hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, wlo, idx.p, lo.p, m);
    hipLaunchKernelGGL(and_not15_kernel, dim3(grid_for(m)), dim3(256), 0, st, lo.p, m);
    if (!wide) {
        SKX_TRY(sidx.alloc(m));
        SKX_TRY(sort_pairs(tmp, lo.p, slo.p, idx.p, sidx.p, m, st));
        hipLaunchKernelGGL(mark_repeats_kernel, dim3(grid_for(m)), dim3(256), 0, st, slo.p, (const uint64_t *)nullptr, sidx.p, (const uint32_t *)nullptr, m, rep.p);
    } else {
        DevBuf<uint32_t> iota, p1, p2; DevBuf<uint64_t> h1;
        SKX_TRY(hi.alloc(m)); SKX_TRY(shi.alloc(m)); SKX_TRY(iota.alloc(m)); SKX_TRY(p1.alloc(m)); SKX_TRY(p2.alloc(m)); SKX_TRY(h1.alloc(m));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, whi, idx.p, hi.p, m);
        hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(m)), dim3(256), 0, st, iota.p, m);
        SKX_TRY(sort_pairs(tmp, lo.p, slo.p, iota.p, p1.p, m, st));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, hi.p, p1.p, h1.p, m);
        SKX_TRY(sort_pairs(tmp, h1.p, shi.p, p1.p, p2.p, m, st));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(m)), dim3(256), 0, st, lo.p, p2.p, slo.p, m);
        hipLaunchKernelGGL(mark_repeats_kernel, dim3(grid_for(m)), dim3(256), 0, st, slo.p, shi.p, p2.p, idx.p, m, rep.p);
    }
    RP(hipStreamSynchronize(st));
    return SKX_OK;
}

// row of the array holding each reference window's split k-mer (0xFFFFFFFF: none / no window), and whether the reference
// strand is the reverse complement of the canonical form (RefKmer::rc): then the canonical middle base is the complement of
// the forward one, i.e. the word's base mask is exactly 1 << (mid ^ 2)
__global__ __launch_bounds__(256) void map_lookup_kernel(const uint64_t *wlo, const uint8_t *flag, const uint8_t *seq, uint64_t len, int h,
                                                        const uint64_t *sorted, const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc)
{
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= len) return;
    uint32_t r = 0xFFFFFFFFu; uint8_t rcf = 0;
    if (flag[p]) {
        const uint64_t w = wlo[p], key = w >> 4;
        uint64_t lo = 0, hi = U;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((sorted[mid] >> 4) < key) lo = mid + 1; else hi = mid; }
        if (lo < U && (sorted[lo] >> 4) == key) r = perm ? perm[lo] : (uint32_t)lo;
        const uint32_t mid_code = (seq[p - h] >> 1) & 3u;
        rcf = ((uint32_t)w & 15u) == (1u << (mid_code ^ 2u));
    }
    row[p] = r; is_rc[p] = rcf;
}
void launch_map_lookup(const uint64_t *wlo, const uint8_t *flag, const uint8_t *seq, uint64_t len, int h, const uint64_t *sorted,
                       const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc, hipStream_t st)
{
    if (!len) return;
    hipLaunchKernelGGL(map_lookup_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, wlo, flag, seq, len, h, sorted, perm, U, row, is_rc);
}
__global__ void row_found_kernel(const uint32_t *row, uint8_t *found, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) found[i] = row[i] != 0xFFFFFFFFu;
}
// stream positions of the mapped windows, in order
int select_mapped(const uint32_t *row, uint64_t len, DevBuf<uint32_t> &mapped, uint64_t *m, hipStream_t st)
{
    Temp tmp; DevBuf<uint8_t> found;
    SKX_TRY(found.alloc(len)); SKX_TRY(mapped.alloc(len));
    hipLaunchKernelGGL(row_found_kernel, dim3(grid_for(len)), dim3(256), 0, st, row, found.p, len);
    (void)tmp;
    return prim_select_index_u8(found.p, mapped.p, len, m, st);
}
// sorted copy of unsorted array keys with the permutation back to rows (arrays loaded from a file keep the file's order)
int sort_words_perm(const uint64_t *words, uint64_t n, DevBuf<uint64_t> &sorted, DevBuf<uint32_t> &perm, hipStream_t st)
{
    Temp tmp; DevBuf<uint32_t> iota;
    SKX_TRY(sorted.alloc(n)); SKX_TRY(perm.alloc(n)); SKX_TRY(iota.alloc(n));
    if (!n) return SKX_OK;
    hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(n)), dim3(256), 0, st, iota.p, n);
    return sort_pairs(tmp, words, sorted.p, iota.p, perm.p, n, st);
}

// RC_IUPAC (bit_encoding.rs:475-510)
__device__ static inline uint8_t rc_iupac(uint8_t b)
{
    switch (b | 0x20) {
    case 'a': return 'T'; case 'b': return 'V'; case 'c': return 'G'; case 'd': return 'H'; case 'g': return 'C';
    case 'h': return 'D'; case 'k': return 'M'; case 'm': return 'K'; case 'n': return 'N'; case 'r': return 'Y';
    case 's': return 'S'; case 't': return 'A'; case 'v': return 'B'; case 'w': return 'W'; case 'y': return 'R';
    default: return '-';
    }
}
// mapped_variants, sample-major: mv[s][m] = (rc ? RC_IUPAC : id)(matrix[s][row of mapped window m])   (ska_ref.rs:519-530)
__global__ __launch_bounds__(256) void gather_mapped_kernel(const uint8_t *matrix, uint64_t pitch, const uint32_t *mapped, const uint32_t *row,
                                                           const uint8_t *is_rc, uint64_t M, uint8_t *mv, uint64_t mpitch)
{
    const uint64_t m = blockIdx.x * 256ull + threadIdx.x;
    if (m >= M) return;
    const uint64_t s = blockIdx.y;
    const uint32_t p = mapped[m];
    const uint8_t b = matrix[s * pitch + row[p]];
    mv[s * mpitch + m] = is_rc[p] ? rc_iupac(b) : b;
}
void launch_gather_mapped(const uint8_t *matrix, uint64_t pitch, int n_samples, const uint32_t *mapped, const uint32_t *row, const uint8_t *is_rc,
                          uint64_t M, uint8_t *mv, uint64_t mpitch, hipStream_t st)
{
    if (!M || !n_samples) return;
    hipLaunchKernelGGL(gather_mapped_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)n_samples), dim3(256), 0, st, matrix, pitch, mapped, row, is_rc, M, mv, mpitch);
}

// AlnWriter (aln_writer.rs) without its sequential walk.  Per sample and chromosome the writer's output is
//   out[x] = the (masked) middle base            where a split k-mer with a non-'-' base of this sample is mapped at x
//          = the reference base                  where such a position lies within `half` of x (write_split_kmer's left flank
//                                                + fill_fwd_bases' right overhang tile exactly the union of [p - half, p + half])
//          = '-'                                 elsewhere,
// plus one artefact that has to be kept for parity: last_mapped / last_written are not reset at a chromosome change, so
// the first fill_fwd_bases of the next chromosome can copy reference bases at the previous chromosome's trailing
// coordinates.  After a chromosome with mapped positions that stale state is (lm, min(lm + half + 1, len)) whatever the
// walk did, so it is reproduced from the first / last mapped position per (sample, chromosome) by one short loop per sample.
__device__ static inline bool is_ambiguous_d(uint8_t b) { b |= 0x20; return !(b == 'a' || b == 'c' || b == 'g' || b == 't' || b == 'u' || b == ('-' | 0x20)); }

// pass 1: middle bases + presence bits
__global__ __launch_bounds__(256) void map_mid_kernel(MapWriteArgs a)
{
    const uint64_t m = blockIdx.x * 256ull + threadIdx.x;
    if (m >= a.M) return;
    const uint64_t s = blockIdx.y;
    const uint8_t base = a.mv[s * a.mpitch + m];
    if (base == '-') return;                                                            // ska_ref.rs:574
    const uint64_t x = a.m_pos[m] + a.coff[a.m_chrom[m]];
    a.out[s * a.opitch + x] = (a.ambig_mask && is_ambiguous_d(base)) ? (uint8_t)'N' : base;
    atomicOr(&a.pres[s * a.ppitch + (x >> 5)], 1u << (x & 31));
}
// first / last present mapped position per (sample, chromosome); mlo/mhi = range of the chromosome in the mapped list
__global__ __launch_bounds__(64) void map_ends_kernel(MapWriteArgs a)
{
    const uint64_t i = blockIdx.x * 64ull + threadIdx.x;
    if (i >= (uint64_t)a.n_samples * a.n_chrom) return;
    const uint64_t s = i / a.n_chrom, c = i % a.n_chrom;
    const uint8_t *mv = a.mv + s * a.mpitch;
    uint32_t first = 0xFFFFFFFFu, last = 0xFFFFFFFFu;
    for (uint64_t m = a.mlo[c]; m < a.mhi[c]; m++) if (mv[m] != '-') { first = a.m_pos[m]; break; }
    if (first != 0xFFFFFFFFu) for (uint64_t m = a.mhi[c]; m-- > a.mlo[c];) if (mv[m] != '-') { last = a.m_pos[m]; break; }
    a.first[i] = first; a.last[i] = last;
}
// pass 2: flanks.  4 output positions per thread; the presence bits of [x - half, x + half] come from three 32-bit words
__global__ __launch_bounds__(256) void map_flank_kernel(MapWriteArgs a)
{
    const uint64_t x0 = (blockIdx.x * 256ull + threadIdx.x) * 4;
    if (x0 >= a.total) return;
    const uint64_t s = blockIdx.y;
    const uint32_t *pres = a.pres + s * a.ppitch;
    uint8_t *out = a.out + s * a.opitch;
    // chromosome of x0 (upper_bound on the output offsets)
    int lo = 0, hi = a.n_chrom;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.coff[mid] <= x0) lo = mid + 1; else hi = mid; }
    int c = lo - 1;
    uint32_t o4 = *reinterpret_cast<const uint32_t *>(out + x0);
    const uint32_t r4 = *reinterpret_cast<const uint32_t *>(a.refcat + x0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t x = x0 + i;
        if (x >= a.total) break;
        while (c + 1 < a.n_chrom && a.coff[c + 1] <= x) c++;
        if ((pres[x >> 5] >> (x & 31)) & 1u) continue;                                   // a middle base
        const uint64_t cb = a.coff[c], ce = cb + a.clen[c];                              // chromosome bounds in output coordinates
        const uint64_t wl = x >= cb + a.half ? x - a.half : cb, wh = x + a.half < ce ? x + a.half : ce - 1;
        const uint64_t wi = wl >> 5; const uint32_t sh = (uint32_t)(wl & 31);
        const uint64_t ab = (uint64_t)pres[wi] | ((uint64_t)pres[wi + 1] << 32);
        uint64_t v = ab >> sh;
        if (sh) v |= (uint64_t)pres[wi + 2] << (64 - sh);
        const uint32_t nb = (uint32_t)(wh - wl + 1);                                     // <= 63
        if (v & ((1ull << nb) - 1ull)) o4 = (o4 & ~(0xFFu << (8 * i))) | (((r4 >> (8 * i)) & 0xFFu) << (8 * i));
    }
    *reinterpret_cast<uint32_t *>(out + x0) = o4;
}
// the stale-state artefact at chromosome changes (see above) + repeat masking; one thread per sample, loops over chromosomes
__global__ __launch_bounds__(64) void map_stale_kernel(MapWriteArgs a)
{
    const uint64_t s = blockIdx.x * 64ull + threadIdx.x;
    if (s >= (uint64_t)a.n_samples) return;
    const uint32_t *pres = a.pres + s * a.ppitch;
    uint8_t *out = a.out + s * a.opitch;
    uint64_t lw = 0, lm = 0;                                                              // last_written, last_mapped
    for (int c = 0; c < a.n_chrom; c++) {
        const uint32_t first = a.first[s * a.n_chrom + c], last = a.last[s * a.n_chrom + c];
        const uint64_t len = a.clen[c], cb = a.coff[c];
        // the fill_fwd_bases that runs with the previous chromosome's state: before the first write (only if it is beyond
        // next_pos = half), or at the end of a chromosome without any write
        const bool has = first != 0xFFFFFFFFu;
        const uint64_t maximum = has ? (uint64_t)first - a.half : len;
        if (lw > 0 && (!has || (uint64_t)first > a.half)) {
            const uint64_t overhang = lm + a.half > lw ? lm + a.half - lw : 0, start = lw + 1;
            uint64_t end = start + overhang; if (end > maximum) end = maximum;
            if (end > start) {
                for (uint64_t x = start; x < end; x++) { const uint64_t X = cb + x; if (!((pres[X >> 5] >> (X & 31)) & 1u)) out[X] = a.refcat[X]; }
                lw = end;
            }
        }
        if (has) { lm = last; lw = (uint64_t)last + a.half + 1 < len ? (uint64_t)last + a.half + 1 : len; }
    }
}
__global__ __launch_bounds__(256) void map_repeat_kernel(MapWriteArgs a)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= a.n_repeat) return;
    uint8_t *o = a.out + (uint64_t)blockIdx.y * a.opitch + a.repeat[i];
    if (*o != '-') *o = 'N';                                                             // aln_writer.rs:150-154
}
void launch_aln_write(const MapWriteArgs &a, hipStream_t st)
{
    if (!a.n_samples || !a.total) return;
    hipLaunchKernelGGL(map_mid_kernel, dim3((unsigned)((a.M + 255) / 256), (unsigned)a.n_samples), dim3(256), 0, st, a);
    hipLaunchKernelGGL(map_ends_kernel, dim3((unsigned)(((uint64_t)a.n_samples * a.n_chrom + 63) / 64)), dim3(64), 0, st, a);
    hipLaunchKernelGGL(map_flank_kernel, dim3((unsigned)(((a.total + 3) / 4 + 255) / 256), (unsigned)a.n_samples), dim3(256), 0, st, a);
    hipLaunchKernelGGL(map_stale_kernel, dim3((unsigned)((a.n_samples + 63) / 64)), dim3(64), 0, st, a);
    if (a.n_repeat) hipLaunchKernelGGL(map_repeat_kernel, dim3((unsigned)((a.n_repeat + 255) / 256), (unsigned)a.n_samples), dim3(256), 0, st, a);
}

}  // namespace skx
