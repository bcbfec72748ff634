// skx_gzdev.hip -- `.fastq.gz` inflated on the device (round 6): feeder threads only read() the compressed bytes into the pinned ring, and
// what the read-set kernels are given is made here.  The format's logic is gz_device.h (bit reader, table builder, block and member walk,
// the block finder's tests: the same functions run on the host in tools/gzdev_host_check.cpp against zlib, tests/test_gz_device_logic.py);
// the kernels around it, a file at a time (two files of a sample on two streams):
//   gzd_find_kernel   : a wavefront per 16 KB chunk of the compressed file -- the first dynamic block that starts in it.  64 bit positions a
//                       turn, a lane each: three header bits and the two counts (one position in nine fits) -> a queue; 64 queued positions: a
//                       lane each sums the code-length code's Kraft sum (complete for one in ~70) -> a second queue; 64 of those: a lane
//                       each decodes the whole header in registers (literal and distance codes exactly complete, an end-of-block code,
//                       literals that are text).  The first position that passes is the chunk's start; no block is decoded here
//                       (SKX_KNOBS=gz_verify: it is, and a block header must follow -- 2.8 -> 15 ms a file)
//   gzd_decode_kernel : a wavefront per chunk walks the blocks from its start to the next chunk's.  Headers by lane 0 (tables into 3.3 KB of
//                       LDS: zlib's root widths, 9 and 6 bits).  Symbols by LOOK-AHEAD: lane i decodes the token that would start at bit
//                       i of the next 64 (two table gathers, extra bits, all in its own 64-bit view); the scalar unit hops along the real
//                       chain (three register reads a token) and pushes tokens into a batch; a batch of 64 is RESOLVED in an LDS buffer:
//                       literals and copies of settled text in parallel, copies of this batch's own output in order; coalesced stores of
//                       16-bit symbols: bytes, or references into the 32 KB in front of the chunk that it cannot know
//   gzd_maps_kernel   : a workgroup per GROUP of 64 chunks -- for every chunk the last 32 KB of text behind it as a map (bytes / references
//                       into the window in front of the group), one chunk after the other within the group
//   gzd_groups_kernel : one workgroup -- the windows in front of the groups, one after the other; then the chunks' places in the text (scan of
//                       their lengths), the members' ends and lengths against their trailers, the file's verdict, its first and last byte
//   gzd_text_kernel   : every symbol to its byte (through the previous chunk's map and the group's window), text written in place
//   gzd_crc_kernel / gzd_crc_check_kernel : CRC-32 of every member from 4 KB pieces joined by multiplication mod P, against the trailers
// Measured (MI355X, one 129 MB file of 150 bp reads at level 1 = 256 MB of text, 72 M tokens; profiles/r06zl): find 2.3 ms, decode
// 10.6-11.3 ms, maps 1.1, groups 1.2, text 0.9, CRC 1.0 ms = 17-18 ms a file (zlib on one core of the box: 0.78 s).  The decode is bound by instruction
// issue, not memory: 55 scalar + 33 vector instructions a token (PMC: 4.0 G + 2.4 G a file), one scalar unit a compute unit; the history
// of its forms -- lane 0 alone 51 ms, a scalar Huffman loop 36 ms, token batches 28 ms, high occupancy 22 ms (and no better however many
// wavefronts: 180 scalar instructions a token), look-ahead 20 -> 15 ms, copies resolved in dependency turns 12 ms, short copies unrolled a lane each and long ones shared 10.6 ms -- is in NOTEBOOK.md.
// HBM traffic per text byte: 2 B written + 2 B read of symbols, 1 B of text written, 1 B read by the CRC; the compressed bytes twice.
#include "skx_internal.h"
#include "skx_device.h"
#include "gz_device.h"

namespace skx {

using namespace gzd;

__global__ void __launch_bounds__(64) gzd_find_lane0_kernel(const uint32_t *w, uint64_t src_bytes, uint32_t chunk_bytes, uint32_t n_chunks, uint64_t *sync)
{
    __shared__ Tables t;
    __shared__ int s_found;
    const uint32_t c = blockIdx.x + 1;
    if (c >= n_chunks) return;
    const uint64_t nwords = (src_bytes + 3) / 4, lo = (uint64_t)c * chunk_bytes * 8;
    const uint64_t hi = lo + (uint64_t)chunk_bytes * 8 < src_bytes * 8 ? lo + (uint64_t)chunk_bytes * 8 : src_bytes * 8;
    uint64_t res = NONE;
    for (uint64_t base = lo; base < hi && res == NONE; base += 64) {
        const uint64_t pos = base + threadIdx.x;
        const bool q = pos < hi && sync_quick(w, nwords, pos);
        uint64_t m = __ballot(q);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (threadIdx.x == 0) s_found = sync_verify(w, src_bytes, base + (uint64_t)l, t) ? 1 : 0;
            __syncthreads();
            const int f = s_found;
            __syncthreads();
            if (f) { res = base + (uint64_t)l; break; }
        }
    }
    if (threadIdx.x == 0) sync[c] = res;
}

// where chunk c's symbols go: the symbol area is `ratio` entries per compressed byte, a chunk's share starts at its first byte's
__device__ static inline void chunk_extent(const uint64_t *sync, uint32_t n_chunks, uint32_t c, uint64_t src_bytes, uint32_t ratio,
                                           uint64_t *start, uint64_t *stop, uint64_t *base, uint64_t *cap)
{
    *start = c ? sync[c] : NONE;
    uint32_t c2 = c + 1;
    while (c2 < n_chunks && sync[c2] == NONE) c2++;
    *stop = c2 < n_chunks ? sync[c2] : NONE;
    const uint64_t sb = c ? (*start >> 3) : 0, eb = *stop == NONE ? src_bytes : (*stop >> 3);
    *base = (uint64_t)ratio * sb;
    *cap = (uint64_t)ratio * (eb - sb);
}

__global__ void __launch_bounds__(64) gzd_decode_lane0_kernel(const uint32_t *w, uint64_t src_bytes, uint32_t n_chunks, uint32_t ratio, const uint64_t *sync,
                                                        uint16_t *sym, ChunkInfo *info, Member *members)
{
    __shared__ Tables t;
    const uint32_t c = blockIdx.x;
    if (threadIdx.x != 0 || c >= n_chunks) return;
    if (c && sync[c] == NONE) { info[c].n_out = 0; info[c].end_bit = 0; info[c].status = OK; info[c].n_members = 0; return; }      // no block starts here: the chunk before walks through
    uint64_t start, stop, base, cap;
    chunk_extent(sync, n_chunks, c, src_bytes, ratio, &start, &stop, &base, &cap);
    decode_chunk(w, src_bytes, start, stop, t, sym + base, cap, &info[c], members + (size_t)c * MAX_MEMBERS);
}


// ---- the wave forms.  What every lane agrees on (positions, counts, the stream's words) is kept in scalar registers: values that pass through
// vector memory or LDS are made uniform again with readfirstlane, so the compiler keeps the control flow on the scalar unit.
__device__ static inline uint64_t uni64(uint64_t x)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
}
__device__ static inline uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// where a wavefront stands in the compressed stream (the words themselves come through uniform loads, a window ahead)
struct WaveBits {
    const uint32_t *w; uint32_t nwords; uint64_t bit;       // (word indices: files below 16 GB; at least 64 zero bytes lie behind the file)
    __device__ void seek(uint64_t at) { bit = uni64(at); }
    __device__ uint64_t pos() const { return bit; }
    __device__ bool past_end() const { return (uint32_t)(bit >> 5) > nwords + 2; }
};

// One block's symbols.  Three things alternate:
//   LOOK AHEAD (the lanes): the next 64 bits of the stream -- lane i decodes the token that WOULD start at bit i: literal/length code (a gather
//     from the table in LDS, a second one for a long code), extra bits, distance code and its extra bits, all from its own 64-bit view of
//     the stream; it ends with the token and the bit where the following one starts.  Most of these tokens do not exist; the ones that do
//     are found by
//   THE CHAIN (the scalar unit): from the bit the last token ended at, hop to `next` of that lane, three or four register reads a token, until
//     the window is left; each token met is pushed into the batch (three vector registers, a lane a token).  A serial Huffman loop costs
//     ~180 scalar instructions a token on ONE scalar unit a compute unit (measured: 22 ms a 129 MB file however many wavefronts); here the
//     table walks are vector work shared by 3-20 tokens a window.
//   RESOLVE (the lanes): a batch of up to 64 tokens into an LDS buffer of GZ_BUF symbols that starts at the batch's first position: literals
//     and the copies whose whole source lies before the batch in parallel, a lane each (their loads from the symbols already in memory
//     overlap); then the copies that read what this very batch produces, one after the other in order, the lanes sharing a copy's elements
//     (LDS to LDS: reads of FASTQ deflated at any level are mostly 3-5 byte copies, a third of them from less than 512 positions back);
//     then the buffer leaves as coalesced stores.
// Counters are 32-bit (a chunk's symbols).  The wavefront's own stores are seen by its later loads (one L1, in order).
constexpr uint32_t GZ_BUF = 512;
constexpr uint32_t TK_LIT = 0, TK_COPY = 1, TK_EOB = 2, TK_ERR = 3;
template <bool DRY>
__device__ static int inflate_block_wave(WaveBits &b, const Tables &t, uint16_t *s_buf, uint32_t *s_tpos, uint16_t *out, uint32_t &n_io, uint32_t cap, int32_t floor)
{
    const uint32_t lane = threadIdx.x;
    uint32_t n = uni32(n_io), bs = n, tc = 0;
    uint32_t v_pos = 0, v_meta = 0, v_src = 0;                         // token `lane` of the batch: destination, length | literal << 16, source (or the literal)
    cap = uni32(cap);
    int st = OK;
    auto resolve = [&]() {
        if (!tc) return;
        const uint32_t len = v_meta & 0x1FFu, rel = v_pos - bs;
        const int32_t src = (int32_t)v_src;
        // a symbol of the text so far: before the chunk (a reference), in this batch's buffer, or in memory
        auto sym_at = [&](int32_t q) -> uint16_t { return q < 0 ? (uint16_t)(SYM0 + (uint32_t)(q + (int32_t)WIN)) : q >= (int32_t)bs ? s_buf[(uint32_t)q - bs] : out[q]; };
        // a long copy (more than SHORT symbols: a read's header, a run), the lanes sharing its elements; one that overlaps itself repeats its first
        // `d` symbols, which are settled when it starts
        constexpr uint32_t SHORT = 8;
        auto copy_long = [&](uint64_t which) {
            while (which) {
                const int k = __ffsll((long long)which) - 1;
                which &= which - 1;
                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)v_pos, k) - bs, l = (uint32_t)__builtin_amdgcn_readlane((int)v_meta, k) & 0x1FFu;
                const int32_t sq = __builtin_amdgcn_readlane((int)v_src, k);
                const uint32_t d = p + bs - (uint32_t)sq;
                for (uint32_t done = 0; done < l; done += 64) {
                    const uint32_t j = done + lane;
                    if (j < l) s_buf[p + j] = sym_at(sq + (int32_t)(d < l ? j % d : j));
                }
            }
        };
        bool near = false, far_long = false;
        if (lane < tc) {
            if (v_meta >> 16) s_buf[rel] = (uint16_t)v_src;
            else if (src + (int32_t)len <= (int32_t)bs) {
                if (len <= SHORT) {
#pragma unroll
                    for (uint32_t h = 0; h < SHORT; h += 4) {
                        uint16_t x[4];
#pragma unroll
                        for (uint32_t u = 0; u < 4; u++) x[u] = h + u < len ? sym_at(src + (int32_t)(h + u)) : (uint16_t)0;      // (four loads overlap)
#pragma unroll
                        for (uint32_t u = 0; u < 4; u++) if (h + u < len) s_buf[rel + h + u] = x[u];
                    }
                } else far_long = true;
            } else near = true;
        }
        s_tpos[lane] = lane < tc ? v_pos : 0xFFFFFFFFu;
        copy_long(__ballot(far_long));
        __syncthreads();
        uint64_t pend = __ballot(near);
        if (pend) {
            // A copy that reads this batch's own output waits for the copies its source runs over (found by their positions: two binary
            // searches), itself excepted -- its own earlier elements are there when it reads them, element by element.  Every turn, the copies
            // that wait for nothing pending go: the short ones a lane each, the long ones one after the other with all lanes; the lowest
            // pending one always goes.  (Reads of FASTQ: two or three turns a batch.)
            uint64_t range = 0;
            if (near) {
                const uint32_t q0 = src > (int32_t)bs ? (uint32_t)src : bs, q1 = (uint32_t)(src + (int32_t)len - 1);
                uint32_t a = 0, b2 = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    if (a + step < 64u && s_tpos[a + step] <= q0) a += step;
                    if (b2 + step < 64u && s_tpos[b2 + step] <= q1) b2 += step;
                }
                range = (b2 >= 63u ? ~0ull : (2ull << b2) - 1ull) & ~((1ull << a) - 1ull) & ~(1ull << lane);
            }
            while (pend) {
                const bool go = near && ((pend >> lane) & 1ull) && (pend & range) == 0;
                if (go && len <= SHORT) {
#pragma unroll
                    for (uint32_t u = 0; u < SHORT; u++) if (u < len) s_buf[rel + u] = sym_at(src + (int32_t)u);      // (in order: it may read its own symbols)
                }
                copy_long(__ballot(go && len > SHORT));
                __syncthreads();
                pend &= ~__ballot(go);
            }
        }
        const uint32_t span = n - bs;
        for (uint32_t i = lane; i < span; i += 64) out[bs + i] = s_buf[i];
        __syncthreads();
        bs = n; tc = 0;
    };
    // Every window is 64 bits further on: two words, and the same shift within a word for the whole block -- lane i's view of a window is a fixed
    // pick of three of its five words.  The two words the NEXT window adds are loaded while this one is worked on.
    const uint32_t sh0 = (uint32_t)b.bit & 31u, tb = sh0 + lane, k = tb >> 5, sh = tb & 31u;
    uint32_t wi = (uint32_t)(b.bit >> 5);
    auto ldw = [&](uint32_t i) { return uni32(b.w[i < b.nwords + 8u ? i : b.nwords + 8u]); };
    uint32_t W0 = ldw(wi), W1 = ldw(wi + 1), W2 = ldw(wi + 2), W3 = ldw(wi + 3), W4 = ldw(wi + 4), P0 = ldw(wi + 5), P1 = ldw(wi + 6);
    uint32_t o = 0;                                                    // where the chain stands within the window
    for (bool last = false; !last;) {
        // ---- look ahead: the window's 64 bit positions, a lane each
        const uint32_t lo = k == 0 ? W0 : k == 1 ? W1 : W2, mid = k == 0 ? W1 : k == 1 ? W2 : W3, hi = k == 0 ? W2 : k == 1 ? W3 : W4;
        uint64_t x = ((uint64_t)lo | ((uint64_t)mid << 32)) >> sh;
        if (sh) x |= (uint64_t)hi << (64 - sh);
        uint32_t used, tk_kind, tk_len = 1, tk_val;
        {
            uint32_t e = t.lit[(uint32_t)x & ((1u << LIT_ROOT) - 1u)];
            used = 0;
            if (e & 0x8000u) { x >>= LIT_ROOT; used = LIT_ROOT; e = t.lit[(e & 2047u) + ((uint32_t)x & ((1u << ((e >> 11) & 15u)) - 1u))]; }
            const uint32_t l1 = e & 15u, s = e >> 4;
            x >>= l1; used += l1;
            tk_val = s;
            if (!l1) tk_kind = TK_ERR;
            else if (s < 256u) tk_kind = TK_LIT;
            else if (s == 256u) tk_kind = TK_EOB;
            else if (s >= 257u + 29u) tk_kind = TK_ERR;
            else {
                const uint32_t sl = s - 257u;
                const uint32_t eb = sl < 8u || sl == 28u ? 0u : (sl - 4u) >> 2;
                const uint32_t base = sl < 8u ? 3u + sl : sl == 28u ? 258u : ((4u + (sl & 3u)) << eb) + 3u;
                tk_len = base + ((uint32_t)x & ((1u << eb) - 1u));
                x >>= eb; used += eb;
                uint32_t e2 = t.dist[(uint32_t)x & ((1u << DIST_ROOT) - 1u)];
                if (e2 & 0x8000u) { x >>= DIST_ROOT; used += DIST_ROOT; e2 = t.dist[(e2 & 2047u) + ((uint32_t)x & ((1u << ((e2 >> 11) & 15u)) - 1u))]; }
                const uint32_t l2 = e2 & 15u, ds = e2 >> 4;
                x >>= l2; used += l2;
                if (!l2 || ds >= 30u) tk_kind = TK_ERR;
                else {
                    const uint32_t eb2 = ds < 4u ? 0u : (ds >> 1) - 1u;
                    const uint32_t dbase = ds < 4u ? 1u + ds : ((2u + (ds & 1u)) << eb2) + 1u;
                    tk_val = dbase + ((uint32_t)x & ((1u << eb2) - 1u));
                    used += eb2;
                    tk_kind = TK_COPY;
                }
            }
        }
        const uint32_t tk_next = lane + used, tk_meta = tk_len | (tk_kind << 16);
        // ---- the chain through this window.  safe: no token of this window can overrun the symbol area or reach before the member's start
        //      (a window holds at most 64 tokens of at most 258 symbols; a copy reaches at most 32768 back): the per-token checks are skipped
        const bool safe = cap - n > 64u * 258u && (int32_t)n - (int32_t)WIN >= floor;
        while (o < 64) {
            const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)tk_meta, (int)o), val = (uint32_t)__builtin_amdgcn_readlane((int)tk_val, (int)o);
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_readlane((int)tk_next, (int)o);
            const uint32_t kind = meta >> 16, len = meta & 0xFFFFu;
            const uint32_t src = kind == TK_LIT ? val : n - val;
            if (kind >= TK_EOB) { if (kind == TK_EOB) o = nxt; else st = E_DATA; last = true; }
            else if (!safe) {
                if (kind == TK_LIT) { if (n >= cap) { st = E_OVERFLOW; last = true; } }
                else if ((int32_t)src < floor) { st = E_DATA; last = true; }
                else if (len > cap - n) { st = E_OVERFLOW; last = true; }
            }
            if (!DRY && (last || tc == 64 || n - bs + len > GZ_BUF)) resolve();
            if (last) break;
            if (!DRY) {
                v_pos = lane == tc ? n : v_pos; v_meta = lane == tc ? (len | ((kind == TK_LIT ? 1u : 0u) << 16)) : v_meta; v_src = lane == tc ? src : v_src;
                tc++;
            }
            n += len;
            o = nxt;
        }
        if (last) { b.bit = ((uint64_t)wi << 5) + sh0 + o; break; }
        W0 = W2; W1 = W3; W2 = W4; W3 = P0; W4 = P1; wi += 2; o -= 64;
        P0 = ldw(wi + 5); P1 = ldw(wi + 6);
        if (wi > b.nwords + 2u) { st = E_DATA; b.bit = (uint64_t)wi << 5; if (!DRY) resolve(); break; }
    }
    if (st == OK && b.past_end()) st = E_DATA;
    n_io = n;
    return st;
}

// 64 bits of the stream at a bit position every lane agrees on
__device__ static inline uint64_t upeek64(const uint32_t *w, uint64_t nwords, uint64_t bit) { return uni64(peek64(w, nwords, bit)); }

struct WaveShared { uint64_t pos; int st; };

// a block's header read by lane 0 with the plain bit reader (tables into LDS), every lane told where the symbols start
__device__ static int wave_block_header(const uint32_t *w, uint64_t nwords, uint64_t pos, uint32_t type, bool strict, Tables &t, WaveShared &sh, uint64_t *after)
{
    if (threadIdx.x == 0) {
        BitIn hb; hb.w = w; hb.nwords = nwords;
        hb.seek(pos);
        sh.st = type == 2 ? read_dynamic(hb, t, strict) : build_fixed(t);
        sh.pos = hb.pos();
    }
    __syncthreads();
    *after = uni64(sh.pos);
    const int st = (int)uni32((uint32_t)sh.st);
    __syncthreads();
    return st;
}

// decode_chunk (gz_device.h) by the whole wavefront
template <bool DRYRUN>
__device__ static void decode_chunk_wave(const uint32_t *w, uint64_t src_bytes, uint64_t start_bit, uint64_t stop_bit, Tables &t, WaveShared &sh, uint16_t *s_buf, uint32_t *s_tpos, uint16_t *out,
                                         uint64_t cap, ChunkInfo *info, Member *members)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t nwords = (src_bytes + 3) / 4, src_bits = src_bytes * 8;
    uint64_t pos; uint32_t n = 0, nm = 0; int32_t floor = -(int32_t)WIN; int st = OK; bool fixed_built = false;
    const uint32_t cap32 = cap > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)cap;
    start_bit = uni64(start_bit); stop_bit = uni64(stop_bit);
    if (start_bit == NONE) {
        const int64_t d = (int64_t)uni64((uint64_t)member_header(w, src_bytes, 0));
        if (d < 0) { if (lane == 0) { info->n_out = 0; info->end_bit = 0; info->status = E_UNUSUAL; info->n_members = 0; } return; }
        start_bit = (uint64_t)d * 8; floor = 0;
    }
    pos = start_bit;
    WaveBits b; b.w = w; b.nwords = (uint32_t)nwords;
    for (;;) {
        if (pos == stop_bit) break;
        if (stop_bit != NONE && pos > stop_bit) { st = E_SYNC; break; }
        if (pos + 3 > src_bits) { st = E_DATA; break; }
        const uint64_t h = upeek64(w, nwords, pos);
        const uint32_t final = (uint32_t)h & 1u, type = (uint32_t)(h >> 1) & 3u;
        pos += 3;
        if (type == 0) {
            pos = (pos + 7) & ~7ull;
            const uint64_t v = upeek64(w, nwords, pos);
            const uint32_t len = (uint32_t)v & 0xFFFFu, nlen = (uint32_t)(v >> 16) & 0xFFFFu;
            if ((len ^ nlen) != 0xFFFFu) { st = E_DATA; break; }
            if (len > cap32 - n) { st = E_OVERFLOW; break; }
            pos += 32;
            if (pos + 8ull * len > src_bits) { st = E_DATA; break; }
            for (uint32_t i = lane; i < len; i += 64) out[n + i] = (uint16_t)byte_at(w, (pos >> 3) + i);
            n += len; pos += 8ull * len;
        } else if (type == 3) { st = E_DATA; break; }
        else {
            if (type == 2 || !fixed_built) { st = wave_block_header(w, nwords, pos, type, false, t, sh, &pos); if (st != OK) break; }
            fixed_built = type == 1;
            b.seek(pos);
            st = inflate_block_wave<DRYRUN>(b, t, s_buf, s_tpos, out, n, cap32, floor);
            pos = b.pos();
            if (st != OK) break;
        }
        if (pos > src_bits) { st = E_DATA; break; }
        if (!final) continue;
        pos = (pos + 7) & ~7ull;
        if (pos + 64 > src_bits) { st = E_DATA; break; }
        const uint64_t tr = upeek64(w, nwords, pos);
        pos += 64;
        if (nm >= (uint32_t)MAX_MEMBERS) { st = E_MEMBERS; break; }
        if (lane == 0) { members[nm].end = n; members[nm].crc = (uint32_t)tr; members[nm].isize = (uint32_t)(tr >> 32); }
        nm++;
        const uint64_t at = pos >> 3;
        if (at == src_bytes) { if (stop_bit != NONE) st = E_SYNC; break; }
        const int64_t d = (int64_t)uni64((uint64_t)member_header(w, src_bytes, at));
        if (d < 0) { st = E_UNUSUAL; break; }
        pos = (uint64_t)d * 8;
        floor = (int32_t)n;
    }
    if (lane == 0) { info->n_out = n; info->end_bit = pos; info->status = (uint32_t)st; info->n_members = nm; }
}

// sync_verify (gz_device.h) by the whole wavefront: the candidate's block is walked by the scalar loop, nothing written
__device__ static bool sync_verify_wave(const uint32_t *w, uint64_t src_bytes, uint64_t bit, Tables &t, WaveShared &sh)
{
    const uint64_t nwords = (src_bytes + 3) / 4, src_bits = src_bytes * 8;
    uint64_t pos;
    if (wave_block_header(w, nwords, bit + 3, 2, true, t, sh, &pos) != OK) return false;
    WaveBits b; b.w = w; b.nwords = (uint32_t)nwords;
    b.seek(pos);
    uint32_t n = 0;
    if (inflate_block_wave<true>(b, t, nullptr, nullptr, nullptr, n, 0xFFFFFF00u, -(int32_t)WIN) != OK) return false;
    pos = b.pos();
    if (n == 0 || pos + 3 > src_bits) return false;
    const uint64_t h = upeek64(w, nwords, pos);
    const uint32_t type = (uint32_t)(h >> 1) & 3u;
    if (type == 3) return false;
    if (type == 0) { const uint64_t v = upeek64(w, nwords, (pos + 3 + 7) & ~7ull); return (((uint32_t)v ^ (uint32_t)(v >> 16)) & 0xFFFFu) == 0xFFFFu; }
    if (type == 2) { uint64_t after; return wave_block_header(w, nwords, pos + 3, 2, false, t, sh, &after) == OK; }
    return true;
}

// A candidate's whole header checked by ONE lane in registers (64 candidates at a time): the code-length code decoded bit by bit from its
// counts (no table), the literal and distance code lengths summed as they come -- complete codes, an end-of-block code, the counts exact.
// What passes here builds the tables and is walked to its end-of-block (sync_verify_wave); what fails cost 1/64 of a wavefront.
__device__ static bool header_ok_lane(const uint32_t *w, uint32_t nwords, uint64_t bit)
{
    uint32_t next = (uint32_t)(bit >> 5);
    uint64_t buf = (uint64_t)(next < nwords ? w[next] : 0u) | ((uint64_t)(next + 1 < nwords ? w[next + 1] : 0u) << 32);
    next += 2;
    int cnt = 64 - (int)(bit & 31);
    buf >>= (bit & 31);
#define GZL_FILL() do { if (cnt <= 32) { buf |= (uint64_t)(next < nwords ? w[next] : 0u) << cnt; cnt += 32; next++; } } while (0)
#define GZL_TAKE(nb) ((uint32_t)buf & ((1u << (nb)) - 1u)); buf >>= (nb); cnt -= (nb)
    buf >>= 3; cnt -= 3;
    const uint32_t hlit = 257u + GZL_TAKE(5);
    const uint32_t hdist = 1u + GZL_TAKE(5);
    const uint32_t hclen = 4u + GZL_TAKE(4);
    if (hlit > 286u || hdist > 30u) return false;
    // the order the code-length code's lengths come in (16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15), five bits each
    const uint64_t ORD_LO = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const uint64_t ORD_HI = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    uint64_t cls = 0;                                                   // three bits a symbol
    for (uint32_t i = 0; i < hclen; i++) {
        GZL_FILL();
        const uint32_t l = GZL_TAKE(3);
        const uint32_t sym = (uint32_t)(i < 12 ? ORD_LO >> (5 * i) : ORD_HI >> (5 * (i - 12))) & 31u;
        cls |= (uint64_t)l << (3 * sym);
    }
    uint64_t cc = 0, so_lo = 0, so_hi = 0;                             // codes per length (five bits each), the symbols in code order
    uint32_t ns = 0;
    for (uint32_t l = 1; l <= 7; l++)
        for (uint32_t sym = 0; sym < 19; sym++)
            if (((uint32_t)(cls >> (3 * sym)) & 7u) == l) {
                if (ns < 12) so_lo |= (uint64_t)sym << (5 * ns); else so_hi |= (uint64_t)sym << (5 * (ns - 12));
                ns++; cc += 1ull << (5 * l);
            }
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0, eob = 0, kl = 0, kd = 0, nd = 0;
    while (i < total) {
        GZL_FILL();
        int code = 0, first = 0, index = 0, sym = -1;
        for (int l = 1; l <= 7; l++) {
            code |= (int)((uint32_t)buf & 1u); buf >>= 1; cnt--;
            const int c = (int)((uint32_t)(cc >> (5 * l)) & 31u);
            if (code - c < first) { const int idx = index + (code - first); sym = (int)((uint32_t)(idx < 12 ? so_lo >> (5 * idx) : so_hi >> (5 * (idx - 12))) & 31u); break; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        if (sym < 0) return false;
        uint32_t rep = 1, val = 0;
        if (sym < 16) val = (uint32_t)sym;
        else if (sym == 16) { if (!i) return false; val = prev; rep = 3u + GZL_TAKE(2); }
        else if (sym == 17) { rep = 3u + GZL_TAKE(3); }
        else { rep = 11u + GZL_TAKE(7); }
        if (i + rep > total) return false;
        // (the finder's rule of read_dynamic(strict): literal codes for printable ASCII, tab, line feed, carriage return only)
        if (val && i < 256u && (i <= 8u || (i <= 12u && i + rep > 11u) || (i <= 31u && i + rep > 14u) || i + rep > 127u)) return false;
        if (val) {
            const uint32_t k = 32768u >> val, n1 = i < hlit ? (rep < hlit - i ? rep : hlit - i) : 0u;
            kl += n1 * k; kd += (rep - n1) * k; nd += rep - n1;
            if (i <= 256u && 256u < i + rep) eob = val;
        }
        prev = val;
        i += rep;
    }
#undef GZL_FILL
#undef GZL_TAKE
    if (next > nwords + 2) return false;
    return kl == 32768u && eob != 0 && (kd == 32768u || nd == 0 || (nd == 1 && kd == 16384u));
}

template <bool VERIFY>
__global__ void __launch_bounds__(64) gzd_find_kernel(const uint32_t *w, uint64_t src_bytes, uint32_t chunk_bytes, uint32_t n_chunks, uint64_t *sync)
{
    __shared__ Tables t;
    __shared__ WaveShared sh;
    __shared__ uint64_t s_q[64], s_p[64];
    const uint32_t c = blockIdx.x + 1, lane = threadIdx.x;
    if (c >= n_chunks) return;
    const uint64_t nwords = (src_bytes + 3) / 4, lo = (uint64_t)c * chunk_bytes * 8;
    const uint64_t hi = lo + (uint64_t)chunk_bytes * 8 < src_bytes * 8 ? lo + (uint64_t)chunk_bytes * 8 : src_bytes * 8;
    // 128 words of the stream, two a lane; a position's 128 bits are four of them, fetched across the lanes
    uint64_t cb = (lo >> 5) & ~63ull;
    uint32_t cw = cb + lane < nwords ? w[cb + lane] : 0u, cw2 = cb + 64 + lane < nwords ? w[cb + 64 + lane] : 0u;
    uint64_t res = NONE;
    uint32_t qn = 0, pn = 0;
    // Two queues, so that every test runs with all lanes busy: positions whose three header bits and two counts fit (one in nine) wait in s_p
    // until there are 64, then a lane each sums a code-length code (complete for one in ~70); those wait in s_q until there are 64 (or the
    // chunk ends), then a lane each checks a whole header
    auto drain_q = [&]() {
        __syncthreads();
        const bool ok = lane < qn && header_ok_lane(w, (uint32_t)nwords, s_q[lane]);
        uint64_t vm = __ballot(ok);
        while (vm) {
            const int l = __ffsll((long long)vm) - 1;
            vm &= vm - 1;
            const uint64_t cand = uni64(s_q[l]);
            // VERIFY: the candidate's block is walked to its end-of-block and a header must follow (as much work as decoding the block).  Without:
            // a header whose three codes are exactly complete (and whose literals are text) is taken at its word -- the chunk before must END
            // exactly there (decode_chunk's E_SYNC), so a false one is noticed, and the file then goes through the reader threads' inflater
            if (!VERIFY || sync_verify_wave(w, src_bytes, cand, t, sh)) { res = cand; break; }
        }
        qn = 0;
        __syncthreads();
    };
    auto drain_p = [&]() {
        __syncthreads();
        bool q = false;
        uint64_t pos = 0;
        if (lane < pn) {
            pos = s_p[lane];
            const uint32_t hclen = ((uint32_t)(peek64(w, nwords, pos) >> 13) & 15u) + 4u;
            uint64_t vv = peek64(w, nwords, pos + 17);
            uint32_t kraft = 0;
            for (uint32_t z = 0; z < hclen; z++) { const uint32_t l = (uint32_t)(vv & 7u); vv >>= 3; if (l) kraft += 128u >> l; }
            q = kraft == 128u;
        }
        pn = 0;
        const uint64_t m = __ballot(q);
        if (m) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (qn + add > 64) { drain_q(); if (res != NONE) return; }
            if (q) s_q[qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pos;
            qn += add;
        }
        __syncthreads();
    };
    for (uint64_t base = lo; base < hi && res == NONE; base += 64) {
        if ((base >> 5) + 6 >= cb + 128) { cw = cw2; cb += 64; cw2 = cb + 64 + lane < nwords ? w[cb + 64 + lane] : 0u; }
        const uint64_t pos = base + lane;
        const uint32_t i = (uint32_t)((pos >> 5) - cb), d = (uint32_t)(pos & 31);
        // the 13 bits at the position: BFINAL 0, BTYPE 10, HLIT and HDIST at most 29
        const uint32_t x0 = (uint32_t)__shfl((int)cw, (int)(i & 63), 64), y0 = (uint32_t)__shfl((int)cw2, (int)(i & 63), 64);
        const uint32_t x1 = (uint32_t)__shfl((int)cw, (int)((i + 1) & 63), 64), y1 = (uint32_t)__shfl((int)cw2, (int)((i + 1) & 63), 64);
        const uint64_t two = (uint64_t)(i < 64 ? x0 : y0) | ((uint64_t)(i + 1 < 64 ? x1 : y1) << 32);
        const uint32_t h = (uint32_t)(two >> d);
        const bool q = pos < hi && (h & 7u) == 4u && ((h >> 3) & 31u) <= 29u && ((h >> 8) & 31u) <= 29u;
        const uint64_t m = __ballot(q);
        if (!m) continue;
        const uint32_t add = (uint32_t)__popcll(m);
        if (pn + add > 64) { drain_p(); if (res != NONE) break; }
        if (q) s_p[pn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pos;
        pn += add;
    }
    if (res == NONE && pn) drain_p();
    if (res == NONE && qn) drain_q();
    if (lane == 0) sync[c] = res;
}

template <bool DRYRUN>
__global__ void __launch_bounds__(64) gzd_decode_kernel(const uint32_t *__restrict__ w, uint64_t src_bytes, uint32_t n_chunks, uint32_t ratio, uint64_t walk_bytes, const uint64_t *sync,
                                                        uint16_t *sym, ChunkInfo *info, Member *members)
{
    __shared__ Tables t;
    __shared__ WaveShared sh;
    __shared__ uint16_t s_buf[GZ_BUF];
    __shared__ uint32_t s_tpos[64];
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    if (c && sync[c] == NONE) { if (threadIdx.x == 0) { info[c].n_out = 0; info[c].end_bit = 0; info[c].status = OK; info[c].n_members = 0; } return; }
    uint64_t start, stop, base, cap;
    chunk_extent(sync, n_chunks, c, src_bytes, ratio, &start, &stop, &base, &cap);
    // (a stretch with no place to start from -- stored or fixed blocks, literals that are not text -- is one wavefront's serial work: beyond
    //  walk_bytes the file is left to the reader threads' inflater)
    if (cap / ratio > walk_bytes) { if (threadIdx.x == 0) { info[c].n_out = 0; info[c].end_bit = 0; info[c].status = E_UNUSUAL; info[c].n_members = 0; } return; }
    decode_chunk_wave<DRYRUN>(w, src_bytes, start, stop, t, sh, s_buf, s_tpos, sym + uni64(base), uni64(cap), &info[c], members + (size_t)c * MAX_MEMBERS);
}

constexpr int GZ_NT = 1024;

__global__ void __launch_bounds__(GZ_NT) gzd_maps_kernel(const uint64_t *sync, uint32_t n_chunks, uint32_t ratio, uint32_t group, const uint16_t *sym,
                                                         const ChunkInfo *info, uint16_t *maps)
{
    const uint32_t c0 = blockIdx.x * group, c1 = c0 + group < n_chunks ? c0 + group : n_chunks;
    for (uint32_t c = c0; c < c1; c++) {
        const uint64_t n = info[c].n_out;
        const uint16_t *s = sym + (uint64_t)ratio * (c ? (sync[c] == NONE ? 0 : sync[c] >> 3) : 0);
        const uint16_t *prev = c == c0 ? nullptr : maps + (size_t)(c - 1) * WIN;
        uint16_t *mine = maps + (size_t)c * WIN;
        {
            uint16_t v[WIN / GZ_NT];                                   // (32 entries a thread: their loads overlap)
#pragma unroll
            for (uint32_t u = 0; u < WIN / GZ_NT; u++) v[u] = map_entry(s, n, prev, threadIdx.x + u * GZ_NT);
#pragma unroll
            for (uint32_t u = 0; u < WIN / GZ_NT; u++) mine[threadIdx.x + u * GZ_NT] = v[u];
        }
        __threadfence_block();                                    // (what is written here is read by this workgroup only, at addresses nobody read before: its own L1 serves)
        __syncthreads();
    }
}

typedef GzDevFileInfo GzFileInfoDev;

__global__ void __launch_bounds__(GZ_NT) gzd_groups_kernel(uint32_t n_chunks, uint32_t group, uint32_t n_groups, const uint16_t *sym, const ChunkInfo *info,
                                                           const Member *members, const uint16_t *maps, uint16_t *gwin, uint64_t *base, uint64_t *m_end,
                                                           uint32_t *m_crc, uint32_t max_members, GzFileInfoDev *fi)
{
    __shared__ uint64_t s_out[GZ_NT];
    __shared__ uint32_t s_mem[GZ_NT];
    __shared__ uint32_t s_status;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_status = OK;
    for (uint32_t i = t; i < WIN; i += GZ_NT) gwin[i] = INVALID;
    __threadfence_block();                                    // (what is written here is read by this workgroup only, at addresses nobody read before: its own L1 serves)
    __syncthreads();
    for (uint32_t g = 1; g < n_groups; g++) {
        const uint16_t *last = maps + (size_t)(g * group - 1) * WIN, *before = gwin + (size_t)(g - 1) * WIN;
        uint16_t *mine = gwin + (size_t)g * WIN;
        {
            uint16_t v[WIN / GZ_NT];
#pragma unroll
            for (uint32_t u = 0; u < WIN / GZ_NT; u++) v[u] = through(before, last[t + u * GZ_NT]);
#pragma unroll
            for (uint32_t u = 0; u < WIN / GZ_NT; u++) mine[t + u * GZ_NT] = v[u];
        }
        __threadfence_block();                                    // (what is written here is read by this workgroup only, at addresses nobody read before: its own L1 serves)
        __syncthreads();
    }
    // the chunks' places in the text, the members' ends: every thread a run of chunks
    const uint32_t per = (n_chunks + GZ_NT - 1) / GZ_NT, a = t * per < n_chunks ? t * per : n_chunks, b = a + per < n_chunks ? a + per : n_chunks;
    uint64_t out = 0; uint32_t nm = 0, st = OK;
    for (uint32_t c = a; c < b; c++) { out += info[c].n_out; nm += info[c].n_members; if (info[c].status && !st) st = info[c].status; }
    s_out[t] = out; s_mem[t] = nm;
    if (st) atomicMax(&s_status, st);
    __syncthreads();
    if (t == 0) {
        uint64_t o = 0; uint32_t m = 0;
        for (int i = 0; i < GZ_NT; i++) { const uint64_t x = s_out[i]; const uint32_t y = s_mem[i]; s_out[i] = o; s_mem[i] = m; o += x; m += y; }
        base[n_chunks] = o;
        fi->total = o; fi->n_members = m;
        if (m > max_members || m == 0) s_status = s_status ? s_status : (m ? E_MEMBERS : E_DATA);
    }
    __syncthreads();
    out = s_out[t]; nm = s_mem[t];
    for (uint32_t c = a; c < b; c++) {
        base[c] = out;
        for (uint32_t j = 0; j < info[c].n_members; j++, nm++)
            if (nm < max_members) { const Member &mb = members[(size_t)c * MAX_MEMBERS + j]; m_end[nm] = out + mb.end; m_crc[nm] = mb.crc; m_crc[max_members + nm] = mb.isize; }
        out += info[c].n_out;
    }
    __threadfence_block();                                    // (what is written here is read by this workgroup only, at addresses nobody read before: its own L1 serves)
    __syncthreads();
    const uint32_t total_m = fi->n_members < max_members ? fi->n_members : max_members;
    for (uint32_t i = t; i < total_m; i += GZ_NT) {
        const uint64_t len = m_end[i] - (i ? m_end[i - 1] : 0);
        if ((uint32_t)len != m_crc[max_members + i]) atomicMax(&s_status, (uint32_t)E_CHECK);
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t total = fi->total;
        if (total_m && m_end[total_m - 1] != total && !s_status) s_status = E_DATA;       // text behind the last member's end
        fi->status = s_status;
        fi->first = total ? sym[0] : 0;
        const uint32_t L = n_chunks - 1;
        fi->last = total ? through(gwin + (size_t)(L / group) * WIN, maps[(size_t)L * WIN + WIN - 1]) : 0;
    }
}

__global__ void __launch_bounds__(256) gzd_text_kernel(const uint64_t *sync, uint32_t ratio, uint32_t group, const uint16_t *sym, const ChunkInfo *info,
                                                       const uint64_t *base, const uint16_t *maps, const uint16_t *gwin, uint8_t *dst, GzFileInfoDev *fi)
{
    const uint32_t c = blockIdx.y;
    const uint64_t n = info[c].n_out;
    if (!n) return;
    const uint16_t *s = sym + (uint64_t)ratio * (c ? sync[c] >> 3 : 0);
    const uint16_t *prev = c % group ? maps + (size_t)(c - 1) * WIN : nullptr, *gw = gwin + (size_t)(c / group) * WIN;
    uint8_t *d = dst + base[c];
    bool bad = false;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        uint16_t v = s[j];
        if (v >= SYM0) {
            if (v != INVALID && prev) v = prev[v - SYM0];
            v = through(gw, v);
            if (v >= SYM0) { bad = true; v = '?'; }
        }
        d[j] = (uint8_t)v;
    }
    if (bad) atomicMax(&fi->status, (uint32_t)E_DATA);
}

constexpr uint32_t CRC_PIECE = 4096;
__global__ void __launch_bounds__(256) gzd_crc_kernel(const uint8_t *text, uint64_t total, const uint64_t *m_end, uint32_t n_members, uint32_t *m_acc)
{
    // four bytes a step through four tables of 256 words in LDS (made here: 8 shift-and-xor steps an entry, then each table from the one before)
    __shared__ uint32_t s_t[4][256];
    {
        uint32_t v = threadIdx.x;
        for (int k = 0; k < 8; k++) v = (v >> 1) ^ ((v & 1u) ? 0xEDB88320u : 0u);
        s_t[0][threadIdx.x] = v;
        __syncthreads();
        for (int tb = 1; tb < 4; tb++) { v = (v >> 8) ^ s_t[0][v & 255u]; s_t[tb][threadIdx.x] = v; }
        __syncthreads();
    }
    const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t pos = q * CRC_PIECE;
    if (pos >= total) return;
    const uint64_t rend = pos + CRC_PIECE < total ? pos + CRC_PIECE : total;
    uint32_t lo = 0, hi = n_members;                                   // the first member that ends behind `pos`
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (m_end[mid] > pos) hi = mid; else lo = mid + 1; }
    uint32_t m = lo;
    while (pos < rend && m < n_members) {
        const uint64_t mend = m_end[m], e = rend < mend ? rend : mend;
        uint32_t state = 0xFFFFFFFFu;
        uint64_t p = pos;
        while (p < e && ((uintptr_t)(text + p) & 3u)) state = crc_bytes(state, text + p, 1), p++;
        for (; p + 4 <= e; p += 4) {
            state ^= *reinterpret_cast<const uint32_t *>(text + p);
            state = s_t[3][state & 255u] ^ s_t[2][(state >> 8) & 255u] ^ s_t[1][(state >> 16) & 255u] ^ s_t[0][state >> 24];
        }
        if (p < e) state = crc_bytes(state, text + p, e - p);
        const uint32_t piece = state ^ 0xFFFFFFFFu;
        atomicXor(&m_acc[m], crc_mul(crc_xpow8(mend - e), piece));
        pos = e;
        if (pos == mend) m++;
    }
}
__global__ void gzd_crc_check_kernel(const uint32_t *m_acc, const uint32_t *m_crc, uint32_t n_members, GzFileInfoDev *fi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_members && m_acc[i] != m_crc[i]) atomicMax(&fi->status, (uint32_t)E_CHECK);
}

template <typename T> static int ensure(DevBuf<T> &b, size_t count) { return b.n >= count ? SKX_OK : b.alloc(count + count / 8); }

// a file's plan (chunks, groups, symbols per compressed byte) and buffers that hold it.  The pipeline reserves for every file of a batch before
// its first decode: a buffer that had to grow later would be freed under kernels still reading it (the next decode is queued behind the last
// sample's text kernels, not after them on the host)
int gz_device_reserve(GzDevWork &wk, uint64_t bytes, uint64_t text_hint)
{
    const long kb = knob("gz_chunk_kb"), kr = knob("gz_ratio"), kg = knob("gz_group");
    wk.chunk_bytes = (uint32_t)(kb > 0 ? kb : 16) << 10;
    wk.src_bytes = bytes;
    wk.n_chunks = (uint32_t)((bytes + wk.chunk_bytes - 1) / wk.chunk_bytes);
    if (wk.n_chunks == 0) wk.n_chunks = 1;
    // symbols per compressed byte: half as much again as the file's own ratio (a chunk that deflates better than that is refused: E_OVERFLOW).
    // Kept tight on purpose: 2 bytes a symbol, and memory another process has just released is slow to get (NOTEBOOK round 6)
    uint64_t ratio = bytes ? (3 * text_hint / 2 + bytes - 1) / bytes + 1 : 8;
    ratio = ratio < 4 ? 4 : ratio > 64 ? 64 : ratio;
    if (kr > 0) ratio = (uint64_t)kr;
    wk.ratio = (uint32_t)ratio;
    wk.group = (uint32_t)(kg > 0 ? kg : 64);
    wk.n_groups = (wk.n_chunks + wk.group - 1) / wk.group;
    wk.max_members = wk.n_chunks * 8 + 64;
    SKX_TRY(ensure(wk.sync, (size_t)wk.n_chunks + 1));
    SKX_TRY(ensure(wk.base, (size_t)wk.n_chunks + 1));
    SKX_TRY(ensure(wk.cinfo, (size_t)wk.n_chunks * sizeof(ChunkInfo)));
    SKX_TRY(ensure(wk.members, (size_t)wk.n_chunks * MAX_MEMBERS * sizeof(Member)));
    SKX_TRY(ensure(wk.sym, (size_t)(ratio * (bytes + 8))));
    SKX_TRY(ensure(wk.maps, (size_t)wk.n_chunks * WIN));
    SKX_TRY(ensure(wk.gwin, (size_t)wk.n_groups * WIN));
    SKX_TRY(ensure(wk.m_end, (size_t)wk.max_members));
    SKX_TRY(ensure(wk.m_crc, (size_t)wk.max_members * 2));
    SKX_TRY(ensure(wk.m_acc, (size_t)wk.max_members));
    SKX_TRY(ensure(wk.finfo, sizeof(GzFileInfoDev)));
    return SKX_OK;
}

// K0..K2 on `st`: src is the file as read (device memory, 4-byte aligned, at least 64 zero bytes behind it).  text_hint: the text's length if the
// trailer can be believed (sizes the symbol area).  wk.finfo holds the verdict when the stream has run.
int gz_device_decode(skx_ctx *ctx, hipStream_t st, const uint8_t *src, uint64_t bytes, uint64_t text_hint, GzDevWork &wk)
{
    (void)ctx;
    SKX_TRY(gz_device_reserve(wk, bytes, text_hint));
    wk.src = src;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
    const uint64_t walk = (uint64_t)(knob("gz_walk_kb") > 0 ? knob("gz_walk_kb") : 1024) << 10;
    if (knob("gz_lane0")) {                                           // (the first form: lane 0 does everything, the plain functions of gz_device.h)
        if (wk.n_chunks > 1) gzd_find_lane0_kernel<<<wk.n_chunks - 1, 64, 0, st>>>(w, bytes, wk.chunk_bytes, wk.n_chunks, wk.sync.p);
        gzd_decode_lane0_kernel<<<wk.n_chunks, 64, 0, st>>>(w, bytes, wk.n_chunks, wk.ratio, wk.sync.p, wk.sym.p, (ChunkInfo *)wk.cinfo.p, (Member *)wk.members.p);
    } else {
        if (wk.n_chunks > 1) {
            if (knob("gz_verify")) gzd_find_kernel<true><<<wk.n_chunks - 1, 64, 0, st>>>(w, bytes, wk.chunk_bytes, wk.n_chunks, wk.sync.p);
            else gzd_find_kernel<false><<<wk.n_chunks - 1, 64, 0, st>>>(w, bytes, wk.chunk_bytes, wk.n_chunks, wk.sync.p);
        }
        if (knob("gz_dry"))       // (measurement only: the symbols are decoded and counted, nothing is copied or written)
            gzd_decode_kernel<true><<<wk.n_chunks, 64, 0, st>>>(w, bytes, wk.n_chunks, wk.ratio, walk, wk.sync.p, wk.sym.p, (ChunkInfo *)wk.cinfo.p, (Member *)wk.members.p);
        else
            gzd_decode_kernel<false><<<wk.n_chunks, 64, 0, st>>>(w, bytes, wk.n_chunks, wk.ratio, walk, wk.sync.p, wk.sym.p, (ChunkInfo *)wk.cinfo.p, (Member *)wk.members.p);
    }
    gzd_maps_kernel<<<wk.n_groups, GZ_NT, 0, st>>>(wk.sync.p, wk.n_chunks, wk.ratio, wk.group, wk.sym.p, (const ChunkInfo *)wk.cinfo.p, wk.maps.p);
    gzd_groups_kernel<<<1, GZ_NT, 0, st>>>(wk.n_chunks, wk.group, wk.n_groups, wk.sym.p, (const ChunkInfo *)wk.cinfo.p, (const Member *)wk.members.p, wk.maps.p,
                                         wk.gwin.p, wk.base.p, wk.m_end.p, wk.m_crc.p, wk.max_members, (GzFileInfoDev *)wk.finfo.p);
    SKX_HIP(hipGetLastError());
    return SKX_OK;
}

// K3 and the members' CRCs: the text to dst[0 .. total) (total, n_members: what wk.finfo said); wk.finfo.status turns E_CHECK / E_DATA if the
// text is not what the trailers say
int gz_device_text(skx_ctx *ctx, hipStream_t st, GzDevWork &wk, uint8_t *dst, uint64_t total, uint32_t n_members)
{
    (void)ctx;
    if (!total) return SKX_OK;
    gzd_text_kernel<<<dim3(32, wk.n_chunks), 256, 0, st>>>(wk.sync.p, wk.ratio, wk.group, wk.sym.p, (const ChunkInfo *)wk.cinfo.p, wk.base.p, wk.maps.p, wk.gwin.p, dst,
                                                          (GzFileInfoDev *)wk.finfo.p);
    if (!knob("gz_no_crc")) {
        SKX_HIP(hipMemsetAsync(wk.m_acc.p, 0, (size_t)n_members * 4, st));
        const uint64_t pieces = (total + CRC_PIECE - 1) / CRC_PIECE;
        gzd_crc_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, st>>>(dst, total, wk.m_end.p, n_members, wk.m_acc.p);
        gzd_crc_check_kernel<<<(n_members + 255) / 256, 256, 0, st>>>(wk.m_acc.p, wk.m_crc.p, n_members, (GzFileInfoDev *)wk.finfo.p);
    }
    SKX_HIP(hipGetLastError());
    return SKX_OK;
}

}  // namespace skx
