// gz_device.h -- DEFLATE (RFC 1951) inside gzip members (RFC 1952), decoded the way a device can: a file is cut into chunks, every chunk's
// wavefront finds the first dynamic block that starts in it (a bit position whose header and whole block decode cleanly), decodes from there
// to the next chunk's block WITHOUT knowing the 32 KB of text before it -- a copy that reaches back there yields a reference into that unknown
// window instead of a byte -- and the references are resolved afterwards, window by window (Kerbiriou & Chikhi 2019 do this on CPU threads).
// This header holds the per-lane logic (bit reader, Huffman tables, block decode, chunk walk, the block finder's tests) as plain functions so
// that tools/gzdev_host_check.cpp runs the SAME code on the host against zlib; skx_gzdev.hip wraps them into kernels.
// What the reference does here: needletail hands `.gz` files to flate2's MultiGzDecoder (io_utils.rs:55-76 opens whatever `parse_fastx_file`
// accepts); its result is the inflated text or an error.  Anything this decoder does not vouch for (a damaged stream, an unusual header, a local
// compression ratio beyond the symbol area) is reported as a status and the sample goes through the reader threads' inflater instead
// (gz_inflate.cpp), which then accepts it or words the error.
#pragma once
#include <stdint.h>

#ifndef GZD_HD
#define GZD_HD __host__ __device__ inline
#endif

namespace gzd {

// (first-level widths as zlib's: 852 and 592 entries are the most a literal/length and a distance code can need with them; small tables are
//  what lets a compute unit hold 32 decoding wavefronts)
constexpr int LIT_ROOT = 9, DIST_ROOT = 6, LIT_SIZE = 864, DIST_SIZE = 608;
constexpr uint32_t WIN = 32768;
// a symbol of a chunk's output: 0..255 a byte; SYM0 + i: the byte i of the 32 KB before the chunk (0 = oldest); INVALID: before any text
constexpr uint16_t SYM0 = 256, INVALID = 0xFFFF;
constexpr uint64_t NONE = ~0ull;
constexpr int MAX_MEMBERS = 32;                       // member ends recorded per chunk (bgzip: ~4 per 64 KB)
enum Status : uint32_t { OK = 0, E_DATA = 1, E_OVERFLOW = 2, E_SYNC = 3, E_UNUSUAL = 4, E_MEMBERS = 5, E_CHECK = 6 };

struct Tables { uint16_t lit[LIT_SIZE]; uint16_t dist[DIST_SIZE]; uint16_t count[16], next_code[16]; uint8_t lens[320]; uint8_t cl[32]; };   // (the builders' small arrays live here too: LDS on the device, not registers)
struct Member { uint64_t end; uint32_t crc, isize; };      // `end`: symbols of the chunk in front of the member's end
struct ChunkInfo { uint64_t n_out; uint64_t end_bit; uint32_t status; uint32_t n_members; };

// ---- bits, least significant first, from 32-bit words (the buffer behind the file is zero for at least 16 bytes)
struct BitIn {
    const uint32_t *w; uint64_t nwords, next; uint64_t buf; int cnt;
    GZD_HD void fill()
    {
        while (cnt <= 32) { const uint32_t x = next < nwords ? w[next] : 0u; buf |= (uint64_t)x << cnt; cnt += 32; next++; }
    }
    GZD_HD void seek(uint64_t bit) { next = bit >> 5; buf = 0; cnt = 0; fill(); const int d = (int)(bit & 31); buf >>= d; cnt -= d; }
    GZD_HD uint64_t pos() const { return next * 32 - (uint64_t)cnt; }
    GZD_HD uint32_t take(int n) { const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1ull)); buf >>= n; cnt -= n; return v; }      // n <= 32 after fill()
    GZD_HD bool past_end() const { return next > nwords + 2; }
};
GZD_HD uint32_t byte_at(const uint32_t *w, uint64_t i) { return (w[i >> 2] >> (8 * (i & 3))) & 255u; }
GZD_HD uint32_t rev_bits(uint32_t c, int l) { uint32_t r = 0; for (int i = 0; i < l; i++) { r = (r << 1) | (c & 1u); c >>= 1; } return r; }

// ---- canonical Huffman code -> two-level table.  Entry: (symbol << 4) | bits for a code of at most `root` bits (or, in a second-level
// table, the bits behind the root); 0x8000 | sub_bits << 11 | offset (< 2048) for a root prefix that longer codes share; 0: no such code.
// kind: 0 complete, 1 a single code of one bit (the only incomplete set inflate accepts), 2 no codes at all
GZD_HD int build_table(const uint8_t *lens, int n, int root, uint16_t *tab, int size, int *kind, uint16_t *count, uint16_t *next_code)
{
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[lens[i]]++;
    const int rootn = 1 << root;
    for (int i = 0; i < rootn; i++) tab[i] = 0;
    if (count[0] == n) { *kind = 2; return OK; }
    int left = 1, maxl = 0;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return E_DATA; if (count[l]) maxl = l; }
    if (left > 0) { if (!(maxl == 1 && count[1] == 1)) return E_DATA; *kind = 1; } else *kind = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t code = 0;
        count[0] = 0;
        for (int l = 1; l < 16; l++) { code = (code + count[l - 1]) << 1; next_code[l] = (uint16_t)code; }
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = rev_bits(next_code[l]++, l);
            if (l <= root) { if (pass == 0) for (uint32_t i = r; i < (uint32_t)rootn; i += 1u << l) tab[i] = (uint16_t)((s << 4) | l); continue; }
            const uint32_t p = r & (uint32_t)(rootn - 1);
            if (pass == 0) { const uint16_t cur = tab[p]; const int sb = l - root; if (!(cur & 0x8000) || (int)(cur & 15) < sb) tab[p] = (uint16_t)(0x8000 | sb); }
            else {
                const uint16_t e = tab[p]; const int sb = (e >> 11) & 15; const uint32_t off = e & 2047u, hi = r >> root;
                for (uint32_t i = hi; i < (1u << sb); i += 1u << (l - root)) tab[off + i] = (uint16_t)((s << 4) | (l - root));
            }
        }
        if (pass == 0) {
            int used = rootn;
            for (int p = 0; p < rootn; p++) {
                if (!(tab[p] & 0x8000)) continue;
                const int sb = tab[p] & 15;
                if (used + (1 << sb) > size || used + (1 << sb) > 2048) return E_UNUSUAL;
                tab[p] = (uint16_t)(0x8000 | (sb << 11) | used);
                for (int i = 0; i < (1 << sb); i++) tab[used + i] = 0;
                used += 1 << sb;
            }
        }
    }
    return OK;
}
// one symbol: 0..: the symbol; -1: a bit pattern no code has
GZD_HD int decode_sym(BitIn &b, const uint16_t *tab, int root)
{
    uint32_t e = tab[b.buf & ((1u << root) - 1u)];
    if (e & 0x8000u) { b.buf >>= root; b.cnt -= root; e = tab[(e & 2047u) + (uint32_t)(b.buf & ((1u << ((e >> 11) & 15u)) - 1u))]; }
    const int l = (int)(e & 15u);
    if (!l) return -1;
    b.buf >>= l; b.cnt -= l;
    return (int)(e >> 4);
}

// the header of a dynamic block (behind its three bits): both tables built.  strict: what the block finder asks of a candidate -- codes that
// are complete the way every compressor writes them
GZD_HD int read_dynamic(BitIn &b, Tables &t, bool strict)
{
    // the order the code-length code's lengths come in (16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15), five bits each
    const uint64_t ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const uint64_t ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    b.fill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return E_DATA;
    uint8_t *cl = t.cl;
    for (int i = 0; i < 19; i++) cl[i] = 0;
    for (int i = 0; i < hclen; i++) { b.fill(); cl[(uint32_t)(i < 12 ? ord_lo >> (5 * i) : ord_hi >> (5 * (i - 12))) & 31u] = (uint8_t)b.take(3); }
    int kind = 0;
    uint16_t *clt = t.dist;                                           // (the distance table's room, until that is built)
    int r = build_table(cl, 19, 7, clt, DIST_SIZE, &kind, t.count, t.next_code);
    if (r != OK) return r;
    if (kind == 2 || (strict && kind != 0)) return E_DATA;
    const int total = hlit + hdist;
    int i = 0;
    while (i < total) {
        b.fill();
        const int s = decode_sym(b, clt, 7);
        if (s < 0) return E_DATA;
        if (s < 16) { t.lens[i++] = (uint8_t)s; continue; }
        int rep, val = 0;
        if (s == 16) { if (i == 0) return E_DATA; val = t.lens[i - 1]; rep = 3 + (int)b.take(2); }
        else if (s == 17) rep = 3 + (int)b.take(3);
        else rep = 11 + (int)b.take(7);
        if (i + rep > total) return E_DATA;
        while (rep--) t.lens[i++] = (uint8_t)val;
    }
    if (b.past_end()) return E_DATA;
    if (t.lens[256] == 0) return E_DATA;                               // no end-of-block code
    if (strict) {
        // the block finder only: a FASTQ file's literals are printable ASCII, tab, line feed, carriage return.  One header in a thousand that
        // passes every other test is not a block's (measured: 1 false among 852 true in 25 MB); a code that gives lengths to other bytes is not
        // taken as a place to start from (a true block refused here is merely walked into from the block before)
        for (int c = 0; c < 256 && c < hlit; c++)
            if (t.lens[c] && !(c == 9 || c == 10 || c == 13 || (c >= 32 && c < 127))) return E_DATA;
    }
    r = build_table(t.lens + hlit, hdist, DIST_ROOT, t.dist, DIST_SIZE, &kind, t.count, t.next_code);
    if (r != OK) return r;
    int kind_l = 0;
    r = build_table(t.lens, hlit, LIT_ROOT, t.lit, LIT_SIZE, &kind_l, t.count, t.next_code);
    if (r != OK) return r;
    if (strict && kind_l != 0) return E_DATA;
    return OK;
}
GZD_HD int build_fixed(Tables &t)
{
    for (int i = 0; i < 144; i++) t.lens[i] = 8;
    for (int i = 144; i < 256; i++) t.lens[i] = 9;
    for (int i = 256; i < 280; i++) t.lens[i] = 7;
    for (int i = 280; i < 288; i++) t.lens[i] = 8;
    int kind = 0;
    int r = build_table(t.lens, 288, LIT_ROOT, t.lit, LIT_SIZE, &kind, t.count, t.next_code);
    if (r != OK) return r;
    for (int i = 0; i < 32; i++) t.lens[i] = 5;
    return build_table(t.lens, 32, DIST_ROOT, t.dist, DIST_SIZE, &kind, t.count, t.next_code);
}

// the symbols of one block, to its end-of-block code.  out[0..n) is the chunk so far; a copy that starts before index 0 reads the unknown
// window (symbolic values), one before `floor` (the member's first byte within this chunk, or -WIN when the chunk began inside the member)
// is the stream's error "distance too far back".  DRY: nothing is written (the block finder's test); n still counts.
template <bool DRY>
GZD_HD int inflate_block(BitIn &b, const Tables &t, uint16_t *out, uint64_t &n, uint64_t cap, int64_t floor)
{
    for (;;) {
        b.fill();
        int s = decode_sym(b, t.lit, LIT_ROOT);
        if (s < 0) return E_DATA;
        if (s < 256) {
            if (n >= cap) return E_OVERFLOW;
            if (!DRY) out[n] = (uint16_t)s;
            n++;
            continue;
        }
        if (s == 256) return b.past_end() ? E_DATA : OK;
        s -= 257;
        if (s >= 29) return E_DATA;
        uint32_t len;
        if (s < 8) len = 3u + (uint32_t)s;
        else if (s == 28) len = 258;
        else { const int eb = (s - 4) >> 2; len = ((4u + (uint32_t)(s & 3)) << eb) + 3u + b.take(eb); }
        b.fill();
        const int ds = decode_sym(b, t.dist, DIST_ROOT);
        if (ds < 0 || ds >= 30) return E_DATA;
        uint32_t dist;
        if (ds < 4) dist = 1u + (uint32_t)ds;
        else { const int eb = (ds >> 1) - 1; dist = ((2u + (uint32_t)(ds & 1)) << eb) + 1u + b.take(eb); }
        const int64_t src = (int64_t)n - (int64_t)dist;
        if (src < floor) return E_DATA;
        if (n + len > cap) return E_OVERFLOW;
        if (b.past_end()) return E_DATA;
        if (!DRY) {
            for (uint32_t j = 0; j < len; j++) {
                const int64_t q = src + (int64_t)j;
                out[n + j] = q < 0 ? (uint16_t)(SYM0 + (uint32_t)(q + (int64_t)WIN)) : out[q];
            }
        }
        n += len;
    }
}

// a gzip member's header at byte `at`: where its deflate data starts, or -1 (not one, or one this decoder leaves to the host's reader)
GZD_HD int64_t member_header(const uint32_t *w, uint64_t src_bytes, uint64_t at)
{
    if (at + 18 > src_bytes) return -1;
    if (byte_at(w, at) != 0x1f || byte_at(w, at + 1) != 0x8b || byte_at(w, at + 2) != 8) return -1;
    const uint32_t flg = byte_at(w, at + 3);
    if (flg & 0xE0u) return -1;
    uint64_t p = at + 10;
    if (flg & 4u) { if (p + 2 > src_bytes) return -1; p += 2 + (byte_at(w, p) | (byte_at(w, p + 1) << 8)); }
    for (int z = 0; z < 2; z++)
        if (flg & (z ? 16u : 8u)) { while (p < src_bytes && byte_at(w, p)) p++; p++; }
    if (flg & 2u) p += 2;
    if (p + 8 > src_bytes) return -1;
    return (int64_t)p;
}

// One chunk: blocks from `start_bit` (NONE: the file's first chunk -- the first member's header is read here) until the block that starts at
// `stop_bit` (NONE: to the end of the file).  Stored, fixed and dynamic blocks, member trailers and the headers of following members are
// walked as they come; `members` gets every member end met on the way.
GZD_HD void decode_chunk(const uint32_t *w, uint64_t src_bytes, uint64_t start_bit, uint64_t stop_bit, Tables &t, uint16_t *out, uint64_t cap,
                         ChunkInfo *info, Member *members)
{
    BitIn b; b.w = w; b.nwords = (src_bytes + 3) / 4;
    uint64_t n = 0; int64_t floor = -(int64_t)WIN; uint32_t nm = 0; int st = OK; bool fixed_built = false;
    const uint64_t src_bits = src_bytes * 8;
    if (start_bit == NONE) {
        const int64_t d = member_header(w, src_bytes, 0);
        if (d < 0) { info->n_out = 0; info->end_bit = 0; info->status = E_UNUSUAL; info->n_members = 0; return; }
        start_bit = (uint64_t)d * 8; floor = 0;
    }
    b.seek(start_bit);
    for (;;) {
        const uint64_t p = b.pos();
        if (p == stop_bit) break;
        if (stop_bit != NONE && p > stop_bit) { st = E_SYNC; break; }
        if (p + 3 > src_bits) { st = E_DATA; break; }
        b.fill();
        const uint32_t final = b.take(1), type = b.take(2);
        if (type == 0) {
            b.take(b.cnt & 7); b.fill();
            const uint32_t len = b.take(16); b.fill(); const uint32_t nlen = b.take(16);
            if ((len ^ nlen) != 0xFFFFu) { st = E_DATA; break; }
            if (n + len > cap) { st = E_OVERFLOW; break; }
            for (uint32_t i = 0; i < len; i++) { b.fill(); out[n++] = (uint16_t)b.take(8); }
            if (b.pos() > src_bits) { st = E_DATA; break; }
        } else if (type == 1) {
            if (!fixed_built) { st = build_fixed(t); if (st != OK) break; }
            fixed_built = true;
            st = inflate_block<false>(b, t, out, n, cap, floor);
            if (st != OK) break;
        } else if (type == 2) {
            fixed_built = false;
            st = read_dynamic(b, t, false);
            if (st == OK) st = inflate_block<false>(b, t, out, n, cap, floor);
            if (st != OK) break;
        } else { st = E_DATA; break; }
        if (b.pos() > src_bits) { st = E_DATA; break; }
        if (!final) continue;
        // the member's trailer: CRC-32 and length of its text, on a byte boundary; then the end of the file or another member
        b.take(b.cnt & 7); b.fill();
        if (b.pos() + 64 > src_bits) { st = E_DATA; break; }
        const uint32_t crc = b.take(32); b.fill(); const uint32_t isize = b.take(32);
        if (nm >= (uint32_t)MAX_MEMBERS) { st = E_MEMBERS; break; }
        members[nm].end = n; members[nm].crc = crc; members[nm].isize = isize; nm++;
        const uint64_t at = b.pos() >> 3;
        if (at == src_bytes) { if (stop_bit != NONE) st = E_SYNC; break; }
        const int64_t d = member_header(w, src_bytes, at);
        if (d < 0) { st = E_UNUSUAL; break; }
        b.seek((uint64_t)d * 8);
        floor = (int64_t)n;
    }
    info->n_out = n; info->end_bit = b.pos(); info->status = (uint32_t)st; info->n_members = nm;
}

// ---- the block finder.  64 bits of the stream at any bit position, from aligned words
GZD_HD uint64_t peek64(const uint32_t *w, uint64_t nwords, uint64_t bit)
{
    const uint64_t i = bit >> 5; const int d = (int)(bit & 31);
    const uint64_t a = i < nwords ? w[i] : 0u, b = i + 1 < nwords ? w[i + 1] : 0u, c = i + 2 < nwords ? w[i + 2] : 0u;
    const uint64_t lo = a | (b << 32);
    return d ? (lo >> d) | (c << (64 - d)) : lo;
}
// cheap test of a candidate position: not the last block, dynamic, counts in range, and a code-length code that is exactly complete
GZD_HD bool sync_quick(const uint32_t *w, uint64_t nwords, uint64_t bit)
{
    const uint64_t h = peek64(w, nwords, bit);
    if ((h & 7u) != 4u) return false;                                  // BFINAL 0, BTYPE 10 (its low bit first)
    const uint32_t hlit = (uint32_t)(h >> 3) & 31u, hdist = (uint32_t)(h >> 8) & 31u, hclen = ((uint32_t)(h >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    uint32_t kraft = 0;
    uint64_t v = peek64(w, nwords, bit + 17);
    for (uint32_t i = 0; i < hclen; i++) { const uint32_t l = (uint32_t)(v & 7u); v >>= 3; if (l) kraft += 128u >> l; }
    return kraft == 128u;
}
// full test: the header builds complete codes, the block decodes to its end within the file, and what follows is a block header too
GZD_HD bool sync_verify(const uint32_t *w, uint64_t src_bytes, uint64_t bit, Tables &t)
{
    BitIn b; b.w = w; b.nwords = (src_bytes + 3) / 4;
    const uint64_t src_bits = src_bytes * 8;
    b.seek(bit); b.take(3);
    if (read_dynamic(b, t, true) != OK) return false;
    uint64_t n = 0;
    if (inflate_block<true>(b, t, nullptr, n, NONE, -(int64_t)WIN) != OK) return false;
    if (n == 0 || b.pos() + 3 > src_bits) return false;
    b.fill();
    b.take(1); const uint32_t type = b.take(2);
    if (type == 3) return false;
    if (type == 0) { b.take(b.cnt & 7); b.fill(); const uint32_t len = b.take(16); b.fill(); const uint32_t nlen = b.take(16); return (len ^ nlen) == 0xFFFFu; }
    if (type == 2) return read_dynamic(b, t, false) == OK;
    return true;
}

// ---- windows.  A chunk's map: entry i = what byte i of the 32 KB behind the chunk's end is -- a byte, a reference into the window `prev`
// is relative to (the text before the chunk's group), or INVALID.  prev == nullptr: the chunk is the first of its group (identity).
GZD_HD uint16_t map_entry(const uint16_t *sym, uint64_t n, const uint16_t *prev, uint32_t i)
{
    uint16_t v;
    if ((uint64_t)i + n < WIN) v = (uint16_t)(SYM0 + i + (uint32_t)n);          // the chunk is shorter than the window: older text shifts down
    else v = sym[n - WIN + i];
    if (v < SYM0 || v == INVALID || !prev) return v;
    return prev[v - SYM0];
}
GZD_HD uint16_t through(const uint16_t *win, uint16_t v) { return (v < SYM0 || v == INVALID) ? v : win[v - SYM0]; }

// ---- CRC-32 (reflected 0xEDB88320) of pieces, joined: crc(A|B) from crc(A), crc(B), len(B) by multiplying in GF(2)[x] mod P
GZD_HD uint32_t crc_mul(uint32_t a, uint32_t b)                        // a * b mod P, reflected representation (bit 31 = x^0)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) r ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return r;
}
GZD_HD uint32_t crc_xpow8(uint64_t nbytes)                              // x^(8 * nbytes) mod P
{
    uint32_t r = 0x80000000u, sq = 0x00800000u;                         // 1, and x^8
    while (nbytes) { if (nbytes & 1u) r = crc_mul(r, sq); sq = crc_mul(sq, sq); nbytes >>= 1; }
    return r;
}
GZD_HD uint32_t crc_bytes(uint32_t state, const uint8_t *p, uint64_t n)      // raw register update (no pre/post inversion), bitwise
{
    for (uint64_t i = 0; i < n; i++) { state ^= p[i]; for (int k = 0; k < 8; k++) state = (state >> 1) ^ ((state & 1u) ? 0xEDB88320u : 0u); }
    return state;
}

}  // namespace gzd
