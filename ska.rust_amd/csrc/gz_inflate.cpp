// gz_inflate.cpp -- gzip members inflated by the reader threads (RFC 1951 / 1952), written for what read sets are: the form FASTQ
// files come in is .fastq.gz (the reference reads it through needletail's `compression` feature: Cargo.toml:34, ska_dict.rs:131-153,
// 356-366), and through zlib's gzread a reader thread produced ~130 MB/s of text against the 6 GB/s it parses and packs.
//
//   * a 64-bit bit buffer refilled without a branch (eight bytes loaded, whole bytes kept): one refill serves a run of literals, a
//     second one the length + distance of a match;
//   * the literal / length table is indexed by 11 bits and its entries are 64 bits wide: an entry holds EVERY literal the 11 bits
//     decode completely -- up to six (the four bases of a read cost two or three bits each in a FASTQ block's code, so one look-up
//     yields three or four of them) --, written with one 8-byte store; lengths carry their base and extra-bit count, longer codes
//     point to sub-tables; the distance table is indexed by 8 bits;
//   * matches are copied 16 bytes at a time when the distance allows it (a quality line repeated from the record before, ~300
//     bytes back, is most of a synthetic read set's matches), 8 at a time down to a distance of 8, a splat for distance 1;
//   * text goes into a window of the reader's own (1 MB; the last 32 KB -- or the caller's unfinished line, if that is longer --
//     are moved to its front when it is full) and is handed out in place: no copy between the inflater and the line parser;
//   * every member's CRC-32 (carry-less multiplication: 64 bytes a step) and length are checked against its trailer, the code
//     length sets against Kraft's sum as zlib checks them, distances against the text produced so far: a damaged or truncated file
//     is an error, never a short input (tests: test_gz_reader_*).  Members follow one another as gzip reads them; bytes behind the
//     last member that are not a member are ignored, as gzip and zlib's gzread ignore them.
#include "skx_internal.h"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <immintrin.h>
#include <unistd.h>

namespace skx {

namespace {

// ---- CRC-32 (IEEE 802.3, reflected: the one gzip trailers hold) ----
uint32_t g_crc_tab[8][256];
bool g_crc_ready = [] {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); g_crc_tab[0][i] = c; }
    for (uint32_t i = 0; i < 256; i++) for (int t = 1; t < 8; t++) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFFu];
    return true;
}();
// table form, eight bytes a step: short pieces, the tail of the folded form, processors without carry-less multiplication
uint32_t crc32_tab(uint32_t crc, const uint8_t *p, size_t n)      // crc: the running value, already inverted
{
    while (n >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        v ^= crc;
        crc = g_crc_tab[7][v & 0xFF] ^ g_crc_tab[6][(v >> 8) & 0xFF] ^ g_crc_tab[5][(v >> 16) & 0xFF] ^ g_crc_tab[4][(v >> 24) & 0xFF] ^
              g_crc_tab[3][(v >> 32) & 0xFF] ^ g_crc_tab[2][(v >> 40) & 0xFF] ^ g_crc_tab[1][(v >> 48) & 0xFF] ^ g_crc_tab[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ g_crc_tab[0][(crc ^ *p++) & 0xFFu];
    return crc;
}
// Folded form: the message is kept as four 128-bit remainders, each multiplied by x^512 mod P (as two 64 x 64 carry-less products) and
// added to the next 64 bytes; then the four are folded into one (x^128 mod P), that into 64 bits, and Barrett reduction gives the 32.
// Constants for the reflected polynomial 0x1DB710641 as published in "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"
// (Gopal et al., Intel 2009): x^(512+32), x^(512-32), x^(128+32), x^(128-32), x^64, P, floor(x^64 / P).  n >= 64, n % 16 == 0.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold(uint32_t crc, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)p), x2 = _mm_loadu_si128((const __m128i *)(p + 16)), x3 = _mm_loadu_si128((const __m128i *)(p + 32)),
            x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64; n -= 64;
#define fold(x, k, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128((x), (k), 0x00), _mm_clmulepi64_si128((x), (k), 0x11)), (next))
    while (n >= 64) {
        x1 = fold(x1, k1k2, _mm_loadu_si128((const __m128i *)p)); x2 = fold(x2, k1k2, _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = fold(x3, k1k2, _mm_loadu_si128((const __m128i *)(p + 32))); x4 = fold(x4, k1k2, _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    x1 = fold(x1, k3k4, x2); x1 = fold(x1, k3k4, x3); x1 = fold(x1, k3k4, x4);
    while (n >= 16) { x1 = fold(x1, k3k4, _mm_loadu_si128((const __m128i *)p)); p += 16; n -= 16; }
    // 128 -> 64 bits
    const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i x0 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x0);
    x0 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), k5, 0x00), x0);
    // Barrett
    x0 = _mm_clmulepi64_si128(_mm_and_si128(x1, m32), poly, 0x10);
    x0 = _mm_clmulepi64_si128(_mm_and_si128(x0, m32), poly, 0x00);
    x1 = _mm_xor_si128(x1, x0);
    return (uint32_t)_mm_extract_epi32(x1, 1);
#undef fold
}
bool crc_has_clmul()
{
    static const bool v = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !knob("no_clmul");
    return v;
}
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)     // zlib's convention: starts at 0, returns the finished value
{
    crc = ~crc;
    if (n >= 128 && crc_has_clmul()) {
        const size_t m = n & ~(size_t)15;
        crc = crc32_fold(crc, p, m);
        p += m; n -= m;
    }
    return ~crc32_tab(crc, p, n);
}

// ---- decode tables ----
constexpr int LIT_BITS = 11, DIST_BITS = 8;
// literal / length entry (64 bits): [7:0] bits consumed, [11:8] literals held, [15:12] kind, [63:16] the literals, first one lowest --
// or, for a length: [31:16] its base, [39:32] extra bits; for a sub-table: [31:16] its first entry, [39:32] its index bits
// distance entry (32 bits): [7:0] bits consumed, [11:8] extra bits / index bits, [15:12] kind, [31:16] base / first entry
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
constexpr uint64_t kind_of(uint64_t e) { return (e >> 12) & 15u; }
constexpr size_t LIT_TABLE = (1u << LIT_BITS) + 288 * 16, DIST_TABLE = (1u << DIST_BITS) + 32 * 128;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bitrev(uint32_t c, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1u); c >>= 1; } return r; }

// Canonical Huffman code -> table.  `payload(sym, len)` = the entry of a symbol (without the bits-consumed field for sub-table
// entries: those count only the bits behind the index).  Returns false for a set of lengths zlib's inflate_table refuses: over-subscribed, or
// incomplete unless it is a single code of length 1 (or no code at all: every entry invalid, an error only if such a code is used).
template <typename E, typename F>
bool build_table(const uint8_t *lens, int nsym, int root, E *tab, size_t tab_cap, F entry, bool single_ok = true)
{
    int count[16] = {0};
    for (int i = 0; i < nsym; i++) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 15; while (maxlen > 0 && !count[maxlen]) maxlen--;
    const E bad = (E)((uint64_t)K_BAD << 12 | 1u);
    if (maxlen == 0) { for (size_t i = 0; i < ((size_t)1 << root); i++) tab[i] = bad; return true; }
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
    if (left > 0 && (maxlen != 1 || !single_ok)) return false;
    uint32_t next[16]; { uint32_t c = 0; for (int l = 1; l <= 15; l++) { c = (c + (uint32_t)count[l - 1]) << 1; next[l] = c; } }
    for (size_t i = 0; i < ((size_t)1 << root); i++) tab[i] = bad;
    size_t used = (size_t)1 << root;
    // sub-tables: one per `root`-bit prefix that longer codes share, as wide as the longest of them
    uint8_t sub_bits[1u << LIT_BITS] = {0};
    uint32_t code_of[320];
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = bitrev(next[l]++, l);
        code_of[s] = c;
        if (l > root) { uint8_t &b = sub_bits[c & ((1u << root) - 1u)]; if (l - root > b) b = (uint8_t)(l - root); }
    }
    for (uint32_t p = 0; p < (1u << root); p++) if (sub_bits[p]) {
        if (used + ((size_t)1 << sub_bits[p]) > tab_cap) return false;
        tab[p] = (E)(((uint64_t)sub_bits[p] << (sizeof(E) == 8 ? 32 : 8)) | ((uint64_t)K_SUB << 12) | ((uint64_t)used << 16) | (uint64_t)root);
        for (size_t i = 0; i < ((size_t)1 << sub_bits[p]); i++) tab[used + i] = bad;
        used += (size_t)1 << sub_bits[p];
    }
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = code_of[s];
        if (l <= root) { const E e = (E)(entry(s) | (uint64_t)l); for (uint32_t i = c; i < (1u << root); i += 1u << l) tab[i] = e; }
        else {
            const E pe = tab[c & ((1u << root) - 1u)];
            const size_t first = (size_t)((pe >> 16) & 0xFFFFu);
            const int sb = sub_bits[c & ((1u << root) - 1u)];
            const E e = (E)(entry(s) | (uint64_t)(l - root));
            for (uint32_t i = c >> root; i < (1u << sb); i += 1u << (l - root)) tab[first + i] = e;
        }
    }
    return true;
}

inline uint64_t lit_entry(int s)
{
    if (s < 256) return ((uint64_t)s << 16) | (1u << 8) | ((uint64_t)K_LIT << 12);
    if (s == 256) return (uint64_t)K_EOB << 12;
    if (s > 285) return (uint64_t)K_BAD << 12;                                   // 286, 287: in the fixed code, never valid
    return ((uint64_t)LEN_BASE[s - 257] << 16) | ((uint64_t)LEN_EXTRA[s - 257] << 32) | ((uint64_t)K_LEN << 12);
}
inline uint64_t dist_entry(int s)
{
    if (s > 29) return (uint64_t)K_BAD << 12;
    return ((uint64_t)DIST_BASE[s] << 16) | ((uint64_t)DIST_EXTRA[s] << 8) | ((uint64_t)K_DIST << 12);
}
// every further literal the index bits decode completely joins its entry (greedily: the code is prefix-free, so what the known bits
// decode is what the stream holds)
void pack_literals(uint64_t *tab)
{
    static thread_local uint64_t one[1u << LIT_BITS];
    memcpy(one, tab, sizeof(one));
    for (uint32_t i = 0; i < (1u << LIT_BITS); i++) {
        uint64_t e = one[i];
        if (kind_of(e) != K_LIT) continue;
        uint32_t used = (uint32_t)(e & 0xFFu), cnt = 1;
        while (cnt < 6 && used < (uint32_t)LIT_BITS) {
            const uint64_t n = one[i >> used];
            const uint32_t l = (uint32_t)(n & 0xFFu);
            if (kind_of(n) != K_LIT || used + l > (uint32_t)LIT_BITS) break;
            e |= ((n >> 16) & 0xFFu) << (16 + 8 * cnt);
            used += l; cnt++;
        }
        tab[i] = (e & ~(uint64_t)0xFFFu) | ((uint64_t)cnt << 8) | used;
    }
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace

uint32_t gz_crc32(uint32_t crc, const uint8_t *p, size_t n) { return crc32_update(crc, p, n); }

struct GzReader::Impl {
    int fd = -1;
    std::vector<uint8_t> in, win;
    size_t in_pos = 0, in_end = 0; bool in_eof = false;
    size_t out_pos = 0, member_start = 0, crc_from = 0;      // window offsets: end of the text, start of this member's text (clamped to 0), first byte not in `crc` yet
    uint64_t member_len = 0;                                  // bytes of the member before crc_from
    uint32_t crc = 0;
    uint64_t bitbuf = 0; int bitcnt = 0;
    enum St { HEADER, BLOCK_HEAD, STORED, HUFF, TRAILER, DONE } st = HEADER;
    bool last_block = false, any_member = false;
    uint32_t stored_left = 0;
    uint64_t lit[LIT_TABLE]; uint32_t dist[DIST_TABLE];
    bool fixed_ready = false; std::vector<uint64_t> fixed_lit; std::vector<uint32_t> fixed_dist;

    static constexpr size_t IN_CAP = 1u << 20, IN_PAD = 64, WIN = 1u << 20, WIN_SLACK = 64, HIST = 32768;

    // more compressed bytes behind the unread ones (moved to the buffer's front); the bytes behind in_end are zero
    int fill_input()
    {
        // the bit buffer holds whole bytes taken from in[in_pos - bitcnt / 8 ...): they stay where they are, in the buffer
        // (and are given back to it by byte_align)
        const size_t held = (size_t)(bitcnt >> 3), base = in_pos - held;
        if (base > 0) { memmove(in.data(), in.data() + base, in_end - base); in_end -= base; in_pos = held; }
        while (!in_eof && in_end < IN_CAP) {
            const ssize_t r = ::read(fd, in.data() + in_end, IN_CAP - in_end);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) return -1;
            if (r == 0) { in_eof = true; break; }
            in_end += (size_t)r;
        }
        memset(in.data() + in_end, 0, IN_PAD);
        return 0;
    }
    // bits: the refill keeps whole bytes; `need` bits (<= 56) are there afterwards unless the input has ended (checked by the callers
    // through overrun())
    inline void refill() { bitbuf |= load64(in.data() + in_pos) << bitcnt; const int add = (63 - bitcnt) >> 3; in_pos += (size_t)add; bitcnt += add * 8; }
    inline uint32_t bits(int n) { const uint32_t v = (uint32_t)(bitbuf & (((uint64_t)1 << n) - 1u)); bitbuf >>= n; bitcnt -= n; return v; }
    inline bool overrun() const { return in_pos - (size_t)(bitcnt >> 3) > in_end; }      // bits were taken from behind the input's end
    void clean_bitbuf() { bitbuf &= bitcnt >= 64 ? ~0ull : (((uint64_t)1 << bitcnt) - 1u); }
    // bytes (header, trailer, stored blocks): the bit buffer's whole bytes go back to the input
    void byte_align() { bits(bitcnt & 7); in_pos -= (size_t)(bitcnt >> 3); bitbuf = 0; bitcnt = 0; }
    int need_bytes(size_t n) { if (in_end - in_pos < n && !in_eof) { if (fill_input() < 0) return -1; } return in_end - in_pos >= n ? 0 : 1; }      // 1: the file ends first

    int header();
    int block_head();
    int huff(size_t out_limit);
    int trailer();
};

int GzReader::Impl::header()
{
    // RFC 1952: ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN + extra] [name 0] [comment 0] [CRC16]
    int r = need_bytes(10);
    if (r < 0) return -1;
    const size_t left = in_end - in_pos;
    const bool magic = left >= 2 ? (in[in_pos] == 0x1f && in[in_pos + 1] == 0x8b) : (left == 1 && in[in_pos] == 0x1f);
    if (r > 0 || !magic) {
        if (!any_member) return -1;
        if (r > 0 && magic) return -1;                        // a further member's header cut short: truncated, as gzip reports it
        st = DONE; return 0;                                  // what follows the last member is not a member: ignored
    }
    if (in[in_pos + 2] != 8 || (in[in_pos + 3] & 0xE0)) return -1;
    const int flg = in[in_pos + 3];
    in_pos += 10;
    if (flg & 4) {
        if (need_bytes(2) != 0) return -1;
        size_t xlen = (size_t)in[in_pos] | ((size_t)in[in_pos + 1] << 8);
        in_pos += 2;
        while (xlen) { if (need_bytes(1) != 0) return -1; const size_t t = std::min(xlen, in_end - in_pos); in_pos += t; xlen -= t; }
    }
    for (int f : {8, 16}) if (flg & f) for (;;) { if (need_bytes(1) != 0) return -1; if (in[in_pos++] == 0) break; }
    if (flg & 2) { if (need_bytes(2) != 0) return -1; in_pos += 2; }
    any_member = true;
    member_start = out_pos; crc_from = out_pos; member_len = 0; crc = 0;
    bitbuf = 0; bitcnt = 0;
    st = BLOCK_HEAD;
    return 0;
}

int GzReader::Impl::block_head()
{
    // a dynamic block's header is at most 14 + 19 x 3 + 320 x 14 bits: make sure it is all there (or the file ends)
    if (in_end - in_pos < 1024 && !in_eof && fill_input() < 0) return -1;
    clean_bitbuf();
    refill();
    last_block = bits(1) != 0;
    const uint32_t type = bits(2);
    if (overrun()) return -1;
    if (type == 0) {
        byte_align();
        if (need_bytes(4) != 0) return -1;
        const uint32_t len = (uint32_t)in[in_pos] | ((uint32_t)in[in_pos + 1] << 8), nlen = (uint32_t)in[in_pos + 2] | ((uint32_t)in[in_pos + 3] << 8);
        if ((len ^ nlen) != 0xFFFFu) return -1;
        in_pos += 4;
        stored_left = len;
        st = STORED;
        return 0;
    }
    if (type == 3) return -1;
    if (type == 1) {
        if (!fixed_ready) {
            uint8_t l[320];
            for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            fixed_lit.resize(LIT_TABLE); fixed_dist.resize(DIST_TABLE);
            if (!build_table(l, 288, LIT_BITS, fixed_lit.data(), LIT_TABLE, lit_entry)) return -1;
            pack_literals(fixed_lit.data());
            for (int i = 0; i < 32; i++) l[i] = 5;
            if (!build_table(l, 32, DIST_BITS, fixed_dist.data(), DIST_TABLE, dist_entry)) return -1;
            fixed_ready = true;
        }
        memcpy(lit, fixed_lit.data(), sizeof(lit)); memcpy(dist, fixed_dist.data(), sizeof(dist));
        st = HUFF;
        return overrun() ? -1 : 0;
    }
    const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    // (every refill below is followed by an end-of-input check before the next one: the zeros behind in_end are IN_PAD bytes, a refill
    // moves in_pos by eight at most, so a file that ends inside this header is refused before a load could leave the buffer)
    for (int i = 0; i < hclen; i++) { if (bitcnt < 3) { refill(); if (overrun()) return -1; } cl[order[i]] = (uint8_t)bits(3); }
    uint32_t pre[128 + 19 * 1];
    const auto pre_entry = [](int s) -> uint64_t { return ((uint64_t)s << 16) | ((uint64_t)K_LIT << 12); };
    if (!build_table(cl, 19, 7, pre, sizeof(pre) / sizeof(pre[0]), pre_entry, false)) return -1;
    uint8_t lens[320 + 140];
    int n = 0;
    while (n < hlit + hdist) {
        refill();
        if (overrun()) return -1;
        const uint32_t e = pre[bitbuf & 127u];
        if (kind_of(e) != K_LIT) return -1;
        bits((int)(e & 0xFFu));
        const int sym = (int)(e >> 16);
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        int rep; uint8_t v = 0;
        if (sym == 16) { if (n == 0) return -1; v = lens[n - 1]; rep = 3 + (int)bits(2); }
        else if (sym == 17) rep = 3 + (int)bits(3);
        else rep = 11 + (int)bits(7);
        if (n + rep > hlit + hdist) return -1;
        memset(lens + n, v, (size_t)rep); n += rep;
    }
    if (overrun()) return -1;
    if (lens[256] == 0) return -1;                             // no end-of-block code
    if (!build_table(lens, hlit, LIT_BITS, lit, LIT_TABLE, lit_entry)) return -1;
    pack_literals(lit);
    if (!build_table(lens + hlit, hdist, DIST_BITS, dist, DIST_TABLE, dist_entry)) return -1;
    st = HUFF;
    return 0;
}

// the symbols of a Huffman block until its end, `out_limit`, or the end of the input in hand.  1: block finished, 0: come again, -1: bad data.
// What a read set's text decodes to is matches, not literals: in a 32 KB window of four letters nearly every stretch of 8 to 14 bases has been
// seen before (a level-6 stream of reads: 14 bytes per match, one literal in fifty symbols), so the loop is built around the match -- the
// next symbol's table entry is looked up BEFORE the copy of this match (the look-up's latency hides behind the copy), a match that
// follows a refill directly decodes length and distance from the same 56 bits, the copy is two unconditional 16-byte moves (a loop only
// beyond 32 bytes: its exit would be the one branch of the iteration that cannot be predicted), and the body is compiled twice, once for
// processors with BMI2 (field extraction and shifts by a register count in one instruction each).
namespace {
template <typename Z>
__attribute__((always_inline)) inline int huff_body(Z &z, size_t out_limit)
{
    uint8_t *const w = z.win.data();
    const uint8_t *const ib = z.in.data();
    const uint64_t *const lit = z.lit; const uint32_t *const dist = z.dist;
    size_t ip = z.in_pos, op = z.out_pos;
    uint64_t bb = z.bitbuf; int bc = z.bitcnt;
    bb &= bc >= 64 ? ~0ull : (((uint64_t)1 << bc) - 1u);
    // with the end of the file in hand the loop may run into the zeros behind it: every symbol is then checked against the input's end
    const bool tail = z.in_eof && z.in_pos + 64 > z.in_end;
    const size_t in_end = z.in_end;
    const size_t in_stop = tail ? in_end + 8 : (in_end >= 40 ? in_end - 40 : 0);
    const size_t hist0 = z.member_start;
    constexpr uint64_t LM = (1u << LIT_BITS) - 1u;
    int rc = 0;
#define GZ_REFILL() do { bb |= load64(ib + ip) << bc; const int add_ = (63 - bc) >> 3; ip += (size_t)add_; bc += add_ * 8; } while (0)
#define GZ_TAKE(n_) do { bb >>= (n_); bc -= (int)(n_); } while (0)
#define GZ_LOW(n_) (bb & (((uint64_t)1 << (n_)) - 1u))
#define GZ_PUT(e_) do { const uint64_t v_ = (e_) >> 16; memcpy(w + op, &v_, 8); op += ((e_) >> 8) & 15u; GZ_TAKE((e_) & 0xFFu); } while (0)
#define GZ_TAILCHECK() if (tail && ip - (size_t)(bc >> 3) > in_end) { rc = -1; break; }
    if (!(op < out_limit && ip <= in_stop)) return 0;
    GZ_REFILL();
    uint64_t e = lit[bb & LM];
    for (;;) {
        // here: at least 56 bits in hand, none of them taken since the refill, e = the entry of the next symbol
        bool fresh = true;
        if ((e & 0xF000u) == 0) {
            // up to three entries of literals on one refill (33 of its 56 bits)
            GZ_PUT(e);
            e = lit[bb & LM];
            if ((e & 0xF000u) == 0) {
                GZ_PUT(e);
                e = lit[bb & LM];
                if ((e & 0xF000u) == 0) {
                    GZ_PUT(e);
                    GZ_TAILCHECK();
                    if (!(op < out_limit && ip <= in_stop)) break;
                    GZ_REFILL();
                    e = lit[bb & LM];
                    continue;
                }
            }
            fresh = false;
        }
        uint32_t k = (uint32_t)kind_of(e);
        if (__builtin_expect(k == K_SUB, 0)) {
            GZ_TAKE(LIT_BITS);
            e = lit[((e >> 16) & 0xFFFFu) + GZ_LOW((e >> 32) & 0xFFu)];
            k = (uint32_t)kind_of(e);
            fresh = false;
            if (k == K_LIT) {
                w[op++] = (uint8_t)(e >> 16); GZ_TAKE(e & 0xFFu);
                GZ_TAILCHECK();
                if (!(op < out_limit && ip <= in_stop)) break;
                GZ_REFILL();
                e = lit[bb & LM];
                continue;
            }
        }
        if (__builtin_expect(k != K_LEN, 0)) { if (k == K_EOB) { GZ_TAKE(e & 0xFFu); rc = 1; } else rc = -1; break; }
        GZ_TAKE(e & 0xFFu);
        const uint32_t xl = (uint32_t)((e >> 32) & 0xFFu);
        const size_t len = (size_t)((e >> 16) & 0xFFFFu) + (size_t)GZ_LOW(xl);
        GZ_TAKE(xl);
        if (!fresh) GZ_REFILL();                         // (a length straight after the refill took 20 bits at most: the 28 of a distance are there)
        uint32_t d = dist[bb & ((1u << DIST_BITS) - 1u)];
        if (__builtin_expect(((d >> 12) & 15u) == K_SUB, 0)) { GZ_TAKE(DIST_BITS); d = dist[(d >> 16) + (uint32_t)GZ_LOW((d >> 8) & 15u)]; }
        if (__builtin_expect(((d >> 12) & 15u) != K_DIST, 0)) { rc = -1; break; }
        GZ_TAKE(d & 0xFFu);
        const uint32_t xd = (d >> 8) & 15u;
        const size_t back = (size_t)(d >> 16) + (size_t)GZ_LOW(xd);
        GZ_TAKE(xd);
        if (__builtin_expect(back > op - hist0, 0)) { rc = -1; break; }
        GZ_TAILCHECK();
        // the next symbol's entry, before this match is copied
        GZ_REFILL();
        e = lit[bb & LM];
        uint8_t *o = w + op; const uint8_t *s = o - back;
        op += len;
        if (__builtin_expect(back >= 16, 1)) {
            _mm_storeu_si128((__m128i *)o, _mm_loadu_si128((const __m128i *)s));
            _mm_storeu_si128((__m128i *)(o + 16), _mm_loadu_si128((const __m128i *)(s + 16)));
            if (__builtin_expect(len > 32, 0)) {
                uint8_t *const oe = o + len;
                o += 32; s += 32;
                do { _mm_storeu_si128((__m128i *)o, _mm_loadu_si128((const __m128i *)s)); o += 16; s += 16; } while (o < oe);
            }
        } else {
            uint8_t *const oe = o + len;
            if (back >= 8) {
                do { uint64_t v; memcpy(&v, s, 8); memcpy(o, &v, 8); o += 8; s += 8; } while (o < oe);
            } else if (back == 1) {
                const uint64_t v = 0x0101010101010101ull * s[0];
                do { memcpy(o, &v, 8); o += 8; } while (o < oe);
            } else {
                do { *o++ = *s++; } while (o < oe);
            }
        }
        if (!(op < out_limit && ip <= in_stop)) break;
    }
#undef GZ_REFILL
#undef GZ_TAKE
#undef GZ_LOW
#undef GZ_PUT
#undef GZ_TAILCHECK
    z.in_pos = ip; z.out_pos = op; z.bitbuf = bb; z.bitcnt = bc;
    if (rc == 1 && z.overrun()) return -1;
    return rc;
}
template <typename Z> __attribute__((target("bmi2"))) int huff_bmi2(Z &z, size_t out_limit) { return huff_body(z, out_limit); }
template <typename Z> int huff_plain(Z &z, size_t out_limit) { return huff_body(z, out_limit); }
}  // namespace
int GzReader::Impl::huff(size_t out_limit)
{
    static const bool bmi2 = __builtin_cpu_supports("bmi2") && !knob("no_bmi2");
    return bmi2 ? huff_bmi2(*this, out_limit) : huff_plain(*this, out_limit);
}

int GzReader::Impl::trailer()
{
    byte_align();
    if (need_bytes(8) != 0) return -1;
    crc = crc32_update(crc, win.data() + crc_from, out_pos - crc_from);
    member_len += out_pos - crc_from; crc_from = out_pos;
    const uint8_t *t = in.data() + in_pos;
    const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    if (want_crc != crc || want_len != (uint32_t)member_len) return -1;
    in_pos += 8;
    st = HEADER;
    return 0;
}

GzReader::GzReader() : impl(new Impl) {}
GzReader::~GzReader() { delete impl; }
void GzReader::open(int fd)
{
    Impl &z = *impl;
    z.fd = fd;
    z.in.assign(Impl::IN_CAP + Impl::IN_PAD, 0);
    z.win.assign(Impl::WIN + Impl::WIN_SLACK + 320, 0);
    z.in_pos = z.in_end = 0; z.in_eof = false; z.out_pos = 0; z.st = Impl::HEADER; z.any_member = false;
    // a reader reopened after an error in mid-stream starts clean (bits held from the old input would shift fill_input's base)
    z.bitbuf = 0; z.bitcnt = 0; z.last_block = false; z.stored_left = 0;
    z.crc = 0; z.member_start = 0; z.crc_from = 0; z.member_len = 0;
}
// More text.  The `keep` bytes in front of the last call's end stay in front of the new text (*p - keep is where they start): the caller's
// unfinished line.  *n == 0: the end of the last member.  -1: not gzip, damaged, truncated, or a read error.
int GzReader::next(const uint8_t **p, size_t *n, size_t keep)
{
    Impl &z = *impl;
    if (keep > z.out_pos) return -1;
    {   // room: the history a match may reach (and the caller's bytes) to the window's front
        const size_t hold = std::min(z.out_pos, std::max<size_t>(Impl::HIST, keep));
        if (z.out_pos + 4096 > Impl::WIN && hold < z.out_pos) {
            if (hold > Impl::WIN / 2) return -1;              // (callers spill lines of that size)
            const size_t shift = z.out_pos - hold;
            memmove(z.win.data(), z.win.data() + shift, hold);
            z.member_start = z.member_start > shift ? z.member_start - shift : 0;
            z.crc_from -= shift; z.out_pos = hold;               // (nothing pending: crc_from == out_pos at every call's end)
        }
    }
    const size_t start = z.out_pos;
    const size_t limit = Impl::WIN - 300;                     // a match is at most 258 bytes and its copy may write 15 more
    while (z.st != Impl::DONE && z.out_pos < limit) {
        int r = 0;
        switch (z.st) {
        case Impl::HEADER: r = z.header(); break;
        case Impl::BLOCK_HEAD: r = z.block_head(); break;
        case Impl::STORED: {
            if (z.stored_left == 0) { z.st = z.last_block ? Impl::TRAILER : Impl::BLOCK_HEAD; break; }
            const int nb = z.need_bytes(1);
            if (nb != 0) { r = -1; break; }
            const size_t t = std::min({(size_t)z.stored_left, z.in_end - z.in_pos, limit - z.out_pos});
            memcpy(z.win.data() + z.out_pos, z.in.data() + z.in_pos, t);
            z.in_pos += t; z.out_pos += t; z.stored_left -= (uint32_t)t;
            break; }
        case Impl::HUFF: {
            if (z.in_end - z.in_pos < 64 && !z.in_eof && z.fill_input() < 0) { r = -1; break; }
            r = z.huff(limit);
            if (r == 1) { z.st = z.last_block ? Impl::TRAILER : Impl::BLOCK_HEAD; r = 0; }
            else if (r == 0 && z.out_pos < limit) {
                // the input in hand ran low (more is read at the loop's top) -- or, with the whole file in hand, out
                if (z.in_eof && z.in_pos > z.in_end + 8) r = -1;
                else if (z.in_eof && z.overrun()) r = -1;
                else if (!z.in_eof && z.fill_input() < 0) r = -1;
            }
            break; }
        case Impl::TRAILER: r = z.trailer(); break;
        default: break;
        }
        if (r < 0) return -1;
    }
    z.crc = crc32_update(z.crc, z.win.data() + z.crc_from, z.out_pos - z.crc_from);
    z.member_len += z.out_pos - z.crc_from; z.crc_from = z.out_pos;
    *p = z.win.data() + start; *n = z.out_pos - start;
    return 0;
}

}  // namespace skx
