// skf_codec.cpp -- `.skf` reader/writer of the engine: snappy-frame( CBOR( MergeSkaArray ) ) as produced by
// MergeSkaArray::save / consumed by ::load (merge_ska_array.rs:191-204).  Field order and encodings follow
// what ciborium 0.2 + ndarray 0.15 serde emit for the struct at merge_ska_array.rs:108-126:
//   map(8){ k, rc, names[], split_kmers[] (u128 > 2^64-1 as tag-2 bignum), variants{v:1, dim:[U,S], data[]},
//           variant_count[], ska_version, k_bits }
// Chunks are written snappy-compressed (type 0x00) with masked CRC-32C so files are loadable by the real `ska`.
//
// the reference encodes and decodes the whole array serially in memory (2 B of CBOR per matrix cell).
// Here both directions stream: the CBOR text exists only as 128 MB super-blocks, whose 64 KB snappy chunks are
// (de)compressed and checksummed by a team of host threads, and the U x S matrix is pulled from / pushed to the caller in
// row blocks (the device gathers / transposes them), so neither the 2UxS-byte CBOR nor a host copy of the matrix is ever
// materialised.
#include "skx_internal.h"
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace skx {
namespace {

// ---------------------------------------------------------------- CRC-32C (Castagnoli), slice-by-8
struct Crc32c {
    uint32_t t[8][256];
    Crc32c()
    {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0); t[0][i] = c; }
        for (uint32_t i = 0; i < 256; i++) for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
    uint32_t operator()(const uint8_t *p, size_t n) const
    {
        uint32_t c = ~0u;
        while (n >= 8) {
            uint64_t v; memcpy(&v, p, 8);
            v ^= c;
            c = t[7][v & 0xFF] ^ t[6][(v >> 8) & 0xFF] ^ t[5][(v >> 16) & 0xFF] ^ t[4][(v >> 24) & 0xFF] ^
                t[3][(v >> 32) & 0xFF] ^ t[2][(v >> 40) & 0xFF] ^ t[1][(v >> 48) & 0xFF] ^ t[0][v >> 56];
            p += 8; n -= 8;
        }
        while (n--) c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
        return ~c;
    }
};
const Crc32c crc32c;
inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

constexpr size_t CHUNK = 65536;                 // uncompressed bytes per snappy-frame chunk (the format's maximum)
// bytes of CBOR in flight per direction: 128 MB (thread teams are forked per super-block); SKX_SKF_BLOCK_MB trades memory for forks
static size_t super_bytes()
{
    static const size_t v = [] { long mb = knob("skf_block_mb", 128); if (mb < 1) mb = 1; if (mb > 4096) mb = 4096; return (size_t)mb * 16 * CHUNK; }();
    return v;
}
#define SUPER (super_bytes())

// does the device take a data section of this many bytes?  SKX_SKF_DEVICE = 0 never, 1 whenever a whole chunk lies inside,
// default: from 1 MB up (below that the host codec is as fast as the launches)
bool device_section(uint64_t bytes)
{
    const long e = knob("skf_device", -1);
    if (e == 0) return false;
    if (e == 1) return bytes >= CHUNK;
    return bytes >= (1u << 20);
}
int n_workers(int threads)
{
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads <= 0) threads = 4; }
    static const int cap = [] { const int v = (int)knob("skf_threads", 64); return v > 0 ? v : 64; }();
    return std::min(threads, cap);
}
template <typename F>
void parallel_for(size_t n, int threads, F &&f)      // f(i) for i in [0, n), dynamic distribution
{
    if (n == 0) return;
    const int nt = (int)std::min<size_t>((size_t)threads, n);
    if (nt <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto &th : pool) th.join();
}

// ---------------------------------------------------------------- snappy block (format_description.txt)
void put_varint(std::vector<uint8_t> &o, uint32_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
void put_literal(std::vector<uint8_t> &o, const uint8_t *p, size_t n)
{
    if (!n) return;
    size_t l = n - 1;
    if (l < 60) o.push_back((uint8_t)(l << 2));
    else if (l < 256) { o.push_back(60 << 2); o.push_back((uint8_t)l); }
    else { o.push_back(61 << 2); o.push_back((uint8_t)l); o.push_back((uint8_t)(l >> 8)); }      // blocks are <= 65536
    o.insert(o.end(), p, p + n);
}
void put_copy(std::vector<uint8_t> &o, size_t off, size_t len)
{
    while (len >= 68) { o.push_back((uint8_t)(((64 - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); len -= 64; }
    if (len > 64) { o.push_back((uint8_t)(((60 - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); len -= 60; }
    if (len >= 12 || off >= 2048) { o.push_back((uint8_t)(((len - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); }
    else { o.push_back((uint8_t)(((off >> 8) << 5) | ((len - 4) << 2) | 1)); o.push_back((uint8_t)off); }
}
void snappy_compress_block(const uint8_t *in, size_t n, std::vector<uint8_t> &o)
{
    put_varint(o, (uint32_t)n);
    if (n < 16) { put_literal(o, in, n); return; }
    static thread_local uint16_t table[1 << 14];
    memset(table, 0, sizeof table);
    auto load32 = [&](size_t i) { uint32_t v; memcpy(&v, in + i, 4); return v; };
    auto load64 = [&](size_t i) { uint64_t v; memcpy(&v, in + i, 8); return v; };
    auto hash = [&](uint32_t v) { return (v * 0x1e35a7bdu) >> 18; };
    size_t lit = 0, i = 1;
    const size_t limit = n - 4;
    while (i <= limit) {
        uint32_t cur = load32(i), h = hash(cur);
        size_t cand = table[h];
        table[h] = (uint16_t)i;
        if (cand < i && load32(cand) == cur) {
            size_t len = 4;
            while (i + len + 8 <= n) {                                  // extend 8 bytes at a time
                const uint64_t x = load64(cand + len) ^ load64(i + len);
                if (x) { len += (size_t)__builtin_ctzll(x) >> 3; goto done; }
                len += 8;
            }
            while (i + len < n && in[cand + len] == in[i + len]) len++;
        done:
            put_literal(o, in + lit, i - lit);
            put_copy(o, i - cand, len);
            i += len; lit = i;
        } else i++;
    }
    put_literal(o, in + lit, n - lit);
}
// uncompressed length announced by a compressed block (0xFFFFFFFF: malformed)
uint32_t snappy_ulen(const uint8_t *in, size_t n, size_t *hdr)
{
    size_t i = 0; uint32_t ulen = 0; int sh = 0;
    for (;;) { if (i >= n || sh > 28) return 0xFFFFFFFFu; uint8_t b = in[i++]; ulen |= (uint32_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; }
    if (hdr) *hdr = i;
    return ulen;
}
// decompress into out[0 .. ulen) (caller sized it from snappy_ulen)
bool snappy_uncompress_block(const uint8_t *in, size_t n, uint8_t *out, size_t ulen)
{
    size_t i = 0;
    if (snappy_ulen(in, n, &i) != ulen) return false;
    size_t o = 0;
    while (i < n) {
        const uint8_t tag = in[i++];
        size_t len, off;
        if ((tag & 3) == 0) {
            len = tag >> 2;
            if (len >= 60) { size_t nb = len - 59; if (i + nb > n) return false; len = 0; for (size_t k = 0; k < nb; k++) len |= (size_t)in[i + k] << (8 * k); i += nb; }
            len++;
            if (i + len > n || o + len > ulen) return false;
            memcpy(out + o, in + i, len); i += len; o += len;
            continue;
        }
        if ((tag & 3) == 1) { if (i >= n) return false; len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i++]; }
        else if ((tag & 3) == 2) { if (i + 2 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8); i += 2; }
        else { if (i + 4 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4; }
        if (!off || off > o || o + len > ulen) return false;
        if (off >= len) memcpy(out + o, out + o - off, len);
        else for (size_t k = 0; k < len; k++) out[o + k] = out[o + k - off];     // overlapping run
        o += len;
    }
    return o == ulen;
}

// ---------------------------------------------------------------- snappy frame, streaming
// Writer side: CBOR bytes are appended; every full super-block is cut into 64 KB chunks that the team compresses.
struct FrameWriter {
    FILE *f = nullptr; int threads = 1; bool ok = true;
    uint64_t total = 0;                             // uncompressed bytes taken so far (written + pending)
    std::vector<uint8_t> cur;
    std::vector<std::vector<uint8_t>> comp;
    bool open(const char *path, int nthreads)
    {
        threads = nthreads; f = fopen(path, "wb");
        if (!f) return false;
        cur.reserve(SUPER + CHUNK);
        ok = fwrite("\xff\x06\x00\x00sNaPpY", 1, 10, f) == 10;
        return ok;
    }
    void emit(size_t nbytes)                        // compress + write cur[0 .. nbytes), keep the rest
    {
        const size_t nch = (nbytes + CHUNK - 1) / CHUNK;
        if (comp.size() < nch) comp.resize(nch);
        std::vector<uint32_t> crc(nch);
        parallel_for(nch, threads, [&](size_t c) {
            const size_t off = c * CHUNK, n = std::min(CHUNK, nbytes - off);
            comp[c].clear();
            snappy_compress_block(cur.data() + off, n, comp[c]);
            crc[c] = mask_crc(crc32c(cur.data() + off, n));
        });
        for (size_t c = 0; c < nch && ok; c++) {
            const size_t off = c * CHUNK, n = std::min(CHUNK, nbytes - off);
            const bool raw = comp[c].size() >= n - n / 8;           // snap's rule of thumb: store incompressible chunks raw
            const uint8_t *payload = raw ? cur.data() + off : comp[c].data();
            const size_t plen = raw ? n : comp[c].size(), clen = plen + 4;
            uint8_t hdr[8] = {(uint8_t)(raw ? 1 : 0), (uint8_t)clen, (uint8_t)(clen >> 8), (uint8_t)(clen >> 16),
                              (uint8_t)crc[c], (uint8_t)(crc[c] >> 8), (uint8_t)(crc[c] >> 16), (uint8_t)(crc[c] >> 24)};
            ok = fwrite(hdr, 1, 8, f) == 8 && fwrite(payload, 1, plen, f) == plen;
        }
        cur.erase(cur.begin(), cur.begin() + (ptrdiff_t)nbytes);
    }
    void flush_all() { if (!cur.empty()) emit(cur.size()); }      // only at a multiple of CHUNK, or at the end
    void append(const uint8_t *p, size_t n)
    {
        total += n;
        while (n) {
            if (cur.size() >= SUPER) emit(cur.size() / CHUNK * CHUNK);
            const size_t take = std::min(n, SUPER - cur.size());
            cur.insert(cur.end(), p, p + take); p += take; n -= take;
        }
    }
    uint8_t *grow(size_t n)                          // room for n more bytes at the end of the pending buffer (n <= SUPER)
    {
        if (cur.size() + n > SUPER) emit(cur.size() / CHUNK * CHUNK);
        total += n;
        const size_t at = cur.size();
        cur.resize(at + n);
        return cur.data() + at;
    }
    bool close()
    {
        if (f) { if (ok && !cur.empty()) emit(cur.size()); ok = (fclose(f) == 0) && ok; f = nullptr; }
        return ok;
    }
    ~FrameWriter() { if (f) fclose(f); }
};

// Reader side: the compressed file is read whole (it is small next to the CBOR it expands to), its chunk directory is
// scanned once, and super-blocks of chunks are decompressed + checked by the team on demand.
// the compressed file, mapped (no copy: the chunks are read by the decoders -- or copied to the device -- straight from the page cache)
struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0;
    ~Mapped() { if (p && n) munmap((void *)p, n); }
    bool open(const char *path, bool populate = true)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); return false; }
        n = (size_t)sb.st_size;
        if (n) { void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) { ::close(fd); n = 0; return false; } p = (const uint8_t *)m; }
        ::close(fd);
        // the chunk walk touches every page of the file (a chunk of this data is about one page long): map them from several
        // threads first instead of taking 700 000 faults one after another (2.8 GB: 0.15-0.27 s -> measured in profiles/)
        if (populate && n >= (64u << 20)) {
            const int T = 8;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([this, t]() {
                const size_t a = n / T * t, b = t + 1 == T ? n : n / T * (t + 1);
#ifdef MADV_POPULATE_READ
                if (madvise((void *)(p + (a & ~(size_t)4095)), b - (a & ~(size_t)4095), MADV_POPULATE_READ) == 0) return;
#endif
                volatile uint8_t sink = 0;
                for (size_t o = a; o < b; o += 4096) sink = sink + p[o];
            });
            for (auto &x : th) x.join();
        }
        return true;
    }
    size_t size() const { return n; }
    const uint8_t *data() const { return p; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
};
struct FrameReader {
    Mapped raw;
    typedef SkfChunk Chunk;
    std::vector<Chunk> chunks;
    size_t next_chunk = 0; int threads = 1;
    std::vector<uint8_t> buf; size_t pos = 0;       // decoded bytes not yet consumed: buf[pos ..)
    uint64_t ubase = 0, total_ulen = 0;             // stream offset of buf[0]; length of the whole uncompressed stream
    uint64_t upos() const { return ubase + pos; }
    // The chunk walk can run on its own thread (open_async): `chunks` is reserved up front, so published entries never move, and
    // readers wait for the entries they need -- the device starts decoding the first rows while the walk is still far from the
    // end of the file.  walk_state: 0 finished, 1 running, 2 failed (err says why).
    std::atomic<size_t> n_pub{0};
    std::atomic<int> walk_state{0};
    std::thread walker;
    ~FrameReader() { if (walker.joinable()) walker.join(); }
    size_t published() const { return n_pub.load(std::memory_order_acquire); }
    bool walking() const { return walk_state.load(std::memory_order_acquire) == 1; }
    bool wait_for(size_t idx)                       // chunk idx published?  (false: the walk ended before it)
    {
        for (;;) {
            if (published() > idx) return true;
            if (!walking()) return published() > idx;
            usleep(50);
        }
    }
    bool wait_covering(uint64_t u)                  // until a published chunk ends beyond stream offset u, or the walk is over
    {
        for (;;) {
            const size_t n = published();
            if (n && chunks[n - 1].uoff + chunks[n - 1].ulen > u) return true;
            if (!walking()) { const size_t m = published(); return m && chunks[m - 1].uoff + chunks[m - 1].ulen > u; }
            usleep(50);
        }
    }
    bool seek(uint64_t u)                           // continue reading at stream offset u
    {
        (void)wait_covering(u);
        if (!walking() && u > total_ulen) return false;
        size_t lo = 0, hi = published();
        const size_t n = hi;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (chunks[mid].uoff + chunks[mid].ulen <= u) lo = mid + 1; else hi = mid; }
        buf.clear(); pos = 0; next_chunk = lo;
        ubase = lo < n ? chunks[lo].uoff : total_ulen;
        if (u > ubase) { if (!fill((size_t)(u - ubase))) return false; pos = (size_t)(u - ubase); }
        return true;
    }
    const char *err = nullptr;
    bool too_many = false;                          // the asynchronous walk ran out of reserved directory entries (not a format error)

    bool walk()                                     // the file's chunk directory; publishes as it goes
    {
        size_t i = 0; bool seen = false; uint64_t tot = 0;
        const size_t cap = chunks.capacity();
        while (i < raw.size()) {
            if (i + 4 > raw.size()) { err = "skf: truncated frame"; return false; }
            const uint8_t type = raw[i]; const size_t len = raw[i + 1] | ((size_t)raw[i + 2] << 8) | ((size_t)raw[i + 3] << 16);
            i += 4;
            if (i + len > raw.size()) { err = "skf: truncated frame"; return false; }
            if (type == 0xff) { if (len != 6 || memcmp(&raw[i], "sNaPpY", 6)) { err = "skf: not a snappy stream"; return false; } seen = true; }
            else if (type <= 0x01) {
                if (!seen || len < 4) { err = "skf: bad chunk"; return false; }
                Chunk c; c.off = i + 4; c.len = len - 4; c.compressed = type == 0x00; memcpy(&c.crc, &raw[i], 4);
                c.ulen = c.compressed ? snappy_ulen(&raw[c.off], c.len, nullptr) : (uint32_t)c.len;
                if (c.ulen == 0xFFFFFFFFu || c.ulen > CHUNK) { err = "skf: corrupt snappy block"; return false; }
                c.uoff = tot; tot += c.ulen;
                // asynchronous walk: entries must not move.  A valid file with smaller chunks than this writer's (another snappy framer,
                // flushed writes) can have more of them than were reserved: the streaming load hands such a file to the general reader
                if (cap && chunks.size() == cap) { err = "skf: more chunks than reserved"; too_many = true; return false; }
                chunks.push_back(c);
                if ((chunks.size() & 255u) == 0) n_pub.store(chunks.size(), std::memory_order_release);
            } else if (type < 0x80) { err = "skf: unsupported chunk type"; return false; }
            i += len;
        }
        if (!seen) { err = "skf: not a snappy stream"; return false; }
        total_ulen = tot;
        n_pub.store(chunks.size(), std::memory_order_release);
        return true;
    }
    bool open(const char *path, int nthreads)
    {
        threads = nthreads;
        if (!raw.open(path)) { err = "open"; return false; }
        const bool ok = walk();
        n_pub.store(chunks.size(), std::memory_order_release);
        return ok;
    }
    bool open_async(const char *path, int nthreads)
    {
        threads = nthreads;
        if (!raw.open(path, false)) { err = "open"; return false; }
        chunks.reserve(raw.size() / 512 + 4096);         // virtual until used; a chunk of 64 KB rarely compresses below 3 KB
        walk_state.store(1, std::memory_order_release);
        walker = std::thread([this]() { const bool ok = walk(); walk_state.store(ok ? 0 : 2, std::memory_order_release); });
        return true;
    }
    size_t avail() const { return buf.size() - pos; }
    const uint8_t *data() const { return buf.data() + pos; }
    void consume(size_t n) { pos += n; }
    // make at least `need` bytes available (fewer only at end of stream); decodes up to one super-block more
    double fill_secs = 0;
    size_t fill_limit = 0;                          // decoded bytes per fill (0: one super-block); small when only a header is wanted
    bool fill(size_t need)
    {
        const auto t_f0 = std::chrono::steady_clock::now();
        struct Acc { double &a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{fill_secs, t_f0};
        while (avail() < need && wait_for(next_chunk)) {
            if (pos) { buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)pos); ubase += pos; pos = 0; }
            size_t last = next_chunk, add = 0;
            const size_t lim = fill_limit ? std::max(fill_limit, need) : SUPER;
            if (!fill_limit && walking()) (void)wait_for(next_chunk + SUPER / CHUNK);      // a whole super-block's chunks, if they are coming
            const size_t n_now = published();
            while (last < n_now && add < lim) add += chunks[last++].ulen;
            const size_t base = buf.size();
            buf.resize(base + add);
            std::vector<size_t> at(last - next_chunk);
            size_t o = base;
            for (size_t c = next_chunk; c < last; c++) { at[c - next_chunk] = o; o += chunks[c].ulen; }
            std::atomic<int> bad{0};
            parallel_for(last - next_chunk, threads, [&](size_t j) {
                const Chunk &c = chunks[next_chunk + j];
                uint8_t *dst = buf.data() + at[j];
                if (c.compressed) { if (!snappy_uncompress_block(&raw[c.off], c.len, dst, c.ulen)) { bad = 1; return; } }
                else memcpy(dst, &raw[c.off], c.len);
                if (mask_crc(crc32c(dst, c.ulen)) != c.crc) bad = 2;
            });
            if (bad) { err = bad == 1 ? "skf: corrupt snappy block" : "skf: checksum mismatch"; return false; }
            next_chunk = last;
        }
        return true;
    }
};

// ---------------------------------------------------------------- CBOR (RFC 8949, definite lengths only)
struct Writer {
    std::vector<uint8_t> b;
    void head(int major, uint64_t v)
    {
        uint8_t m = (uint8_t)(major << 5);
        if (v < 24) b.push_back(m | (uint8_t)v);
        else if (v <= 0xFF) { b.push_back(m | 24); b.push_back((uint8_t)v); }
        else if (v <= 0xFFFF) { b.push_back(m | 25); b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }
        else if (v <= 0xFFFFFFFFull) { b.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
        else { b.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
    }
    void text(const std::string &s) { head(3, s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void key(const skx_key &kk)
    {
        if (!kk.hi) { head(0, kk.lo); return; }
        uint8_t be[16]; int nb = 0;
        for (int s = 56; s >= 0; s -= 8) { uint8_t x = (uint8_t)(kk.hi >> s); if (nb || x) be[nb++] = x; }
        for (int s = 56; s >= 0; s -= 8) be[nb++] = (uint8_t)(kk.lo >> s);
        head(6, 2); head(2, (uint64_t)nb); b.insert(b.end(), be, be + nb);
    }
};
// pull parser over the frame reader
struct Reader {
    FrameReader &fr; bool ok = true;
    explicit Reader(FrameReader &r) : fr(r) {}
    bool need(size_t n) { if (fr.avail() < n && !fr.fill(n)) return ok = false; if (fr.avail() < n) return ok = false; return true; }
    bool head(int &major, uint64_t &v)
    {
        if (!need(1)) return false;
        const uint8_t c = fr.data()[0]; major = c >> 5; const uint8_t ai = c & 31;
        if (ai < 24) { v = ai; fr.consume(1); return true; }
        const int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : 0;
        if (!nb || !need(1 + (size_t)nb)) return ok = false;
        v = 0; for (int k = 0; k < nb; k++) v = (v << 8) | fr.data()[1 + k];
        fr.consume(1 + (size_t)nb);
        return true;
    }
    uint64_t uint() { int m; uint64_t v = 0; if (!head(m, v) || m != 0) ok = false; return v; }
    uint64_t array() { int m; uint64_t v = 0; if (!head(m, v) || m != 4) ok = false; return v; }
    std::string text()
    {
        int m; uint64_t v = 0;
        if (!head(m, v) || m != 3 || v > (1u << 20) || !need(v)) { ok = false; return {}; }
        std::string s((const char *)fr.data(), v); fr.consume(v);
        return s;
    }
    bool boolean() { if (!need(1)) return false; const uint8_t c = fr.data()[0]; if (c != 0xf4 && c != 0xf5) { ok = false; return false; } fr.consume(1); return c == 0xf5; }
    skx_key key()
    {
        skx_key k{0, 0}; int m; uint64_t v;
        if (!head(m, v)) return k;
        if (m == 0) { k.lo = v; return k; }
        if (m == 6 && v == 2 && head(m, v) && m == 2 && v <= 16 && need(v)) {
            unsigned __int128 x = 0; for (uint64_t t = 0; t < v; t++) x = (x << 8) | fr.data()[t];
            fr.consume(v);
            k.lo = (uint64_t)x; k.hi = (uint64_t)(x >> 64); return k;
        }
        ok = false; return k;
    }
};

}  // namespace

int skf_read_stream(const char *path, SkfMeta &m, std::vector<skx_key> &keys, std::vector<uint64_t> &counts,
                    const std::function<int(uint64_t, uint64_t)> &begin_rows, const RowSink &sink, int threads, const DevDecode *dev)
{
    threads = n_workers(threads);
    FrameReader fr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_open = now();
    const bool opened = fr.open(path, threads);
    phase_add("load.map_file_chunk_directory", secs(t_open, now()));
    auto t_mark = now();
    if (!opened) {
        if (fr.err && !strcmp(fr.err, "open")) { set_error("cannot open %s", path); return SKX_EIO; }
        set_error("%s", fr.err ? fr.err : "skf: read failed"); return SKX_EFORMAT;
    }
    Reader rd(fr);
    int mj; uint64_t nf = 0;
    if (!rd.head(mj, nf) || mj != 5) { set_error("%s", fr.err ? fr.err : "skf: CBOR decode failed"); return SKX_EFORMAT; }
    uint64_t dim0 = 0, dim1 = 0; bool have_var = false;
    for (uint64_t fidx = 0; fidx < nf && rd.ok; fidx++) {
        const std::string name = rd.text();
        if (name == "k") m.k = (int)rd.uint();
        else if (name == "rc") m.rc = rd.boolean();
        else if (name == "names") { uint64_t n = rd.array(); for (uint64_t j = 0; j < n && rd.ok; j++) m.names.push_back(rd.text()); }
        else if (name == "split_kmers") {
            uint64_t n = rd.array();
            if (n > (1ull << 40)) { rd.ok = false; continue; }
            phase_add("load.dbg_before_keys", secs(t_mark, now()));
            { const auto tq = now(); keys.resize(n); phase_add("load.dbg_keys_resize", secs(tq, now())); }
            double team_s = 0, generic_s = 0; uint64_t generic_n = 0;
            // nearly every 64-bit split k-mer is a 9-byte uint (0x1b + 8 bytes): whole runs of those are decoded by the team,
            // anything else one at a time by the generic parser
            uint64_t j = 0;
            // keys of any width straight from the buffered bytes: a uint of 1 to 9 bytes, or tag 2 + a byte string of up to 16 bytes (what serde
            // writes for a u128 above 2^64: lib.rs:592-622) -- the forms Reader::key takes, without its per-key calls (128-bit lists went through
            // those one key at a time: 0.8 s of a 4.1 s `ska merge` of four 5 M-key files, profiles/r06d_reads_1000.log).  Stops at anything
            // else and 19 bytes before the buffer's end; returns the keys taken.
            auto inline_keys = [&](uint64_t limit) -> uint64_t {
                const uint8_t *p = fr.data(); const size_t av = fr.avail(); size_t o = 0; uint64_t got = 0;
                while (got < limit && o + 19 <= av) {
                    const uint8_t c = p[o];
                    uint64_t lo = 0, hi = 0;
                    if (c == 0x1b) { uint64_t v; memcpy(&v, p + o + 1, 8); lo = __builtin_bswap64(v); o += 9; }
                    else if (c == 0xC2) {
                        const uint8_t b = p[o + 1];
                        if (b < 0x40 || b > 0x50) break;
                        const size_t len = (size_t)(b - 0x40);
                        unsigned __int128 x = 0;
                        for (size_t t = 0; t < len; t++) x = (x << 8) | p[o + 2 + t];
                        lo = (uint64_t)x; hi = (uint64_t)(x >> 64); o += 2 + len;
                    }
                    else if (c < 0x18) { lo = c; o += 1; }
                    else if (c == 0x18) { lo = p[o + 1]; o += 2; }
                    else if (c == 0x19) { lo = ((uint64_t)p[o + 1] << 8) | p[o + 2]; o += 3; }
                    else if (c == 0x1a) { uint32_t v; memcpy(&v, p + o + 1, 4); lo = __builtin_bswap32(v); o += 5; }
                    else break;
                    keys[j + got] = skx_key{lo, hi};
                    got++;
                }
                fr.consume(o);
                return got;
            };
            bool wide_list = false;                                                       // a run that was not all 9-byte uints has been seen: no more team passes
            while (j < n && rd.ok) {
                const uint64_t can = wide_list ? 0 : std::min<uint64_t>(n - j, fr.avail() / 9);
                if (can >= 4096) {
                    const uint8_t *src = fr.data();
                    const size_t parts = (size_t)std::min<uint64_t>((uint64_t)threads, can / 1024);
                    const uint64_t per = can / parts;
                    std::atomic<int> other{0};
                    const auto tq = now();
                    parallel_for(parts, threads, [&](size_t pt) {
                        const uint64_t a = pt * per, b = pt + 1 == parts ? can : a + per;
                        uint8_t bad = 0;
                        for (uint64_t c = a; c < b; c++) {
                            const uint8_t *q = src + 9 * c;
                            bad |= (uint8_t)(q[0] ^ 0x1b);
                            uint64_t v; memcpy(&v, q + 1, 8);
                            keys[j + c] = skx_key{__builtin_bswap64(v), 0};
                        }
                        if (bad) other = 1;
                    });
                    team_s += secs(tq, now());
                    if (!other) { fr.consume(9 * can); j += can; continue; }
                    wide_list = true;                                                     // a run with shorter / wider keys in it
                }
                { const auto tq = now();
                  const uint64_t got = inline_keys(n - j);
                  j += got;
                  if (j < n && got == 0) { keys[j++] = rd.key(); generic_n++; }           // near the buffer's end, or a form the inline parser leaves: the generic one (it refills)
                  generic_s += secs(tq, now()); }
            }
            phase_add("load.dbg_keys_team", team_s); phase_add("load.dbg_keys_generic", generic_s); phase_add("load.dbg_fill", fr.fill_secs);
            if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] generic keys %llu\n", (unsigned long long)generic_n);
        }
        else if (name == "variants") {
            uint64_t n3 = 0; if (!rd.head(mj, n3) || mj != 5) rd.ok = false;
            for (uint64_t g = 0; g < n3 && rd.ok; g++) {
                const std::string sub = rd.text();
                if (sub == "v") rd.uint();
                else if (sub == "dim") { if (rd.array() != 2) rd.ok = false; dim0 = rd.uint(); dim1 = rd.uint(); }
                else if (sub == "data") {
                    const uint64_t n = rd.array();
                    if (!rd.ok || n != dim0 * dim1) { rd.ok = false; break; }          // ndarray serde writes dim before data
                    m.n_rows = dim0;
                    int r = begin_rows(dim0, dim1);
                    if (r != SKX_OK) return r;
                    // cells are CBOR uints: one byte below 24, else 0x18 + the byte (every base letter): the usual case is
                    // extracted by the whole team, two bytes per cell; anything else falls back to the generic decoder
                    const uint64_t S = dim1;
                    const uint64_t block_rows = S ? std::max<uint64_t>(1, (uint64_t)(SUPER / 2) / S) : 1;
                    std::vector<uint8_t> rows;
                    uint64_t row0 = 0;
                    if (dev && device_section(2 * n) && fr.upos() + 2 * n <= fr.total_ulen) {
                        // the whole section on the device: compressed chunks in, sample-major matrix out
                        const uint64_t upos = fr.upos();
                        phase_add("load.header_split_kmers", secs(t_mark, now()));
                        t_mark = now();
                        r = (*dev)(fr.raw.data(), fr.chunks.data(), fr.chunks.size(), upos, dim0, dim1);
                        phase_add("load.data_section_device", secs(t_mark, now()));
                        t_mark = now();
                        if (r == SKX_OK) { if (!fr.seek(upos + 2 * n)) rd.ok = false; row0 = dim0; }
                        else if (r != SKF_NOT_TAKEN) return r;
                    }
                    while (row0 < dim0 && rd.ok) {
                        const uint64_t nr = std::min(block_rows, dim0 - row0), cells = nr * S;
                        rows.resize(cells);
                        uint64_t done = 0;
                        while (done < cells && rd.ok) {
                            if (fr.avail() < 2 && !rd.need(1)) break;
                            const uint64_t can = std::min<uint64_t>(cells - done, fr.avail() / 2);
                            const uint8_t *src = fr.data();
                            uint64_t fast = 0;
                            if (can >= 4096) {
                                const size_t parts = (size_t)std::min<uint64_t>((uint64_t)threads, can / 4096);
                                const uint64_t per = can / parts;
                                std::atomic<int> odd{0};
                                parallel_for(parts, threads, [&](size_t pt) {
                                    const uint64_t a = pt * per, b = pt + 1 == parts ? can : a + per;
                                    uint8_t bad = 0;
                                    for (uint64_t c = a; c < b; c++) { bad |= (uint8_t)(src[2 * c] ^ 0x18); rows[done + c] = src[2 * c + 1]; }
                                    if (bad) odd = 1;
                                });
                                if (!odd) fast = can;
                            }
                            if (fast) { fr.consume(2 * fast); done += fast; continue; }
                            // generic cell by cell (also the tail of a block)
                            const uint64_t lim = std::min<uint64_t>(cells - done, std::max<uint64_t>(can, 1));
                            for (uint64_t c = 0; c < lim && rd.ok; c++) rows[done + c] = (uint8_t)rd.uint();
                            done += lim;
                        }
                        if (!rd.ok) break;
                        r = sink(row0, nr, rows.data());
                        if (r != SKX_OK) return r;
                        row0 += nr;
                    }
                    have_var = rd.ok;
                }
                else rd.ok = false;
            }
        }
        else if (name == "variant_count") { uint64_t n = rd.array(); if (n > (1ull << 40)) rd.ok = false; else counts.resize(n); for (uint64_t j = 0; j < n && rd.ok; j++) counts[j] = rd.uint(); }
        else if (name == "ska_version") m.version = rd.text();
        else if (name == "k_bits") m.k_bits = (int)rd.uint();
        else rd.ok = false;
    }
    if (!rd.ok || !have_var || dim1 != m.names.size()) { set_error("%s", fr.err ? fr.err : "skf: CBOR decode failed"); return SKX_EFORMAT; }
    return SKX_OK;
}

// ---------------------------------------------------------------- layout-only open (the streaming load of `ska align x.skf`)
struct SkfFile::Impl { FrameReader fr; };
SkfFile::SkfFile() : impl(new Impl()) {}
SkfFile::~SkfFile() { delete impl; }
const uint8_t *SkfFile::file() const { return impl->fr.raw.data(); }
const SkfChunk *SkfFile::chunks() const { return impl->fr.chunks.data(); }
size_t SkfFile::n_chunks() const { return impl->fr.published(); }
bool SkfFile::wait_chunk(size_t idx) { return impl->fr.wait_for(idx); }
int SkfFile::walk_result()
{
    FrameReader &fr = impl->fr;
    while (fr.walking()) usleep(50);
    if (fr.walk_state.load() == 2) { if (fr.too_many) return SKF_NOT_TAKEN; set_error("%s", fr.err ? fr.err : "skf: read failed"); return SKX_EFORMAT; }
    if (upos_data + 2 * m.n_rows * (uint64_t)m.names.size() > fr.total_ulen) { set_error("skf: truncated frame"); return SKX_EFORMAT; }
    return SKX_OK;
}
size_t SkfFile::chunk_of(uint64_t u)
{
    FrameReader &fr = impl->fr;
    (void)fr.wait_covering(u);
    size_t lo = 0, hi = fr.published();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (fr.chunks[mid].uoff + fr.chunks[mid].ulen <= u) lo = mid + 1; else hi = mid; }
    return lo;
}
// Header, then the split k-mer list is stepped over without being decoded: for k <= 31 it is U uints, 9 bytes each unless a key is below
// 2^32 (then the bytes at +9U are not the "variants" field and the caller takes the general reader); for k > 31 its items are walked.
int SkfFile::open(const char *path)
{
    FrameReader &fr = impl->fr;
    if (!fr.open_async(path, n_workers(0))) { set_error("cannot open %s", path); return SKX_EIO; }      // the chunk walk runs beside everything below
    PhaseTimer t_hdr("load.header");
    fr.fill_limit = 1u << 20;
    Reader rd(fr);
    int mj; uint64_t nf = 0;
    if (!rd.head(mj, nf) || mj != 5 || nf != 8) return SKF_NOT_TAKEN;
    if (rd.text() != "k") return SKF_NOT_TAKEN;
    m.k = (int)rd.uint();
    if (rd.text() != "rc") return SKF_NOT_TAKEN;
    m.rc = rd.boolean();
    if (rd.text() != "names") return SKF_NOT_TAKEN;
    { const uint64_t n = rd.array(); if (!rd.ok || n > 65535) return SKF_NOT_TAKEN; for (uint64_t j = 0; j < n && rd.ok; j++) m.names.push_back(rd.text()); }
    if (!rd.ok || rd.text() != "split_kmers") return SKF_NOT_TAKEN;
    n_keys = rd.array();
    if (!rd.ok || n_keys > (1ull << 40)) return SKF_NOT_TAKEN;
    upos_keys = fr.upos();
    if (m.k <= 31) {
        if (!fr.seek(upos_keys + 9 * n_keys)) return SKF_NOT_TAKEN;
    } else {
        // 128-bit keys (lib.rs:592-622): uints below 2^64, tag-2 bignums of up to 16 bytes above -- variable length, so the list is walked
        // (header bytes only: ~2 ns per key) instead of jumped over
        fr.fill_limit = 0;
        uint64_t j = 0;
        while (j < n_keys) {
            const uint8_t *p = fr.data(); const size_t av = fr.avail(); size_t o = 0;
            while (j < n_keys && o + 19 <= av) {
                const uint8_t c = p[o];
                if (c == 0xC2) { const uint8_t b = p[o + 1]; if (b < 0x40 || b > 0x50) return SKF_NOT_TAKEN; o += 2 + (size_t)(b - 0x40); }
                else if (c < 0x18) o += 1;
                else if (c <= 0x1B) o += 1 + ((size_t)1 << (c - 0x18));
                else return SKF_NOT_TAKEN;
                j++;
            }
            fr.consume(o);
            if (j < n_keys && o == 0) { (void)rd.key(); if (!rd.ok) return SKF_NOT_TAKEN; j++; }      // near the end of what is buffered: the general decoder refills
        }
        fr.fill_limit = 1u << 20;
    }
    rd.ok = true;
    if (rd.text() != "variants" || !rd.ok) return SKF_NOT_TAKEN;
    uint64_t n3 = 0;
    if (!rd.head(mj, n3) || mj != 5 || n3 != 3) return SKF_NOT_TAKEN;
    if (rd.text() != "v") return SKF_NOT_TAKEN;
    rd.uint();
    if (rd.text() != "dim" || rd.array() != 2) return SKF_NOT_TAKEN;
    const uint64_t d0 = rd.uint(), d1 = rd.uint();
    if (rd.text() != "data") return SKF_NOT_TAKEN;
    const uint64_t nd = rd.array();
    if (!rd.ok || d0 != n_keys || d1 != m.names.size() || nd != d0 * d1) return SKF_NOT_TAKEN;
    m.n_rows = d0;
    upos_data = fr.upos();
    return SKX_OK;                                   // whether the stream really holds 2 nd more bytes: walk_result()
}
// what follows the data section: variant_count, ska_version, k_bits
int SkfFile::read_tail(std::vector<uint32_t> &counts)
{
    FrameReader &fr = impl->fr;
    PhaseTimer t("load.variant_counts");
    fr.fill_limit = 0;
    if (!fr.seek(upos_data + 2 * m.n_rows * (uint64_t)m.names.size())) { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    Reader rd(fr);
    if (rd.text() != "variant_count") { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    const uint64_t n = rd.array();
    if (!rd.ok || n != m.n_rows) { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    counts.resize(n);
    // 1 to 3 bytes per count while the samples number below 65 536: straight from the decoded bytes when a run is fully buffered
    uint64_t j = 0;
    while (j < n && rd.ok) {
        const uint8_t *p = fr.data(); const size_t av = fr.avail(); size_t o = 0;
        while (j < n && o + 3 <= av) {
            const uint8_t c = p[o];
            if (c < 24) { counts[j++] = c; o += 1; }
            else if (c == 24) { counts[j++] = p[o + 1]; o += 2; }
            else if (c == 25) { counts[j++] = ((uint32_t)p[o + 1] << 8) | p[o + 2]; o += 3; }
            else break;
        }
        fr.consume(o);
        if (j < n) { const uint64_t v = rd.uint(); if (v > 0xFFFFFFFFull) rd.ok = false; counts[j++] = (uint32_t)v; }
    }
    if (!rd.ok || rd.text() != "ska_version") { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    m.version = rd.text();
    if (!rd.ok || rd.text() != "k_bits") { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    m.k_bits = (int)rd.uint();
    if (!rd.ok) { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    return SKX_OK;
}

int skf_write_stream(const char *path, const SkfMeta &m, const std::vector<skx_key> &keys, const std::vector<uint64_t> &counts,
                     const RowFetch &fetch, int threads, const DevEncode *dev, const SkfFastSections *fast)
{
    threads = n_workers(threads);
    FrameWriter fw;
    if (!fw.open(path, threads)) { set_error("cannot create %s", path); return SKX_EIO; }
    const uint64_t S = m.names.size(), U = m.n_rows;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto t_mark = now();
    Writer w;
    auto flush = [&]() { fw.append(w.b.data(), w.b.size()); w.b.clear(); };
    w.head(5, 8);
    w.text("k"); w.head(0, (uint64_t)m.k);
    w.text("rc"); w.b.push_back(m.rc ? 0xf5 : 0xf4);
    w.text("names"); w.head(4, S); for (auto &s : m.names) w.text(s);
    const bool fast_keys = fast && fast->keys_cbor;
    w.text("split_kmers"); w.head(4, fast_keys ? fast->n_keys : keys.size());
    flush();
    if (fast_keys) fw.append(fast->keys_cbor, fast->keys_cbor_len);
    for (size_t i0 = 0; !fast_keys && i0 < keys.size(); i0 += (4u << 20)) {              // slices of the list are encoded by the team, then appended in order
        const size_t cnt = std::min<size_t>(4u << 20, keys.size() - i0), parts = std::max<size_t>(1, std::min<size_t>((size_t)threads, cnt / 4096));
        std::vector<Writer> ws(parts);
        parallel_for(parts, threads, [&](size_t pt) {
            const size_t a = i0 + cnt * pt / parts, b = i0 + cnt * (pt + 1) / parts;
            ws[pt].b.reserve((b - a) * 9 + 32);
            for (size_t i = a; i < b; i++) ws[pt].key(keys[i]);
        });
        for (auto &x : ws) fw.append(x.b.data(), x.b.size());
    }
    w.text("variants"); w.head(5, 3);
    w.text("v"); w.head(0, 1);
    w.text("dim"); w.head(4, 2); w.head(0, U); w.head(0, S);
    w.text("data"); w.head(4, U * S);
    flush();
    phase_add("save.header_split_kmers", secs(t_mark, now()));
    t_mark = now();
    const uint64_t block_rows = S ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(SUPER / 2) / S, SUPER / (2 * S) ? SUPER / (2 * S) : 1)) : 1;
    std::vector<uint8_t> rows;
    uint64_t first_row = 0;
    if (dev && S && device_section(2 * U * S)) {
        // whole 64 KB chunks inside the section are produced on the device; the ragged ends (< 64 KB each) here
        const uint64_t upos = fw.total, bytes = 2 * U * S;
        const uint64_t A = (upos + CHUNK - 1) / CHUNK * CHUNK, Z = (upos + bytes) / CHUNK * CHUNK;
        auto host_bytes = [&](uint64_t b0, uint64_t b1) -> int {           // section bytes [b0, b1)
            if (b0 >= b1) return SKX_OK;
            const uint64_t c0 = b0 / 2, c1 = (b1 + 1) / 2, r0 = c0 / S, r1 = (c1 - 1) / S;
            rows.resize((r1 - r0 + 1) * S);
            int r = fetch(r0, r1 - r0 + 1, rows.data());
            if (r != SKX_OK) return r;
            std::vector<uint8_t> tmp(b1 - b0);
            for (uint64_t q = b0; q < b1; q++) tmp[q - b0] = (q & 1) ? rows[(q >> 1) - r0 * S] : 0x18;
            fw.append(tmp.data(), tmp.size());
            return SKX_OK;
        };
        if (Z > A) {
            int r = host_bytes(0, A - upos);
            if (r != SKX_OK) return r;
            fw.flush_all();
            if (!fw.ok) { set_error("short write %s", path); return SKX_EIO; }
            r = (*dev)(fw.f, upos, A, (Z - A) / CHUNK);
            if (r != SKX_OK) return r;
            fw.total = Z;
            r = host_bytes(Z - upos, bytes);
            if (r != SKX_OK) return r;
            first_row = U;
        }
    }
    for (uint64_t row0 = first_row; row0 < U && fw.ok; row0 += block_rows) {
        const uint64_t nr = std::min(block_rows, U - row0), cells = nr * S;
        rows.resize(cells);
        int r = fetch(row0, nr, rows.data());
        if (r != SKX_OK) return r;
        if (2 * cells <= SUPER) {
            // every cell the engine holds is a letter or '-' (>= 24): two bytes per cell, written in place by the team
            uint8_t *dst = fw.grow(2 * cells);
            const size_t parts = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, cells / 65536));
            const uint64_t per = cells / parts;
            std::atomic<int> small{0};
            parallel_for(parts, threads, [&](size_t pt) {
                const uint64_t a = pt * per, b = pt + 1 == parts ? cells : a + per;
                uint8_t lo = 0xFF;
                for (uint64_t c = a; c < b; c++) { const uint8_t v = rows[c]; lo = std::min(lo, v); dst[2 * c] = 0x18; dst[2 * c + 1] = v; }
                if (lo < 24) small = 1;
            });
            if (!small) continue;
            fw.cur.resize(fw.cur.size() - 2 * cells); fw.total -= 2 * cells;       // a value below 24 is a single byte: redo this block generically
        }
        for (uint64_t c = 0; c < cells; c++) { w.head(0, rows[c]); if (w.b.size() >= (1u << 20)) flush(); }
        flush();
    }
    phase_add("save.data_section", secs(t_mark, now()));
    t_mark = now();
    if (fast && fast->counts32) {
        w.text("variant_count"); w.head(4, fast->n_counts);
        flush();
        const size_t cnt = fast->n_counts, parts = std::max<size_t>(1, std::min<size_t>((size_t)threads, cnt / 65536));
        std::vector<Writer> ws(parts);
        parallel_for(parts, threads, [&](size_t pt) {
            const size_t a = cnt * pt / parts, b = cnt * (pt + 1) / parts;
            ws[pt].b.reserve((b - a) * 3 + 16);
            for (size_t i = a; i < b; i++) ws[pt].head(0, fast->counts32[i]);
        });
        for (auto &x : ws) fw.append(x.b.data(), x.b.size());
    } else {
        w.text("variant_count"); w.head(4, counts.size());
        for (size_t i = 0; i < counts.size(); i++) { w.head(0, counts[i]); if (w.b.size() >= (1u << 20)) flush(); }
    }
    w.text("ska_version"); w.text(m.version);
    w.text("k_bits"); w.head(0, (uint64_t)m.k_bits);
    flush();
    if (!fw.close()) { set_error("short write %s", path); return SKX_EIO; }
    phase_add("save.counts_close", secs(t_mark, now()));
    return SKX_OK;
}

// the `k` of a file, from its first fields only (0 if the split k-mer list comes first or the file cannot be read): what width its keys need
// without loading it
int skf_peek_k(const char *path)
{
    FrameReader fr;
    if (!fr.open(path, 1)) return 0;
    Reader rd(fr);
    int mj; uint64_t nf = 0;
    if (!rd.head(mj, nf) || mj != 5) return 0;
    for (uint64_t f = 0; f < nf && rd.ok; f++) {
        const std::string name = rd.text();
        if (name == "k") { const uint64_t k = rd.uint(); return rd.ok && k < 1000 ? (int)k : 0; }
        else if (name == "rc") rd.boolean();
        else if (name == "names") { uint64_t n = rd.array(); for (uint64_t j = 0; j < n && rd.ok; j++) rd.text(); }
        else return 0;
    }
    return 0;
}

// whole-array forms (small inputs, tests)
int skf_read(const char *path, SkfData &d)
{
    SkfMeta m; uint64_t S = 0;
    int r = skf_read_stream(path, m, d.keys, d.counts,
                            [&](uint64_t U, uint64_t cols) { S = cols; d.variants.resize(U * S); return SKX_OK; },
                            [&](uint64_t row0, uint64_t nr, const uint8_t *src) { memcpy(d.variants.data() + row0 * S, src, nr * S); return SKX_OK; }, 0);
    if (r != SKX_OK) return r;
    d.k = m.k; d.rc = m.rc; d.k_bits = m.k_bits; d.names = m.names; d.version = m.version; d.n_rows = m.n_rows;
    return SKX_OK;
}
int skf_write(const char *path, const SkfData &d)
{
    SkfMeta m; m.k = d.k; m.rc = d.rc; m.k_bits = d.k_bits; m.names = d.names; m.version = d.version; m.n_rows = d.n_rows;
    const uint64_t S = d.names.size();
    return skf_write_stream(path, m, d.keys, d.counts,
                            [&](uint64_t row0, uint64_t nr, uint8_t *dst) { memcpy(dst, d.variants.data() + row0 * S, nr * S); return SKX_OK; }, 0);
}

}  // namespace skx
