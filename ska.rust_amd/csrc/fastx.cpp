// fastx.cpp -- the engine's FASTA/FASTQ(.gz) reader: replaces the needletail record iterator used at
// ska_dict.rs:131-153,356-366.  Produces the *record stream* the device kernels consume: each record's bases
// with line breaks removed, followed by one '\n' ('\n' can never occur inside a record, so it doubles as the
// record terminator the end-of-record rule of split_kmer.rs:89 needs).  For FASTQ a parallel quality stream
// is produced.  Format is decided by the first byte of file 1 ('>' / '@'), as ska_dict.rs:356-366 does via the
// first record; file 2 is parsed with the same mode.  `proportion_reads` keeps record n iff n % round(1/p) == 0
// (ska_dict.rs:125-141), per file.
#include "skx_internal.h"
#include <cmath>
#include <functional>
#include <cstring>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <memory>
#include <dlfcn.h>

namespace skx {

// bzip2 / xz / zstd inputs (needletail's `compression` feature, Cargo.toml:32, takes them besides gzip): inflated through the system's
// libbz2 / liblzma / libzstd, opened at run time and bound by their published C interfaces (the image has the libraries, not their headers).
// A file of one of these kinds whose library is missing is refused with the reference's message instead of being parsed as text.
namespace {
struct BzStream { char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32; char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
                  void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque; };
struct LzmaStream { const uint8_t *next_in; size_t avail_in; uint64_t total_in; uint8_t *next_out; size_t avail_out; uint64_t total_out;
                    const void *allocator; void *internal; void *rp1, *rp2, *rp3, *rp4; uint64_t ri1, ri2; size_t ri3, ri4; int re1, re2; };
struct ZstdIn { const void *src; size_t size, pos; };
struct ZstdOut { void *dst; size_t size, pos; };
void *open_lib(const char *const *names) { for (; *names; names++) if (void *h = dlopen(*names, RTLD_NOW | RTLD_LOCAL)) return h; return nullptr; }
void grow(std::vector<uint8_t> &out, size_t used) { if (out.size() - used < (1u << 20)) out.resize(out.size() * 2 + (8u << 20)); }
// 0 = done, 1 = the library (or a symbol) is missing, 2 = the data is bad
int inflate_bz2(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    static const char *const L[] = {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so", nullptr};
    static void *h = open_lib(L);
    if (!h) return 1;
    auto init = (int (*)(BzStream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
    auto run = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompress");
    auto end = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompressEnd");
    if (!init || !run || !end) return 1;
    size_t ipos = 0, used = 0;
    out.clear();
    while (ipos < in.size()) {                                     // concatenated streams, as bzip2 itself reads them
        BzStream z{};
        if (init(&z, 0, 0) != 0) return 2;
        for (;;) {
            grow(out, used);
            z.next_in = (char *)in.data() + ipos; z.avail_in = (unsigned)std::min<size_t>(in.size() - ipos, 1u << 30);
            z.next_out = (char *)out.data() + used; z.avail_out = (unsigned)std::min<size_t>(out.size() - used, 1u << 30);
            const unsigned ai = z.avail_in, ao = z.avail_out;
            const int r = run(&z);
            ipos += ai - z.avail_in; used += ao - z.avail_out;
            if (r == 4) break;                                     // BZ_STREAM_END
            if (r != 0 || (ai == z.avail_in && ao == z.avail_out)) { end(&z); return 2; }
        }
        end(&z);
    }
    out.resize(used);
    return 0;
}
int inflate_xz(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    static const char *const L[] = {"liblzma.so.5", "liblzma.so", nullptr};
    static void *h = open_lib(L);
    if (!h) return 1;
    auto dec = (int (*)(LzmaStream *, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
    auto code = (int (*)(LzmaStream *, int))dlsym(h, "lzma_code");
    auto end = (void (*)(LzmaStream *))dlsym(h, "lzma_end");
    if (!dec || !code || !end) return 1;
    LzmaStream z{};
    if (dec(&z, ~0ull, 0x08u /* LZMA_CONCATENATED */) != 0) return 2;
    size_t used = 0;
    out.clear();
    z.next_in = in.data(); z.avail_in = in.size();
    for (;;) {
        grow(out, used);
        z.next_out = out.data() + used; z.avail_out = out.size() - used;
        const size_t ao = z.avail_out;
        const int r = code(&z, z.avail_in ? 0 /* LZMA_RUN */ : 3 /* LZMA_FINISH */);
        used += ao - z.avail_out;
        if (r == 1) break;                                         // LZMA_STREAM_END
        if (r != 0) { end(&z); return 2; }
    }
    end(&z);
    out.resize(used);
    return 0;
}
int inflate_zstd(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    static const char *const L[] = {"libzstd.so.1", "libzstd.so", nullptr};
    static void *h = open_lib(L);
    if (!h) return 1;
    auto mk = (void *(*)())dlsym(h, "ZSTD_createDStream");
    auto fr = (size_t (*)(void *))dlsym(h, "ZSTD_freeDStream");
    auto run = (size_t (*)(void *, ZstdOut *, ZstdIn *))dlsym(h, "ZSTD_decompressStream");
    auto bad = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    if (!mk || !fr || !run || !bad) return 1;
    void *d = mk();
    if (!d) return 2;
    ZstdIn zi{in.data(), in.size(), 0};
    size_t used = 0, last = 1;
    out.clear();
    while (zi.pos < zi.size || last != 0) {
        grow(out, used);
        ZstdOut zo{out.data() + used, out.size() - used, 0};
        const size_t before = zi.pos;
        last = run(d, &zo, &zi);
        if (bad(last)) { fr(d); return 2; }
        used += zo.pos;
        if (zi.pos == zi.size && zo.pos == 0 && before == zi.pos) { if (last != 0) { fr(d); return 2; } break; }      // input ended inside a frame
    }
    fr(d);
    out.resize(used);
    return 0;
}
}  // namespace

static int slurp(const char *path, std::vector<uint8_t> &buf)
{
    // (read once: POSIX_FADV_NOREUSE keeps the kernel from promoting every page on first access -- one lock for all reader threads)
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
    // a plain (not gzip) regular file is read straight into a buffer of its size -- no inflate layer, no doubling of the buffer
    struct stat sb{};
    unsigned char magic[2] = {0, 0};
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
        const size_t n = (size_t)sb.st_size;
        if (buf.size() < n) buf.resize(n);                    // (callers keep `buf` across files: it only ever grows)
        size_t got = 0;
        while (got < n) { const ssize_t r = ::read(fd, buf.data() + got, n - got); if (r < 0 && errno == EINTR) continue; if (r <= 0) break; got += (size_t)r; }
        ::close(fd);
        buf.resize(got);
        if (got == 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
        // bzip2 ("BZh"), xz (FD 37 7A 58 5A 00), zstd (28 B5 2F FD)
        const int kind = got >= 3 && !memcmp(buf.data(), "BZh", 3) ? 1 : got >= 6 && !memcmp(buf.data(), "\xFD" "7zXZ\0", 6) ? 2 : got >= 4 && !memcmp(buf.data(), "\x28\xB5\x2F\xFD", 4) ? 3 : 0;
        if (kind) {
            std::vector<uint8_t> plain;
            const int r = kind == 1 ? inflate_bz2(buf, plain) : kind == 2 ? inflate_xz(buf, plain) : inflate_zstd(buf, plain);
            if (r != 0 || plain.empty()) { set_error("Invalid path/file: %s", path); return SKX_EIO; }      // (library missing or data bad: never parsed as text)
            buf.swap(plain);
        }
        return SKX_OK;
    }
    // a gzip file: through the reader threads' own inflater (gz_inflate.cpp)
    if (S_ISREG(sb.st_mode) && magic[0] == 0x1f && magic[1] == 0x8b && !knob("zlib_reader")) {
        GzReader zr;
        zr.open(fd);
        buf.clear();
        size_t n = 0;
        for (;;) {
            const uint8_t *p; size_t got;
            if (zr.next(&p, &got, 0) != 0) { ::close(fd); set_error("Invalid path/file: %s", path); return SKX_EIO; }      // damaged or truncated: an error, not a short input
            if (got == 0) break;
            if (buf.size() - n < got) buf.resize(std::max(buf.size() * 2, n + got + (4u << 20)));
            memcpy(buf.data() + n, p, got);
            n += got;
        }
        ::close(fd);
        buf.resize(n);
        if (n == 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
        return SKX_OK;
    }
    // what is not a regular file (a pipe cannot be looked at first), and SKX_KNOBS=zlib_reader (A/B measurements, differential tests): zlib's gzread
    gzFile g = gzdopen(fd, "rb");            // transparent for uncompressed input that is not a regular file; closes fd with gzclose
    if (!g) { ::close(fd); set_error("Invalid path/file: %s", path); return SKX_EIO; }
    gzbuffer(g, 1 << 20);
    buf.clear();
    size_t n = 0;
    for (;;) {
        if (buf.size() - n < (1u << 20)) buf.resize(buf.size() * 2 + (4u << 20));
        int r = gzread(g, buf.data() + n, (unsigned)std::min<size_t>(buf.size() - n, 1u << 30));
        if (r < 0) { gzclose(g); set_error("Invalid path/file: %s", path); return SKX_EIO; }
        if (r == 0) break;
        n += (size_t)r;
    }
    {   // a gzip stream that stops before its end (a truncated file) is an error, not a short input
        int zerr = Z_OK; (void)gzerror(g, &zerr);
        if (zerr != Z_OK && zerr != Z_STREAM_END) { gzclose(g); set_error("Invalid path/file: %s", path); return SKX_EIO; }
    }
    gzclose(g);
    buf.resize(n);
    if (n == 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
    return SKX_OK;
}

static int parse_fasta(const std::vector<uint8_t> &b, size_t step, HostStream &out)
{
    const uint8_t *p = b.data(), *end = p + b.size();
    size_t rec = 0;
    if (*p != '>') { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
    out.seq.reserve(out.seq.size() + b.size());
    while (p < end) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));     // header line
        const bool keep = rec % step == 0;
        if (keep) { const uint8_t *q = p + 1, *he = nl ? nl : end; while (q < he && *q != ' ' && *q != '\t' && *q != '\r') q++; out.ids.emplace_back((const char *)p + 1, (size_t)(q - p - 1)); }
        p = nl ? nl + 1 : end;
        while (p < end && *p != '>') {                                                // sequence lines
            nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            const uint8_t *le = nl ? nl : end;
            if (keep) for (const uint8_t *q = p; q < le; q++) if (*q != '\r') out.seq.push_back(*q);
            p = nl ? nl + 1 : end;
        }
        if (keep) out.seq.push_back('\n');
        rec++;
    }
    return SKX_OK;
}

static int parse_fastq(const std::vector<uint8_t> &b, size_t step, HostStream &out)
{
    const uint8_t *p = b.data(), *end = p + b.size();
    size_t rec = 0;
    out.seq.reserve(out.seq.size() + b.size() / 2 + 4096); out.qual.reserve(out.qual.size() + b.size() / 2 + 4096);
    auto line = [&](const uint8_t *&s, const uint8_t *&e) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
        s = p; e = nl ? nl : end; p = nl ? nl + 1 : end;
        if (e > s && e[-1] == '\r') e--;
    };
    while (p < end) {
        if (*p == '\n' || *p == '\r') { p++; continue; }
        const uint8_t *hs, *he, *ss, *se, *ps, *pe, *qs, *qe;
        line(hs, he);
        if (*hs != '@' || p >= end) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
        line(ss, se);
        if (p >= end) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
        line(ps, pe);
        if (ps == pe || *ps != '+') { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
        line(qs, qe);
        if (qe - qs != se - ss) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
        if (rec % step == 0) {
            out.seq.insert(out.seq.end(), ss, se); out.seq.push_back('\n');
            out.qual.insert(out.qual.end(), qs, qe); out.qual.push_back('\n');
        }
        rec++;
    }
    return SKX_OK;
}

// A plain FASTQ file as two streams of lines (sequence, quality; the sink appends the '\n' that ends a record), handed to `emit` line by line: nothing of
// the file's size is allocated (the reader threads of a batch of 50x isolates each took ~0.8 GB of fresh memory through the whole-file
// parser, and 32 of them together ran at a sixth of the rate of 8).  Same checks as parse_fastq.  SKF_NOT_TAKEN: not a plain FASTQ file.
static size_t newline_offsets(const uint8_t *p, size_t n, uint32_t *out, size_t cap, size_t *used);      // (below, beside the other vector code)
int stream_fastq_file(const char *path, const std::function<int(int which, const uint8_t *p, size_t n)> &emit)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
    struct Close { int fd; ~Close() { ::close(fd); } } cl{fd};
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
    // a gzip file is inflated piece by piece as it is read, by the reader thread's own inflater (gz_inflate.cpp: members one after the other, as
    // gzip reads them; the text is walked where the inflater put it); SKX_KNOBS=zlib_reader: through zlib's gzread on a descriptor of its own
    unsigned char mg[2] = {0, 0};
    gzFile gz = nullptr;
    std::unique_ptr<GzReader> zr;
    if (pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1f && mg[1] == 0x8b) {
        if (!knob("zlib_reader")) { zr.reset(new GzReader); zr->open(fd); }
        else {
            const int fd2 = dup(fd);
            if (fd2 < 0 || !(gz = gzdopen(fd2, "rb"))) { if (fd2 >= 0) ::close(fd2); set_error("Invalid path/file: %s", path); return SKX_EIO; }
            gzbuffer(gz, 1u << 20);
        }
    }
    struct GzClose { gzFile &g; ~GzClose() { if (g) gzclose(g); } } gzc{gz};
    static thread_local std::vector<uint8_t> chunk;
    constexpr size_t CH = 4u << 20;
    if (chunk.size() < CH + 1) chunk.resize(CH + 1);
    size_t have = 0, line_no = 0, seq_len = 0;          // bytes in chunk not yet consumed start at 0 (leftover of a split line is moved to the front)
    bool first = true, in_record = false;
    std::vector<uint8_t> spill;                         // a line longer than the chunk (never for reads; kept correct)
    auto take_line = [&](const uint8_t *s, const uint8_t *e) -> int {      // one line without its terminator
        if (e > s && e[-1] == '\r') e--;
        const size_t n = (size_t)(e - s);
        switch (line_no & 3u) {
        case 0:
            if (n == 0) return SKX_OK;                                      // blank line between records
            if (*s != '@') { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
            in_record = true; break;
        case 1: { seq_len = n; const int r = emit(0, s, n); if (r != SKX_OK) return r; break; }
        case 2: if (n == 0 || *s != '+') { set_error("Invalid FASTA/Q record"); return SKX_EIO; } break;
        default: {
            if (n != seq_len) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
            const int r = emit(1, s, n); if (r != SKX_OK) return r;
            in_record = false; break; }
        }
        line_no++;
        return SKX_OK;
    };
    // the complete lines of [p, end): line ends found in bulk (newline_offsets), a whole record at a time where its four lines are in hand and
    // plain (no blank line before it, no carriage returns, the lengths agree) -- two calls of the sink instead of four trips through the
    // line-by-line rules above; *rest = where the unfinished last line starts.  (Walking a mapping of the file instead of read()ing it was
    // tried: 1.4 x faster on one thread, 2.5 x slower on sixteen -- page faults and unmapping share the process's address-space lock.)
    static thread_local std::vector<uint32_t> ends;
    constexpr size_t PIECE = 1u << 20, CAP = 1u << 15;
    if (ends.size() < CAP) ends.resize(CAP);
    auto walk = [&](const uint8_t *p, const uint8_t *end, const uint8_t **rest) -> int {
        const uint8_t *line = p;
        while (p < end) {
            size_t used = 0;
            const size_t c = newline_offsets(p, std::min<size_t>(PIECE, (size_t)(end - p)), ends.data(), CAP, &used);
            for (size_t j = 0; j < c;) {
                if ((line_no & 3u) == 0 && j + 4 <= c) {
                    const uint8_t *e0 = p + ends[j], *e1 = p + ends[j + 1], *e2 = p + ends[j + 2], *e3 = p + ends[j + 3];
                    const uint8_t *sq = e0 + 1, *pl = e1 + 1, *ql = e2 + 1;
                    if (line < e0 && *line == '@' && pl < e2 && *pl == '+' && e0[-1] != '\r' && e2[-1] != '\r' && (sq == e1 || e1[-1] != '\r') &&
                        (ql == e3 || e3[-1] != '\r') && e1 - sq == e3 - ql) {
                        int rc = emit(0, sq, (size_t)(e1 - sq));
                        if (rc == SKX_OK) rc = emit(1, ql, (size_t)(e3 - ql));
                        if (rc != SKX_OK) return rc;
                        line = e3 + 1; line_no += 4; j += 4;
                        continue;
                    }
                }
                const uint8_t *nl = p + ends[j];
                const int rc = take_line(line, nl);
                if (rc != SKX_OK) return rc;
                line = nl + 1; j++;
            }
            p += used;
        }
        *rest = line;
        return SKX_OK;
    };
    for (;;) {
        ssize_t r;
        const uint8_t *base = chunk.data();                                  // where the unconsumed bytes (`have` of them) start
        if (zr) {
            const uint8_t *np; size_t got;
            if (zr->next(&np, &got, have) != 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }      // damaged or truncated: an error, not a short input
            base = np - have; r = (ssize_t)got;
        } else r = gz ? (ssize_t)gzread(gz, chunk.data() + have, (unsigned)(CH - have)) : ::read(fd, chunk.data() + have, CH - have);
        if (r < 0 && !gz && errno == EINTR) continue;
        if (r < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
        if (first) {
            if (r == 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
            if (base[0] != '@') return SKF_NOT_TAKEN;
            first = false;
        }
        const size_t n = have + (size_t)r;
        const uint8_t *p = base, *end = p + n;
        if (!spill.empty()) {                                              // the end of a line that began in an earlier chunk
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            if (nl) {
                spill.insert(spill.end(), p, nl);
                const int rc = take_line(spill.data(), spill.data() + spill.size());
                spill.clear();
                if (rc != SKX_OK) return rc;
                p = nl + 1;
            }
        }
        if (spill.empty()) { const int rc = walk(p, end, &p); if (rc != SKX_OK) return rc; }
        have = (size_t)(end - p);
        if (r == 0 && gz) {                                                 // a gzip stream that stops before its end (a truncated file) is an error
            int zerr = Z_OK; (void)gzerror(gz, &zerr);
            if (zerr != Z_OK && zerr != Z_STREAM_END) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
        }
        if (r == 0) {                                                       // end of file: a last line without terminator
            if (have || !spill.empty()) {
                spill.insert(spill.end(), p, end);
                const int rc = take_line(spill.data(), spill.data() + spill.size());
                if (rc != SKX_OK) return rc;
            }
            break;
        }
        if (zr) { if (have > GzReader::KEEP_MAX) { spill.insert(spill.end(), p, end); have = 0; } }      // (the inflater keeps the unfinished line in front of its next text)
        else if (have == CH) { spill.insert(spill.end(), p, end); have = 0; }    // no line end in a whole chunk
        else if (have) memmove(chunk.data(), p, have);
    }
    if (in_record || (line_no & 3u) != 0) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
    return SKX_OK;
}

// ---- read sets as bit planes (the reader threads of a batch of read sets: 5 bits per position cross PCIe instead of two bytes) ----
// A sequence line as three planes, one 64-bit word per 64 bases, bits beyond the line zero: lo / hi = bits 1 / 2 of the byte (the code
// encode_base gives: A 0, C 1, T 2, G 3 -- bit_encoding.rs:42-51), bad = the bytes valid_base rejects (low nibble 14: N, n;
// bit_encoding.rs:52-54).  A quality line as one plane: the bases the quality filters reject, !(q - 33 > min_qual) in u8 arithmetic
// (split_kmer.rs:98-101).  AVX-512 (64 bytes a step, the compare masks are the plane words) or AVX2 (32) where the processor has them,
// byte by byte otherwise.
#include <immintrin.h>
__attribute__((target("avx512f,avx512bw"))) static void pack_bases_avx512(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad)
{
    const __m512i nib = _mm512_set1_epi8(0x0F), v14 = _mm512_set1_epi8(14);
    size_t w = 0;
    for (size_t i = 0; i < n; i += 64, w++) {
        const size_t m = n - i;
        const __mmask64 in = m >= 64 ? ~0ull : (1ull << m) - 1ull;
        const __m512i b = _mm512_maskz_loadu_epi8(in, s + i);                    // (masked: nothing is read beyond the line)
        const uint64_t bm = _mm512_mask_cmpeq_epi8_mask(in, _mm512_and_si512(b, nib), v14);
        lo[w] = (uint64_t)_mm512_movepi8_mask(_mm512_slli_epi16(b, 6)) & in & ~bm;          // bit 1 of every byte at its top
        hi[w] = (uint64_t)_mm512_movepi8_mask(_mm512_slli_epi16(b, 5)) & in & ~bm;          // bit 2
        bad[w] = bm;
    }
}
__attribute__((target("avx512f,avx512bw"))) static void pack_qual_avx512(const uint8_t *q, size_t n, int min_qual, uint64_t *qb)
{
    const __m512i v33 = _mm512_set1_epi8(33), vm = _mm512_set1_epi8((char)(uint8_t)min_qual);
    size_t w = 0;
    for (size_t i = 0; i < n; i += 64, w++) {
        const size_t m = n - i;
        const __mmask64 in = m >= 64 ? ~0ull : (1ull << m) - 1ull;
        const __m512i b = _mm512_maskz_loadu_epi8(in, q + i);
        qb[w] = _mm512_mask_cmple_epu8_mask(in, _mm512_sub_epi8(b, v33), vm);             // q - 33 <= min_qual, unsigned
    }
}
__attribute__((target("avx2"))) static inline void planes32_avx2(const uint8_t *s, size_t m, uint32_t &lo, uint32_t &hi, uint32_t &bad)
{
    const __m256i nib = _mm256_set1_epi8(0x0F), v14 = _mm256_set1_epi8(14);
    __m256i b;
    if (m >= 32) b = _mm256_loadu_si256((const __m256i *)s);
    else { alignas(32) uint8_t tmp[32] = {0}; memcpy(tmp, s, m); b = _mm256_load_si256((const __m256i *)tmp); }
    const uint32_t in = m >= 32 ? 0xFFFFFFFFu : (1u << m) - 1u;
    bad = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(b, nib), v14)) & in;
    lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 6)) & in & ~bad;
    hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 5)) & in & ~bad;
}
__attribute__((target("avx2"))) static void pack_bases_avx2(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad)
{
    size_t w = 0;
    for (size_t i = 0; i < n; i += 64, w++) {
        uint32_t l0, h0, b0, l1 = 0, h1 = 0, b1 = 0;
        planes32_avx2(s + i, n - i, l0, h0, b0);
        if (n - i > 32) planes32_avx2(s + i + 32, n - i - 32, l1, h1, b1);
        lo[w] = ((uint64_t)l1 << 32) | l0; hi[w] = ((uint64_t)h1 << 32) | h0; bad[w] = ((uint64_t)b1 << 32) | b0;
    }
}
__attribute__((target("avx2"))) static inline uint32_t qual32_avx2(const uint8_t *q, size_t m, int min_qual)
{
    const __m256i v33 = _mm256_set1_epi8(33), vm = _mm256_set1_epi8((char)(uint8_t)min_qual);
    __m256i b;
    if (m >= 32) b = _mm256_loadu_si256((const __m256i *)q);
    else { alignas(32) uint8_t tmp[32] = {0}; memcpy(tmp, q, m); b = _mm256_load_si256((const __m256i *)tmp); }
    const uint32_t in = m >= 32 ? 0xFFFFFFFFu : (1u << m) - 1u;
    const __m256i t = _mm256_sub_epi8(b, v33);
    return (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(t, vm), vm)) & in;      // t <= min_qual, unsigned
}
__attribute__((target("avx2"))) static void pack_qual_avx2(const uint8_t *q, size_t n, int min_qual, uint64_t *qb)
{
    size_t w = 0;
    for (size_t i = 0; i < n; i += 64, w++) {
        const uint32_t a = qual32_avx2(q + i, n - i, min_qual), b = n - i > 32 ? qual32_avx2(q + i + 32, n - i - 32, min_qual) : 0u;
        qb[w] = ((uint64_t)b << 32) | a;
    }
}
static void pack_bases_plain(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad)
{
    for (size_t w = 0; w * 64 < n; w++) {
        uint64_t l = 0, h = 0, b = 0;
        const size_t m = std::min<size_t>(64, n - w * 64);
        for (size_t j = 0; j < m; j++) {
            const uint8_t c = s[w * 64 + j];
            const uint64_t isbad = (c & 0xF) == 14;
            b |= isbad << j; l |= (((uint64_t)c >> 1) & 1u & ~isbad) << j; h |= (((uint64_t)c >> 2) & 1u & ~isbad) << j;
        }
        lo[w] = l; hi[w] = h; bad[w] = b;
    }
}
static void pack_qual_plain(const uint8_t *q, size_t n, int min_qual, uint64_t *qb)
{
    for (size_t w = 0; w * 64 < n; w++) {
        uint64_t v = 0;
        const size_t m = std::min<size_t>(64, n - w * 64);
        for (size_t j = 0; j < m; j++) v |= (uint64_t)((uint8_t)(q[w * 64 + j] - 33) <= (uint8_t)min_qual) << j;
        qb[w] = v;
    }
}
// 2 = AVX-512 (F + BW), 1 = AVX2, 0 = neither; SKX_KNOBS=simd_cap=<n> caps it at level n - 1 (tests run all three)
static int simd_level()
{
    static const int v = [] {
        int lv = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
        if (knob("simd_cap")) lv = std::min<int>(lv, (int)knob("simd_cap") - 1);
        return lv;
    }();
    return v;
}
// the line ends of a piece of text, in bulk: offsets of '\n' in [p, p + n), at most `cap` of them (cap >= 64); *used = the bytes looked at (n unless the
// list filled up first).  64 / 32 bytes a step where the processor has AVX-512 / AVX2 (a memchr per 76-byte line was a third of a reader thread's time)
__attribute__((target("avx512f,avx512bw"))) static size_t newlines_avx512(const uint8_t *p, size_t n, uint32_t *out, size_t cap, size_t *used)
{
    const __m512i nl = _mm512_set1_epi8('\n');
    size_t c = 0, i = 0;
    for (; i < n && c + 64 <= cap; i += 64) {
        const size_t m = n - i;
        const __mmask64 in = m >= 64 ? ~0ull : (1ull << m) - 1ull;
        uint64_t k = _mm512_mask_cmpeq_epi8_mask(in, _mm512_maskz_loadu_epi8(in, p + i), nl);
        while (k) { out[c++] = (uint32_t)(i + (size_t)__builtin_ctzll(k)); k &= k - 1; }
    }
    *used = i < n ? i : n;
    return c;
}
__attribute__((target("avx2"))) static size_t newlines_avx2(const uint8_t *p, size_t n, uint32_t *out, size_t cap, size_t *used)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t c = 0, i = 0;
    for (; i + 32 <= n && c + 32 <= cap; i += 32) {
        uint32_t k = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl));
        while (k) { out[c++] = (uint32_t)(i + (size_t)__builtin_ctz(k)); k &= k - 1; }
    }
    for (; i < n && c < cap && n - i < 32; i++) if (p[i] == '\n') out[c++] = (uint32_t)i;      // the tail, byte by byte
    *used = i;
    return c;
}
static size_t newlines_plain(const uint8_t *p, size_t n, uint32_t *out, size_t cap, size_t *used)
{
    size_t c = 0, i = 0;
    while (i < n && c < cap) {
        const uint8_t *q = (const uint8_t *)memchr(p + i, '\n', n - i);
        if (!q) { i = n; break; }
        out[c++] = (uint32_t)(q - p); i = (size_t)(q - p) + 1;
    }
    *used = i;
    return c;
}
static size_t newline_offsets(const uint8_t *p, size_t n, uint32_t *out, size_t cap, size_t *used)
{
    const int lv = simd_level();
    return lv == 2 ? newlines_avx512(p, n, out, cap, used) : lv == 1 ? newlines_avx2(p, n, out, cap, used) : newlines_plain(p, n, out, cap, used);
}
void pack_bases_planes(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad)
{
    const int lv = simd_level();
    if (lv == 2) pack_bases_avx512(s, n, lo, hi, bad); else if (lv == 1) pack_bases_avx2(s, n, lo, hi, bad); else pack_bases_plain(s, n, lo, hi, bad);
}
void pack_qual_plane(const uint8_t *q, size_t n, int min_qual, uint64_t *qb)
{
    const int lv = simd_level();
    if (lv == 2) pack_qual_avx512(q, n, min_qual, qb); else if (lv == 1) pack_qual_avx2(q, n, min_qual, qb); else pack_qual_plain(q, n, min_qual, qb);
}

int read_sample_stream(const char *file1, const char *file2, double proportion_reads, HostStream &out)
{
    size_t step = 1;
    if (proportion_reads > 0.0) { step = (size_t)std::llround(1.0 / proportion_reads); if (step == 0) step = 1; }
    out.seq.clear(); out.qual.clear(); out.ids.clear();
    static thread_local std::vector<uint8_t> buf;              // a reader thread's file buffer lives across its samples (no fresh pages per file)
    SKX_TRY(slurp(file1, buf));
    if (buf[0] == '@') out.is_fastq = true;
    else if (buf[0] == '>') out.is_fastq = false;
    else { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
    SKX_TRY(out.is_fastq ? parse_fastq(buf, step, out) : parse_fasta(buf, step, out));
    if (file2) {
        SKX_TRY(slurp(file2, buf));
        SKX_TRY(out.is_fastq ? parse_fastq(buf, step, out) : parse_fasta(buf, step, out));
    }
    return SKX_OK;
}

}  // namespace skx
