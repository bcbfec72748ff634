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
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
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
    struct stat sb;
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
int stream_fastq_file(const char *path, const std::function<int(int which, const uint8_t *p, size_t n)> &emit)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
    struct Close { int fd; ~Close() { ::close(fd); } } cl{fd};
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
    // a gzip file is inflated piece by piece as it is read (zlib's gzread on a descriptor of its own: members one after the other, as gzip reads them)
    unsigned char mg[2] = {0, 0};
    gzFile gz = nullptr;
    if (pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1f && mg[1] == 0x8b) {
        const int fd2 = dup(fd);
        if (fd2 < 0 || !(gz = gzdopen(fd2, "rb"))) { if (fd2 >= 0) ::close(fd2); set_error("Invalid path/file: %s", path); return SKX_EIO; }
        gzbuffer(gz, 1u << 20);
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
    for (;;) {
        ssize_t r = gz ? (ssize_t)gzread(gz, chunk.data() + have, (unsigned)(CH - have)) : ::read(fd, chunk.data() + have, CH - have);
        if (r < 0 && !gz && errno == EINTR) continue;
        if (r < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
        if (first) {
            if (r == 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
            if (chunk[0] != '@') return SKF_NOT_TAKEN;
            first = false;
        }
        const size_t n = have + (size_t)r;
        const uint8_t *p = chunk.data(), *end = p + n;
        while (p < end) {
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            if (!nl) break;
            int rc;
            if (!spill.empty()) { spill.insert(spill.end(), p, nl); rc = take_line(spill.data(), spill.data() + spill.size()); spill.clear(); }
            else rc = take_line(p, nl);
            if (rc != SKX_OK) return rc;
            p = nl + 1;
        }
        have = (size_t)(end - p);
        if (r == 0) {                                                       // end of file: a last line without terminator
            if (have || !spill.empty()) {
                spill.insert(spill.end(), p, end);
                const int rc = take_line(spill.data(), spill.data() + spill.size());
                if (rc != SKX_OK) return rc;
            }
            break;
        }
        if (have == CH) { spill.insert(spill.end(), p, end); have = 0; }    // no line end in a whole chunk
        else if (have) memmove(chunk.data(), p, have);
    }
    if (in_record || (line_no & 3u) != 0) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
    return SKX_OK;
}

// ---- read sets as bit planes (the reader threads of a batch of read sets: 5 bits per position cross PCIe instead of two bytes) ----
// A sequence line as three planes, one 32-bit word per 32 bases, bits beyond the line zero: lo / hi = bits 1 / 2 of the byte (the code
// encode_base gives: A 0, C 1, T 2, G 3 -- bit_encoding.rs:42-51), bad = the bytes valid_base rejects (low nibble 14: N, n;
// bit_encoding.rs:52-54).  A quality line as one plane: the bases the quality filters reject, !(q - 33 > min_qual) in u8 arithmetic
// (split_kmer.rs:98-101).  AVX2 where the processor has it (32 bytes a step), byte by byte otherwise.
#include <immintrin.h>
__attribute__((target("avx2"))) static void pack_bases_avx2(const uint8_t *s, size_t n, uint32_t *lo, uint32_t *hi, uint32_t *bad)
{
    const __m256i nib = _mm256_set1_epi8(0x0F), v14 = _mm256_set1_epi8(14);
    size_t w = 0;
    for (size_t i = 0; i < n; i += 32, w++) {
        const size_t m = n - i;
        __m256i b;
        if (m >= 32) b = _mm256_loadu_si256((const __m256i *)(s + i));
        else { alignas(32) uint8_t tmp[32] = {0}; memcpy(tmp, s + i, m); b = _mm256_load_si256((const __m256i *)tmp); }
        const uint32_t in = m >= 32 ? 0xFFFFFFFFu : (1u << m) - 1u;
        const uint32_t bm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(b, nib), v14)) & in;
        lo[w] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 6)) & in & ~bm;          // bit 1 of every byte at its top
        hi[w] = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 5)) & in & ~bm;          // bit 2
        bad[w] = bm;
    }
}
__attribute__((target("avx2"))) static void pack_qual_avx2(const uint8_t *q, size_t n, int min_qual, uint32_t *qb)
{
    const __m256i v33 = _mm256_set1_epi8(33), vm = _mm256_set1_epi8((char)(uint8_t)min_qual);
    size_t w = 0;
    for (size_t i = 0; i < n; i += 32, w++) {
        const size_t m = n - i;
        __m256i b;
        if (m >= 32) b = _mm256_loadu_si256((const __m256i *)(q + i));
        else { alignas(32) uint8_t tmp[32] = {0}; memcpy(tmp, q + i, m); b = _mm256_load_si256((const __m256i *)tmp); }
        const uint32_t in = m >= 32 ? 0xFFFFFFFFu : (1u << m) - 1u;
        const __m256i t = _mm256_sub_epi8(b, v33);
        qb[w] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(t, vm), vm)) & in;      // t <= min_qual, unsigned
    }
}
static void pack_bases_plain(const uint8_t *s, size_t n, uint32_t *lo, uint32_t *hi, uint32_t *bad)
{
    for (size_t w = 0; w * 32 < n; w++) {
        uint32_t l = 0, h = 0, b = 0;
        const size_t m = std::min<size_t>(32, n - w * 32);
        for (size_t j = 0; j < m; j++) {
            const uint8_t c = s[w * 32 + j];
            const uint32_t isbad = (c & 0xF) == 14;
            b |= isbad << j; l |= (((uint32_t)c >> 1) & 1u & ~isbad) << j; h |= (((uint32_t)c >> 2) & 1u & ~isbad) << j;
        }
        lo[w] = l; hi[w] = h; bad[w] = b;
    }
}
static void pack_qual_plain(const uint8_t *q, size_t n, int min_qual, uint32_t *qb)
{
    for (size_t w = 0; w * 32 < n; w++) {
        uint32_t v = 0;
        const size_t m = std::min<size_t>(32, n - w * 32);
        for (size_t j = 0; j < m; j++) v |= (uint32_t)((uint8_t)(q[w * 32 + j] - 33) <= (uint8_t)min_qual) << j;
        qb[w] = v;
    }
}
static bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2") && !knob("no_avx2"); return v; }
void pack_bases_planes(const uint8_t *s, size_t n, uint32_t *lo, uint32_t *hi, uint32_t *bad)
{
    if (have_avx2()) pack_bases_avx2(s, n, lo, hi, bad); else pack_bases_plain(s, n, lo, hi, bad);
}
void pack_qual_plane(const uint8_t *q, size_t n, int min_qual, uint32_t *qb)
{
    if (have_avx2()) pack_qual_avx2(q, n, min_qual, qb); else pack_qual_plain(q, n, min_qual, qb);
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
