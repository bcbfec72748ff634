// fastx.cpp -- the engine's FASTA/FASTQ(.gz) reader: replaces the needletail record iterator used at
// ska_dict.rs:131-153,356-366.  Produces the *record stream* the device kernels consume: each record's bases
// with line breaks removed, followed by one '\n' ('\n' can never occur inside a record, so it doubles as the
// record terminator the end-of-record rule of split_kmer.rs:89 needs).  For FASTQ a parallel quality stream
// is produced.  Format is decided by the first byte of file 1 ('>' / '@'), as ska_dict.rs:356-366 does via the
// first record; file 2 is parsed with the same mode.  `proportion_reads` keeps record n iff n % round(1/p) == 0
// (ska_dict.rs:125-141), per file.
#include "skx_internal.h"
#include <cmath>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

namespace skx {

static int slurp(const char *path, std::vector<uint8_t> &buf)
{
    // (read once: POSIX_FADV_NOREUSE keeps the kernel from promoting every page on first access -- one lock for all reader threads)
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_error("Invalid path/file: %s", path); return SKX_EIO; }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
    gzFile g = gzdopen(fd, "rb");            // transparent for uncompressed files; closes fd with gzclose
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

int read_sample_stream(const char *file1, const char *file2, double proportion_reads, HostStream &out)
{
    size_t step = 1;
    if (proportion_reads > 0.0) { step = (size_t)std::llround(1.0 / proportion_reads); if (step == 0) step = 1; }
    out.seq.clear(); out.qual.clear(); out.ids.clear();
    std::vector<uint8_t> buf;
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
