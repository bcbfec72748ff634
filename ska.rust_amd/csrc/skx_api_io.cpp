// skx_api_io.cpp -- the C ABI entry points that move text and files: `ska cov` histogram, `ska map`, .skf save / load.
// (skx_api.cpp holds the build -> merge -> filter -> distance path.)
#include "skx_internal.h"
#include "../../include/skx_host.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>
#include <unistd.h>
#include <fcntl.h>

using namespace skx;

// ------------------------------------------------------------------------------------------ ska cov (N4)
// CoverageHistogram::new (coverage.rs:70-148) + the histogram of fit_histogram (:158-163): both files must be FASTQ, qualities
// are ignored, hist[c - 1] = number of split k-mers occurring c times over both files (c <= 1000)
extern "C" int skx_cov_histogram(skx_ctx *ctx, const char *fastq_fwd, const char *fastq_rev, int k, int rc, uint32_t *hist)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !fastq_fwd || !fastq_rev || !hist) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    HostStream hs[2];
    const char *files[2] = {fastq_fwd, fastq_rev};
    for (int f = 0; f < 2; f++) {
        SKX_TRY(read_sample_stream(files[f], nullptr, 0.0, hs[f]));
        if (!hs[f].is_fastq) { set_error("%s appears to be FASTA.\nCoverage can only be used with FASTQ files, not FASTA.", files[f]); return SKX_EINVAL; }   // :97-99
    }
    const uint64_t L = hs[0].seq.size() + hs[1].seq.size();
    DevBuf<uint8_t> d_seq; DevBuf<uint32_t> d_hist;
    SKX_TRY(d_seq.alloc(L + 16)); SKX_TRY(d_hist.alloc(1000)); SKX_TRY(d_hist.zero(st));
    SKX_HIP(hipMemcpyAsync(d_seq.p, hs[0].seq.data(), hs[0].seq.size(), hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_seq.p + hs[0].seq.size(), hs[1].seq.data(), hs[1].seq.size(), hipMemcpyHostToDevice, st));
    SKX_TRY(cov_histogram(ctx, d_seq.p, L, k, rc, d_hist.p));
    SKX_HIP(hipMemcpyAsync(hist, d_hist.p, 1000 * 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ ska map (N3)
// generic_modes::map (generic_modes.rs:56-84): RefSka::new(k, reference, rc, ambig_mask, repeat_mask) (ska_ref.rs:189-311), map
// (:508-533), write_aln | write_vcf (:622-765).  Device: reference windows, look-up of every window's split k-mer in the array,
// gather of the mapped rows (reverse-complemented where the reference strand is not the canonical one), one AlnWriter state
// machine per sample.  Host: record bookkeeping, repeat coordinates, text.
extern "C" int skx_array_map(skx_array *a, const char *reference, int ambig_mask, int repeat_mask, int format, int threads, char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    if (!a || !reference || !buf || !len) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const int k = a->k, h = (k - 1) / 2, S = (int)a->names.size();
    if (a->n_kmers != a->n_rows || a->keys_absent) { set_error("split k-mers and variants are out of step (filtered without update_kmers)"); return SKX_EINVAL; }
    SKX_TRY(array_materialize(a));
    HostStream hs;
    SKX_TRY(read_sample_stream(reference, nullptr, 0.0, hs));
    if (hs.is_fastq) { set_error("Cannot create reference from FASTQ files"); return SKX_EINVAL; }                                // ska_ref.rs:206-208
    const uint64_t L = hs.seq.size();
    if (L > 0xFFFFFFF0ull) { set_error("reference longer than 4 G bases"); return SKX_EUNSUP; }
    // chromosomes: start in the record stream, length, offset in the concatenated output
    std::vector<uint64_t> cstart, clen, coff;
    { uint64_t b0 = 0, off = 0; for (uint64_t i = 0; i < L; i++) if (hs.seq[i] == '\n') { cstart.push_back(b0); clen.push_back(i - b0); coff.push_back(off); off += i - b0; b0 = i + 1; } }
    const size_t n_chrom = cstart.size();
    if (n_chrom > 1 && !format) skh_log(1, "ska::ska_ref", "Reference contained multiple contigs, in the output they will be concatenated");      // ska_ref.rs:641-645 write_aln
    uint64_t total = 0; for (auto l : clen) total += l;
    DevBuf<uint8_t> d_seq; SKX_TRY(d_seq.alloc(L + 16));
    SKX_HIP(hipMemcpyAsync(d_seq.p, hs.seq.data(), L, hipMemcpyHostToDevice, st));
    DevBuf<uint64_t> wlo, whi; DevBuf<uint8_t> flag;
    SKX_TRY(ref_windows(ctx, d_seq.p, L, k, a->rc, wlo, whi, flag));
    std::vector<uint8_t> hflag(L);
    SKX_HIP(hipMemcpyAsync(hflag.data(), flag.p, L, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    uint64_t n_windows = 0; for (uint64_t p = 0; p < L; p++) n_windows += hflag[p] != 0;
    if (n_windows == 0) { set_error("%s has no valid sequence", reference); return SKX_EEMPTY; }                                  // ska_ref.rs:255-257
    auto chrom_of = [&](uint64_t sp) { return (size_t)(std::upper_bound(cstart.begin(), cstart.end(), sp) - cstart.begin()) - 1; };
    // repeat coordinates (ska_ref.rs:259-293): every window whose split k-mer occurs more than once in the reference
    std::vector<uint64_t> repeat;
    if (repeat_mask) {
        // which windows: on the device (sorted with the engine's radix sort, neighbours compared); the coordinates below are bookkeeping
        DevBuf<uint8_t> d_rep;
        SKX_TRY(ref_repeat_flags(ctx, wlo.p, k > 31 ? whi.p : nullptr, flag.p, L, d_rep));
        std::vector<uint8_t> rep(L, 0);
        SKX_HIP(hipMemcpy(rep.data(), d_rep.p, L, hipMemcpyDeviceToHost));
        uint64_t last_chrom = 0, last_end = 0, chrom_offset = 0;
        for (uint64_t p = 0; p < L; p++) {
            if (!hflag[p]) continue;
            const size_t c = chrom_of(p - h);
            if (c > last_chrom) { chrom_offset += clen[last_chrom]; last_chrom = c; }
            if (!rep[p]) continue;
            const uint64_t pos = p - h - cstart[c], start = pos - h + chrom_offset, end = pos + h + chrom_offset;
            for (uint64_t x = (start > last_end || start == 0) ? start : last_end + 1; x < end + 1; x++) repeat.push_back(x);
            last_chrom = c; last_end = end;
        }
    }
    // look-up
    DevBuf<uint32_t> row; DevBuf<uint8_t> is_rc;
    SKX_TRY(row.alloc(L)); SKX_TRY(is_rc.alloc(L));
    if (k <= 31) {
        DevBuf<uint64_t> sorted; DevBuf<uint32_t> perm;
        const uint64_t *skeys = a->keys.p; const uint32_t *sperm = nullptr;
        if (!a->engine_order) { SKX_TRY(sort_words_perm(a->keys.p, a->n_rows, sorted, perm, st)); skeys = sorted.p; sperm = perm.p; }
        launch_map_lookup(wlo.p, flag.p, d_seq.p, L, h, skeys, sperm, a->n_rows, row.p, is_rc.p, st);
        SKX_HIP(hipStreamSynchronize(st));
    } else {
        DevBuf<uint64_t> tmp, sorted; DevBuf<uint32_t> perm; const u128 *w = nullptr;
        if (a->n_rows) SKX_TRY(array_wide_words(a, tmp, &w));
        const u128 *skeys = w; const uint32_t *sperm = nullptr;
        if (!a->engine_order) { SKX_TRY(sort_wide_perm(w, a->n_rows, sorted, perm, st)); skeys = (const u128 *)sorted.p; sperm = perm.p; }
        launch_map_lookup_wide(wlo.p, whi.p, flag.p, d_seq.p, L, h, skeys, sperm, a->n_rows, row.p, is_rc.p, st);
        SKX_HIP(hipStreamSynchronize(st));
    }
    DevBuf<uint32_t> mapped; uint64_t M = 0;
    SKX_TRY(select_mapped(row.p, L, mapped, &M, st));
    if (M == 0) { set_error("No split k-mers mapped to reference"); return SKX_EINVAL; }                                          // ska_ref.rs:553-555
    std::vector<uint32_t> hm(M), mpos(M), mchrom(M);
    SKX_HIP(hipMemcpy(hm.data(), mapped.p, M * 4, hipMemcpyDeviceToHost));
    for (uint64_t m = 0; m < M; m++) { const uint64_t mid = (uint64_t)hm[m] - h; const size_t c = chrom_of(mid); mchrom[m] = (uint32_t)c; mpos[m] = (uint32_t)(mid - cstart[c]); }
    // per chromosome: its range in the mapped list; the reference with the chromosomes concatenated (= output coordinates)
    std::vector<uint64_t> mlo(n_chrom, 0), mhi(n_chrom, 0);
    for (uint64_t m = 0; m < M; m++) { const uint32_t c = mchrom[m]; if (mhi[c] == 0) mlo[c] = m; mhi[c] = m + 1; }
    std::vector<uint8_t> refcat; refcat.reserve(total + 8);
    for (size_t c = 0; c < n_chrom; c++) refcat.insert(refcat.end(), hs.seq.begin() + (ptrdiff_t)cstart[c], hs.seq.begin() + (ptrdiff_t)(cstart[c] + clen[c]));
    refcat.resize(total + 8, '-');
    DevBuf<uint32_t> d_mpos, d_mchrom, d_pres, d_first, d_last; DevBuf<uint64_t> d_clen, d_coff, d_rep, d_mlo, d_mhi; DevBuf<uint8_t> d_refcat;
    const uint64_t ppitch = (total + 31) / 32 + 4;
    SKX_TRY(d_mpos.alloc(M)); SKX_TRY(d_mchrom.alloc(M)); SKX_TRY(d_clen.alloc(n_chrom)); SKX_TRY(d_coff.alloc(n_chrom));
    SKX_TRY(d_mlo.alloc(n_chrom)); SKX_TRY(d_mhi.alloc(n_chrom)); SKX_TRY(d_rep.alloc(repeat.size() + 1)); SKX_TRY(d_refcat.alloc(total + 8));
    SKX_TRY(d_pres.alloc((uint64_t)S * ppitch)); SKX_TRY(d_first.alloc((uint64_t)S * n_chrom)); SKX_TRY(d_last.alloc((uint64_t)S * n_chrom));
    SKX_TRY(d_pres.zero(st));
    SKX_HIP(hipMemcpyAsync(d_mpos.p, mpos.data(), M * 4, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_mchrom.p, mchrom.data(), M * 4, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_clen.p, clen.data(), n_chrom * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_coff.p, coff.data(), n_chrom * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_mlo.p, mlo.data(), n_chrom * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_mhi.p, mhi.data(), n_chrom * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_refcat.p, refcat.data(), total + 8, hipMemcpyHostToDevice, st));
    if (!repeat.empty()) SKX_HIP(hipMemcpyAsync(d_rep.p, repeat.data(), repeat.size() * 8, hipMemcpyHostToDevice, st));
    const uint64_t mpitch = (M + 255) / 256 * 256, opitch = (total + 255) / 256 * 256 + 256;
    DevBuf<uint8_t> mv, out;
    SKX_TRY(mv.alloc((uint64_t)S * mpitch)); SKX_TRY(out.alloc((uint64_t)S * opitch));
    SKX_HIP(hipMemsetAsync(out.p, '-', (uint64_t)S * opitch, st));
    launch_gather_mapped(a->matrix.p, a->pitch, S, mapped.p, row.p, is_rc.p, M, mv.p, mpitch, st);
    MapWriteArgs wa{mv.p, mpitch, M, d_mpos.p, d_mchrom.p, d_mlo.p, d_mhi.p, d_refcat.p, total, d_clen.p, d_coff.p, (int)n_chrom, (uint64_t)h, ambig_mask,
                    d_rep.p, (uint64_t)repeat.size(), out.p, opitch, S, d_pres.p, ppitch, d_first.p, d_last.p};
    launch_aln_write(wa, st);
    std::vector<uint8_t> aln((uint64_t)S * opitch);
    SKX_HIP(hipMemcpyAsync(aln.data(), out.p, aln.size(), hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    // ---- text
    std::string o;
    if (format == 0) {                                                                                                             // write_aln, ska_ref.rs:622-645
        o.reserve((uint64_t)S * (total + 64));
        for (int s = 0; s < S; s++) { o += '>'; o += a->names[s]; o += '\n'; o.append((const char *)aln.data() + (uint64_t)s * opitch, total); o += '\n'; }
    } else {                                                                                                                       // write_vcf, :648-765
        o += "##fileformat=VCFv4.4\n";
        for (size_t c = 0; c < n_chrom; c++) { o += "##contig=<ID="; o += hs.ids[c]; o += ">\n"; }
        o += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
        for (int s = 0; s < S; s++) { o += '\t'; o += a->names[s]; }
        o += '\n';
        auto vbase = [](uint8_t b) { return (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? (char)b : 'N'; };                    // u8_to_base, :137-146
        const uint64_t BLK = 4096;
        const uint64_t nblk = (total + BLK - 1) / BLK;
        std::vector<std::string> parts(nblk);
        int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 64) nt = 64;
        std::atomic<uint64_t> next{0};
        auto work = [&]() {
            std::vector<uint8_t> col((uint64_t)S * BLK);
            std::vector<int> gt(S);
            for (uint64_t b; (b = next.fetch_add(1)) < nblk;) {
                const uint64_t x0 = b * BLK, nx = std::min(BLK, total - x0);
                for (int s = 0; s < S; s++) memcpy(col.data() + (uint64_t)s * BLK, aln.data() + (uint64_t)s * opitch + x0, nx);
                std::string &t = parts[b];
                for (uint64_t i = 0; i < nx; i++) {
                    const uint64_t idx = x0 + i;
                    const size_t c = (size_t)(std::upper_bound(coff.begin(), coff.end(), idx) - coff.begin()) - 1;               // IdxCheck (idx_check.rs)
                    const uint64_t pos = idx - coff[c];
                    const uint8_t ref_base = hs.seq[cstart[c] + pos];
                    char alts[5]; int n_alt = 0; bool variant = false;
                    for (int s = 0; s < S; s++) {
                        const uint8_t mb = col[(uint64_t)s * BLK + i];
                        if (mb == ref_base) gt[s] = 0;
                        else if (mb == '-') { variant = true; gt[s] = -1; }
                        else {
                            variant = true;
                            const char ab = vbase(mb);
                            int at = -1;
                            for (int q = 0; q < n_alt; q++) if (alts[q] == ab) at = q;
                            if (at < 0) { alts[n_alt] = ab; at = n_alt++; }
                            gt[s] = at + 1;
                        }
                    }
                    if (!variant) continue;
                    t += hs.ids[c]; t += '\t'; t += std::to_string(pos + 1); t += "\t.\t"; t += vbase(ref_base); t += '\t';
                    if (!n_alt) t += '.';
                    for (int q = 0; q < n_alt; q++) { if (q) t += ','; t += alts[q]; }
                    t += "\t.\t.\t.\tGT";
                    for (int s = 0; s < S; s++) { t += '\t'; if (gt[s] < 0) t += '.'; else t += std::to_string(gt[s]); }
                    t += '\n';
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto &th : pool) th.join();
        for (auto &p : parts) o += p;
    }
    char *mem = (char *)malloc(o.size() + 1);
    if (!mem) return SKX_ENOMEM;
    memcpy(mem, o.data(), o.size()); mem[o.size()] = 0;
    *buf = mem; *len = o.size();
    return SKX_OK;
    });
}

// The compressed bytes of a group of chunks, from the file into pinned memory by a small team of pread()s, a group ahead of the device
// (round 6).  The loaders used to hand the runtime a pointer into the file's mapping: a pageable copy, one thread taking the page faults --
// 8 GB/s on a file read before, 2 GB/s on one another process has just written (`ska align x.skf` behind `ska build`, `ska distance`
// behind `ska merge`: the first touch of fresh tmpfs pages; load.stream_decode_filter 1.76 s of a 2.5 s `ska distance`,
// profiles/r06d_reads_1000.log -- 0.44 s on the same file read a second time).  SKX_KNOBS=no_load_stager: the mapping, as before.
namespace {
struct GroupStager {
    int fd = -1, nt = 4;
    uint8_t *pin[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0};
    std::thread th[2]; bool ok[2] = {true, true};
    size_t lo[2] = {0, 0}, hi[2] = {0, 0};
    double t_alloc = 0.0; std::atomic<long long> us_pread{0};       // (what pinning the buffers and the team's reads cost: phases load.stager_*)
    explicit GroupStager(const char *path)
    {
        if (knob("no_load_stager")) return;
        fd = ::open(path, O_RDONLY);
        nt = std::max(1, std::min(8, cpu_budget()));
    }
    ~GroupStager()
    {
        for (int w = 0; w < 2; w++) { if (th[w].joinable()) th[w].join(); if (pin[w]) (void)hipHostFree(pin[w]); }
        if (fd >= 0) { ::close(fd); phase_add("load.stager_pin_alloc", t_alloc); phase_add("load.stager_reads_wall", us_pread.load() * 1e-6); }
    }
    bool on() const { return fd >= 0; }
    // file bytes [a, b) into buffer w (which nothing on the device reads any more)
    bool start(int w, size_t a, size_t b)
    {
        if (th[w].joinable()) th[w].join();
        const size_t n = b - a;
        if (n + 512 > cap[w]) {
            if (pin[w]) (void)hipHostFree(pin[w]);
            pin[w] = nullptr; cap[w] = 0;
            const size_t want = n + n / 4 + (1u << 20);
            const auto ta = std::chrono::steady_clock::now();
            if (hipHostMalloc((void **)&pin[w], want, hipHostMallocDefault) != hipSuccess) { pin[w] = nullptr; (void)hipGetLastError(); return false; }
            t_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
            cap[w] = want;
        }
        lo[w] = a; hi[w] = b; ok[w] = true;
        uint8_t *dst = pin[w]; const int fdc = fd; const int team = (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, n >> 20));
        bool *okp = &ok[w];
        std::atomic<long long> *usp = &us_pread;
        th[w] = std::thread([dst, fdc, a, n, team, okp, usp]() {
            const auto tr = std::chrono::steady_clock::now();
            struct Add { std::atomic<long long> *u; std::chrono::steady_clock::time_point t0; ~Add() { *u += (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } } add{usp, tr};
            std::atomic<bool> good{true};
            auto slice = [&](int t) {
                size_t p = n * (size_t)t / (size_t)team, e = n * (size_t)(t + 1) / (size_t)team;
                while (p < e) {
                    const ssize_t r = ::pread(fdc, dst + p, e - p, (off_t)(a + p));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { good = false; return; }
                    p += (size_t)r;
                }
            };
            std::vector<std::thread> helpers;
            for (int t = 1; t < team; t++) helpers.emplace_back(slice, t);
            slice(0);
            for (auto &h : helpers) h.join();
            *okp = good.load();
        });
        return true;
    }
    // the bytes [a, b) if buffer w holds (or is being filled with) exactly those; nullptr otherwise
    const uint8_t *get(int w, size_t a, size_t b)
    {
        if (!pin[w] || lo[w] != a || hi[w] != b || a == b) return nullptr;
        if (th[w].joinable()) th[w].join();
        return ok[w] ? pin[w] : nullptr;
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------ .skf
// 64 KB chunks per launch of the device codec (512 MB of CBOR; SKX_SKF_GROUP_CHUNKS overrides, tests use small groups)
static uint64_t skf_group_chunks()
{
    const long e_ = knob("skf_group_chunks"); char ebuf_[24]; snprintf(ebuf_, sizeof ebuf_, "%ld", e_); const char *e = e_ ? ebuf_ : nullptr;
    const long v = e ? atol(e) : 8192;
    return (uint64_t)std::max<long>(1, std::min<long>(v, 1 << 20));
}
// MergeSkaArray::save (merge_ska_array.rs:191-199), streamed: rows go out in the array's own order (the
// order of H for arrays built here, the file's order for loaded ones; the reference's order is its hash map's), one
// transposed row block at a time, so the host never holds the U x S matrix or its 2-bytes-per-cell CBOR text.
extern "C" int skx_array_save(skx_array *a, const char *path)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const uint64_t U = a->n_rows, S = a->names.size();
    if (a->n_kmers != a->n_rows || a->keys_absent) { set_error("split k-mers and variants are out of step (filtered without update_kmers): such an array cannot be loaded back"); return SKX_EINVAL; }
    SkfMeta m; m.k = a->k; m.rc = a->rc; m.k_bits = a->k_bits; m.names = a->names; m.version = a->version; m.n_rows = U;
    // split k-mers: 64-bit keys held on the device leave as finished CBOR (9 bytes each) through a pinned buffer; wider ones, keys
    // kept on the host and lists with a key below 2^32 (shorter minimal form) are encoded by the host team
    std::vector<skx_key> keys;
    SkfFastSections fast;
    struct PinnedBuf { uint8_t *p = nullptr; ~PinnedBuf() { if (p) (void)hipHostFree(p); } } pin_keys, pin_counts;
    const auto t_k0 = std::chrono::steady_clock::now();
    if (a->k <= 31 && a->keys.p && U && !a->keys_absent && a->n_kmers == U) {
        DevBuf<uint8_t> d_kc; DevBuf<int> d_short;
        SKX_TRY(d_kc.alloc(9 * U + 64)); SKX_TRY(d_short.alloc(1)); SKX_TRY(d_short.zero(st));
        launch_keys_cbor(a->keys.p, U, a->hp, d_kc.p, d_short.p, st);
        if (hipHostMalloc((void **)&pin_keys.p, 9 * U + 64, hipHostMallocDefault) != hipSuccess) { pin_keys.p = nullptr; set_error("out of host memory"); return SKX_ENOMEM; }
        int is_short = 0;
        SKX_HIP(hipMemcpyAsync(pin_keys.p, d_kc.p, 9 * U, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipMemcpyAsync(&is_short, d_short.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        if (!is_short) { fast.keys_cbor = pin_keys.p; fast.keys_cbor_len = 9 * U; fast.n_keys = U; }
    }
    if (!fast.keys_cbor) SKX_TRY(array_host_keys(a, keys));
    phase_add("save.keys_to_host", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_k0).count());
    SKX_TRY(array_lazy_stats(a));              // a lazily held array: the stored counts come from a statistics-only pass
    std::vector<uint64_t> counts;
    if (U) {
        if (hipHostMalloc((void **)&pin_counts.p, U * 4, hipHostMallocDefault) != hipSuccess) { pin_counts.p = nullptr; set_error("out of host memory"); return SKX_ENOMEM; }
        SKX_HIP(hipMemcpy(pin_counts.p, a->vcount.p, U * 4, hipMemcpyDeviceToHost));
        fast.counts32 = (const uint32_t *)pin_counts.p; fast.n_counts = U;
    }
    DevBuf<uint8_t> d_blk, d_win;
    uint64_t blk_cap = 0;
    // rows [r0, r0 + nr) sample-major: a slice of the matrix, or -- lazily held array -- a window assembled for the occasion
    auto rows_view = [&](uint64_t r0, uint64_t nr, const uint8_t **p, uint64_t *pitch) -> int {
        if (a->lazy()) return array_lazy_window(a, r0, nr, d_win, p, pitch);
        *p = a->matrix.p + r0; *pitch = a->pitch;
        return SKX_OK;
    };
    auto fetch = [&](uint64_t row0, uint64_t nr, uint8_t *dst) -> int {
        if (nr * S > blk_cap) { blk_cap = nr * S; SKX_TRY(d_blk.alloc(blk_cap)); }
        const uint8_t *src; uint64_t sp;
        SKX_TRY(rows_view(row0, nr, &src, &sp));
        launch_transpose(src, sp, S, nr, d_blk.p, S, st);                                // [S][nr] slice -> [nr][S]
        SKX_HIP(hipMemcpyAsync(dst, d_blk.p, nr * S, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        return SKX_OK;
    };
    // the data section on the device (skx_snappy.hip): groups of 64 KB chunks -> row-major cells -> finished frame chunks
    struct Pinned { uint8_t *p = nullptr; size_t cap = 0; ~Pinned() { if (p) (void)hipHostFree(p); }
                    int need(size_t n) { if (n <= cap) return SKX_OK; if (p) (void)hipHostFree(p); p = nullptr; cap = 0;
                                         if (hipHostMalloc((void **)&p, n, hipHostMallocDefault) != hipSuccess) { p = nullptr; return SKX_ENOMEM; } cap = n; return SKX_OK; } } pin[2];
    const DevEncode dev_encode = [&](FILE *f, uint64_t upos, uint64_t uoff0, uint64_t n_chunks) -> int {
        const uint64_t G = skf_group_chunks();
        if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] save: %llu chunks of the data section encoded on the device\n", (unsigned long long)n_chunks);
        DevBuf<uint8_t> d_cells, d_slots, d_dense; DevBuf<uint32_t> d_sizes; DevBuf<uint64_t> d_off;
        SKX_TRY(d_slots.alloc(std::min(G, n_chunks) * (uint64_t)SKF_SLOT)); SKX_TRY(d_sizes.alloc(std::min(G, n_chunks))); SKX_TRY(d_off.alloc(std::min(G, n_chunks)));
        std::vector<uint32_t> sizes; std::vector<uint64_t> off;
        uint64_t cells_cap = 0, dense_cap = 0;
        // the file write of one group runs beside the device work of the next (two pinned buffers)
        struct Writer { std::thread th; bool ok = true; void join() { if (th.joinable()) th.join(); } ~Writer() { join(); } } wr;
        int flip = 0;
        for (uint64_t g0 = 0; g0 < n_chunks; g0 += G, flip ^= 1) {
            const uint64_t ng = std::min(G, n_chunks - g0);
            const uint64_t rel_lo = uoff0 + g0 * 65536ull - upos, rel_hi = rel_lo + ng * 65536ull;        // section bytes of this group
            const uint64_t c_lo = rel_lo >> 1, c_hi = ((rel_hi - 1) >> 1) + 1;
            const uint64_t r0 = c_lo / S, r1 = (c_hi - 1) / S + 1, nr = r1 - r0;
            if (nr * S + 16 > cells_cap) { cells_cap = nr * S + 16; SKX_TRY(d_cells.alloc(cells_cap)); }
            const uint8_t *src; uint64_t sp;
            SKX_TRY(rows_view(r0, nr, &src, &sp));
            launch_transpose(src, sp, S, nr, d_cells.p, S, st);                                             // [S][nr] slice -> [nr][S]
            SKX_TRY(launch_skf_encode_cells(ctx->device, d_cells.p, r0 * S, upos, uoff0 + g0 * 65536ull, (uint32_t)ng, d_slots.p, d_sizes.p, st));
            sizes.resize(ng); off.resize(ng);
            SKX_HIP(hipMemcpyAsync(sizes.data(), d_sizes.p, ng * 4, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            uint64_t total = 0;
            for (uint64_t c = 0; c < ng; c++) { off[c] = total; total += sizes[c]; }
            if (total > dense_cap) { dense_cap = total + total / 4; SKX_TRY(d_dense.alloc(dense_cap)); }
            SKX_TRY(pin[flip].need(total));
            SKX_HIP(hipMemcpyAsync(d_off.p, off.data(), ng * 8, hipMemcpyHostToDevice, st));
            launch_skf_gather(d_slots.p, d_sizes.p, d_off.p, (uint32_t)ng, d_dense.p, st);
            SKX_HIP(hipMemcpyAsync(pin[flip].p, d_dense.p, total, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            SKX_HIP(hipGetLastError());
            wr.join();
            if (!wr.ok) break;
            const uint8_t *buf = pin[flip].p;
            wr.th = std::thread([&wr, buf, total, f]() { wr.ok = fwrite(buf, 1, total, f) == total; });
        }
        wr.join();
        if (!wr.ok) { set_error("short write %s", path); return SKX_EIO; }
        return SKX_OK;
    };
    const auto t_w0 = std::chrono::steady_clock::now();
    const int r = skf_write_stream(path, m, keys, counts, fetch, 0, &dev_encode, &fast);
    phase_add("save.total_stream", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count());
    return r;
    });
}

// MergeSkaArray::load (merge_ska_array.rs:201-204), streamed: row blocks are transposed into the sample-major matrix as
// they are decoded
extern "C" int skx_skf_peek_k(const char *path) { return path ? skx::skf_peek_k(path) : 0; }

extern "C" int skx_array_load(skx_ctx *ctx, const char *path, int want_bits, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !path || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx;
    SkfMeta m; std::vector<skx_key> keys; std::vector<uint64_t> counts;
    DevBuf<uint8_t> d_blk; uint64_t blk_cap = 0, S = 0;
    auto begin_rows = [&](uint64_t U, uint64_t cols) -> int {
        S = cols;
        if (S == 0 || S > 65535) { set_error("skf: unsupported number of samples"); return SKX_EFORMAT; }
        // (the split k-mer list comes before the matrix in the file: a list that does not fit the width asked for is refused HERE, before the data
        //  section is decoded -- `ska merge` of k > 31 files tries 64 bits first, as lib.rs:635-661 does, and was loading its first file twice)
        if (want_bits == 64)
            for (auto &kk : keys) if (kk.hi) { set_error("split k-mer does not fit 64 bits"); return SKX_EFORMAT; }
        a->n_rows = U; a->pitch = pitch_for(U);
        { PhaseTimer t_al("load.matrix_alloc"); SKX_TRY(a->matrix.alloc(S * a->pitch)); }
        SKX_HIP(hipMemsetAsync(a->matrix.p, '-', S * a->pitch, st));
        return SKX_OK;
    };
    auto sink = [&](uint64_t row0, uint64_t nr, const uint8_t *src) -> int {
        if (nr * S > blk_cap) { blk_cap = nr * S; SKX_TRY(d_blk.alloc(blk_cap)); }
        SKX_HIP(hipMemcpyAsync(d_blk.p, src, nr * S, hipMemcpyHostToDevice, st));
        launch_transpose(d_blk.p, S, nr, S, a->matrix.p + row0, a->pitch, st);          // [nr][S] -> columns row0.. of [S][pitch]
        SKX_HIP(hipStreamSynchronize(st));                                                // src is reused by the decoder
        return SKX_OK;
    };
    // the data section on the device: groups of compressed chunks -> row-major cells -> transposed into the matrix
    const DevDecode dev_decode = [&](const uint8_t *file, const SkfChunk *ch, size_t nch, uint64_t upos, uint64_t U, uint64_t cols) -> int {
        if (!U || !cols) return SKF_NOT_TAKEN;
        const uint64_t uend = upos + 2 * U * cols, G = skf_group_chunks();
        size_t c0 = 0, c1 = nch;
        { size_t lo = 0, hi = nch; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (ch[mid].uoff + ch[mid].ulen <= upos) lo = mid + 1; else hi = mid; } c0 = lo; }
        { size_t lo = c0, hi = nch; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (ch[mid].uoff < uend) lo = mid + 1; else hi = mid; } c1 = lo; }
        if (c0 >= c1) return SKF_NOT_TAKEN;
        if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] load: %zu chunks of the data section decoded on the device\n", c1 - c0);
        DevBuf<uint8_t> d_src, d_cells[2], d_scratch; DevBuf<SnapChunk> d_chunks; DevBuf<int> d_status;
        SKX_TRY(d_scratch.alloc(std::min<uint64_t>(G, c1 - c0) * 65536ull + 16));
        SKX_TRY(d_status.alloc(1)); SKX_TRY(d_status.zero(st));
        const uint64_t cells_cap = std::min<uint64_t>(G, c1 - c0) * 32768ull + cols + 64;
        SKX_TRY(d_cells[0].alloc(cells_cap)); SKX_TRY(d_cells[1].alloc(cells_cap)); SKX_TRY(d_chunks.alloc(std::min<uint64_t>(G, c1 - c0)));
        std::vector<SnapChunk> tab;
        uint64_t src_cap = 0, row_lo = 0, have_hi = 0;          // rows < row_lo are in the matrix; cells [row_lo * cols, have_hi) wait in the current buffer
        int cur = 0;
        GroupStager stg(path);
        int sw = 0;
        for (size_t g0 = c0; g0 < c1; g0 += G, sw ^= 1) {
            const size_t g1 = std::min<size_t>(c1, g0 + G);
            const size_t f_lo = ch[g0].off, f_hi = ch[g1 - 1].off + ch[g1 - 1].len;
            if (f_hi - f_lo + 512 > src_cap) { src_cap = f_hi - f_lo + 512; SKX_TRY(d_src.alloc(src_cap)); }      // + the decoder's read-ahead window
            tab.resize(g1 - g0);
            for (size_t c = g0; c < g1; c++) tab[c - g0] = SnapChunk{ch[c].off - f_lo, ch[c].uoff, (uint32_t)ch[c].len, ch[c].ulen, ch[c].crc, ch[c].compressed ? 1u : 0u};
            // this group's bytes: staged by the pread team (started a group ago; now, for the first), or straight from the mapping
            const uint8_t *src_bytes = nullptr;
            if (stg.on()) { src_bytes = stg.get(sw, f_lo, f_hi); if (!src_bytes && stg.start(sw, f_lo, f_hi)) src_bytes = stg.get(sw, f_lo, f_hi); }
            if (stg.on() && g1 < c1) { const size_t n1 = std::min<size_t>(c1, g1 + G); (void)stg.start(sw ^ 1, ch[g1].off, ch[n1 - 1].off + ch[n1 - 1].len); }
            SKX_HIP(hipMemcpyAsync(d_src.p, src_bytes ? src_bytes : file + f_lo, f_hi - f_lo, hipMemcpyHostToDevice, st));
            SKX_HIP(hipMemcpyAsync(d_chunks.p, tab.data(), tab.size() * sizeof(SnapChunk), hipMemcpyHostToDevice, st));
            const uint64_t base_cell = (row_lo * cols) & ~7ull;
            SKX_TRY(launch_skf_decode_cells(ctx->device, d_src.p, d_chunks.p, (uint32_t)(g1 - g0), upos, uend, d_scratch.p, d_cells[cur].p, base_cell, d_status.p, st));
            // cells this group delivered: value bytes (odd section offsets) below the end of its last chunk
            const uint64_t s_hi = std::min(ch[g1 - 1].uoff + ch[g1 - 1].ulen, uend);
            have_hi = (s_hi - upos) >> 1;
            const uint64_t row_done = g1 == c1 ? U : have_hi / cols;
            const uint8_t *in = d_cells[cur].p + (row_lo * cols - base_cell);
            for (uint64_t r = row_lo; r < row_done; r += 4000000ull) {                          // grid.y of the transpose: 65 535 tiles of 64 rows
                const uint64_t nr = std::min<uint64_t>(4000000ull, row_done - r);
                launch_transpose(in + (r - row_lo) * cols, cols, nr, cols, a->matrix.p + r, a->pitch, st);
            }
            if (g1 < c1) {                                                                      // the unfinished row moves to the other buffer
                const uint64_t nb = (row_done * cols) & ~7ull, left = have_hi - row_done * cols;
                if (left) SKX_HIP(hipMemcpyAsync(d_cells[cur ^ 1].p + (row_done * cols - nb), d_cells[cur].p + (row_done * cols - base_cell), left, hipMemcpyDeviceToDevice, st));
            }
            SKX_HIP(hipStreamSynchronize(st));                                                   // tab / d_src are reused
            row_lo = row_done; cur ^= 1;
        }
        int status = 0;
        SKX_HIP(hipMemcpy(&status, d_status.p, 4, hipMemcpyDeviceToHost));
        SKX_HIP(hipGetLastError());
        if (status == 3) {                                                                       // cells that are not (0x18, byte): the generic decoder's case
            SKX_HIP(hipMemsetAsync(a->matrix.p, '-', cols * a->pitch, st));
            return SKF_NOT_TAKEN;
        }
        if (status) { set_error(status == 2 ? "skf: checksum mismatch" : "skf: corrupt snappy block"); return SKX_EFORMAT; }
        return SKX_OK;
    };
    const auto t_r0 = std::chrono::steady_clock::now();
    SKX_TRY(skf_read_stream(path, m, keys, counts, begin_rows, sink, 0, &dev_decode));
    phase_add("load.total_stream", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_r0).count());
    PhaseTimer t_fin("load.keys_upload_stats");
    SKX_TRY(check_k(m.k));
    if (want_bits == 64)       // serde into Vec<u64> fails on wider values; lib.rs:635-661 then retries as u128
        for (auto &kk : keys) if (kk.hi) { set_error("split k-mer does not fit 64 bits"); return SKX_EFORMAT; }
    const uint64_t U = a->n_rows;
    if (keys.size() != U) { set_error("skf: split_kmers and variants disagree"); return SKX_EFORMAT; }
    a->k = m.k; a->rc = m.rc; a->k_bits = m.k_bits; a->hp = make_hash_params(std::min(m.k, 31)); a->wh = make_wide_hash(m.k);
    a->version = m.version.empty() ? skx_version() : m.version; a->names = m.names;
    a->n_kmers = U; a->engine_order = false;
    SKX_TRY(a->present.alloc(U)); SKX_TRY(a->unambig.alloc(U)); SKX_TRY(a->mask.alloc(U)); SKX_TRY(a->vcount.alloc(U));
    if (m.k <= 31) {
        SKX_TRY(a->keys.alloc(U));
        std::vector<uint64_t> lo(U);
        for (uint64_t i = 0; i < U; i++) lo[i] = keys[i].lo;
        DevBuf<uint64_t> tmp; SKX_TRY(tmp.alloc(U));
        if (U) SKX_HIP(hipMemcpyAsync(tmp.p, lo.data(), U * 8, hipMemcpyHostToDevice, st));
        launch_hash_keys(tmp.p, a->keys.p, U, a->hp, st);
        SKX_HIP(hipStreamSynchronize(st));
    } else a->host_keys.swap(keys);
    DevBuf<int> d_bad; SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st));
    if (U) {
        launch_col_stats(a->matrix.p, a->pitch, (int)S, U, a->present.p, a->unambig.p, a->mask.p, d_bad.p, st);
        if (counts.size() == U) {
            std::vector<uint32_t> vc(U);
            for (uint64_t i = 0; i < U; i++) vc[i] = (uint32_t)counts[i];
            SKX_HIP(hipMemcpyAsync(a->vcount.p, vc.data(), U * 4, hipMemcpyHostToDevice, st));
            SKX_HIP(hipStreamSynchronize(st));
        } else SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, U * 4, hipMemcpyDeviceToDevice, st));
    }
    int bad = 0;
    SKX_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    if (bad) { set_error("variants contain a byte outside -ACGTMRWSYKVHDBN (not supported on the device path)"); return SKX_EUNSUP; }
    *out = a.release();
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ load + filter in one pass
// `ska align x.skf` / `ska distance x.skf` (generic_modes.rs:22-50,136-189 on io_utils::load_array's single-file branch).  The
// general form -- skx_array_load, then the filter(s) -- holds the unfiltered rows x samples matrix (22.6 GB for 1 000 x 5 Mbp) only
// to drop four fifths of its rows.  Here each group of rows leaves the decoder row-major, gets its statistics and its filter
// verdict, and only the kept rows are transposed into the (small) sample-major matrix.  The filter's min_count test reads the
// file's variant_count, which is stored AFTER the matrix: the pass runs on the count the rows imply (what every writer stores)
// while a host thread parses the stored list, and a mismatch sends the file through the general path instead.
static int load_then_filter(skx_ctx *ctx, const char *path, const skx_filter_spec *f, skx_array **out, int64_t *removed, int64_t *constant)
{
    skx_array *a = nullptr;
    int r = skx_array_load(ctx, path, 64, &a);
    if (r == SKX_EFORMAT) r = skx_array_load(ctx, path, 128, &a);
    if (r != SKX_OK) return r;
    const uint64_t S = a->names.size();
    int32_t rem = 0, cst = 0;
    if (f->two_stage) {
        if (f->min_freq * (double)S >= 1.0) r = skx_array_filter(a, (uint64_t)std::ceil((double)S * f->min_freq), 0, SKX_FILTER_NONE, 0, 0, 0, &rem);
        if (r == SKX_OK) r = skx_array_filter(a, 0, 0, SKX_FILTER_NO_CONST, 0, 0, 0, &cst);
    } else
        r = skx_array_filter(a, (uint64_t)std::ceil((double)S * f->min_freq), f->filter_ambig_as_missing, f->filter_type, f->mask_ambig, f->ignore_const_gaps, 0, &rem);
    if (r != SKX_OK) { skx_array_free(a); return r; }
    if (removed) *removed = rem;
    if (constant) *constant = cst;
    *out = a;
    return SKX_OK;
}

extern "C" int skx_array_load_filtered(skx_ctx *ctx, const char *path, const skx_filter_spec *f, skx_array **out, int64_t *removed, int64_t *constant)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !path || !f || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    // the output hint applies to exactly this call, whichever path it takes (a stale descriptor number may belong to another file by
    // the time of a later call)
    const int expect_fd = ctx->expect_fd; ctx->expect_fd = -1;
    if (knob("no_stream_load")) return load_then_filter(ctx, path, f, out, removed, constant);
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    struct Release { std::chrono::steady_clock::time_point t; bool on = false;
                     ~Release() { if (on) phase_add("load.release_buffers_file", std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); } } rel;
    std::unique_ptr<SkfFile> sfp(new SkfFile());
    SkfFile &sf = *sfp;
    {
        PhaseTimer t_open("load.open_header");
        const int r = sf.open(path);
        if (r == SKF_NOT_TAKEN) return load_then_filter(ctx, path, f, out, removed, constant);
        if (r != SKX_OK) return r;
    }
    const uint64_t U = sf.m.n_rows, S = sf.m.names.size();
    if (!U || !S) return load_then_filter(ctx, path, f, out, removed, constant);
    SKX_TRY(check_k(sf.m.k));
    // the stored counts, parsed beside the device work
    std::vector<uint32_t> counts; int tail_rc = SKX_OK; std::string tail_err;
    std::thread tail([&]() { tail_rc = sf.read_tail(counts); if (tail_rc != SKX_OK) tail_err = skx_last_error(); });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_tail{tail};

    const uint64_t upos = sf.upos_data, uend = upos + 2 * U * S, G = skf_group_chunks();
    const SkfChunk *ch = sf.chunks(); const uint8_t *file = sf.file();
    const size_t c0 = sf.chunk_of(upos);                     // first chunk of the data section (the walk is past it: the header was read there)
    if (c0 >= sf.n_chunks()) { const int wr = sf.walk_result(); if (wr != SKX_OK && wr != SKF_NOT_TAKEN) return SKX_EFORMAT; return load_then_filter(ctx, path, f, out, removed, constant); }

    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx; a->k = sf.m.k; a->rc = sf.m.rc; a->hp = make_hash_params(std::min(sf.m.k, 31)); a->wh = make_wide_hash(sf.m.k);
    a->names = sf.m.names; a->engine_order = false; a->keys_absent = true; a->n_kmers = U;
    const uint64_t thr = f->two_stage ? (f->min_freq * (double)S >= 1.0 ? (uint64_t)std::ceil((double)S * f->min_freq) : 0) : (uint64_t)std::ceil((double)S * f->min_freq);

    PhaseTimer t_data("load.stream_decode_filter");
    DevBuf<uint8_t> d_src, d_cells[2], d_scratch, keep; DevBuf<SnapChunk> d_chunks; DevBuf<int> d_status, d_bad;
    DevBuf<uint32_t> present, unambig, mask, sc_sums; DevBuf<uint64_t> pos, sc_offs;
    const uint64_t gmax = std::min<uint64_t>(G, (uend - upos) / 65536 + 2);
    SKX_TRY(d_scratch.alloc(gmax * 65536ull + 16));
    SKX_TRY(d_status.alloc(1)); SKX_TRY(d_status.zero(st)); SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st));
    const uint64_t cells_cap = gmax * 32768ull + S + 64, rows_cap = gmax * 32768ull / S + 2;
    SKX_TRY(d_cells[0].alloc(cells_cap)); SKX_TRY(d_cells[1].alloc(cells_cap)); SKX_TRY(d_chunks.alloc(gmax));
    SKX_TRY(present.alloc(U)); SKX_TRY(unambig.alloc(U)); SKX_TRY(mask.alloc(U)); SKX_TRY(keep.alloc(U));
    SKX_TRY(pos.alloc(rows_cap + 1)); SKX_TRY(sc_sums.alloc(scan_u8_blocks(rows_cap) + 1)); SKX_TRY(sc_offs.alloc(scan_u8_blocks(rows_cap) + 2));
    // kept rows: capacity grows when a file keeps more than the first guess (a quarter of its rows)
    // (`ska distance` without --min-freq keeps every row that varies -- nearly all of them: sized for that from the start)
    uint64_t cap = pitch_for(f->two_stage && thr <= 1 ? U : std::max<uint64_t>(U / 4, 1024)), kept = 0;
    SKX_TRY(a->matrix.alloc(S * cap));
    std::vector<SnapChunk> tab;
    uint64_t src_cap = 0, row_lo = 0, have_hi = 0;
    int cur = 0;
    // the alignment's destination, when the caller announced it: its pages are allocated as the kept rows accumulate
    uint64_t name_bytes = 0;
    for (auto &nm : a->names) name_bytes += nm.size() + 3;
    if (expect_fd >= 0 && !f->two_stage) {
        off_t opos;
        if (mappable_output_fd(expect_fd, &opos)) a->prealloc = std::make_shared<Preallocator>(expect_fd, opos);
    }
    bool last_group = false;
    GroupStager stg(path);
    int sw = 0; size_t planned_g1 = 0;                                                      // the group whose bytes the stager is reading ahead ends here (0: none)
    for (size_t g0 = c0; !last_group; sw ^= 1) {
        // the group's chunks as far as the walker has come (it stays ahead of the device: ~3 M chunks/s against ~2.5 M decoded)
        // (at most gmax chunks: the buffers above are sized for gmax chunks of up to 64 KB each; a file framed in smaller chunks has more
        // chunks than 64 KB pieces and simply takes more groups)
        if (!planned_g1) (void)sf.wait_chunk(g0 + gmax - 1);
        size_t g1 = planned_g1 ? planned_g1 : std::min<size_t>(sf.n_chunks(), g0 + gmax);
        planned_g1 = 0;
        if (g1 <= g0) {                                                                      // the walk ended before the data section did
            const int wr = sf.walk_result();
            if (wr == SKF_NOT_TAKEN) { a->prealloc.reset(); a.reset(); return load_then_filter(ctx, path, f, out, removed, constant); }
            if (wr != SKX_OK) return wr;
            set_error("skf: truncated frame"); return SKX_EFORMAT;
        }
        for (size_t c = g0; c < g1; c++) if (ch[c].uoff + ch[c].ulen >= uend) { g1 = c + 1; last_group = true; break; }
        const size_t f_lo = ch[g0].off, f_hi = ch[g1 - 1].off + ch[g1 - 1].len;
        if (f_hi - f_lo + 512 > src_cap) { src_cap = f_hi - f_lo + 512; SKX_TRY(d_src.alloc(src_cap)); }
        tab.resize(g1 - g0);
        for (size_t c = g0; c < g1; c++) tab[c - g0] = SnapChunk{ch[c].off - f_lo, ch[c].uoff, (uint32_t)ch[c].len, ch[c].ulen, ch[c].crc, ch[c].compressed ? 1u : 0u};
        // this group's bytes: staged by the pread team (started a group ago; now, for the first), or straight from the mapping; then the next
        // group as far as the walker has come is planned and its bytes are asked for
        const uint8_t *src_bytes = nullptr;
        if (stg.on()) { src_bytes = stg.get(sw, f_lo, f_hi); if (!src_bytes && stg.start(sw, f_lo, f_hi)) src_bytes = stg.get(sw, f_lo, f_hi); }
        if (stg.on() && !last_group) {
            size_t n1 = std::min<size_t>(sf.n_chunks(), g1 + gmax);
            for (size_t c = g1; c < n1; c++) if (ch[c].uoff + ch[c].ulen >= uend) { n1 = c + 1; break; }
            if (n1 > g1 && stg.start(sw ^ 1, ch[g1].off, ch[n1 - 1].off + ch[n1 - 1].len)) planned_g1 = n1;
        }
        SKX_HIP(hipMemcpyAsync(d_src.p, src_bytes ? src_bytes : file + f_lo, f_hi - f_lo, hipMemcpyHostToDevice, st));
        SKX_HIP(hipMemcpyAsync(d_chunks.p, tab.data(), tab.size() * sizeof(SnapChunk), hipMemcpyHostToDevice, st));
        const uint64_t base_cell = (row_lo * S) & ~7ull;
        SKX_TRY(launch_skf_decode_cells(ctx->device, d_src.p, d_chunks.p, (uint32_t)(g1 - g0), upos, uend, d_scratch.p, d_cells[cur].p, base_cell, d_status.p, st));
        const uint64_t s_hi = std::min(ch[g1 - 1].uoff + ch[g1 - 1].ulen, uend);
        have_hi = (s_hi - upos) >> 1;
        const uint64_t row_done = last_group ? U : have_hi / S;
        const uint8_t *in = d_cells[cur].p + (row_lo * S - base_cell);
        const uint64_t nr = row_done - row_lo;
        uint64_t gk = 0;
        if (a->prealloc && kept) a->prealloc->raise(name_bytes + S * kept);
        if (nr) {
            launch_row_stats_rm(in, S, nr, present.p + row_lo, unambig.p + row_lo, mask.p + row_lo, d_bad.p, st);
            FilterArgs fa{present.p + row_lo, present.p + row_lo, unambig.p + row_lo, mask.p + row_lo, nr, (uint32_t)S, thr, f->two_stage ? 0 : f->filter_ambig_as_missing,
                          f->two_stage ? SKX_FILTER_NO_CONST : f->filter_type, f->two_stage ? 0 : f->ignore_const_gaps, keep.p + row_lo, f->two_stage};
            launch_filter_flags(fa, st);
            launch_scan_u8(keep.p + row_lo, pos.p, nr, sc_sums.p, sc_offs.p, st);
            SKX_HIP(hipMemcpyAsync(&gk, pos.p + nr, 8, hipMemcpyDeviceToHost, st));
        }
        if (!last_group) {                                                                  // the unfinished row moves to the other buffer
            const uint64_t nb = (row_done * S) & ~7ull, left = have_hi - row_done * S;
            if (left) SKX_HIP(hipMemcpyAsync(d_cells[cur ^ 1].p + (row_done * S - nb), d_cells[cur].p + (row_done * S - base_cell), left, hipMemcpyDeviceToDevice, st));
        }
        SKX_HIP(hipStreamSynchronize(st));                                                   // gk; tab / d_src are reused
        if (kept + gk > cap - 256) {                                                        // more rows survive than guessed: a wider matrix
            const uint64_t ncap = pitch_for(std::max((kept + gk) * 2, cap * 2));
            DevBuf<uint8_t> nm; SKX_TRY(nm.alloc(S * ncap));
            if (kept) SKX_HIP(hipMemcpy2DAsync(nm.p, ncap, a->matrix.p, cap, kept, S, hipMemcpyDeviceToDevice, st));
            SKX_HIP(hipStreamSynchronize(st));
            a->matrix = std::move(nm); cap = ncap;
        }
        if (gk) launch_compact_rm(in, S, nr, keep.p + row_lo, pos.p, a->matrix.p, cap, kept, f->two_stage ? 0 : f->mask_ambig, st);
        kept += gk;
        row_lo = row_done; cur ^= 1;
        g0 = g1;
    }
    {   // framing errors anywhere in the file
        const int wr = sf.walk_result();
        if (wr == SKF_NOT_TAKEN) { a->prealloc.reset(); a.reset(); return load_then_filter(ctx, path, f, out, removed, constant); }
        if (wr != SKX_OK) return wr;
    }
    int status = 0, bad = 0;
    SKX_HIP(hipMemcpyAsync(&status, d_status.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    t_data.stop();
    if (status == 3) { tail.join(); a.reset(); return load_then_filter(ctx, path, f, out, removed, constant); }    // cells that are not (0x18, byte)
    if (status) { set_error(status == 2 ? "skf: checksum mismatch" : "skf: corrupt snappy block"); return SKX_EFORMAT; }
    if (bad) { set_error("variants contain a byte outside -ACGTMRWSYKVHDBN (not supported on the device path)"); return SKX_EUNSUP; }
    // the stored counts: must be what the rows imply for the verdicts above to stand
    tail.join();
    if (tail_rc != SKX_OK) { set_error("%s", tail_err.c_str()); return tail_rc; }
    PhaseTimer t_fin("load.counts_check_statistics");
    DevBuf<uint32_t> d_counts; SKX_TRY(d_counts.alloc(U));
    SKX_HIP(hipMemcpyAsync(d_counts.p, counts.data(), U * 4, hipMemcpyHostToDevice, st));
    if (!(f->filter_ambig_as_missing && !f->two_stage)) {                                   // update_counts(true) ignores the stored counts
        SKX_TRY(d_status.zero(st));
        launch_differ_u32(d_counts.p, present.p, U, d_status.p, st);
        int differ = 0;
        SKX_HIP(hipMemcpyAsync(&differ, d_status.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        if (differ) { a.reset(); return load_then_filter(ctx, path, f, out, removed, constant); }
    }
    // statistics of the kept rows, in row order
    unsigned long long n_silent = 0, n_const = 0;
    {
        DevBuf<unsigned long long> d_cnt; SKX_TRY(d_cnt.alloc(2)); SKX_TRY(d_cnt.zero(st));
        launch_count_u8(keep.p, U, 2, d_cnt.p, st); launch_count_u8(keep.p, U, 3, d_cnt.p + 1, st);
        unsigned long long h[2] = {0, 0};
        SKX_HIP(hipMemcpyAsync(h, d_cnt.p, 16, hipMemcpyDeviceToHost, st));
        DevBuf<uint64_t> gpos; DevBuf<uint32_t> s2; DevBuf<uint64_t> o2;
        SKX_TRY(gpos.alloc(U + 1)); SKX_TRY(s2.alloc(scan_u8_blocks(U) + 1)); SKX_TRY(o2.alloc(scan_u8_blocks(U) + 2));
        launch_scan_u8(keep.p, gpos.p, U, s2.p, o2.p, st);
        SKX_TRY(a->present.alloc(kept)); SKX_TRY(a->unambig.alloc(kept)); SKX_TRY(a->mask.alloc(kept)); SKX_TRY(a->vcount.alloc(kept));
        launch_compact_u32(present.p, a->present.p, U, keep.p, gpos.p, st);
        launch_compact_u32(unambig.p, a->unambig.p, U, keep.p, gpos.p, st);
        launch_compact_u32(mask.p, a->mask.p, U, keep.p, gpos.p, st);
        launch_compact_u32((f->filter_ambig_as_missing && !f->two_stage) ? unambig.p : d_counts.p, a->vcount.p, U, keep.p, gpos.p, st);
        if (f->mask_ambig && !f->two_stage) launch_mask_ambig_stats(a->mask.p, kept, st);
        SKX_HIP(hipStreamSynchronize(st));
        n_silent = h[0]; n_const = h[1];
    }
    SKX_HIP(hipGetLastError());
    a->k_bits = sf.m.k_bits; a->version = sf.m.version.empty() ? skx_version() : sf.m.version;
    a->n_rows = kept; a->pitch = cap;
    if (removed) *removed = (int64_t)(U - kept - n_silent - n_const);
    if (constant) *constant = (int64_t)n_const;
    rel.t = std::chrono::steady_clock::now(); rel.on = true;                              // what follows is the destructors
    // (unmapping the 700 000 pages of the file takes 0.07-0.13 s; handing it to a helper thread moved that time into the write that
    // follows -- its page faults and pinned allocations wait for the same address-space lock -- so it stays here)
    *out = a.release();
    return SKX_OK;
    });
}
