// ska_host.cpp -- host side above the C ABI: the reference's mode glue for the hot path
// (generic_modes.rs: align / apply_filters / distance / save_skf; io_utils.rs: read_input_fastas /
// get_input_list / load_array / set_ostream; cli.rs + lib.rs:557-727,808-827 for build | align | distance | nk).
// Written in C++ because the image has no Rust toolchain; it only talks to the engine through include/skx.h.
#include "../../include/skx_host.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <sstream>
#include <atomic>
#include <thread>
#include <ctime>
#include <string>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

namespace {

void put(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void put(std::string &s, const char *fmt, ...)
{
    char tmp[256];
    va_list ap; va_start(ap, fmt); int n = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (n < (int)sizeof tmp) { s.append(tmp, (size_t)n); return; }
    std::vector<char> big((size_t)n + 1);
    va_start(ap, fmt); vsnprintf(big.data(), big.size(), fmt, ap); va_end(ap);
    s.append(big.data(), (size_t)n);
}
int to_buf(const std::string &s, char **buf, uint64_t *len)
{
    char *p = (char *)malloc(s.size() + 1);
    if (!p) return SKX_ENOMEM;
    memcpy(p, s.data(), s.size()); p[s.size()] = 0;
    *buf = p; *len = s.size();
    return SKX_OK;
}
bool ends_with_ci(const std::string &s, const char *suf)
{
    size_t m = strlen(suf);
    if (s.size() < m) return false;
    for (size_t i = 0; i < m; i++) if (tolower((unsigned char)s[s.size() - m + i]) != suf[i]) return false;
    return true;
}
// Rust `{:?}` of a String
void rust_debug(std::string &o, const std::string &s)
{
    o += '"';
    for (unsigned char c : s) {
        if (c == '"') o += "\\\""; else if (c == '\\') o += "\\\\"; else if (c == '\n') o += "\\n";
        else if (c == '\r') o += "\\r"; else if (c == '\t') o += "\\t";
        else if (c < 0x20 || c == 0x7f) put(o, "\\u{%x}", c);
        else o += (char)c;
    }
    o += '"';
}
void skx_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void skx_set_error(const char *fmt, ...)
{
    char tmp[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    skx_set_last_error(tmp);
}
struct Phase {       // wall-clock phase recorded through the ABI (skx_phase_add)
    const char *name; std::chrono::steady_clock::time_point t0;
    explicit Phase(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    void stop() { if (name) { skx_phase_add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); name = nullptr; } }
    ~Phase() { stop(); }
};
template <typename F>
int skx_guarded(F &&f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc &) { return SKX_ENOMEM; }
    catch (...) { return SKX_EINVAL; }
}
}  // namespace

extern "C" char *skh_sample_name(const char *path)
{
    // ^.+/(.+)\.(?i:fa|fasta|fastq|fastq\.gz)$  then  ^(.+)\.(?i:...)$  else the whole argument (io_utils.rs:31-46)
    static const char *exts[] = {".fa", ".fasta", ".fastq", ".fastq.gz"};     // greedy (.+): shortest extension wins
    const std::string s(path);
    for (const char *e : exts) {
        if (!ends_with_ci(s, e)) continue;
        const size_t stem_end = s.size() - strlen(e);
        for (size_t i = stem_end; i-- > 1;)
            if (s[i] == '/' && i + 1 < stem_end) return strdup(s.substr(i + 1, stem_end - i - 1).c_str());
    }
    for (const char *e : exts)
        if (ends_with_ci(s, e) && s.size() > strlen(e)) return strdup(s.substr(0, s.size() - strlen(e)).c_str());
    return strdup(path);
}

extern "C" int skh_apply_filters(skx_array *a, double min_freq, int filter_ambig_as_missing, int filter_type, int ambig_mask,
                                 int ignore_const_gaps, int32_t *removed)
{
    return skx_guarded([&]() -> int {
    skx_array_info_t info; skx_array_info(a, &info);
    const uint64_t threshold = (uint64_t)std::ceil((double)info.total_samples * min_freq);   // generic_modes.rs:121 (nsamples() over all slabs)
    return skx_array_filter(a, threshold, filter_ambig_as_missing, filter_type, ambig_mask, ignore_const_gaps, /*update_kmers=*/0, removed);
    });
}

extern "C" int skh_align(skx_array *a, int filter_type, int mask_ambig, int ignore_const_gaps, double min_freq, int filter_ambig_as_missing,
                         char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    int32_t removed = 0;
    int r = skh_apply_filters(a, min_freq, filter_ambig_as_missing, filter_type, mask_ambig, ignore_const_gaps, &removed);
    if (r != SKX_OK) return r;
    return skx_array_fasta(a, buf, len);
    });
}

// the same, written straight to a file descriptor (the alignment of a large array is never held on the host)
extern "C" int skh_align_fd(skx_array *a, int filter_type, int mask_ambig, int ignore_const_gaps, double min_freq, int filter_ambig_as_missing, int fd)
{
    return skx_guarded([&]() -> int {
    int32_t removed = 0;
    Phase pf("align.filter");
    int r = skh_apply_filters(a, min_freq, filter_ambig_as_missing, filter_type, mask_ambig, ignore_const_gaps, &removed);
    pf.stop();
    if (r != SKX_OK) return r;
    Phase pw("align.write_fasta");
    return skx_array_write_fasta(a, fd);
    });
}

static int distance_text(skx_array *a, const std::vector<skx_dist> &d, char **buf, uint64_t *len);
static int distance_text_names(const std::vector<const char *> &names, const skx_dist *d, char **buf, uint64_t *len);
extern "C" int skh_distance_tsv(skx_array *a, double min_freq, int filt_ambig, char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    skx_array_info_t info; skx_array_info(a, &info);
    const uint64_t S = info.n_samples;
    std::vector<skx_dist> d(S * (S - 1) / 2 + 1);
    int64_t constant = 0; uint64_t rows = 0; int r;
    // the two filters of generic_modes.rs:149-168 are applied while the bit planes are built; the array itself stays as it is
    { Phase pd("distance.pair_sweep"); if ((r = skx_array_distance_filtered(a, min_freq, filt_ambig, d.data(), &constant, &rows)) != SKX_OK) return r; }
    return distance_text(a, d, buf, len);
    });
}

static int distance_text(skx_array *a, const std::vector<skx_dist> &d, char **buf, uint64_t *len)
{
    skx_array_info_t info; skx_array_info(a, &info);
    std::vector<const char *> names(info.n_samples);
    for (uint64_t i = 0; i < info.n_samples; i++) names[i] = skx_array_name(a, i);
    return distance_text_names(names, d.data(), buf, len);
}
static int distance_text_names(const std::vector<const char *> &names, const skx_dist *d, char **buf, uint64_t *len)
{
    const uint64_t S = names.size();
    Phase pt("distance.table_text");
    // the rows of one first sample are formatted by one thread (a table of 1 000 samples is half a million printf calls, 20 MB)
    const int T = (int)std::min<uint64_t>(std::max<uint64_t>(1, S / 16), std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
    std::vector<std::string> part(S);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        for (uint64_t i; (i = next.fetch_add(1)) < S;) {
            size_t n = (size_t)(i * (2 * S - i - 1) / 2);               // pairs before row i
            std::string &o = part[i];
            for (uint64_t j = i + 1; j < S; j++, n++)
                put(o, "%s\t%s\t%.2f\t%.5f\t%llu\t%llu\n", names[i], names[j], d[n].distance, d[n].mismatch_prop,
                    (unsigned long long)d[n].match_count, (unsigned long long)d[n].mismatch_count);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    std::string out = "Sample1\tSample2\tDistance\tMismatches (proportion)\tMatch count\tMismatch count\n";
    size_t total = out.size();
    for (auto &x : part) total += x.size();
    out.reserve(total);
    for (auto &x : part) out += x;
    return to_buf(out, buf, len);
}
static int distance_table(skx_array *a, double constant, int filt_ambig, char **buf, uint64_t *len)
{
    skx_array_info_t info; skx_array_info(a, &info);
    const uint64_t S = info.n_samples;
    std::vector<skx_dist> d(S * (S - 1) / 2 + 1);
    int r;
    { Phase pd("distance.pair_sweep"); if ((r = skx_array_distance(a, constant, filt_ambig, d.data())) != SKX_OK) return r; }
    return distance_text(a, d, buf, len);
}

extern "C" int skh_distance_skf_tsv(skx_ctx *ctx, const char *skf_file, double min_freq, int filt_ambig, char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    skh_log(2, "ska::generic_modes", "Calculating distances");                                            // generic_modes.rs:170
    skx_filter_spec fs{min_freq, 0, SKX_FILTER_NO_CONST, 0, 0, 1};
    skx_array *a = nullptr; int64_t removed = 0, constant = 0;
    int r = skx_array_load_filtered(ctx, skf_file, &fs, &a, &removed, &constant);
    if (r != SKX_OK) return r;
    r = distance_table(a, (double)constant, filt_ambig, buf, len);
    skx_array_free(a);
    return r;
    });
}

extern "C" int skh_align_inputs_fd(skx_ctx *ctx, const char *const *inputs, int n_inputs, int threads, int filter_type, int mask_ambig, int ignore_const_gaps,
                                   double min_freq, int filter_ambig_as_missing, int fd)
{
    return skx_guarded([&]() -> int {
    skx_array *a = nullptr; int r;
    if (n_inputs == 1) {
        skx_filter_spec fs{min_freq, filter_ambig_as_missing, filter_type, mask_ambig, ignore_const_gaps, 0};
        int64_t removed = 0;
        skx_ctx_expect_output(ctx, fd);                               // the file's pages are allocated while the rows are read
        skh_log(2, "ska::io_utils", "Single file as input, trying to load as skf 64-bits");                  // io_utils.rs:66-69
        { Phase pl("align.load_filtered"); if ((r = skx_array_load_filtered(ctx, inputs[0], &fs, &a, &removed, nullptr)) != SKX_OK) return r; }
        {   // generic_modes.rs:121-122, merge_ska_array.rs:385 (here the rows are filtered as they leave the decoder: the lines follow the work)
            skx_array_info_t info; skx_array_info(a, &info);
            static const char *const FN[] = {"No filtering", "No constant sites", "No ambiguous sites", "No constant sites or ambiguous bases"};
            char msg[320];
            snprintf(msg, sizeof msg, "Applying filters: threshold=%llu constant_site_filter=%s filter_ambig_as_missing=%s ambig_mask=%s no_gap_only_sites=%s",
                     (unsigned long long)std::ceil((double)info.n_samples * min_freq), FN[filter_type & 3], filter_ambig_as_missing ? "true" : "false", mask_ambig ? "true" : "false",
                     ignore_const_gaps ? "true" : "false");
            skh_log(2, "ska::generic_modes", msg);
            snprintf(msg, sizeof msg, "Filtering removed %lld split k-mers", (long long)removed);
            skh_log(2, "ska::merge_ska_array", msg);
        }
        skh_log(2, "ska::generic_modes", "Writing alignment");                                              // generic_modes.rs:45
        Phase pw("align.write_fasta");
        r = skx_array_write_fasta(a, fd);
    } else {
        skh_log(2, "ska::io_utils", "Multiple files as input, running ska build with default settings");      // io_utils.rs:76
        if ((r = skh_load_array(ctx, inputs, n_inputs, threads, &a)) != SKX_OK) return r;
        skh_log(2, "ska::generic_modes", "Writing alignment");
        r = skh_align_fd(a, filter_type, mask_ambig, ignore_const_gaps, min_freq, filter_ambig_as_missing, fd);
    }
    { Phase pf("align.free_array"); skx_array_free(a); }
    return r;
    });
}

extern "C" int skh_nk(skx_array *a, int full_info, char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    skx_array_info_t info; skx_array_info(a, &info);
    std::string o;
    put(o, "ska_version=%s\nk=%d\nk_bits=%d\nrc=%s\nk-mers=%llu\nsamples=%llu\n", skx_array_version(a), info.k, info.k_bits,
        info.rc ? "true" : "false", (unsigned long long)info.n_kmers, (unsigned long long)info.n_samples);
    o += "sample_names=[";
    for (uint64_t s = 0; s < info.n_samples; s++) { if (s) o += ", "; rust_debug(o, skx_array_name(a, s)); }
    o += "]\nsample_kmers=[";
    std::vector<int64_t> sk(info.n_samples);
    int r = skx_array_sample_kmers(a, sk.data());
    if (r != SKX_OK) return r;
    for (uint64_t s = 0; s < info.n_samples; s++) put(o, "%s%lld", s ? ", " : "", (long long)sk[s]);
    o += "]\n\n";
    if (full_info) {
        std::vector<skx_key> keys(info.n_kmers);
        std::vector<uint8_t> var(info.n_rows * info.n_samples);
        if ((r = skx_array_export(a, keys.data(), var.data(), nullptr)) != SKX_OK) return r;
        const int half = (info.k - 1) / 2;
        static const char L[] = "ACTG";
        for (uint64_t row = 0; row < info.n_kmers && row < info.n_rows; row++) {
            unsigned __int128 key = ((unsigned __int128)keys[row].hi << 64) | keys[row].lo;
            std::string up(half, 'A'), lo(half, 'A');
            for (int i = 0; i < half; i++) { lo[half - 1 - i] = L[(int)(key & 3)]; key >>= 2; }
            for (int i = 0; i < half; i++) { up[half - 1 - i] = L[(int)(key & 3)]; key >>= 2; }
            o += up; o += '\t'; o += lo; o += '\t';
            for (uint64_t s = 0; s < info.n_samples; s++) { if (s) o += ','; uint8_t b = var[row * info.n_samples + s]; o += b ? (char)b : '-'; }
            o += '\n';
        }
        o += '\n';
    }
    return to_buf(o, buf, len);
    });
}

extern "C" int skh_save_skf(skx_array *a, const char *out_prefix)
{
    return skx_guarded([&]() -> int {
    std::string p(out_prefix);
    if (p.size() < 4 || p.compare(p.size() - 4, 4, ".skf") != 0) p += ".skf";           // generic_modes.rs:272-276
    return skx_array_save(a, p.c_str());
    });
}

extern "C" int skh_load_array(skx_ctx *ctx, const char *const *inputs, int n_inputs, int threads, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (n_inputs == 1) {                                                                // io_utils.rs:65-75, lib.rs:635-661
        int r = skx_array_load(ctx, inputs[0], 64, out);
        if (r != SKX_EFORMAT) return r;                                                 // only "does not fit u64" falls through to u128; a missing
        return skx_array_load(ctx, inputs[0], 128, out);                                // or corrupt file keeps its own message
    }
    // >1 inputs: `ska build` with defaults (k=31, rc, min_count 5, min_qual 20, strict), io_utils.rs:76-92
    std::vector<char *> names(n_inputs);
    for (int i = 0; i < n_inputs; i++) names[i] = skh_sample_name(inputs[i]);
    skx_qual q{5, 20, SKX_QUAL_STRICT};
    int r = skx_build_and_merge(ctx, names.data(), inputs, nullptr, n_inputs, 31, 1, &q, threads, 0.0, out);
    for (auto p : names) free(p);
    return r;
    });
}

extern "C" int skh_merge(skx_ctx *ctx, const char *const *skf_files, int n_files, const char *out_prefix)
{
    return skx_guarded([&]() -> int {
    if (n_files < 2) { skx_set_error("Need at least two files to merge"); return SKX_EINVAL; }                    // lib.rs:729-731
    std::vector<skx_array *> arrs(n_files, nullptr);
    auto cleanup = [&]() { for (auto p : arrs) if (p) skx_array_free(p); };
    const auto t_m0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    // The files side by side, a thread each (up to four): a load is half host work (the chunk walk, the split k-mer list, the stored counts)
    // and half device work on the context's one stream, so they fill each other's gaps (`ska merge` of four 5 M-row files: loads 1.6 of 3.4 s
    // one after the other, profiles/r06h_reads_1000.log).  The first file's `k` says which key width to ask for; without it (or if the first
    // file then refuses) the reference's order is followed: the first file as 64-bit keys, then as 128-bit keys, then the others
    // (lib.rs:635-661, generic_modes.rs:99-100).  SKX_KNOBS=serial_loads: one after the other.
    const char *kn = getenv("SKX_KNOBS");
    const bool serial = kn && strstr(kn, "serial_loads");
    const int k0 = skx_skf_peek_k(skf_files[0]);
    int bits = k0 > 31 ? 128 : 64;
    auto load_team = [&](int first, std::vector<int> &rc) {
        std::atomic<int> next{first};
        auto work = [&]() { for (int i; (i = next.fetch_add(1)) < n_files;) rc[i] = skx_array_load(ctx, skf_files[i], bits, &arrs[i]); };
        const int team = serial ? 1 : std::min(4, n_files - first);
        std::vector<std::thread> th;
        for (int t = 1; t < team; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    };
    std::vector<int> rc(n_files, SKX_OK);
    bool have_first = false;
    if (k0 > 0) {
        load_team(0, rc);
        have_first = rc[0] == SKX_OK;
        if (!have_first) { cleanup(); for (auto &p : arrs) p = nullptr; }
    }
    skx_phase_add("merge.first_file_wall", since(t_m0));
    const auto t_m1 = std::chrono::steady_clock::now();
    if (!have_first) {
        bits = 64;
        if (skx_array_load(ctx, skf_files[0], 64, &arrs[0]) != SKX_OK) {
            bits = 128;
            if (skx_array_load(ctx, skf_files[0], 128, &arrs[0]) != SKX_OK) { skx_set_error("Could not read input file: %s", skf_files[0]); return SKX_EIO; }
        }
        std::fill(rc.begin(), rc.end(), SKX_OK);
        load_team(1, rc);
    }
    for (int i = 1; i < n_files; i++)
        if (rc[i] != SKX_OK) {                                                                                // generic_modes.rs:99-100
            cleanup(); skx_set_error("Failed to load input file (inconsistent k-mer lengths?): %s", skf_files[i]); return SKX_EINVAL;
        }
    skx_phase_add("merge.other_files_wall", since(t_m1));
    const auto t_m2 = std::chrono::steady_clock::now();
    skx_array *m = nullptr;
    int r = skx_array_merge(ctx, arrs.data(), n_files, &m);
    skx_phase_add("merge.union_scatter_wall", since(t_m2));
    const auto t_m3 = std::chrono::steady_clock::now();
    cleanup();
    skx_phase_add("merge.release_inputs_wall", since(t_m3));
    if (r != SKX_OK) return r;
    r = skh_save_skf(m, out_prefix);
    skx_array_free(m);
    return r;
    });
}

extern "C" int skh_delete(skx_array *a, const char *const *names, int n_names, const char *out_file)
{
    return skx_guarded([&]() -> int {
    int r = skx_array_delete_samples(a, names, n_names);
    if (r != SKX_OK) return r;
    return skh_save_skf(a, out_file);                                                                             // generic_modes.rs:200-209 (same suffix rule)
    });
}

extern "C" int skh_weed(skx_array *a, const char *weed_file, int reverse, double min_freq, int filter_ambig_as_missing, int filter_type,
                        int ambig_mask, int ignore_const_gaps, const char *out_file)
{
    return skx_guarded([&]() -> int {
    skx_array_info_t info; skx_array_info(a, &info);
    if (weed_file) {
        // RefSka::new(k, file, rc, ..) (generic_modes.rs:222-235): FASTA only; its split k-mers = the keys of the file's dictionary
        skx_keyset *ks = nullptr;
        int r = skx_keyset_from_fasta(skx_array_ctx(a), weed_file, info.k, info.rc, &ks);
        if (r != SKX_OK) return r;
        uint64_t removed = 0;
        r = skx_array_weed(a, ks, reverse, &removed);
        skx_keyset_free(ks);
        if (r != SKX_OK) return r;
    }
    const uint64_t threshold = (uint64_t)std::floor((double)info.n_samples * min_freq);                           // generic_modes.rs:249
    if (threshold > 0 || filter_type != SKX_FILTER_NONE || ambig_mask || ignore_const_gaps) {
        int32_t removed = 0;
        int r = skx_array_filter(a, threshold, filter_ambig_as_missing, filter_type, ambig_mask, ignore_const_gaps, /*update_kmers=*/1, &removed);
        if (r != SKX_OK) return r;
    }
    return out_file ? skx_array_save(a, out_file) : SKX_OK;                                                       // :263-266 (no suffix rule here)
    });
}

// ------------------------------------------------------------------------------------------ ska cov (coverage.rs)
namespace {
double cov_lse(double a, double b) { const double x = a > b ? a : b; return x + std::log(std::exp(a - x) + std::exp(b - x)); }           // :289-292
double cov_ln_dpois(double x, double lambda) { return x * std::log(lambda) - std::lgamma(x + 1.0) - lambda; }                                // :295-297
double cov_a(double w0, double i) { return std::log(w0) + cov_ln_dpois(i, 1.0); }                                                            // :300-302
double cov_b(double w0, double c, double i) { return std::log(1.0 - w0) + cov_ln_dpois(i, c); }                                              // :305-307
double cov_ll(const double p[2], const double *counts, uint64_t n)                                                                            // :310-326
{
    if (!(p[0] >= 0.0 && p[0] <= 1.0) || p[1] < 1.0) return -1.7976931348623157e308;
    double ll = 0.0;
    for (uint64_t i = 0; i < n; i++) { const double x = (double)i + 1.0; ll += counts[i] * cov_lse(cov_a(p[0], x), cov_b(p[0], p[1], x)); }
    return ll;
}
void cov_grad(const double p[2], const double *counts, uint64_t n, double g[2])                                                              // :329-346
{
    double g0 = 0.0, g1 = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        const double x = (double)i + 1.0, a = cov_a(p[0], x), b = cov_b(p[0], p[1], x);
        const double dlda = 1.0 / (1.0 + std::exp(b - a)), dldb = 1.0 / (1.0 + std::exp(a - b));
        g0 += counts[i] * (dlda / p[0] - dldb / (1.0 - p[0]));
        g1 += counts[i] * (dldb * (x / p[1] - 1.0));
    }
    g[0] = g0; g[1] = g1;
}
// Rust `{:e}`: shortest digits that round-trip, exponent without padding
std::string lower_exp(double v)
{
    char tmp[64]; int prec = 0;
    for (; prec < 17; prec++) { snprintf(tmp, sizeof tmp, "%.*e", prec, v); if (strtod(tmp, nullptr) == v) break; }
    snprintf(tmp, sizeof tmp, "%.*e", prec, v);
    char *e = strchr(tmp, 'e');
    const int ex = atoi(e + 1);
    *e = 0;
    std::string m(tmp);
    if (m.find('.') != std::string::npos) { while (!m.empty() && m.back() == '0') m.pop_back(); if (!m.empty() && m.back() == '.') m.pop_back(); }
    return m + "e" + std::to_string(ex);
}
}  // namespace

extern "C" int skh_cov_fit(const double *counts, uint64_t n, double *w0_out, double *c_out, uint64_t *cutoff)
{
    return skx_guarded([&]() -> int {
    // argmin BFGS (inverse-Hessian form) + BacktrackingLineSearch(ArmijoCondition(1e-4), contraction 0.9) from (0.8, 20), H0 = I,
    // tolerance_cost 1e-6, max_iters 20 (coverage.rs:176-196)
    double x[2] = {0.8, 20.0}, H[2][2] = {{1.0, 0.0}, {0.0, 1.0}}, g[2];
    double f = -cov_ll(x, counts, n);
    cov_grad(x, counts, n, g); g[0] = -g[0]; g[1] = -g[1];
    bool converged = false;
    for (int it = 0; it < 20 && !converged; it++) {
        const double p[2] = {-(H[0][0] * g[0] + H[0][1] * g[1]), -(H[1][0] * g[0] + H[1][1] * g[1])};
        const double gp = g[0] * p[0] + g[1] * p[1];
        double alpha = 1.0, xn[2], fn;
        for (;;) {
            xn[0] = x[0] + alpha * p[0]; xn[1] = x[1] + alpha * p[1];
            fn = -cov_ll(xn, counts, n);
            if (fn <= f + 1e-4 * alpha * gp) break;
            alpha *= 0.9;
            if (alpha == 0.0) break;
        }
        double gn[2];
        cov_grad(xn, counts, n, gn); gn[0] = -gn[0]; gn[1] = -gn[1];
        const double y[2] = {gn[0] - g[0], gn[1] - g[1]}, s[2] = {xn[0] - x[0], xn[1] - x[1]};
        const double ys = y[0] * s[0] + y[1] * s[1], prev = f;
        x[0] = xn[0]; x[1] = xn[1]; f = fn; g[0] = gn[0]; g[1] = gn[1];
        if (std::sqrt(g[0] * g[0] + g[1] * g[1]) < 1.4901161193847656e-8 || std::fabs(prev - f) < 1e-6) { converged = true; break; }
        const double rho = 1.0 / ys;
        double t1[2][2], t2[2][2], m[2][2], r[2][2];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { t1[a][b] = (a == b) - rho * s[a] * y[b]; t2[a][b] = (a == b) - rho * y[a] * s[b]; }
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) m[a][b] = t1[a][0] * H[0][b] + t1[a][1] * H[1][b];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) r[a][b] = m[a][0] * t2[0][b] + m[a][1] * t2[1][b] + rho * s[a] * s[b];
        memcpy(H, r, sizeof r);
    }
    if (!converged) { skx_set_error("Couldn't fit coverage model: Optimiser did not converge: Maximum number of iterations reached"); return SKX_EINVAL; }   // :216-220
    uint64_t cut = 1;                                                                                            // find_cutoff, :349-363
    while (cut < n) { if (cov_a(x[0], (double)cut) - cov_b(x[0], x[1], (double)cut) < 0.0) break; cut++; }
    *w0_out = x[0]; *c_out = x[1]; *cutoff = cut;
    return SKX_OK;
    });
}

extern "C" int skh_cov(skx_ctx *ctx, const char *fastq_fwd, const char *fastq_rev, int k, int rc, char **text, uint64_t *len, uint64_t *cutoff)
{
    return skx_guarded([&]() -> int {
    std::vector<uint32_t> hist(1000);
    int r = skx_cov_histogram(ctx, fastq_fwd, fastq_rev, k, rc, hist.data());
    if (r != SKX_OK) return r;
    uint64_t n = 1000;
    while (n && hist[n - 1] < 50u) n--;                                                                           // MIN_FREQ, coverage.rs:166-173
    std::vector<double> cf(hist.begin(), hist.begin() + (ptrdiff_t)n);
    double w0 = 0, c = 0; uint64_t cut = 0;
    if ((r = skh_cov_fit(cf.data(), n, &w0, &c, &cut)) != SKX_OK) return r;
    if (cutoff) *cutoff = cut;
    if (text) {                                                                                                   // plot_hist, :227-250
        std::string o = "Count\tK_mers\tMixture_density\tComponent\n";
        for (uint64_t i = 0; i < n; i++) {
            put(o, "%llu\t%u\t", (unsigned long long)(i + 1), hist[i]);
            o += lower_exp(std::exp(cov_lse(cov_a(w0, (double)i + 1.0), cov_b(w0, c, (double)i + 1.0))));
            o += (i + 1) < cut ? "\tError\n" : "\tCoverage\n";
        }
        return to_buf(o, text, len);
    }
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ several GPUs, one process each
namespace {
// the common head of the three sharded modes: this rank's dictionaries -> key-table all-gather -> the rank's columns over the global rows
int sharded_array(skx_ctx *ctx, skx_comm *comm, const skh_job *job, uint64_t *lo_out, uint64_t *hi_out, skx_array **out)
{
    if (!ctx || !comm || !job || job->n_samples <= 0) { skx_set_error("bad arguments"); return SKX_EINVAL; }
    const int rank = skx_comm_rank(comm), world = skx_comm_world(comm);
    uint64_t lo = 0, hi = 0;
    // (a rank that cannot start -- no share of the samples -- goes through the status agreement below like one whose build failed: its
    // peers must not be left waiting in the all-reduce)
    int r = skx_shard_range((uint64_t)job->n_samples, rank, world, &lo, &hi);
    if (r == SKX_OK && hi <= lo) { skx_set_error("rank %d: no samples (fewer samples than ranks)", rank); r = SKX_EINVAL; }
    skx_dictset *ds = nullptr; skx_keyset *ks = nullptr, *rows = nullptr;
    if (r == SKX_OK) {
        Phase p("sharded.dictionaries");
        std::vector<const char *> f2(hi - lo, nullptr);
        if (job->file2) for (uint64_t i = lo; i < hi; i++) f2[i - lo] = job->file2[i];
        r = skx_dictset_build_files(ctx, job->file1 + lo, f2.data(), (int)(hi - lo), job->k, job->rc, &job->qual, job->threads, job->proportion_reads, &ds);
    }
    // a rank whose build failed must not leave the others waiting in the key-table exchange: the ranks agree on a status first, and
    // everybody returns the failure (the failing rank reports its own message, the others name the rank)
    {
        std::vector<uint32_t> st(world, 0u);
        st[rank] = r == SKX_OK ? 0u : 1u;
        const std::string mine = r == SKX_OK ? "" : skx_last_error();
        const int ra = skx_comm_allreduce_u32(comm, st.data(), (uint64_t)world, 0);
        if (ra != SKX_OK) { if (ds) skx_dictset_free(ds); return ra; }
        if (r != SKX_OK) { skx_set_last_error(mine.c_str()); return r; }
        for (int q = 0; q < world; q++)
            if (st[q]) { skx_dictset_free(ds); skx_set_last_error(("rank " + std::to_string(q) + " could not build its samples").c_str()); return SKX_EINVAL; }
    }
    { Phase p("sharded.local_union"); r = skx_keyset_union_notes(ctx, ds, &ks); }      // (assemblies: the append pass -- its pieces travel with the key set)
    if (r == SKX_OK) { Phase p("sharded.key_table_exchange"); r = skx_keyset_allgather(comm, ks, &rows); }
    if (ks) skx_keyset_free(ks);
    if (r != SKX_OK) { skx_dictset_free(ds); return r; }
    // rows + dictionaries: the rank's column slab over the global rows is never allocated (8 000 samples: 100+ GB per rank)
    r = skx_array_assemble_lazy(ctx, ds, rows, job->names + lo, out);          // takes ds and rows, also on failure
    *lo_out = lo; *hi_out = hi;
    return r;
}
std::string part_name(const char *prefix, int rank, int world) { return std::string(prefix) + ".part" + std::to_string(rank) + "of" + std::to_string(world) + ".skf"; }
}  // namespace

extern "C" int skh_build_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job)
{
    return skx_guarded([&]() -> int {
    skx_array *arr = nullptr; uint64_t lo, hi;
    int r = sharded_array(ctx, comm, job, &lo, &hi, &arr);
    if (r != SKX_OK) return r;
    if (!job->output) { skx_array_free(arr); skx_set_error("-o <output> is required"); return SKX_EINVAL; }
    const int rank = skx_comm_rank(comm), world = skx_comm_world(comm);
    r = skx_array_save(arr, part_name(job->output, rank, world).c_str());      // local counts: each part is a self-consistent MergeSkaArray
    skx_array_free(arr);
    const int rb = skx_comm_barrier(comm);
    if (r != SKX_OK) return r;
    if (rb != SKX_OK) return rb;
    if (rank == 0) {
        std::vector<std::string> parts; std::vector<const char *> cp;
        for (int p = 0; p < world; p++) parts.push_back(part_name(job->output, p, world));
        for (auto &x : parts) cp.push_back(x.c_str());
        if (job->merge_parts) {
            if (world == 1) {                                                    // one part is the array itself
                std::string o(job->output); if (o.size() < 4 || o.compare(o.size() - 4, 4, ".skf") != 0) o += ".skf";
                if (rename(parts[0].c_str(), o.c_str()) != 0) { skx_set_error("cannot rename %s", parts[0].c_str()); r = SKX_EIO; }
            } else if ((r = skh_merge(ctx, cp.data(), world, job->output)) == SKX_OK)
                for (auto &x : parts) unlink(x.c_str());
        } else {
            std::string msg = "wrote";
            for (auto &x : parts) { msg += ' '; msg += x; }
            fprintf(stderr, "%s\njoin them with: ska merge -o %s <parts>\n", msg.c_str(), job->output);
        }
    }
    return r;
    });
}

extern "C" int skh_align_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job)
{
    return skx_guarded([&]() -> int {
    skx_array *arr = nullptr; uint64_t lo, hi;
    int r = sharded_array(ctx, comm, job, &lo, &hi, &arr);
    if (r != SKX_OK) return r;
    auto done = [&](int rc) { skx_array_free(arr); return rc; };
    { Phase p("sharded.row_stats_exchange"); if ((r = skx_array_reduce_stats(comm, arr, (uint64_t)job->n_samples)) != SKX_OK) return done(r); }
    int32_t removed = 0;
    { Phase p("align.filter"); if ((r = skh_apply_filters(arr, job->min_freq, job->filter_ambig_as_missing, job->filter_type, job->mask_ambig, job->ignore_const_gaps, &removed)) != SKX_OK) return done(r); }
    skx_array_info_t info; skx_array_info(arr, &info);
    const uint64_t kept = info.n_rows;
    // every rank writes its samples' records at their place in the one file: ">name\n" + kept bases + "\n" per sample
    uint64_t offset = 0, total = 0;
    for (int i = 0; i < job->n_samples; i++) { const uint64_t sz = strlen(job->names[i]) + kept + 3; if ((uint64_t)i < lo) offset += sz; total += sz; }
    const int rank = skx_comm_rank(comm);
    if (!job->output) { skx_set_error("with several GPUs the alignment goes to a file: give -o"); return done(SKX_EINVAL); }
    if (rank == 0) {
        int fd0 = open(job->output, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd0 < 0 || ftruncate(fd0, (off_t)total) != 0) { if (fd0 >= 0) close(fd0); skx_set_error("cannot create output file"); r = SKX_EIO; }
        else close(fd0);
    }
    const int rb = skx_comm_barrier(comm);                                       // the file exists at its full size
    if (r != SKX_OK) return done(r);
    if (rb != SKX_OK) return done(rb);
    int fd = open(job->output, O_RDWR);
    if (fd < 0) { skx_set_error("cannot open output file"); return done(SKX_EIO); }
    if (lseek(fd, (off_t)offset, SEEK_SET) < 0) { close(fd); skx_set_error("cannot seek in output file"); return done(SKX_EIO); }
    { Phase p("align.write_fasta"); r = skx_array_write_fasta(arr, fd); }
    close(fd);
    const int rb2 = skx_comm_barrier(comm);
    return done(r != SKX_OK ? r : rb2);
    });
}

extern "C" int skh_distance_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job)
{
    return skx_guarded([&]() -> int {
    skx_array *arr = nullptr; uint64_t lo, hi;
    int r = sharded_array(ctx, comm, job, &lo, &hi, &arr);
    if (r != SKX_OK) return r;
    auto done = [&](int rc) { skx_array_free(arr); return rc; };
    const uint64_t S = (uint64_t)job->n_samples;
    { Phase p("sharded.row_stats_exchange"); if ((r = skx_array_reduce_stats(comm, arr, S)) != SKX_OK) return done(r); }
    int32_t removed = 0, constant = 0;
    if (job->min_freq * (double)S >= 1.0)                                          // generic_modes.rs:149-159
        if ((r = skx_array_filter(arr, (uint64_t)std::ceil((double)S * job->min_freq), 0, SKX_FILTER_NONE, 0, 0, 0, &removed)) != SKX_OK) return done(r);
    if ((r = skx_array_filter(arr, 0, 0, SKX_FILTER_NO_CONST, 0, 0, 0, &constant)) != SKX_OK) return done(r);      // :161-168
    const int rank = skx_comm_rank(comm);
    std::vector<skx_dist> d(rank == 0 ? S * (S - 1) / 2 + 1 : 1);
    if ((r = skx_array_distance_sharded(comm, arr, job->filt_ambig, (double)constant, d.data(), d.size())) != SKX_OK) return done(r);
    if (rank == 0) {
        std::vector<const char *> names(job->names, job->names + S);
        char *buf = nullptr; uint64_t len = 0;
        if ((r = distance_text_names(names, d.data(), &buf, &len)) != SKX_OK) return done(r);
        FILE *f = job->output ? fopen(job->output, "wb") : stdout;
        if (!f) { skx_free(buf); skx_set_error("cannot create output file"); return done(SKX_EIO); }
        fwrite(buf, 1, len, f);
        if (f != stdout) fclose(f); else fflush(stdout);
        skx_free(buf);
    }
    return done(SKX_OK);
    });
}

// ------------------------------------------------------------------------------------------ CLI
extern char **environ;
namespace {
struct Args {
    std::vector<std::string> pos;
    std::vector<std::pair<std::string, std::string>> opt;
    bool has(const std::string &k) const { for (auto &o : opt) if (o.first == k) return true; return false; }
    std::string get(const std::string &k, const std::string &d = "") const { for (auto &o : opt) if (o.first == k) return o.second; return d; }
};
const char *VALUE_OPTS[] = {"-o", "-k", "-f", "--threads", "--min-count", "--min-qual", "--qual-filter", "--proportion-reads",
                            "--min-freq", "-m", "--filter", "-s", "--skf-file", "--format", "--gpus", nullptr};
bool takes_value(const std::string &s) { for (int i = 0; VALUE_OPTS[i]; i++) if (s == VALUE_OPTS[i]) return true; return false; }
int fail(const char *msg) { fprintf(stderr, "error: %s\n", msg); return 2; }
}
const char *skh_usage_line(const char *cmd);       // ska_help.cpp
// simple_logger's line (lib.rs:559-563: Info with -v, Warn without): "<UTC time> <LEVEL> [<target>] <message>" on stderr
static bool g_verbose = false;
extern "C" void skh_log(int level, const char *target, const char *message)         // level: 1 = WARN, 2 = INFO
{
    if (level > (g_verbose ? 2 : 1)) return;
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    struct tm tmv; gmtime_r(&ts.tv_sec, &tmv);
    char tb[40]; strftime(tb, sizeof tb, "%Y-%m-%dT%H:%M:%S", &tmv);
    fprintf(stderr, "%s.%03ldZ %-5s [%s] %s\n", tb, ts.tv_nsec / 1000000, level == 1 ? "WARN" : "INFO", target, message);
}
namespace {
// clap's wording for what its derive macros refuse (cli.rs: value_parser / value_enum / required): exit code 2, the reason, the hint
int clap_invalid(const std::string &value, const char *arg, const char *why)
{
    fprintf(stderr, "error: invalid value '%s' for '%s': %s\n\nFor more information, try '--help'.\n", value.c_str(), arg, why);
    return 2;
}
int clap_possible(const std::string &value, const char *arg, const char *values)
{
    fprintf(stderr, "error: invalid value '%s' for '%s'\n  [possible values: %s]\n\nFor more information, try '--help'.\n", value.c_str(), arg, values);
    return 2;
}
int clap_missing(const char *cmd, const char *args)
{
    fprintf(stderr, "error: the following required arguments were not provided:\n  %s\n\nUsage: %s\n\nFor more information, try '--help'.\n", args, skh_usage_line(cmd));
    return 2;
}
int engine_fail() { fprintf(stderr, "error: %s\n", skx_last_error()); return 101; }   // Rust panics exit with 101
int parse_filter(const std::string &s)
{
    if (s == "no-filter") return SKX_FILTER_NONE;
    if (s == "no-const") return SKX_FILTER_NO_CONST;
    if (s == "no-ambig") return SKX_FILTER_NO_AMBIG;
    if (s == "no-ambig-or-const") return SKX_FILTER_NO_AMBIG_OR_CONST;
    return -1;
}
// `--gpus N`: this executable once per GPU, the ranks find each other through SKX_RANK / SKX_WORLD and a rendezvous directory
// (rank 0 leaves the RCCL id there; SKX_COMM=local makes the ranks exchange through the directory itself, for ranks that share a device)
int launch_ranks(int world, char **argv)
{
    char dir[64];
    struct stat sb;
    snprintf(dir, sizeof dir, "%s/skx_ranks_XXXXXX", stat("/dev/shm", &sb) == 0 ? "/dev/shm" : "/tmp");
    if (!mkdtemp(dir)) return fail("cannot create the rendezvous directory");
    const bool local = getenv("SKX_COMM") && !strcmp(getenv("SKX_COMM"), "local");
    std::vector<pid_t> pids;
    int rcode = 0;
    for (int r = 0; r < world && !rcode; r++) {
        std::vector<std::string> env;
        for (char **e = environ; *e; e++) if (strncmp(*e, "SKX_RANK=", 9) && strncmp(*e, "SKX_WORLD=", 10) && strncmp(*e, "SKX_COMM_", 9)) env.push_back(*e);
        env.push_back("SKX_RANK=" + std::to_string(r)); env.push_back("SKX_WORLD=" + std::to_string(world));
        env.push_back(std::string(local ? "SKX_COMM_DIR=" : "SKX_COMM_ID_FILE=") + dir + (local ? "" : "/id"));
        std::vector<char *> envp; for (auto &x : env) envp.push_back(&x[0]); envp.push_back(nullptr);
        pid_t pid;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, argv, envp.data()) != 0) {
            rcode = fail("cannot start a rank");
            for (pid_t q : pids) kill(q, SIGTERM);                                          // (they would wait for the missing peer)
        } else pids.push_back(pid);
    }
    size_t left = pids.size();
    while (left) {
        int st = 0; const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) { if (errno == EINTR) continue; break; }
        size_t i = 0; while (i < pids.size() && pids[i] != p) i++;
        if (i == pids.size()) continue;
        pids[i] = 0; left--;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        if (code && !rcode) { rcode = code; for (pid_t q : pids) if (q) kill(q, SIGTERM); }      // the others wait in a collective for a rank that is gone
    }
    (void)!system((std::string("rm -rf '") + dir + "'").c_str());
    return rcode;
}
// the communicator of this rank (RCCL unless SKX_COMM_DIR names a directory for the host-staged transport)
int open_comm(skx_ctx *ctx, int rank, int world, skx_comm **out)
{
    if (const char *d = getenv("SKX_COMM_DIR")) return skx_comm_create_local(ctx, rank, world, d, out);
    uint8_t id[SKX_COMM_ID_BYTES];
    const char *idf = getenv("SKX_COMM_ID_FILE");
    if (world == 1 && !idf) return skx_comm_create_local(ctx, 0, 1, nullptr, out);      // a single rank exchanges with nobody: no RCCL needed (the library's own directory, removed with the communicator)
    if (world > 1 && !idf) { skx_set_error("SKX_WORLD > 1 needs SKX_COMM_ID_FILE (where rank 0 leaves the RCCL id) or SKX_COMM_DIR"); return SKX_EINVAL; }
    if (rank == 0) {
        int r = skx_comm_unique_id(id);
        if (r != SKX_OK) return r;
        if (idf) {
            const std::string tmp = std::string(idf) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); skx_set_error("cannot write %s", tmp.c_str()); return SKX_EIO; }
            fclose(f);
            if (rename(tmp.c_str(), idf) != 0) { skx_set_error("cannot publish %s", idf); return SKX_EIO; }
        }
    } else {
        FILE *f = nullptr;
        for (int i = 0; i < 6000 && !(f = fopen(idf, "rb")); i++) usleep(20000);               // two minutes
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); skx_set_error("rank %d: no RCCL id at %s", rank, idf); return SKX_EIO; }
        fclose(f);
    }
    return skx_comm_create(ctx, rank, world, id, out);
}
struct BuildOpts { int k = 31; skx_qual q{5, 20, SKX_QUAL_STRICT}; bool auto_count = false; double prop = 0.0; };
int parse_build_opts(const Args &a, BuildOpts &o)                                      // cli.rs:27-108 (Build)
{
    o.k = atoi(a.get("-k", "31").c_str());
    if (o.k < 5 || o.k > 63 || o.k % 2 == 0) return clap_invalid(a.get("-k", "31"), "-k <K>", "K-mer must be an odd number between 5 and 63 (inclusive)");   // cli.rs:38-47
    if (a.has("--min-count")) {
        const std::string mc = a.get("--min-count");
        if (mc == "auto") o.auto_count = true; else {
        char *end; long v = strtol(mc.c_str(), &end, 10);
        if (*end || v < 1 || v > 65535) return clap_invalid(mc, "--min-count <MIN_COUNT>", "Minimum kmer count must be >= 1");   // cli.rs:94-108
        o.q.min_count = (uint16_t)v; }
    }
    if (a.has("--min-qual")) o.q.min_qual = (uint8_t)atoi(a.get("--min-qual").c_str());
    if (a.has("--qual-filter")) {
        const std::string f = a.get("--qual-filter");
        o.q.qual_filter = f == "no-filter" ? SKX_QUAL_NOFILTER : f == "middle" ? SKX_QUAL_MIDDLE : f == "strict" ? SKX_QUAL_STRICT : -1;
        if (o.q.qual_filter < 0) return clap_possible(f, "--qual-filter <QUAL_FILTER>", "no-filter, middle, strict");
    }
    o.prop = a.has("--proportion-reads") ? atof(a.get("--proportion-reads").c_str()) : 0.0;
    if (o.prop < 0.0 || o.prop > 1.0) return clap_invalid(a.get("--proportion-reads"), "--proportion-reads <PROPORTION_READS>", "K-mer must be between 0 and 1 (inclusive)");   // cli.rs:49-58 (the reference's own wording)
    return 0;
}
struct Inputs { std::vector<std::string> names, f1, f2; std::vector<const char *> cn, c1, c2; };
int read_inputs(const Args &a, Inputs &in)                                             // io_utils.rs:31-46,116-146
{
    if (a.has("-f")) {
        std::ifstream f(a.get("-f"));
        if (!f) return fail("Unable to open file_list");
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ls(line); std::vector<std::string> fld; std::string t;
            while (ls >> t) fld.push_back(t);
            if (fld.size() < 2 || fld.size() > 3) return fail("Unable to parse line in file_list");
            in.names.push_back(fld[0]); in.f1.push_back(fld[1]); in.f2.push_back(fld.size() == 3 ? fld[2] : "");
        }
    } else
        for (auto &p : a.pos) { char *n = skh_sample_name(p.c_str()); in.names.push_back(n); free(n); in.f1.push_back(p); in.f2.push_back(""); }
    for (size_t i = 0; i < in.names.size(); i++) { in.cn.push_back(in.names[i].c_str()); in.c1.push_back(in.f1[i].c_str()); in.c2.push_back(in.f2[i].empty() ? nullptr : in.f2[i].c_str()); }
    return 0;
}
// --min-count auto (io_utils::kmer_min_cutoff, io_utils.rs:175-212): the FIRST file of the first two paired inputs
int auto_min_count(skx_ctx *ctx, const Inputs &in, int k, bool rc, skx_qual &q, bool print)
{
    std::vector<std::string> fq;
    for (size_t i = 0; i < in.names.size() && fq.size() < 2; i++) if (!in.f2[i].empty()) fq.push_back(in.f1[i]);
    if (fq.size() >= 2) {
        char *buf = nullptr; uint64_t len = 0, cutoff = 0;
        if (skh_cov(ctx, fq[0].c_str(), fq[1].c_str(), k, rc, &buf, &len, &cutoff) != SKX_OK) return 101;
        if (print) { fwrite(buf, 1, len, stdout); fflush(stdout); }                                               // cov.plot_hist()
        skx_free(buf);
        q.min_count = (uint16_t)cutoff;
        if (print) fprintf(stderr, "Using inferred minimum kmer value of %llu\n", (unsigned long long)cutoff);
    } else if (print) fprintf(stderr, "Not enough fastq files to fit mixture model, using default kmer count of 5\n");
    return 0;
}
// build | align | distance as one rank of a job over several GPUs
// `ska selftest --gpus N`: what a sharded job needs of the node, in a few seconds and with the engine's own message when something is missing --
// the hand-off of the RCCL id, one all-gather of key tables of unequal sizes (skx_keyset_allgather), one all-reduce, one gather of unequal
// pieces to rank 0, each compared with the same exchange over the host-staged transport in the same group of ranks.  60 s, then it gives up.
static const char *g_selftest_stage = "start";
static void selftest_alarm(int) { char b[160]; const int n = snprintf(b, sizeof b, "error: selftest: no progress for 60 s in: %s\n", g_selftest_stage); if (n > 0) (void)!write(2, b, (size_t)n); _exit(3); }
int selftest(skx_ctx *ctx, int rank, int world)
{
    signal(SIGALRM, selftest_alarm);
    alarm(60);
    auto stage = [&](const char *s) { g_selftest_stage = s; if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] selftest rank %d: %s\n", rank, s); };
    stage("communicator (RCCL id hand-off)");
    skx_comm *comm = nullptr, *local = nullptr;
    if (open_comm(ctx, rank, world, &comm) != SKX_OK) return engine_fail();
    // the same ranks over the host-staged transport, in a directory next to the id file
    std::string ldir;
    if (const char *idf = getenv("SKX_COMM_ID_FILE")) { ldir = idf; ldir = ldir.substr(0, ldir.find_last_of('/')) + "/selftest_local"; if (rank == 0) mkdir(ldir.c_str(), 0700); }
    stage("barrier");
    if (skx_comm_barrier(comm) != SKX_OK) return engine_fail();
    if (!ldir.empty() && world > 1) { stage("host-staged communicator"); if (skx_comm_create_local(ctx, rank, world, ldir.c_str(), &local) != SKX_OK) return engine_fail(); }
    int bad = 0;
    // all-reduce
    stage("all-reduce");
    std::vector<uint32_t> v(4096), v2;
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint32_t)(rank + 1) * (uint32_t)(i % 7 + 1);
    v2 = v;
    if (skx_comm_allreduce_u32(comm, v.data(), v.size(), 0) != SKX_OK) return engine_fail();
    for (size_t i = 0; i < v.size(); i++) if (v[i] != (uint32_t)(world * (world + 1) / 2) * (uint32_t)(i % 7 + 1)) bad |= 1;
    if (local) { if (skx_comm_allreduce_u32(local, v2.data(), v2.size(), 0) != SKX_OK) return engine_fail(); if (v2 != v) bad |= 2; }
    // gather of unequal pieces to rank 0
    stage("gather to rank 0");
    std::vector<uint64_t> sizes(world); uint64_t total = 0;
    for (int q = 0; q < world; q++) { sizes[q] = 1000 + 333 * (uint64_t)q; total += sizes[q]; }
    std::vector<uint8_t> piece(sizes[rank]), all(rank == 0 ? total : 1), all2(rank == 0 ? total : 1);
    for (size_t i = 0; i < piece.size(); i++) piece[i] = (uint8_t)(i * 31 + rank);
    if (skx_comm_gather_root(comm, piece.data(), sizes.data(), all.data()) != SKX_OK) return engine_fail();
    if (rank == 0) { uint64_t o = 0; for (int q = 0; q < world; q++) { for (uint64_t i = 0; i < sizes[q]; i++) if (all[o + i] != (uint8_t)(i * 31 + q)) bad |= 4; o += sizes[q]; } }
    if (local) { if (skx_comm_gather_root(local, piece.data(), sizes.data(), all2.data()) != SKX_OK) return engine_fail(); if (rank == 0 && all2 != all) bad |= 8; }
    // key tables of unequal sizes: every rank's own sequence plus one all share
    stage("key tables (all-gather of unequal tables + union)");
    {
        char tmpl[] = "/tmp/skx_selftest_XXXXXX";
        const int fd = mkstemp(tmpl);
        if (fd < 0) return fail("selftest: cannot write a temporary file");
        std::string fa = ">shared\n";
        uint64_t x = 88172645463325252ull;
        auto base = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return "ACGT"[x & 3]; };
        for (int i = 0; i < 20000; i++) fa += base();
        fa += "\n>own\n";
        x = 0x9E3779B97F4A7C15ull * (uint64_t)(rank + 1);
        for (int i = 0; i < 5000 * (rank + 1); i++) fa += base();
        fa += "\n";
        (void)!write(fd, fa.data(), fa.size()); close(fd);
        skx_keyset *ks = nullptr, *rows = nullptr, *ks2 = nullptr, *rows2 = nullptr;
        int r = skx_keyset_from_fasta(ctx, tmpl, 31, 1, &ks);
        if (r == SKX_OK && local) r = skx_keyset_from_fasta(ctx, tmpl, 31, 1, &ks2);
        unlink(tmpl);
        if (r != SKX_OK) return engine_fail();
        if (skx_keyset_allgather(comm, ks, &rows) != SKX_OK) return engine_fail();
        uint64_t n1 = 0, n2 = 0, mine_n = 0;
        skx_keyset_size(rows, &n1); skx_keyset_size(ks, &mine_n);
        if (n1 < mine_n || (world > 1 && n1 == mine_n)) bad |= 16;
        if (local) { if (skx_keyset_allgather(local, ks2, &rows2) != SKX_OK) return engine_fail(); skx_keyset_size(rows2, &n2); if (n2 != n1) bad |= 32; }
        std::vector<uint32_t> agree(1, (uint32_t)(n1 & 0xFFFFFFu));
        if (skx_comm_allreduce_u32(comm, agree.data(), 1, 0) != SKX_OK) return engine_fail();
        if (agree[0] != (uint32_t)(n1 & 0xFFFFFFu) * (uint32_t)world) bad |= 64;              // every rank derived the same number of rows
        if (rank == 0) fprintf(stderr, "selftest: %d rank(s): id hand-off, all-reduce, gather to rank 0, all-gather of unequal key tables (%llu rows)%s: %s\n", world,
                               (unsigned long long)n1, local ? ", each also over the host-staged transport" : "", bad ? "MISMATCH" : "ok");
        skx_keyset_free(ks); skx_keyset_free(rows); if (ks2) skx_keyset_free(ks2); if (rows2) skx_keyset_free(rows2);
    }
    // BASELINE config 4's exchange at its size: every rank's key table holds ~8 M keys (a 5 Mbp genome all ranks share + 3 to 3.7 Mbp of its
    // own, as 1 000 related assemblies with private SNPs leave it), unequal from rank to rank; the union of eight such tables is ~32 M rows.  The
    // number of rows is known in advance -- shared + the sum of the private parts -- and every rank must arrive at it.  (SKX_SELFTEST_SMALL: skipped.)
    if (!getenv("SKX_SELFTEST_SMALL")) {
        alarm(180);
        stage("config-4 shape: key tables of ~8 M keys per rank (all-gather + union)");
        auto write_fa = [&](std::string &path, uint64_t shared_bases, uint64_t own_bases) -> bool {
            char tmpl[] = "/tmp/skx_selftest_big_XXXXXX";
            const int fd = mkstemp(tmpl);
            if (fd < 0) return false;
            std::string fa; fa.reserve(shared_bases + own_bases + 64);
            uint64_t x = 0x2545F4914F6CDD1Dull;
            auto base = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return "ACGT"[(x >> 11) & 3]; };
            if (shared_bases) { fa += ">shared\n"; for (uint64_t i = 0; i < shared_bases; i++) fa += base(); fa += "\n"; }
            if (own_bases) { fa += ">own\n"; x = 0xD1B54A32D192ED03ull * (uint64_t)(rank + 7); for (uint64_t i = 0; i < own_bases; i++) fa += base(); fa += "\n"; }
            const bool ok = write(fd, fa.data(), fa.size()) == (ssize_t)fa.size();
            close(fd);
            path = tmpl;
            return ok;
        };
        const uint64_t shared_bases = 5000000, own_bases = 3000000 + 100000 * (uint64_t)rank;
        std::string p_all, p_shared;
        if (!write_fa(p_all, shared_bases, own_bases) || !write_fa(p_shared, shared_bases, 0)) return fail("selftest: cannot write a temporary file");
        skx_keyset *ks = nullptr, *kshared = nullptr, *rows = nullptr;
        int r = skx_keyset_from_fasta(ctx, p_all.c_str(), 31, 1, &ks);
        if (r == SKX_OK) r = skx_keyset_from_fasta(ctx, p_shared.c_str(), 31, 1, &kshared);
        unlink(p_all.c_str()); unlink(p_shared.c_str());
        if (r != SKX_OK) return engine_fail();
        uint64_t mine_n = 0, shared_n = 0, n1 = 0;
        skx_keyset_size(ks, &mine_n); skx_keyset_size(kshared, &shared_n);
        stage("config-4 shape: the all-gather");
        if (skx_keyset_allgather(comm, ks, &rows) != SKX_OK) return engine_fail();
        skx_keyset_size(rows, &n1);
        stage("config-4 shape: agreement on the row count");
        // (sums in 16-bit halves: the all-reduce adds 32-bit words)
        const uint64_t own_n = mine_n - shared_n;
        std::vector<uint32_t> parts = {(uint32_t)(own_n & 0xFFFF), (uint32_t)((own_n >> 16) & 0xFFFF), (uint32_t)(own_n >> 32), (uint32_t)(n1 & 0xFFFF), (uint32_t)((n1 >> 16) & 0xFFFF), (uint32_t)(n1 >> 32)};
        if (skx_comm_allreduce_u32(comm, parts.data(), parts.size(), 0) != SKX_OK) return engine_fail();
        const uint64_t own_sum = (uint64_t)parts[0] + ((uint64_t)parts[1] << 16) + ((uint64_t)parts[2] << 32), n_sum = (uint64_t)parts[3] + ((uint64_t)parts[4] << 16) + ((uint64_t)parts[5] << 32);
        if (mine_n <= shared_n || n1 != shared_n + own_sum) bad |= 128;                      // the union is what the tables add up to
        if (n_sum != n1 * (uint64_t)world) bad |= 256;                                       // and every rank holds the same one
        if (rank == 0) fprintf(stderr, "selftest: config-4 shape: %llu keys on rank 0 (%llu shared), union of %d tables %llu rows (expected %llu): %s\n", (unsigned long long)mine_n,
                               (unsigned long long)shared_n, world, (unsigned long long)n1, (unsigned long long)(shared_n + own_sum), (bad & 384) ? "MISMATCH" : "ok");
        skx_keyset_free(ks); skx_keyset_free(kshared); skx_keyset_free(rows);
    }
    stage("closing");
    if (local) skx_comm_destroy(local);
    skx_comm_destroy(comm);
    alarm(0);
    if (bad) { fprintf(stderr, "error: selftest: rank %d: results differ (mask %d)\n", rank, bad); return 4; }
    return 0;
}
int main_sharded(skx_ctx *ctx, const std::string &cmd, const Args &a, int rank, int world, int threads)
{
    BuildOpts bo; Inputs in;
    if (a.pos.empty() == !a.has("-f")) return fail("give either sequence files or -f <file_list>");
    if (int e = parse_build_opts(a, bo)) return e;
    if (int e = read_inputs(a, in)) return e;
    for (auto &p : in.f1) if (p.size() > 4 && p.compare(p.size() - 4, 4, ".skf") == 0) return fail("with --gpus give the sequence files (or -f <file_list>), not a .skf");
    if ((int)in.names.size() < world) return fail("fewer samples than GPUs");
    const bool rc = !a.has("--single-strand");
    if (bo.auto_count && auto_min_count(ctx, in, bo.k, rc, bo.q, rank == 0)) return engine_fail();    // every rank fits the same two files: same cutoff
    skh_job job{};
    job.names = in.cn.data(); job.file1 = in.c1.data(); job.file2 = in.c2.data(); job.n_samples = (int)in.cn.size();
    job.k = bo.k; job.rc = rc; job.qual = bo.q; job.threads = threads; job.proportion_reads = bo.prop;
    const std::string out = a.get("-o");
    job.output = a.has("-o") ? out.c_str() : nullptr;
    skx_comm *comm = nullptr;
    if (open_comm(ctx, rank, world, &comm) != SKX_OK) return engine_fail();
    int r;
    if (cmd == "build") {
        if (!a.has("-o")) return fail("-o <output> is required");
        job.merge_parts = a.has("--merge");
        r = skh_build_sharded(ctx, comm, &job);
    } else if (cmd == "align") {
        job.filter_type = parse_filter(a.get("--filter", "no-const"));
        if (job.filter_type < 0) return fail("invalid --filter");
        job.min_freq = atof(a.get("--min-freq", a.get("-m", "0.9")).c_str());
        if (job.min_freq < 0 || job.min_freq > 1) return fail("Frequency must be between 0 and 1 (inclusive)");
        job.mask_ambig = a.has("--ambig-mask"); job.ignore_const_gaps = a.has("--no-gap-only-sites"); job.filter_ambig_as_missing = a.has("--filter-ambig-as-missing");
        r = skh_align_sharded(ctx, comm, &job);
    } else {
        job.min_freq = atof(a.get("--min-freq", a.get("-m", "0")).c_str());
        job.filt_ambig = !a.has("--allow-ambiguous");
        r = skh_distance_sharded(ctx, comm, &job);
    }
    const int rcode = r == SKX_OK ? 0 : engine_fail();
    if (rank == 0 && getenv("SKX_DEBUG")) fprintf(stderr, "[skx] rank 0 received %llu bytes through the communicator\n", (unsigned long long)skx_comm_bytes_received(comm));
    skx_comm_destroy(comm);
    return rcode;
}
int emit(const std::string &out_path, const char *buf, uint64_t len)                 // io_utils::set_ostream
{
    if (out_path.empty()) { fwrite(buf, 1, len, stdout); fflush(stdout); return 0; }
    FILE *f = fopen(out_path.c_str(), "wb");
    if (!f) return fail("cannot create output file");
    fwrite(buf, 1, len, f); fclose(f);
    return 0;
}
}  // namespace

namespace {
// What clap refuses before lib.rs::main runs (cli.rs: required arguments, the argument groups of build / delete, value_parser and value_enum
// of every option): the same refusals in clap's wording, before the banner and before a device is touched.  0: the command line stands.
int validate_cli(const std::string &cmd, const Args &a, bool multi)
{
    auto number = [](const std::string &v) { if (v.empty()) return false; char *e; (void)strtod(v.c_str(), &e); return *e == 0; };
    auto kmer = [&](const char *arg) -> int {
        if (!a.has("-k")) return 0;
        const std::string v = a.get("-k");
        if (!number(v) || v.find_first_not_of("0123456789") != std::string::npos) return clap_invalid(v, arg, ("`" + v + "` isn't a valid k-mer").c_str());
        const int k = atoi(v.c_str());
        return (k < 5 || k > 63 || k % 2 == 0) ? clap_invalid(v, arg, "K-mer must be an odd number between 5 and 63 (inclusive)") : 0;
    };
    auto freq = [&]() -> int {
        const std::string v = a.has("--min-freq") ? a.get("--min-freq") : a.has("-m") ? a.get("-m") : "";
        if (v.empty()) return 0;
        if (!number(v)) return clap_invalid(v, "--min-freq <MIN_FREQ>", ("`" + v + "` isn't a valid frequency").c_str());
        const double f = atof(v.c_str());
        return (f < 0.0 || f > 1.0) ? clap_invalid(v, "--min-freq <MIN_FREQ>", "Frequency must be between 0 and 1 (inclusive)") : 0;
    };
    auto filter = [&]() -> int {
        return a.has("--filter") && parse_filter(a.get("--filter")) < 0 ? clap_possible(a.get("--filter"), "--filter <FILTER>", "no-filter, no-const, no-ambig, no-ambig-or-const") : 0;
    };
    if (a.has("--threads")) {
        const std::string v = a.get("--threads");
        if (v.find_first_not_of("0123456789") != std::string::npos || v.empty()) return clap_invalid(v, "--threads <THREADS>", ("`" + v + "` isn't a valid number of cores").c_str());
        if (atoi(v.c_str()) < 1) return clap_invalid(v, "--threads <THREADS>", "Threads must be one or higher");
    }
    const bool files_from_build = multi && (cmd == "align" || cmd == "distance");        // (--gpus N: sequence files or -f, and the build options)
    if (cmd == "build" || files_from_build) {
        if (cmd == "build" && !a.has("-o")) return clap_missing("build", "-o <OUTPUT>");
        if (a.pos.empty() && !a.has("-f")) return clap_missing(cmd.c_str(), "<SEQ_FILES|-f <FILE_LIST>>");
        if (!a.pos.empty() && a.has("-f")) { fprintf(stderr, "error: the argument '[SEQ_FILES]...' cannot be used with '-f <FILE_LIST>'\n\nUsage: %s\n\nFor more information, try '--help'.\n", skh_usage_line("build")); return 2; }
        BuildOpts bo;
        if (int e = kmer("-k <K>")) return e;
        if (int e = parse_build_opts(a, bo)) return e;
    }
    if (cmd == "align") { if (!multi && a.pos.empty()) return clap_missing("align", "<INPUT>..."); if (int e = filter()) return e; if (int e = freq()) return e; }
    else if (cmd == "distance") { if (!multi && a.pos.size() != 1) return a.pos.empty() ? clap_missing("distance", "<SKF_FILE>") : fail("one .skf file required"); if (int e = freq()) return e; }
    else if (cmd == "nk") { if (a.pos.empty()) return clap_missing("nk", "<SKF_FILE>"); }
    else if (cmd == "cov") { if (a.pos.size() < 2) return clap_missing("cov", a.pos.empty() ? "<FASTQ_FWD>\n  <FASTQ_REV>" : "<FASTQ_REV>"); if (int e = kmer("-k <K>")) return e; }
    else if (cmd == "map") {
        if (a.pos.empty()) return clap_missing("map", "<REFERENCE>");
        const std::string fmt = a.get("--format", a.has("-f") ? a.get("-f") : "aln");
        if (fmt != "aln" && fmt != "vcf") return clap_possible(fmt, "--format <FORMAT>", "vcf, aln");
    }
    else if (cmd == "merge") { if (!a.has("-o")) return clap_missing("merge", "-o <OUTPUT>"); }
    else if (cmd == "delete") {
        if (!a.has("-s") && !a.has("--skf-file")) return clap_missing("delete", "--skf-file <SKF_FILE>");
        if (a.pos.empty() && !a.has("-f")) return clap_missing("delete", "<-f <FILE_LIST>|NAMES>");
        if (!a.pos.empty() && a.has("-f")) { fprintf(stderr, "error: the argument '[NAMES]...' cannot be used with '-f <FILE_LIST>'\n\nUsage: %s\n\nFor more information, try '--help'.\n", skh_usage_line("delete")); return 2; }
    }
    else if (cmd == "weed") { if (a.pos.empty()) return clap_missing("weed", "<SKF_FILE>"); if (int e = filter()) return e; if (int e = freq()) return e; }
    else if (cmd == "lo") return fail("`ska lo` is not part of this engine (build, align, map, distance, nk, merge, delete, weed, cov are)");
    else if (cmd != "build" && cmd != "selftest") {
        fprintf(stderr, "error: unrecognized subcommand '%s'\n\nUsage: ska [OPTIONS] <COMMAND>\n\nFor more information, try '--help'.\n", cmd.c_str());
        return 2;
    }
    return 0;
}
}  // namespace

extern "C" int skh_main(int argc, char **argv)
{
    const int env_world = getenv("SKX_WORLD") ? atoi(getenv("SKX_WORLD")) : 0, env_rank = getenv("SKX_RANK") ? atoi(getenv("SKX_RANK")) : 0;
    // clap answers --help / --version (and refuses what it cannot parse) before lib.rs::main prints its banner (lib.rs:558-565)
    if (const int h = skh_help(argc, argv)) return h == 1 ? 0 : h;
    if (argc < 2) {                                                                        // clap: help on stderr, exit code 2
        fprintf(stderr, "Split k-mer analysis\n\nUsage: ska [OPTIONS] <COMMAND>\n\nFor more information, try '--help'.\n");
        return 2;
    }
    const std::string cmd = argv[1];
    Args a;
    for (int i = 2; i < argc; i++) {
        std::string s = argv[i];
        if (s.size() > 1 && s[0] == '-' && !(s.size() > 1 && isdigit((unsigned char)s[1]))) {
            if (takes_value(s)) { if (i + 1 >= argc) return fail("missing option value"); a.opt.emplace_back(s, argv[++i]); }
            else a.opt.emplace_back(s, "");
        } else a.pos.push_back(s);
    }
    // several GPUs (an extension; the reference has --threads only): `--gpus N` starts N ranks of this executable, a rank finds
    // SKX_WORLD / SKX_RANK in its environment; align / distance then take sequence files or -f and the build options
    const bool shards = cmd == "build" || cmd == "align" || cmd == "distance" || cmd == "selftest";
    const bool multi = a.has("--gpus") || (env_world > 0 && shards);                       // (a stray SKX_WORLD does not reach merge / nk / map / weed / cov)
    {   // clap rejects what a subcommand does not declare (cli.rs:109-330); -v / --verbose is global (cli.rs:103-104)
        static const struct { const char *cmd; const char *flags; } KNOWN[] = {
            {"build", " -o -k -f --proportion-reads --single-strand --min-count --min-qual --qual-filter --threads --gpus --merge "},
            {"align", " -o -m --min-freq --filter-ambig-as-missing --filter --ambig-mask --no-gap-only-sites --threads --gpus "},
            {"map", " -o -f --format --ambig-mask --repeat-mask --threads "},
            {"distance", " -o -m --min-freq --allow-ambiguous --threads --gpus "},
            {"merge", " -o "}, {"delete", " -s --skf-file -o -f "},
            {"weed", " -o --reverse -m --min-freq --filter-ambig-as-missing --filter --ambig-mask --no-gap-only-sites "},
            {"nk", " --full-info "}, {"cov", " -k --single-strand "}, {"selftest", " --gpus "}};
        for (auto &kc : KNOWN)
            if (cmd == kc.cmd)
                for (auto &o : a.opt)
                    if (o.first != "-v" && o.first != "--verbose" && !strstr(kc.flags, (" " + o.first + " ").c_str()) &&
                        !(multi && (cmd == "align" || cmd == "distance") && strstr(" -f -k --single-strand --min-count --min-qual --qual-filter --proportion-reads ", (" " + o.first + " ").c_str()))) {
                        fprintf(stderr, "error: unexpected argument '%s' found\n\nUsage: ska %s [OPTIONS]\n\nFor more information, try '--help'.\n", o.first.c_str(), cmd.c_str());
                        return 2;
                    }
    }
    if (const int bad = validate_cli(cmd, a, multi)) return bad;
    g_verbose = a.has("-v") || a.has("--verbose");
    if (env_rank == 0) fprintf(stderr, "SKA: Split K-mer Analysis (the alignment-free aligner)\n");      // lib.rs:565: after the arguments stand
    if (env_rank == 0) {
        char msg[256];
        // cli.rs:86-91 check_threads (build, align, map, distance: lib.rs:580,634,672,714)
        const int asked = atoi(a.get("--threads", "1").c_str()), cores = (int)std::thread::hardware_concurrency();
        if ((cmd == "build" || cmd == "align" || cmd == "map" || cmd == "distance") && cores > 0 && asked > cores) {
            snprintf(msg, sizeof msg, "%d threads is greater than available cores %d", asked, cores); skh_log(1, "ska::cli", msg);
        }
        // io_utils.rs:66-73 load_array: one input = a .skf, where --threads has nothing to do
        if (!multi && ((cmd == "align" && a.pos.size() == 1) || (cmd == "map" && a.pos.size() == 2)) && asked > 1) skh_log(1, "ska::io_utils", "--threads only used if building skf, setting to 1");
        // merge_ska_array.rs:298-300 filter
        if ((cmd == "align" || cmd == "weed") && a.has("--no-gap-only-sites")) {
            const std::string f = a.get("--filter", cmd == "align" ? "no-const" : "no-filter");
            if (f == "no-ambig" || f == "no-filter") skh_log(1, "ska::merge_ska_array", "--no-gap-only-sites can only be applied when filtering constant bases");
        }
        if (cmd == "build") { const int k = atoi(a.get("-k", "31").c_str()); snprintf(msg, sizeof msg, "k=%d: using %d-bit representation", k, k <= 31 ? 64 : 128); skh_log(2, "ska", msg); }   // lib.rs:592,607
    }
    int threads = atoi(a.get("--threads", "1").c_str());
    if (threads < 1) return fail("Threads must be one or higher");
    const auto t_main = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(); };
    const bool dbg = getenv("SKX_DEBUG") != nullptr;
    int world = 1, rank = 0;
    if (multi) {
        if (!shards) return fail("--gpus applies to build, align, distance and selftest");
        const int gpus = a.has("--gpus") ? atoi(a.get("--gpus").c_str()) : env_world;
        if (gpus < 1) return fail("--gpus must be one or higher");
        if (env_world <= 0 && gpus > 1) return launch_ranks(gpus, argv);                   // the parent only starts and reaps the ranks
        world = env_world > 0 ? env_world : 1; rank = env_rank;
        if (rank < 0 || rank >= world) return fail("SKX_RANK outside SKX_WORLD");
    }
    skx_ctx *ctx = nullptr;
    const int device = getenv("SKX_DEVICE") ? atoi(getenv("SKX_DEVICE")) : getenv("SKX_LOCAL_RANK") ? atoi(getenv("SKX_LOCAL_RANK")) : rank;
    if (skx_ctx_create(device, &ctx) != SKX_OK) return engine_fail();
    skx_phase_add("main.device_context", since());
    int rcode = 0;
    skx_array *arr = nullptr;
    if (multi && cmd == "selftest") rcode = selftest(ctx, rank, world);
    else if (cmd == "selftest") rcode = selftest(ctx, 0, 1);
    else if (multi) rcode = main_sharded(ctx, cmd, a, rank, world, threads);
    else if (cmd == "build") {
        if (!a.has("-o")) return fail("-o <output> is required");
        if (a.pos.empty() == !a.has("-f")) return fail("give either sequence files or -f <file_list>");
        BuildOpts bo; Inputs in;
        if (int e = parse_build_opts(a, bo)) return e;
        if (int e = read_inputs(a, in)) return e;
        const int k = bo.k; skx_qual &q = bo.q; const bool auto_count = bo.auto_count; const double prop = bo.prop;
        std::vector<std::string> &names = in.names, &f1 = in.f1, &f2 = in.f2;
        std::vector<const char *> &cn = in.cn, &c1 = in.c1, &c2 = in.c2;
        (void)names; (void)f1; (void)f2;
        if (auto_count && auto_min_count(ctx, in, k, !a.has("--single-strand"), q, true)) { skx_ctx_destroy(ctx); return engine_fail(); }
        if (skx_build_and_merge(ctx, cn.data(), c1.data(), c2.data(), (int)cn.size(), k, !a.has("--single-strand"), &q, threads, prop, &arr) != SKX_OK ||
            skh_save_skf(arr, a.get("-o").c_str()) != SKX_OK)
            rcode = engine_fail();
    } else if (cmd == "align") {
        if (a.pos.empty()) return fail("input required");
        std::vector<const char *> in; for (auto &p : a.pos) in.push_back(p.c_str());
        const int filter = parse_filter(a.get("--filter", "no-const"));
        if (filter < 0) return fail("invalid --filter");
        const double mf = atof(a.get("--min-freq", a.get("-m", "0.9")).c_str());
        if (mf < 0 || mf > 1) return fail("Frequency must be between 0 and 1 (inclusive)");
        int fd = 1;                                                                                               // io_utils::set_ostream: stdout or -o
        if (a.has("-o")) { fd = open(a.get("-o").c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644)   /* read-write: the writer maps the file */; if (fd < 0) return fail("cannot create output file"); }
        if (skh_align_inputs_fd(ctx, in.data(), (int)in.size(), threads, filter, a.has("--ambig-mask"), a.has("--no-gap-only-sites"), mf,
                                a.has("--filter-ambig-as-missing"), fd) != SKX_OK)
            rcode = engine_fail();
        if (fd != 1) close(fd);
    } else if (cmd == "distance") {
        if (a.pos.size() != 1) return fail("one .skf file required");
        const char *in[1] = {a.pos[0].c_str()};
        const double mf = atof(a.get("--min-freq", a.get("-m", "0")).c_str());
        char *buf = nullptr; uint64_t len = 0;
        (void)in;
        if (skh_distance_skf_tsv(ctx, a.pos[0].c_str(), mf, !a.has("--allow-ambiguous"), &buf, &len) != SKX_OK)
            rcode = engine_fail();
        else { rcode = emit(a.get("-o"), buf, len); skx_free(buf); }
    } else if (cmd == "nk") {
        if (a.pos.size() != 1) return fail("one .skf file required");
        const char *in[1] = {a.pos[0].c_str()};
        char *buf = nullptr; uint64_t len = 0;
        if (skh_load_array(ctx, in, 1, 1, &arr) != SKX_OK || skh_nk(arr, a.has("--full-info"), &buf, &len) != SKX_OK) rcode = engine_fail();
        else { rcode = emit("", buf, len); skx_free(buf); }
    } else if (cmd == "cov") {                                                                                    // cli.rs Cov, lib.rs:828-851
        if (a.pos.size() != 2) return fail("usage: ska cov <fastq_fwd> <fastq_rev> [-k K] [--single-strand]");
        const int k = atoi(a.get("-k", "31").c_str());
        if (k < 5 || k > 63 || k % 2 == 0) return fail("K-mer must be an odd number between 5 and 63 (inclusive)");
        char *buf = nullptr; uint64_t len = 0, cutoff = 0;
        if (skh_cov(ctx, a.pos[0].c_str(), a.pos[1].c_str(), k, !a.has("--single-strand"), &buf, &len, &cutoff) != SKX_OK) rcode = engine_fail();
        else { rcode = emit("", buf, len); skx_free(buf); fprintf(stderr, "Estimated cutoff\t%llu\n", (unsigned long long)cutoff); }
    } else if (cmd == "map") {                                                                                    // cli.rs Map, lib.rs:663-709
        if (a.pos.size() < 2) return fail("usage: ska map <reference.fa> <input.skf | sequence files...> [-o out] [-f aln|vcf]");
        const std::string fmt = a.get("--format", a.has("-f") ? a.get("-f") : "aln");                             // -f is the format here, not a file list
        if (fmt != "aln" && fmt != "vcf") return fail("invalid --format (aln | vcf)");
        std::vector<const char *> in; for (size_t i = 1; i < a.pos.size(); i++) in.push_back(a.pos[i].c_str());
        char *buf = nullptr; uint64_t len = 0;
        if (skh_load_array(ctx, in.data(), (int)in.size(), threads, &arr) != SKX_OK ||
            skx_array_map(arr, a.pos[0].c_str(), a.has("--ambig-mask"), a.has("--repeat-mask"), fmt == "vcf", threads, &buf, &len) != SKX_OK)
            rcode = engine_fail();
        else { rcode = emit(a.get("-o"), buf, len); skx_free(buf); }
    } else if (cmd == "merge") {                                                                                  // cli.rs Merge, lib.rs:728-741
        if (!a.has("-o")) return fail("-o <output> is required");
        std::vector<const char *> in; for (auto &p : a.pos) in.push_back(p.c_str());
        if (skh_merge(ctx, in.data(), (int)in.size(), a.get("-o").c_str()) != SKX_OK) rcode = engine_fail();
    } else if (cmd == "delete") {                                                                                 // cli.rs Delete, lib.rs:742-759
        const std::string skf = a.get("-s", a.get("--skf-file"));
        if (skf.empty()) return fail("-s <skf_file> is required");
        if (a.pos.empty() == !a.has("-f")) return fail("give either sample names or -f <file_list>");
        std::vector<std::string> names;
        if (a.has("-f")) {                                                                                        // io_utils::get_input_list: first column
            std::ifstream in(a.get("-f"));
            if (!in) return fail("Unable to open file_list");
            std::string line;
            while (std::getline(in, line)) { std::istringstream ls(line); std::string t; if (ls >> t) names.push_back(t); }
        } else names = a.pos;
        std::vector<const char *> cn; for (auto &n : names) cn.push_back(n.c_str());
        const char *in[1] = {skf.c_str()};
        const std::string out = a.get("-o", skf);
        if (skh_load_array(ctx, in, 1, 1, &arr) != SKX_OK || skh_delete(arr, cn.data(), (int)cn.size(), out.c_str()) != SKX_OK) rcode = engine_fail();
    } else if (cmd == "weed") {                                                                                   // cli.rs Weed, lib.rs:760-806
        if (a.pos.empty() || a.pos.size() > 2) return fail("usage: ska weed <skf_file> [weed_file]");
        const int filter = parse_filter(a.get("--filter", "no-filter"));
        if (filter < 0) return fail("invalid --filter");
        const double mf = atof(a.get("--min-freq", a.get("-m", "0.9")).c_str());
        if (mf < 0 || mf > 1) return fail("Frequency must be between 0 and 1 (inclusive)");
        const char *in[1] = {a.pos[0].c_str()};
        const std::string out = a.get("-o", a.pos[0]);
        if (skh_load_array(ctx, in, 1, 1, &arr) != SKX_OK ||
            skh_weed(arr, a.pos.size() == 2 ? a.pos[1].c_str() : nullptr, a.has("--reverse"), mf, a.has("--filter-ambig-as-missing"), filter,
                     a.has("--ambig-mask"), a.has("--no-gap-only-sites"), out.c_str()) != SKX_OK)
            rcode = engine_fail();
    } else {
        rcode = fail("unknown subcommand (this engine provides build, align, map, distance, nk, merge, delete, weed, cov)");
    }
    const double t_done = since();
    if (dbg) fprintf(stderr, "[skx] main: %s done after %.2f s\n", cmd.c_str(), t_done);
    if (arr) skx_array_free(arr);
    skx_ctx_destroy(ctx);
    skx_phase_add("main.release_device", since() - t_done);
    skx_phase_add("main.total", since());
    // lib.rs:888-891: what the reference says on stderr when a command has run through
    if (rank == 0 && rcode == 0 && cmd != "selftest") fprintf(stderr, "SKA done in %llus\n\xE2\xAC\x9B\xE2\xAC\x9C\xE2\xAC\x9B\xE2\xAC\x9C\xE2\xAC\x9B\xE2\xAC\x9C\xE2\xAC\x9B\n\xE2\xAC\x9C\xE2\xAC\x9B\xE2\xAC\x9C\xE2\xAC\x9B\xE2\xAC\x9C\xE2\xAC\x9B\xE2\xAC\x9C\n", (unsigned long long)since());
    if (rank == 0 && rcode == 0 && cmd != "selftest") skh_log(2, "ska", "Complete");
    if (const char *pp = rank == 0 ? getenv("SKX_PHASES") : nullptr) {                  // phase table of this invocation as JSON (bench.py's end_to_end leg)
        char *js = nullptr; uint64_t jl = 0;
        if (skx_phases_json(&js, &jl, 0) == SKX_OK) { if (FILE *f = fopen(pp, "w")) { fwrite(js, 1, jl, f); fputc('\n', f); fclose(f); } skx_free(js); }
    }
    return rcode;
}
