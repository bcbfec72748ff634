// skx_api.cpp -- the C ABI (include/skx.h): host orchestration of the gfx950 kernels for build -> merge -> filter -> distance and the
// .skf life-cycle operations (cov / map / .skf files: skx_api_io.cpp).
// No CPU fallback exists: without a usable HIP device every compute entry point fails with SKX_ENODEV.
#include "skx_internal.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <unordered_map>
#include <thread>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace skx;

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024];
void skx::set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
long skx::knob(const char *name, long absent)
{
    const char *e = getenv("SKX_KNOBS");
    if (!e) return absent;
    const size_t n = strlen(name);
    for (const char *p = e; *p;) {
        const char *q = strchr(p, ','); if (!q) q = p + strlen(p);
        if ((size_t)(q - p) >= n && !strncmp(p, name, n) && (p[n] == '=' || p + n == q)) return p[n] == '=' ? atol(p + n + 1) : 1;
        p = *q ? q + 1 : q;
    }
    return absent;
}
int skx::hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? SKX_ENOMEM : SKX_ENODEV;
}
extern "C" const char *skx_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ phases
namespace {
struct Phases { std::mutex mu; std::vector<std::pair<std::string, double>> v; } g_phases;
}
void skx::phase_add(const char *name, double secs)
{
    {
        std::lock_guard<std::mutex> lk(g_phases.mu);
        bool found = false;
        for (auto &p : g_phases.v) if (p.first == name) { p.second += secs; found = true; break; }
        if (!found) g_phases.v.emplace_back(name, secs);
    }
    if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] %-28s %.3f s\n", name, secs);
}
extern "C" void skx_phase_add(const char *name, double seconds) { if (name) skx::phase_add(name, seconds); }
extern "C" int skx_phases_json(char **buf, uint64_t *len, int reset)
{
    return skx_guarded([&]() -> int {
    std::string o = "{";
    {
        std::lock_guard<std::mutex> lk(g_phases.mu);
        for (size_t i = 0; i < g_phases.v.size(); i++) {
            char tmp[64]; snprintf(tmp, sizeof tmp, "%.6f", g_phases.v[i].second);
            o += (i ? ", \"" : "\"") + g_phases.v[i].first + "\": " + tmp;
        }
        if (reset) g_phases.v.clear();
    }
    o += "}";
    char *p = (char *)malloc(o.size() + 1);
    if (!p) { set_error("out of host memory"); return SKX_ENOMEM; }
    memcpy(p, o.c_str(), o.size() + 1);
    *buf = p; if (len) *len = o.size();
    return SKX_OK;
    });
}
extern "C" const char *skx_version(void) { return "0.5.2"; }      // Cargo.toml:3, written as ska_version
extern "C" void skx_free(void *p) { free(p); }

// ------------------------------------------------------------------------------------------ device memory cache
namespace {
// one pool per device: a block is only ever handed back to a request made while its own device is current
struct DevCache {
    std::mutex mu;
    std::map<int, std::multimap<size_t, void *>> free_blocks;      // device -> size -> block
    std::unordered_map<void *, std::pair<size_t, int>> live;       // block -> (size, device)
    size_t cached_bytes = 0;
} g_cache;
int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
}
void *skx::dev_alloc(size_t bytes, hipError_t *err)
{
    bytes = (bytes + 255) & ~(size_t)255;
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        auto &pool = g_cache.free_blocks[dev];
        auto it = pool.lower_bound(bytes);
        if (it != pool.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {
            void *p = it->second; size_t sz = it->first;
            pool.erase(it); g_cache.cached_bytes -= sz; g_cache.live[p] = {sz, dev};
            return p;
        }
    }
    void *p = nullptr;
    const auto t_malloc = std::chrono::steady_clock::now();
    struct Took { std::chrono::steady_clock::time_point t0; ~Took() { skx::phase_add("alloc.hipMalloc_all_threads", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } } took{t_malloc};      // (what the driver took to hand out memory: the phase table's answer to "where did the seconds go" behind a process that has just released its own)
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                                // out of memory: drop the cache and retry once
        (void)hipGetLastError();
        skx::dev_trim();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); if (err) *err = e; return nullptr; }
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.live[p] = {bytes, dev};
    return p;
}
void skx::dev_free(void *p)
{
    std::lock_guard<std::mutex> lk(g_cache.mu);
    auto it = g_cache.live.find(p);
    if (it == g_cache.live.end()) { (void)hipFree(p); return; }
    g_cache.free_blocks[it->second.second].emplace(it->second.first, p); g_cache.cached_bytes += it->second.first;
    g_cache.live.erase(it);
}
void skx::dev_trim()
{
    std::lock_guard<std::mutex> lk(g_cache.mu);
    for (auto &pool : g_cache.free_blocks) for (auto &kv : pool.second) (void)hipFree(kv.second);      // hipFree takes a block of any device
    g_cache.free_blocks.clear(); g_cache.cached_bytes = 0;
}

// CPUs this process may keep busy: the hardware's count or, where a control group caps the job (cgroup v2 cpu.max, v1 cfs quota), the cap --
// thread teams larger than that only take turns, and a team that overruns the quota has the whole process (the thread that feeds the GPU
// included) stopped for the rest of the scheduler's period
int skx::cpu_budget()
{
    static const int v = [] {
        int hc = (int)std::thread::hardware_concurrency(); if (hc < 1) hc = 1;
        double q = 0, per = 0; char w[64] = {0};
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { if (fscanf(f, "%63s %lf", w, &per) == 2 && strcmp(w, "max") != 0) q = atof(w); fclose(f); }
        if (q <= 0) {
            long long qq = -1, pp = 0;
            if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &qq) != 1) qq = -1; fclose(f); }
            if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &pp) != 1) pp = 0; fclose(f); }
            if (qq > 0 && pp > 0) { q = (double)qq; per = (double)pp; }
        }
        if (q > 0 && per > 0) hc = std::min(hc, std::max(1, (int)std::ceil(q / per)));
        return hc;
    }();
    return v;
}

int skx::check_k(int k)
{
    if (k < 5 || k > 63 || (k & 1) == 0) { set_error("Invalid k-mer length"); return SKX_EINVAL; }   // ska_dict.rs:342-344
    return SKX_OK;
}

// ------------------------------------------------------------------------------------------ output page allocation
skx::Preallocator::Preallocator(int fd_, off_t base_) : fd(fd_), base(base_)
{
    th = std::thread([this]() {
        constexpr uint64_t STEP = 256ull << 20;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return stop || done < target; });
            if (stop && done >= target) return;
            const uint64_t o = done, n = std::min(STEP, target - done);
            lk.unlock();
            // a failure (ENOSPC, quota) is recorded: a writer that stores through a mapping of pages that do not exist dies of SIGBUS
            if (posix_fallocate(fd, base + (off_t)o, (off_t)n) != 0) failed.store(true, std::memory_order_release);
            lk.lock();
            done = o + n;
            cv.notify_all();
        }
    });
}
skx::Preallocator::~Preallocator() { { std::lock_guard<std::mutex> lk(mu); stop = true; target = done; } cv.notify_all(); if (th.joinable()) th.join(); }
void skx::Preallocator::raise(uint64_t bytes) { { std::lock_guard<std::mutex> lk(mu); if (bytes > target) target = bytes; } cv.notify_all(); }
void skx::Preallocator::finish(uint64_t bytes)
{
    std::unique_lock<std::mutex> lk(mu);
    if (bytes > target) target = bytes;
    cv.notify_all();
    cv.wait(lk, [&] { return done >= bytes; });
}
// a regular file, opened read-write and not in append mode, can be written through a mapping
bool skx::mappable_output_fd(int fd, off_t *pos)
{
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
    const int fl = fcntl(fd, F_GETFL);
    if (fl < 0 || (fl & O_APPEND) || (fl & O_ACCMODE) != O_RDWR) return false;
    *pos = lseek(fd, 0, SEEK_CUR);
    return *pos >= 0;
}
extern "C" int skx_ctx_expect_output(skx_ctx *ctx, int fd)
{
    if (!ctx) return SKX_EINVAL;
    off_t pos;
    ctx->expect_fd = (fd >= 0 && mappable_output_fd(fd, &pos)) ? fd : -1;
    return SKX_OK;
}

// ------------------------------------------------------------------------------------------ ctx
extern "C" int skx_ctx_create(int device, skx_ctx **out)
{
    return skx_guarded([&]() -> int {
    if (!out) { set_error("null out"); return SKX_EINVAL; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { set_error("no HIP device available (%s); the engine has no CPU path", e == hipSuccess ? "count 0" : hipGetErrorString(e)); return SKX_ENODEV; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(device));
    skx_ctx *c = new skx_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev[0]) != hipSuccess || hipEventCreate(&c->ev[1]) != hipSuccess || hipEventCreate(&c->ev[2]) != hipSuccess ||
        hipEventCreate(&c->ev[3]) != hipSuccess) {
        delete c; set_error("cannot create HIP stream/events"); return SKX_ENODEV;
    }
    *out = c;
    return SKX_OK;
    });
}
extern "C" void skx_ctx_destroy(skx_ctx *c)
{
    if (!c) return;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    skx::dev_trim();
    delete c;
}
extern "C" int skx_ctx_sync(skx_ctx *c) { SKX_HIP(hipSetDevice(c->device)); SKX_HIP(hipStreamSynchronize(c->stream)); return SKX_OK; }
extern "C" void *skx_ctx_stream(skx_ctx *c) { return (void *)c->stream; }
uint64_t skx::next_object_id() { static std::atomic<uint64_t> n{0}; return ++n; }
extern "C" const char *skx_ctx_merge_path(skx_ctx *c) { return c ? c->merge_path.c_str() : ""; }
extern "C" int skx_ctx_timings(skx_ctx *c, skx_timings *t, int reset)
{
    return skx_guarded([&]() -> int {
    if (t) *t = c->tm;
    if (reset) c->tm = skx_timings{};
    return SKX_OK;
    });
}

namespace {
struct StageTimer {        // HIP-event bracket around one stage on the ctx stream
    skx_ctx *c; double *slot;
    StageTimer(skx_ctx *c_, double *s) : c(c_), slot(s) { if (c->timing) (void)hipEventRecord(c->ev[0], c->stream); }
    ~StageTimer()
    {
        if (!c->timing) return;
        (void)hipEventRecord(c->ev[1], c->stream);
        (void)hipEventSynchronize(c->ev[1]);
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) *slot += ms;
    }
};
struct KernelTimer {       // the same around one kernel inside a stage (its own pair of events)
    skx_ctx *c; double *slot;
    KernelTimer(skx_ctx *c_, double *s) : c(c_), slot(s) { if (c->timing) (void)hipEventRecord(c->ev[2], c->stream); }
    ~KernelTimer()
    {
        if (!c->timing) return;
        (void)hipEventRecord(c->ev[3], c->stream);
        (void)hipEventSynchronize(c->ev[3]);
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) *slot += ms;
    }
};
int ilog2_ceil(uint64_t x) { int l = 0; while ((1ull << l) < x) l++; return l; }
constexpr uint32_t LDS_TABLE_MAX = 18000;     // slots (+ pad) of 8 B must stay below 160 KiB
constexpr uint32_t LDS_SORT_MAX = 6144;       // words per region of the counting sort (24 per thread in registers)
constexpr uint32_t LDS_SORT_MAX_WIDE = 4096;  // 128-bit path (16 per thread)
}  // namespace

// ------------------------------------------------------------------------------------------ dictset
extern "C" void skx_dictset_free(skx_dictset *d) { delete d; }
extern "C" int skx_dictset_nsamples(const skx_dictset *d) { return d->n; }
extern "C" int skx_dictset_key_bits(const skx_dictset *d) { return d->key_bits; }

// The passing windows' packed words of every sample (reads_sample_words) -> (sample, bucket) regions like an assembly's -> the same dedupe
// kernel: sorted, folded, sub-indexed regions, so union / assemble treat reads and assemblies alike.  SKF_NOT_TAKEN: regions the LDS
// sort cannot hold (the sort-based form takes the batch).
static int reads_words_to_dictset(skx_ctx *ctx, std::vector<DevBuf<uint64_t>> &wl, std::vector<DevBuf<uint64_t>> &wh2, const std::vector<uint64_t> &cnt,
                                  int k, int rc, skx_dictset **out)
{
    const int n = (int)wl.size();
    hipStream_t st = ctx->stream;
    const bool wide_r = k > 31;
    const int wpk_r = wide_r ? 2 : 1, kbits = 2 * (k - 1);
    uint64_t maxn = 0;
    for (auto c : cnt) maxn = std::max(maxn, c);
    const uint64_t per = wide_r ? 2500 : 3500, lds_max = wide_r ? LDS_SORT_MAX_WIDE : LDS_SORT_MAX;
    int logB = std::min({ilog2_ceil((maxn + per - 1) / per), kbits, MAX_LOGB});
    if (logB < 0) logB = 0;
    std::unique_ptr<skx_dictset> d(new skx_dictset());
    d->ctx = ctx; d->n = n; d->k = k; d->rc = rc; d->logB = logB; d->hp = make_hash_params(std::min(k, 31)); d->wh = make_wide_hash(k);
    d->key_bits = wide_r ? 128 : 64;
    const uint64_t nreg = (uint64_t)n << logB;
    SKX_TRY(d->raw.alloc(nreg)); SKX_TRY(d->ucnt.alloc(nreg)); SKX_TRY(d->off.alloc(nreg + 1)); SKX_TRY(d->sidx.alloc(nreg * skx::SUBIDX));
    d->sb = std::min(4, kbits - logB);
    SKX_TRY(d->raw.zero(st));
    for (int s = 0; s < n; s++) launch_words_regions(true, wl[s].p, wide_r ? wh2[s].p : nullptr, cnt[s], kbits, logB, (uint64_t)s << logB, d->raw.p, nullptr, nullptr, nullptr, st);
    DevBuf<uint32_t> d_max, cursor; SKX_TRY(d_max.alloc(1)); SKX_TRY(cursor.alloc(nreg)); SKX_TRY(cursor.zero(st));
    launch_scan_u32(d->raw.p, d->off.p, nreg, d_max.p, st);
    uint64_t total = 0; uint32_t max_raw = 0;
    SKX_HIP(hipMemcpyAsync(&total, d->off.p + nreg, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(&max_raw, d_max.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    if (max_raw > lds_max) { if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] reads: region of %u words, batch left to the sort-based form\n", max_raw); return SKF_NOT_TAKEN; }   // e.g. min_count 2 on deep reads: every repeat occurrence is a word
    SKX_TRY(d->words.alloc(std::max<uint64_t>(total, 1) * wpk_r));
    for (int s = 0; s < n; s++) {
        launch_words_regions(false, wl[s].p, wide_r ? wh2[s].p : nullptr, cnt[s], kbits, logB, (uint64_t)s << logB, nullptr, d->off.p, cursor.p, d->words.p, st);
        wl[s].release(); wh2[s].release();
    }
    const uint32_t cap = std::max<uint32_t>(512, (uint32_t)std::min<uint64_t>(((uint64_t)max_raw + 255) / 256 * 256, lds_max));
    DevBuf<int> d_flag2; SKX_TRY(d_flag2.alloc(1)); SKX_TRY(d_flag2.zero(st));
    SKX_HIP(hipMemsetAsync(d->sidx.p, 0xFF, nreg * skx::SUBIDX * sizeof(uint16_t), st));
    { StageTimer t(ctx, &ctx->tm.dedupe);
      if (wide_r) launch_dedupe_wide((u128 *)d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, cap, kbits - logB, d_flag2.p, d->sidx.p, d->sb, st);
      else launch_dedupe_mb(d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, cap, d->hp.bits - logB, d_flag2.p, d->sidx.p, d->sb, st); }
    int overflow = 0;
    std::vector<uint32_t> ucnt(nreg);
    SKX_HIP(hipMemcpyAsync(&overflow, d_flag2.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(ucnt.data(), d->ucnt.p, nreg * 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    if (overflow) { if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] reads: dedupe overflow, batch left to the sort-based form\n"); return SKF_NOT_TAKEN; }
    d->sample_size.assign(n, 0);
    for (int s = 0; s < n; s++) for (uint64_t b = 0; b < (1ull << logB); b++) d->sample_size[s] += ucnt[((uint64_t)s << logB) + b];
    *out = d.release();
    return SKX_OK;

}

// the regions of a dictset sorted and folded in place (a sample's SkaDict per bucket: dedupe_mb_kernel / dedupe_wide_kernel, the table
// form for repeat-rich regions); *overflow != 0: a region does not fit, the caller builds again with more buckets
static int dictset_sort_flat(skx_ctx *ctx, skx_dictset *d);
static int dedupe_regions(skx_ctx *ctx, skx_dictset *d, uint32_t lds_cap, uint64_t maxlen, int *overflow_out, std::vector<uint32_t> &ucnt)
{
    hipStream_t st = ctx->stream;
    const bool wide = d->wide();
    const int logB = d->logB;
    const uint64_t nreg = (uint64_t)d->n << logB;
    const HashParams hp = d->hp;
    const int key_bits_used = 2 * (d->k - 1);
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1));
    {
    // LDS capacity (words) of the per-region counting sort: 12 B per word, <= 160 KiB
    uint32_t cap = std::max<uint32_t>(512, (uint32_t)std::min<uint64_t>(((uint64_t)lds_cap + 255) / 256 * 256, wide ? LDS_SORT_MAX_WIDE : LDS_SORT_MAX));
    // windows per region are Poisson around len / B: all but ~5 in 10 000 regions stay below mean + 3.3 sigma, which may need a
    // smaller launch shape than the regions' capacity does (launch_dedupe_mb)
    const double mean_r = (double)(maxlen >> logB) + 1.0;
    const uint32_t typical = wide ? 0u : (uint32_t)(mean_r + 3.3 * std::sqrt(mean_r) + 1.0);
    DevBuf<uint32_t> d_big;
    if (typical) SKX_TRY(d_big.alloc(nreg + 1));
    SKX_TRY(d_flag.zero(st));
    SKX_HIP(hipMemsetAsync(d->sidx.p, 0xFF, nreg * skx::SUBIDX * sizeof(uint16_t), st));     // 0xFFFF = sub-range without words
    { StageTimer t(ctx, &ctx->tm.dedupe);
      if (wide) launch_dedupe_wide((u128 *)d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, cap, key_bits_used - logB, d_flag.p, d->sidx.p, d->sb, st);
      else launch_dedupe_mb(d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, cap, hp.bits - logB, d_flag.p, d->sidx.p, d->sb, st, typical, d_big.p, 0); }
    int overflow = 0;
    SKX_HIP(hipMemcpyAsync(&overflow, d_flag.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    for (uint32_t from = dedupe_spill_grid(); !wide && (overflow & 4); from += dedupe_spill_grid()) {     // more listed regions than one grid of the second stage
        int keep = overflow & ~4;
        SKX_HIP(hipMemcpyAsync(d_flag.p, &keep, 4, hipMemcpyHostToDevice, st));
        { StageTimer t(ctx, &ctx->tm.dedupe); launch_dedupe_mb(d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, cap, hp.bits - logB, d_flag.p, d->sidx.p, d->sb, st, typical, d_big.p, from); }
        SKX_HIP(hipMemcpyAsync(&overflow, d_flag.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
    }
    if (getenv("SKX_DEBUG") && typical) {
        uint32_t listed = 0;
        SKX_HIP(hipMemcpy(&listed, d_big.p, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[skx] dedupe: typical region %u words, capacity %u; %u of %llu regions above the typical launch shape\n", typical, cap, listed, (unsigned long long)nreg);
    }
    if (!wide && overflow == 2) {
        // regions beyond the counting sort's capacity (repeat-rich buckets): table-based dedupe, only distinct keys must fit
        SKX_TRY(d_flag.zero(st));
        { StageTimer t(ctx, &ctx->tm.dedupe); launch_dedupe(d->words.p, d->off.p, d->raw.p, d->ucnt.p, nreg, LDS_TABLE_MAX, hp.bits - logB, d_flag.p, cap, d->sidx.p, d->sb, st); }
        SKX_HIP(hipMemcpyAsync(&overflow, d_flag.p, 4, hipMemcpyDeviceToHost, st));
    }
    ucnt.resize(nreg);
    SKX_HIP(hipMemcpyAsync(ucnt.data(), d->ucnt.p, nreg * 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    *overflow_out = overflow;
    return SKX_OK;
    }
}

// Samples whose regions are beyond the per-region LDS sort (assemblies above ~5 Mbp keep 2^10 regions that grow with them, so that the
// extraction kernel's (tile, bucket) chunks stay whole lines: round 6): every sample's words as ONE sorted, folded list -- the form the
// sort-based read-set path leaves (logB = 0), which every consumer of sorted dictionaries takes.  Radix sort + fold (skx_prims, skx_reads.hip).
static int dictset_sort_flat(skx_ctx *ctx, skx_dictset *d)
{
    hipStream_t st = ctx->stream;
    const bool wide = d->wide();
    const int wpk = wide ? 2 : 1, n = d->n;
    const uint64_t B = 1ull << d->logB, nreg = (uint64_t)n << d->logB;
    std::vector<uint32_t> raw(nreg);
    SKX_HIP(hipMemcpyAsync(raw.data(), d->raw.p, nreg * 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    std::vector<DevBuf<uint64_t>> lists(n);
    std::vector<uint64_t> sizes(n, 0), offs(n + 1, 0), dst(B);
    DevBuf<uint64_t> d_dst, lo, hi;
    SKX_TRY(d_dst.alloc(B));
    for (int s = 0; s < n; s++) {
        uint64_t m = 0;
        for (uint64_t b = 0; b < B; b++) { dst[b] = m; m += raw[(uint64_t)s * B + b]; }
        if (m == 0) continue;
        if (lo.n < m) { SKX_TRY(lo.alloc(m + m / 8)); if (wide) SKX_TRY(hi.alloc(m + m / 8)); }
        SKX_HIP(hipMemcpyAsync(d_dst.p, dst.data(), B * 8, hipMemcpyHostToDevice, st));
        launch_gather_regions(d->words.p, d->off.p, d->raw.p, d_dst.p, (uint64_t)s * B, B, wpk, lo.p, wide ? hi.p : nullptr, st);
        SKX_TRY(sort_fold_words(ctx, lo.p, wide ? hi.p : nullptr, m, lists[s], &sizes[s]));      // (returns with the stream idle: dst may be refilled)
        if (sizes[s] > 0xFFFFFFFFull) { set_error("sample too large"); return SKX_EUNSUP; }
    }
    for (int s = 0; s < n; s++) offs[s + 1] = offs[s] + sizes[s];
    lo.release(); hi.release();
    DevBuf<uint64_t> words, off; DevBuf<uint32_t> rawn, ucnt;
    SKX_TRY(words.alloc(offs[n] * wpk)); SKX_TRY(off.alloc(n + 1)); SKX_TRY(rawn.alloc(n)); SKX_TRY(ucnt.alloc(n));
    std::vector<uint32_t> uc(n);
    for (int s = 0; s < n; s++) {
        uc[s] = (uint32_t)sizes[s];
        if (sizes[s]) SKX_HIP(hipMemcpyAsync(words.p + offs[s] * wpk, lists[s].p, sizes[s] * 8 * wpk, hipMemcpyDeviceToDevice, st));
    }
    SKX_HIP(hipMemcpyAsync(off.p, offs.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(ucnt.p, uc.data(), n * 4, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(rawn.p, uc.data(), n * 4, hipMemcpyHostToDevice, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    d->words = std::move(words); d->off = std::move(off); d->raw = std::move(rawn); d->ucnt = std::move(ucnt);
    d->sidx.release(); d->sb = 0; d->logB = 0;
    d->sample_size = sizes;
    d->sorted = true;
    return SKX_OK;
}

int skx::dictset_sort(skx_dictset *d)
{
    if (d->sorted) return SKX_OK;
    skx_ctx *ctx = d->ctx;
    SKX_HIP(hipSetDevice(ctx->device));
    if (d->region_cap > (d->wide() ? LDS_SORT_MAX_WIDE : LDS_SORT_MAX)) return dictset_sort_flat(ctx, d);
    int overflow = 0;
    std::vector<uint32_t> ucnt;
    SKX_TRY(dedupe_regions(ctx, d, d->region_cap, d->maxlen, &overflow, ucnt));
    if (overflow) { set_error("internal: a fixed-capacity region did not fit the counting sort (%d)", overflow); return SKX_EUNSUP; }
    d->sample_size.assign(d->n, 0);
    for (int s = 0; s < d->n; s++) for (uint64_t b = 0; b < (1ull << d->logB); b++) d->sample_size[s] += ucnt[((uint64_t)s << d->logB) + b];
    d->sorted = true;
    return SKX_OK;
}

static int dictset_build_device(skx_ctx *ctx, const std::vector<const uint8_t *> &seqs, const std::vector<const uint8_t *> &quals,
                                const std::vector<uint64_t> &lens, int k, int rc, const skx_qual *q, skx_dictset **out)
{
    const int n = (int)seqs.size();
    hipStream_t st = ctx->stream;
    bool any_qual = false;
    for (auto p : quals) any_qual |= p != nullptr;
    // reads: per-sample quality gates + KmerFilter (skx_reads.hip); the dictset is then one sorted region per sample.  The same
    // sort-based path takes assemblies too large for the bucketed one (more than 2^MAX_LOGB regions' worth of windows: a 40 Mbp
    // sample and up, to 4 Gbp), with the filters switched off.
    auto build_sorted = [&]() -> int {
        const bool wide_r = k > 31;
        const int wpk_r = wide_r ? 2 : 1;
        std::vector<DevBuf<uint64_t>> lists(n);
        std::vector<uint64_t> sizes(n, 0), offs(n + 1, 0);
        for (int s = 0; s < n; s++) {
            skx_qual qs = q ? *q : skx_qual{5, 20, SKX_QUAL_STRICT};
            if (!quals[s]) { qs.min_count = 0; qs.qual_filter = SKX_QUAL_NOFILTER; }        // FASTA sample in a mixed batch: no filtering
            SKX_TRY(reads_sample_dict(ctx, seqs[s], quals[s], lens[s], k, rc, qs, lists[s], &sizes[s]));
            if (sizes[s] > 0xFFFFFFFFull) { set_error("sample too large"); return SKX_EUNSUP; }
            offs[s + 1] = offs[s] + sizes[s];
        }
        std::unique_ptr<skx_dictset> d(new skx_dictset());
        d->ctx = ctx; d->n = n; d->k = k; d->rc = rc; d->logB = 0; d->hp = make_hash_params(std::min(k, 31)); d->wh = make_wide_hash(k);
        d->key_bits = wide_r ? 128 : 64;
        SKX_TRY(d->words.alloc(offs[n] * wpk_r)); SKX_TRY(d->off.alloc(n + 1)); SKX_TRY(d->raw.alloc(n)); SKX_TRY(d->ucnt.alloc(n));
        std::vector<uint32_t> uc(n);
        for (int s = 0; s < n; s++) {
            uc[s] = (uint32_t)sizes[s];
            if (sizes[s]) SKX_HIP(hipMemcpyAsync(d->words.p + offs[s] * wpk_r, lists[s].p, sizes[s] * 8 * wpk_r, hipMemcpyDeviceToDevice, st));
        }
        SKX_HIP(hipMemcpyAsync(d->off.p, offs.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
        SKX_HIP(hipMemcpyAsync(d->ucnt.p, uc.data(), n * 4, hipMemcpyHostToDevice, st));
        SKX_HIP(hipMemcpyAsync(d->raw.p, uc.data(), n * 4, hipMemcpyHostToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));
        d->sample_size = sizes;
        *out = d.release();
        return SKX_OK;
    };
    // The same samples through the engine's own kernels (skx_reads2.hip): the windows that pass the gates and the count filter
    // come back as packed words, which go into (sample, bucket) regions like an assembly's and through the same dedupe kernel
    // -- sorted, folded, sub-indexed regions, so union / assemble treat reads and assemblies alike.  A sample the partition
    // kernels do not take (or regions the LDS sort cannot hold) sends the batch to the sort-based form above.
    auto build_bucketed_reads = [&]() -> int {
        std::vector<DevBuf<uint64_t>> wl(n), wh2(n);
        std::vector<uint64_t> cnt(n, 0);
        uint64_t maxn = 0;
        for (int s = 0; s < n; s++) {
            skx_qual qs = q ? *q : skx_qual{5, 20, SKX_QUAL_STRICT};
            if (!quals[s]) { qs.min_count = 0; qs.qual_filter = SKX_QUAL_NOFILTER; }        // FASTA sample in a mixed batch: no filtering
            const int r = reads_sample_words(ctx, seqs[s], quals[s], lens[s], k, rc, qs, wl[s], wh2[s], &cnt[s]);
            if (r != SKX_OK) return r;                                                       // SKF_NOT_TAKEN included
            maxn = std::max(maxn, cnt[s]);
        }
        return reads_words_to_dictset(ctx, wl, wh2, cnt, k, rc, out);
    };
    auto build_reads = [&]() -> int {
        if (!knob("reads_sort")) {
            const int r = build_bucketed_reads();
            if (r != SKF_NOT_TAKEN) return r;
        }
        return build_sorted();
    };
    const bool wide = k > 31;                                   // lib.rs:592: u64 for k <= 31, u128 above
    uint64_t maxlen = 0;
    for (auto l : lens) maxlen = std::max(maxlen, l);
    HashParams hp = make_hash_params(std::min(k, 31));
    WideHash wh = make_wide_hash(k);
    const int key_bits_used = 2 * (k - 1);
    // windows per bucket (upper bound): the 64-bit dedupe sorts up to 6 144 words per region in LDS and the regions get 20 % + 256
    // words of head-room, so 4 900 is the largest mean that fits -- and the largest buckets give the scatter its widest chunks
    // (128-bit keys: 3 200, so that a region's fixed capacity -- 20 % + 256 above the mean -- stays within the 4 096 words of the wide counting sort
    // and the samples can stay as extracted for the append pass whatever their length)
    uint64_t per_region = wide ? 3200 : 4900;
    if (knob("per_region") > 0) per_region = (uint64_t)knob("per_region");       // (measurements: 9800 = 2^9 regions for 5 Mbp samples, 2450 = 2^11)
    if (any_qual) return build_reads();
    // Longer samples (round 6): the bucket count stops growing at the 5 Mbp shape (2^10 regions; 2^11 for 128-bit keys) and the regions grow
    // instead -- the extraction kernel's (tile, bucket) chunks stay whole lines (with regions capped at 4 900 words a 20 Mbp sample took 13 ps per
    // base against 2.5, a 40 Mbp one 43, and beyond that the assembly kernels were left altogether), the append pass reads regions of any
    // size unsorted (2^(logQ - logB) row blocks per region: up to 32 readers a region), and who asks for sorted dictionaries of such samples
    // gets the flat sorted form (dictset_sort_flat).  Beyond 32 row blocks per region the buckets grow again (to 2^13: ~1.3 Gbp).
    const int need = std::max(0, ilog2_ceil((maxlen + per_region - 1) / per_region));
    const int base_logB = knob("per_region") > 0 ? std::min(need, MAX_LOGB) : (wide ? 11 : 10);
    const bool grow_regions = !knob("small_regions");
    if ((grow_regions ? need - 5 : need) > MAX_LOGB) return build_reads();
    int logB = std::min({grow_regions && need > base_logB ? std::max(base_logB, need - 5) : need, key_bits_used, MAX_LOGB});
    if (logB < 0) logB = 0;

    DevBuf<const uint8_t *> d_seqs, d_quals;
    DevBuf<uint64_t> d_lens;
    SKX_TRY(d_seqs.alloc(n)); SKX_TRY(d_lens.alloc(n));
    SKX_HIP(hipMemcpyAsync(d_seqs.p, seqs.data(), n * sizeof(void *), hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(d_lens.p, lens.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    if (any_qual) { SKX_TRY(d_quals.alloc(n)); SKX_HIP(hipMemcpyAsync(d_quals.p, quals.data(), n * sizeof(void *), hipMemcpyHostToDevice, st)); }
    DevBuf<int> d_flag;
    SKX_TRY(d_flag.alloc(1));

    bool exact = false;        // single pass with fixed-capacity regions first; exact two-pass layout if a region overflows
    for (;;) {
        std::unique_ptr<skx_dictset> d(new skx_dictset());
        d->ctx = ctx; d->n = n; d->k = k; d->rc = rc; d->logB = logB; d->hp = hp; d->wh = wh; d->key_bits = wide ? 128 : 64;
        const int wpk = wide ? 2 : 1;
        const int tile_bases = wide ? extract_tile_bases_wide() : extract_tile_bases(logB);
        const uint64_t nreg = (uint64_t)n << logB;
        SKX_TRY(d->raw.alloc(nreg)); SKX_TRY(d->ucnt.alloc(nreg)); SKX_TRY(d->off.alloc(nreg + 1));
        SKX_TRY(d->sidx.alloc(nreg * skx::SUBIDX)); d->sb = std::min(4, (wide ? key_bits_used : hp.bits) - logB);
        SKX_TRY(d->raw.zero(st)); SKX_TRY(d_flag.zero(st));

        ExtractArgs a{};
        a.seqs = d_seqs.p; a.quals = any_qual ? d_quals.p : nullptr; a.lens = d_lens.p; a.n_samples = n;
        a.tiles_max = (int)((maxlen + tile_bases - 1) / tile_bases);
        a.k = k; a.rc = rc; a.min_qual = q ? q->min_qual : 0; a.qual_filter = q ? q->qual_filter : 0;
        a.logB = logB; a.hp = hp; a.wh = wh; a.overflow = d_flag.p;
        uint32_t lds_cap;
        if (!exact) {
            // hashed buckets are Poisson around len/B: 20 % + 256 words of head-room covers ordinary repeat content
            const uint64_t mean = (maxlen >> logB) + 1;
            const uint32_t region_cap = (uint32_t)std::min<uint64_t>(((mean + mean / 5 + 256) + 63) / 64 * 64, 0x7FFFFFFFull);
            launch_fill_offsets(d->off.p, nreg, region_cap, st);
            SKX_TRY(d->words.alloc(nreg * (uint64_t)region_cap * wpk + 2048));      // (+ slack: the append pass reads whole chunks of up to 12 KB)
            { StageTimer t(ctx, &ctx->tm.scatter); a.hist = d->raw.p; a.off = d->off.p; a.words = d->words.p; a.capacity = region_cap;
              if (wide) launch_scatter_wide(a, st); else launch_scatter(a, st); }
            int over = 0;
            SKX_HIP(hipMemcpyAsync(&over, d_flag.p, 4, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            if (over) { exact = true; continue; }
            lds_cap = region_cap;
        } else {
            DevBuf<uint32_t> d_cursor, d_max;
            SKX_TRY(d_cursor.alloc(nreg)); SKX_TRY(d_max.alloc(1)); SKX_TRY(d_cursor.zero(st));
            { StageTimer t(ctx, &ctx->tm.hist); a.hist = d->raw.p; if (wide) launch_hist_wide(a, st); else launch_hist(a, st); }
            launch_scan_u32(d->raw.p, d->off.p, nreg, d_max.p, st);
            uint64_t total = 0; uint32_t max_raw = 0;
            SKX_HIP(hipMemcpyAsync(&total, d->off.p + nreg, 8, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipMemcpyAsync(&max_raw, d_max.p, 4, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            SKX_TRY(d->words.alloc(total * wpk));
            { StageTimer t(ctx, &ctx->tm.scatter); a.hist = d_cursor.p; a.off = d->off.p; a.words = d->words.p; a.capacity = 0xFFFFFFFFu;
              if (wide) launch_scatter_wide(a, st); else launch_scatter(a, st); }
            lds_cap = max_raw;
        }
        // Assemblies whose regions kept their fixed capacity stay as the extraction kernel left them (both key widths): MergeSkaDict::append
        // (skx_append.hip) reads them unsorted, and the sorted, folded form (a sample's SkaDict) is made when something asks for it
        // (skx::dictset_sort: skx_dictset_size / _export, the key-set union of a sharded job).  A fixed-capacity region always fits
        // the counting sort (region_cap <= LDS_SORT_MAX), so that later sort cannot come back for a finer split.
        d->maxlen = maxlen; d->region_cap = lds_cap;
        const bool big_regions = lds_cap > (wide ? LDS_SORT_MAX_WIDE : LDS_SORT_MAX);
        if (!exact && !any_qual && (big_regions || (!knob("sorted_dicts") && !(wide && knob("sorted_wide"))))) {
            DevBuf<unsigned long long> d_tot; SKX_TRY(d_tot.alloc(n));
            launch_region_totals(d->raw.p, n, logB, d_tot.p, st);
            d->raw_total.resize(n);
            SKX_HIP(hipMemcpyAsync(d->raw_total.data(), d_tot.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            SKX_HIP(hipGetLastError());
            d->sorted = false;
            // (regions beyond the per-region sort stay as extracted whatever the knobs say -- that is what keeps the extraction's chunks whole --
            // and the knobs that ask for sorted dictionaries get them from the flat sort)
            if (big_regions && (knob("sorted_dicts") || (wide && knob("sorted_wide")))) SKX_TRY(dictset_sort(d.get()));
            *out = d.release();
            return SKX_OK;
        }
        if (big_regions) {
            // an exact layout (a region overflowed its fixed capacity: repeat content) of regions the LDS sort cannot hold: the flat sort
            d->sorted = false;
            SKX_TRY(dictset_sort(d.get()));
            *out = d.release();
            return SKX_OK;
        }
        int overflow = 0;
        std::vector<uint32_t> ucnt;
        SKX_TRY(dedupe_regions(ctx, d.get(), lds_cap, maxlen, &overflow, ucnt));
        if (overflow) {
            if (logB >= std::min(key_bits_used, MAX_LOGB)) return build_reads();       // repeat content beyond every region size
            logB++;
            continue;
        }
        d->sample_size.assign(n, 0);
        for (int s = 0; s < n; s++) for (uint64_t b = 0; b < (1ull << logB); b++) d->sample_size[s] += ucnt[((uint64_t)s << logB) + b];
        *out = d.release();
        return SKX_OK;
    }
}

extern "C" int skx_dictset_build(skx_ctx *ctx, const skx_stream *samples, int n, int on_device, int k, int rc, const skx_qual *q, skx_dictset **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !samples || n <= 0 || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    SKX_HIP(hipSetDevice(ctx->device));
    std::vector<const uint8_t *> seqs(n), quals(n, nullptr);
    std::vector<uint64_t> lens(n);
    std::vector<DevBuf<uint8_t>> own;
    if (on_device) {
        for (int i = 0; i < n; i++) {
            if (((uintptr_t)samples[i].seq & 15) || ((uintptr_t)samples[i].qual & 15)) { set_error("device record streams must be 16-byte aligned"); return SKX_EINVAL; }
            seqs[i] = samples[i].seq; quals[i] = samples[i].qual; lens[i] = samples[i].len;
        }
    } else {
        own.resize(2 * (size_t)n);
        for (int i = 0; i < n; i++) {
            lens[i] = samples[i].len;
            SKX_TRY(own[2 * i].alloc(samples[i].len + 16));
            SKX_HIP(hipMemcpyAsync(own[2 * i].p, samples[i].seq, samples[i].len, hipMemcpyHostToDevice, ctx->stream));
            seqs[i] = own[2 * i].p;
            if (samples[i].qual) {
                SKX_TRY(own[2 * i + 1].alloc(samples[i].len + 16));
                SKX_HIP(hipMemcpyAsync(own[2 * i + 1].p, samples[i].qual, samples[i].len, hipMemcpyHostToDevice, ctx->stream));
                quals[i] = own[2 * i + 1].p;
            }
        }
    }
    skx_dictset *d = nullptr;
    SKX_TRY(dictset_build_device(ctx, seqs, quals, lens, k, rc, q, &d));
    for (int s = 0; s < n; s++)
        if ((d->sorted ? d->sample_size[s] : d->raw_total[s]) == 0) { set_error("sample %d has no valid sequence", s); delete d; return SKX_EEMPTY; }
    *out = d;
    return SKX_OK;
    });
}


// A read set's records -> the five bit planes (groups of 64 positions x 5 words: two code bits, the bytes valid_base rejects, line ends,
// quality verdicts), fed line by line from stream_fastq_file; `push` takes a finished group.  Used by the reader threads that pack on the
// host and by the consumer when the device's framing calls a sample irregular (then it is this code that accepts or refuses the file).
namespace {
struct PlanePacker {
    std::vector<uint64_t> pl;                                           // a record's four planes (sequence line, then its quality line)
    size_t line_n = 0;
    uint64_t cur[5] = {0, 0, 0, 0, 0}, pos = 0, cap = 0;                // the group being filled; positions so far; the most that may come
    int min_qual = 20; bool gz = false;
    std::function<int(const uint64_t *)> push;
    static constexpr int OVER_BOUND = -1002;                            // a gzip file longer than its trailer says: not an error of the input
    int emit(int which, const uint8_t *p, size_t nb)
    {
        const size_t words = (nb + 1 + 63) / 64;                        // the line and its end
        if (which == 0) {
            if (pos + nb + 1 > cap) { if (gz) return OVER_BOUND; skx::set_error("Invalid FASTA/Q record"); return SKX_EIO; }
            if (pl.size() < 4 * words) pl.resize(4 * words + 64);
            line_n = nb;
            for (int pln = 0; pln < 4; pln++) pl[pln * words + words - 1] = 0;
            skx::pack_bases_planes(p, nb, &pl[0], &pl[words], &pl[2 * words]);
            return SKX_OK;
        }
        if (nb != line_n) { skx::set_error("Invalid FASTA/Q record"); return SKX_EIO; }
        skx::pack_qual_plane(p, nb, min_qual, &pl[3 * words]);
        const uint64_t *lo = &pl[0], *hi = &pl[words], *bd = &pl[2 * words], *qb = &pl[3 * words];
        for (size_t w = 0; w < words; w++) {
            const unsigned take = (unsigned)std::min<size_t>(64, nb + 1 - 64 * w), off = (unsigned)(pos & 63);
            const uint64_t v[5] = {lo[w], hi[w], bd[w], nb / 64 == w ? 1ull << (nb & 63) : 0ull, qb[w]};
            for (int pln = 0; pln < 5; pln++) cur[pln] |= v[pln] << off;
            pos += take;
            if (off + take >= 64) {
                const int pr = push(cur); if (pr != SKX_OK) return pr;
                for (int pln = 0; pln < 5; pln++) cur[pln] = off ? v[pln] >> (64 - off) : 0ull;      // (what did not fit; bits beyond `take` are zero)
            }
        }
        return SKX_OK;
    }
    int finish() { return (pos & 63) ? push(cur) : SKX_OK; }           // the last, partly filled group
};
}  // namespace

// Read sets (every sample plain FASTQ, one or two files), pipelined.  The one-shot form below reads every sample, allocates stream buffers the
// size of all files together (24 GB for 96 isolates of BASELINE config 5's shape: a 1-3 s allocation when the memory has just been released
// by another process) and then filters one isolate after the other (11 ms each) on an idle PCIe link.  Here a small pool of stream slots
// (two device buffers per slot, sized for the largest sample) is filled by the reader threads through the pinned ring, and this thread runs a
// sample's window / count-filter kernels (reads_sample_words) as soon as its last piece has arrived, then hands the slot back: reading,
// upload and kernels overlap, and the device holds a few samples' text instead of all of it.  Results are those of the one-shot form
// (the per-sample kernels do not depend on the order samples arrive in).  SKF_NOT_TAKEN: not this kind of input, or a sample the
// partition kernels leave to the sort-based form -- the caller takes the one-shot path from the start.
static int build_reads_pipelined(skx_ctx *ctx, const char *const *file1, const char *const *file2, int n, int k, int rc, const skx_qual *q, int threads,
                                 skx_dictset **out)
{
    if (knob("no_reads_pipeline") || n < 2) return SKF_NOT_TAKEN;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> bound(n, 0), text_bytes(n, 0);
    struct GzSizes { uint64_t comp[2] = {0, 0}, hint[2] = {0, 0}; int files = 0, gz_files = 0; };
    std::vector<GzSizes> gzs(n);                                                // (a sample whose files are all gzip may be inflated on the device)
    uint64_t slot_bytes = 0, raw_cap = 0, comp_cap = 0;
    bool any_gz = false;
    constexpr int SKF_OVER_BOUND = PlanePacker::OVER_BOUND;
    for (int i = 0; i < n; i++) {
        uint64_t bytes = 0;
        for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
            if (!f) continue;
            struct stat sb; unsigned char c0[2] = {0, 0}, tail[4] = {0, 0, 0, 0};
            const int fd = ::open(f, O_RDONLY);
            bool ok = fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && ::read(fd, c0, 2) == 2;
            uint64_t plain = ok ? (uint64_t)sb.st_size : 0;
            if (ok && c0[0] == 0x1f && c0[1] == 0x8b) {
                // gzip: the reader thread inflates as it goes; the stream's size from the trailer (ISIZE, the last member's length mod 2^32).  A
                // batch in which a file turns out longer than that says is left to the one-shot form (SKF_OVER_BOUND below)
                ok = sb.st_size > 18 && pread(fd, tail, 4, sb.st_size - 4) == 4;
                plain = (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) | ((uint64_t)tail[3] << 24);
                // a trailer that cannot be the whole text (shorter than the file itself): several members -- bgzip's 64 KB blocks, files
                // joined with cat -- or 4 GB and more.  Six times the file's size then stands for the text's length (reads deflate 3-5 x);
                // a text that turns out longer sends the batch to the one-shot form like any file longer than its bound
                if (ok && plain < (uint64_t)sb.st_size) plain = 6 * (uint64_t)sb.st_size;
                any_gz = true;
                gzs[i].gz_files++;
            } else ok = ok && c0[0] == '@';
            if (fd >= 0) ::close(fd);
            if (!ok) return SKF_NOT_TAKEN;
            if (gzs[i].files < 2) { gzs[i].comp[gzs[i].files] = (uint64_t)sb.st_size; gzs[i].hint[gzs[i].files] = plain; }
            gzs[i].files++;
            bytes += plain;
        }
        if (gzs[i].gz_files == gzs[i].files) comp_cap = std::max<uint64_t>(comp_cap, ((gzs[i].comp[0] + 64 + 255) & ~255ull) + (gzs[i].files > 1 ? ((gzs[i].comp[1] + 64 + 255) & ~255ull) : 0ull));
        bound[i] = (bytes / 2 + 64 + 255) & ~255ull;                             // plain FASTQ holds at most half its bytes in either stream
        text_bytes[i] = bytes;
        slot_bytes = std::max(slot_bytes, bound[i]);
        raw_cap = std::max(raw_cap, bytes);
    }
    SKX_HIP(hipSetDevice(ctx->device));
    const int nt = std::max(1, std::min({threads, n, 64, cpu_budget()}));      // (parsing + packing: a reader keeps a CPU busy)
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // a slot per reader thread and a few waiting for their kernels: more only costs allocation time (64 slots = 17 GB took 4.7 s right after
    // another process had released the memory, 32 slots 0.26 s: profiles/r03zr_reads_pipeline_512.log)
    // A sample crosses PCIe in one of two forms, chosen by its reader thread when it starts on it (round 6):
    //   * PACKED -- bit planes, groups of 64 positions, five words each: two code bits, the bases valid_base rejects, the line ends, the quality
    //     verdicts (fastx.cpp pack_*_planes) -- framed and packed by the reader thread: 5 bits per position instead of two bytes, 157 MB per 50x
    //     isolate, 0.1-0.3 s of a CPU;
    //   * RAW -- the file's bytes as read() delivers them (0.55 GB per 50x isolate), the reader thread does nothing else; the device frames the
    //     records and makes the same planes (skx_fastq.hip).
    // Raw text alone is bound by the link (~32 GB/s with the readers running = 60 isolates/s), packing alone by the CPUs (16 of them: 40-80
    // isolates/s); a reader takes RAW while the pinned ring has room -- the link is keeping up -- and PACKED when it is filling up, so
    // both are busy.  SKX_KNOBS=reads_raw=1: never raw; =2: always.  The window pass and the rebuild of the passing windows' words read the planes
    // as they are, whoever made them.
    //   * GZDEV (gzip files) -- the COMPRESSED bytes as read() delivers them, half to a fifth of the text: the reader thread does nothing else, and
    //     the device inflates (skx_gzdev.hip: block finder, symbolic decode per 64 KB chunk, window maps, text, member lengths and CRCs), frames and
    //     packs.  A file the device does not vouch for (damaged, unusual header, a stretch that deflates beyond the symbol area) goes through the
    //     reader threads' inflater on this thread, which accepts it or words the error.  SKX_KNOBS=reads_gz=1: inflate on the reader threads.
    const long raw_knob = knob("reads_raw");
    const bool raw_possible = raw_knob != 1 && raw_cap + 2 < 0xFFFFFF00ull;
    //     Both inflaters work at once: a few reader threads (gz_feed of them) only feed the device -- a 50x isolate is 0.1-0.3 s of read() for them
    //     and ~50 ms of the device's inflater -- and the others inflate and hand over text or planes as before (~1.1 s of a thread an isolate);
    //     all take their samples from the same counter, so the split follows the two rates.  reads_gz=2: the device only.
    const long gz_knob = knob("reads_gz");
    const bool gz_device = any_gz && raw_possible && comp_cap > 0 && gz_knob != 1;
    const int gz_tail = (int)(knob("reads_gz_tail") > 0 ? knob("reads_gz_tail") : 20);
    const int gz_feed = !gz_device ? 0 : gz_knob == 2 ? 1 << 30 : (int)std::max<long>(1, knob("reads_gz_feed") > 0 ? knob("reads_gz_feed") : 3);
    int n_gz_samples = 0;
    for (int i = 0; i < n; i++) if (gz_device && gzs[i].gz_files == gzs[i].files) n_gz_samples++;
    const uint64_t pslot_bytes = ((slot_bytes / 64 + 2) * READ_GROUP_BYTES + 255) & ~255ull;
    const uint64_t rslot_bytes = raw_possible ? ((raw_cap + 2 + 64 + 255) & ~255ull) : 0;
    // Two pools of device slots: packed samples (157 MB each at 50x of 5 Mbp; a reader each and a few waiting for their kernels) and raw ones
    // (0.55 GB each: a few -- the link carries about one at a time -- plus ONE buffer for the planes the device makes of them, since the
    // kernels take a sample at a time).  Kept small on purpose: a pool of 24 slots that hold either form is 17 GB, and allocating that right
    // after another process has released its memory took 1.4-1.9 s of a 3 s build (profiles/r06b_reads_modes.log).
    int P = (int)std::min<uint64_t>((uint64_t)n, std::max<uint64_t>(2, std::min<uint64_t>((uint64_t)nt + 8, (free_b / 8) / (pslot_bytes + 1))));
    // raw slots: up to one per reader and two waiting for their kernels (a reader holds its slot for as long as it reads -- 0.1 s of a file read
    // before, 0.25 s of one read for the first time, when sixteen read()s contend for the page cache's LRU lock -- so six slots carried 24 raw
    // samples a second at most and the link idled at 12 GB/s: profiles/r06d_reads_1000.log).  They are allocated one by one by a helper thread
    // while the pipeline already runs on the packed pool: 11 GB taken at once right after another process released its memory cost 1.4-1.9 s.
    const int R = raw_possible ? (int)std::min<uint64_t>((uint64_t)n, std::max<uint64_t>(2, std::min<uint64_t>(raw_knob == 2 || (any_gz && !gz_device) ? (uint64_t)nt + 2 : gz_device ? (uint64_t)std::max(1, (nt - std::min(gz_feed, nt)) / 2) + 2 : (uint64_t)std::max(1, nt / 8) + 3, (free_b / 8) / (rslot_bytes + 1)))) : 0;      // (beside the device's inflater: half the inflating readers send text, the others planes)
    // slots for compressed samples: one per reader and a few waiting for the inflater (0.27 GB each at 50x of 5 Mbp); the text the device makes of
    // a sample lives in ONE buffer (the kernels take a sample at a time)
    const uint64_t gslot_bytes = gz_device ? comp_cap : 0;
    const int G = gz_device ? (int)std::min<uint64_t>((uint64_t)n_gz_samples, std::max<uint64_t>(2, std::min<uint64_t>((uint64_t)std::min(gz_feed, nt) + 3, (free_b / 8) / (gslot_bytes + 1)))) : 0;
    if (raw_knob == 2) P = 1;
    DevBuf<uint8_t> packed_pool, raw_planes, gz_text;
    std::vector<DevBuf<uint8_t>> raw_slots((size_t)R), gz_slots((size_t)G);
    SKX_TRY(packed_pool.alloc((uint64_t)P * pslot_bytes));
    if (R) { SKX_TRY(raw_planes.alloc(pslot_bytes)); SKX_TRY(raw_slots[0].alloc(rslot_bytes)); }
    if (G) { SKX_TRY(gz_text.alloc(rslot_bytes)); SKX_TRY(gz_slots[0].alloc(gslot_bytes)); }
    constexpr size_t SLOT = ((8u << 20) / READ_GROUP_BYTES) * READ_GROUP_BYTES;          // whole groups
    constexpr size_t RAW_CHUNK = (SLOT - 1) / 256 * 256;                                 // raw text leaves in pieces that keep their destinations aligned
    const int min_qual_host = q ? (int)q->min_qual : 20;
    const int n_slots = 2 * nt + 8;
    struct Sample { int slot = -1; int pending = 0; bool read_done = false, queued = false, raw = false, gzdev = false; uint64_t len = 0, junction = 0, coff[2] = {0, 0}; };
    struct Ring {
        uint8_t *base = nullptr; std::mutex mu; std::condition_variable cv_free, cv_work, cv_stream, cv_ready;
        std::vector<int> free_slots, free_stream, free_raw, free_gz; struct Req { int slot; uint8_t *dst; size_t bytes; int sample; }; std::deque<Req> work;
        std::deque<int> ready; int readers_left = 0, up_pending = 0, raw_active = 0, gz_active = 0; bool failed = false, abort = false, prefer_packed = false;
        ~Ring() { if (base) (void)hipHostFree(base); }
    } ring;
    if (hipHostMalloc((void **)&ring.base, (size_t)n_slots * SLOT, hipHostMallocDefault) != hipSuccess) { ring.base = nullptr; return SKF_NOT_TAKEN; }
    for (int b = 0; b < n_slots; b++) ring.free_slots.push_back(b);
    for (int p = 0; p < P; p++) ring.free_stream.push_back(p);
    if (R) ring.free_raw.push_back(0);
    if (G) ring.free_gz.push_back(0);
    ring.readers_left = nt;
    std::thread raw_alloc([&]() {
        (void)hipSetDevice(ctx->device);
        for (int p = 1; p < G; p++) {
            { std::lock_guard<std::mutex> lk(ring.mu); if (ring.abort || ring.readers_left == 0) break; }
            if (gz_slots[(size_t)p].alloc(gslot_bytes) != SKX_OK) break;
            { std::lock_guard<std::mutex> lk(ring.mu); ring.free_gz.push_back(p); }
            ring.cv_stream.notify_all();
        }
        for (int p = 1; p < R; p++) {
            { std::lock_guard<std::mutex> lk(ring.mu); if (ring.abort || ring.readers_left == 0) break; }
            if (raw_slots[(size_t)p].alloc(rslot_bytes) != SKX_OK) break;               // (no room: the pipeline goes on with what there is)
            { std::lock_guard<std::mutex> lk(ring.mu); ring.free_raw.push_back(p); }
            ring.cv_stream.notify_all();
        }
    });
    struct JoinAlloc { std::thread &t; ~JoinAlloc() { if (t.joinable()) t.join(); } } join_alloc{raw_alloc};
    phase_add("build.alloc_text_pin_ring", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    std::vector<Sample> smp(n);
    std::vector<int> rcodes(n, SKX_OK);
    std::vector<std::string> errs(n);
    auto mark_ready_locked = [&](int i) { Sample &x = smp[i]; if (x.read_done && x.pending == 0 && !x.queued) { x.queued = true; ring.ready.push_back(i); } };
    std::vector<std::thread> uploaders;
    const int n_up = 2;
    for (int u = 0; u < n_up; u++) uploaders.emplace_back([&]() {
        (void)hipSetDevice(ctx->device);
        hipStream_t up = nullptr;
        if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) up = nullptr;
        std::vector<Ring::Req> batch;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(ring.mu);
                ring.cv_work.wait(lk, [&] { return !ring.work.empty() || ring.readers_left == 0; });
                if (ring.work.empty() && ring.readers_left == 0) break;
                const size_t take = std::max<size_t>(1, ring.work.size() / 2);
                batch.assign(ring.work.begin(), ring.work.begin() + (ptrdiff_t)take); ring.work.erase(ring.work.begin(), ring.work.begin() + (ptrdiff_t)take);
            }
            bool bad = false;
            for (auto &r : batch) bad |= hipMemcpyAsync(r.dst, ring.base + (size_t)r.slot * SLOT, r.bytes, hipMemcpyHostToDevice, up) != hipSuccess;
            bad |= hipStreamSynchronize(up) != hipSuccess;
            {
                std::lock_guard<std::mutex> lk(ring.mu);
                if (bad) { ring.failed = true; ring.abort = true; }
                for (auto &r : batch) { ring.free_slots.push_back(r.slot); smp[r.sample].pending--; ring.up_pending--; mark_ready_locked(r.sample); }
            }
            ring.cv_free.notify_all(); ring.cv_ready.notify_all();
            if (bad) { ring.cv_stream.notify_all(); }
        }
        if (up) (void)hipStreamDestroy(up);
    });
    std::atomic<long long> us_wait_stream{0}, us_wait_ring{0}, us_files{0}, n_raw{0}, n_gzdev{0}, bytes_up{0};      // summed over the reader threads
    auto us_since = [](std::chrono::steady_clock::time_point t) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); };
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&, t]() {
            const bool feeder = t < gz_feed;                              // (this thread hands gzip samples to the device's inflater)
            struct Leave { Ring &r; ~Leave() { { std::lock_guard<std::mutex> lk(r.mu); r.readers_left--; } r.cv_work.notify_all(); r.cv_ready.notify_all(); } } leave{ring};
            PlanePacker pk;
            pk.min_qual = min_qual_host; pk.gz = any_gz;
            for (int i;;) {
                // the batch's last samples are left to the feeders: a thread that starts inflating one now (~1.1 s) would finish after the device has
                // been through all of them (~50 ms each)
                if (!feeder && gz_feed > 0 && n_gz_samples == n && n - next.load() < gz_tail) { std::lock_guard<std::mutex> lk(ring.mu); if (!ring.prefer_packed) break; }
                if ((i = next.fetch_add(1)) >= n) break;
                int sslot = -1; bool raw = false, gzdev = false;
                {
                    const auto tw = std::chrono::steady_clock::now();
                    std::unique_lock<std::mutex> lk(ring.mu);
                    // the link keeps up (few filled pieces of the pinned ring wait for their copy) and a raw slot is to be had: this sample goes as it
                    // is; otherwise it is packed here.  (reads_raw=2: raw whatever the ring says -- then a raw slot is waited for.)
                    // Measured (profiles/r06e_reads_modes.log, 16 readers): files read before -- packed 124 isolates/s through the pipeline, raw 91 (the link:
                    // 46 GB/s), and a reader's time is the read() either way (0.10 s of its 0.11 s per isolate: packing is what fits beside it); files
                    // read for the first time -- 35 isolates/s in every form (sixteen read()s of fresh tmpfs pages share 26 GB/s).  So raw text is
                    // what relieves a processor that packs slowly or inflates (gzip: every sample raw), and beside fast packers only a sample or two
                    // at a time travel raw, on bandwidth the link has left.
                    const int raw_cap = any_gz ? nt : std::max(1, nt / 8);
                    auto want_raw = [&] { return R > 0 && !ring.prefer_packed && (raw_knob == 2 || (!ring.free_raw.empty() && ring.raw_active < raw_cap && ring.up_pending * 4 <= n_slots)); };
                    // a sample of gzip files: its compressed bytes, the device inflates (unless a sample before it turned out irregular: then the
                    // readers inflate and pack, as they do for plain files)
                    auto want_gzdev = [&] { return G > 0 && feeder && gzs[i].gz_files == gzs[i].files && !ring.prefer_packed; };
                    ring.cv_stream.wait(lk, [&] { return ring.abort || (want_gzdev() ? !ring.free_gz.empty() : want_raw() ? !ring.free_raw.empty() : !ring.free_stream.empty()); });
                    us_wait_stream += us_since(tw);
                    if (ring.abort) return;
                    gzdev = want_gzdev();
                    raw = !gzdev && want_raw();
                    std::vector<int> &fl = gzdev ? ring.free_gz : raw ? ring.free_raw : ring.free_stream;
                    sslot = fl.back(); fl.pop_back();
                    smp[i].slot = sslot; smp[i].raw = raw; smp[i].gzdev = gzdev;
                    if (raw) ring.raw_active++;
                    if (gzdev) ring.gz_active++;
                }
                const auto t_files = std::chrono::steady_clock::now();
                struct Out { int slot = -1; size_t used = 0; uint8_t *dst = nullptr; uint64_t off = 0; } x;
                x.dst = gzdev ? gz_slots[(size_t)sslot].p : raw ? raw_slots[(size_t)sslot].p : packed_pool.p + (uint64_t)sslot * pslot_bytes;
                auto flush = [&]() {
                    if (x.slot < 0) return;
                    { std::lock_guard<std::mutex> lk(ring.mu); ring.work.push_back({x.slot, x.dst + x.off, x.used, i}); smp[i].pending++; ring.up_pending++; }
                    ring.cv_work.notify_one();
                    bytes_up += (long long)x.used;
                    x.off += x.used; x.slot = -1; x.used = 0;
                };
                auto give_back = [&]() {
                    if (x.slot >= 0) { { std::lock_guard<std::mutex> lk(ring.mu); ring.free_slots.push_back(x.slot); } ring.cv_free.notify_one(); x.slot = -1; }
                };
                auto take_slot = [&]() -> int {
                    if (x.slot >= 0) return SKX_OK;
                    const auto tw = std::chrono::steady_clock::now();
                    std::unique_lock<std::mutex> lk(ring.mu);
                    ring.cv_free.wait(lk, [&] { return !ring.free_slots.empty() || ring.abort; });
                    us_wait_ring += us_since(tw);
                    if (ring.abort) return SKF_ABORTED;                   // somebody else stopped the pipeline: not this reader's failure
                    x.slot = ring.free_slots.back(); ring.free_slots.pop_back(); x.used = 0;
                    return SKX_OK;
                };
                int r = SKX_OK;
                uint64_t sample_len = 0, junction = 0;
                if (gzdev) {
                    // the files as they are, each followed by zeros to the next multiple of 256 bytes (at least 64: the inflater's bit reader looks ahead)
                    n_gzdev++;
                    uint64_t total = 0;
                    int fno = 0;
                    for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
                        if (!f) continue;
                        const uint64_t want = gzs[i].comp[fno];
                        smp[i].coff[fno] = total;
                        const int fd = ::open(f, O_RDONLY);
                        if (fd < 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                        struct Close { int fd; ~Close() { ::close(fd); } } cl{fd};
                        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
                        uint64_t got_file = 0;
                        while (got_file < want) {
                            if ((r = take_slot()) != SKX_OK) break;
                            uint8_t *dstp = ring.base + (size_t)x.slot * SLOT + x.used;
                            const ssize_t rd = ::read(fd, dstp, std::min<uint64_t>(RAW_CHUNK - x.used, want - got_file));
                            if (rd < 0 && errno == EINTR) continue;
                            if (rd < 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                            if (rd == 0) { r = SKF_OVER_BOUND; break; }              // (the file shrank since it was measured: the one-shot form takes the batch)
                            x.used += (size_t)rd; got_file += (uint64_t)rd; total += (uint64_t)rd;
                            if (x.used >= RAW_CHUNK) flush();
                        }
                        if (r != SKX_OK) break;
                        uint64_t zeros = ((want + 64 + 255) & ~255ull) - want;
                        while (zeros) {
                            if ((r = take_slot()) != SKX_OK) break;
                            const size_t c = (size_t)std::min<uint64_t>(zeros, RAW_CHUNK - x.used);
                            memset(ring.base + (size_t)x.slot * SLOT + x.used, 0, c);
                            x.used += c; zeros -= c; total += c;
                            if (x.used >= RAW_CHUNK) flush();
                        }
                        if (r != SKX_OK) break;
                        fno++;
                    }
                    sample_len = total;
                } else if (raw) {
                    // the files' bytes into the pinned ring, nothing else: plain files by read() straight into a slot, gzip files inflated by this
                    // thread's inflater and copied there.  A '\n' is put behind a file that lacks its last one (the device frames lines by their ends)
                    n_raw++;
                    uint64_t total = 0; const uint64_t cap = text_bytes[i] + 2;
                    int fno = 0;
                    for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
                        if (!f) continue;
                        if (fno++ == 1) junction = total;
                        const int fd = ::open(f, O_RDONLY);
                        if (fd < 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                        struct Close { int fd; ~Close() { ::close(fd); } } cl{fd};
                        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
                        unsigned char mg[2] = {0, 0};
                        const bool gz = pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
                        std::unique_ptr<GzReader> zr;
                        if (gz) { zr.reset(new GzReader); zr->open(fd); }
                        uint8_t last = '\n'; bool first = true; uint64_t got_file = 0;
                        for (;;) {
                            if ((r = take_slot()) != SKX_OK) break;
                            uint8_t *dstp = ring.base + (size_t)x.slot * SLOT + x.used;
                            const size_t room = RAW_CHUNK - x.used;               // (> 0: a full piece has left; a byte beyond it stays free for the '\n' a file may lack)
                            size_t got = 0;
                            if (zr) {
                                const uint8_t *np; size_t ng;
                                // (the inflater hands out what it has, up to its window: copied piecewise into the slot)
                                if (zr->next(&np, &ng, 0) != 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                                if (ng == 0) break;
                                size_t done = 0;
                                while (done < ng && r == SKX_OK) {
                                    if ((r = take_slot()) != SKX_OK) break;
                                    dstp = ring.base + (size_t)x.slot * SLOT + x.used;
                                    const size_t c = std::min(ng - done, RAW_CHUNK - x.used);
                                    if (total + c > cap) { r = SKF_OVER_BOUND; break; }
                                    memcpy(dstp, np + done, c);
                                    if (first) { first = false; if (np[0] != '@') { r = SKF_OVER_BOUND; break; } }      // (a gzip file that is not FASTQ: the one-shot form takes the batch)
                                    x.used += c; done += c; total += c; got_file += c; last = np[done - 1];
                                    if (x.used >= RAW_CHUNK) flush();
                                }
                                if (r != SKX_OK) break;
                                continue;
                            }
                            const ssize_t rd = ::read(fd, dstp, std::min<uint64_t>(room, cap - total));
                            if (rd < 0 && errno == EINTR) continue;
                            if (rd < 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                            got = (size_t)rd;
                            if (got == 0) {
                                // (the file grew since it was measured: what the bound was made from no longer holds)
                                if (total >= cap) { char c1; if (::read(fd, &c1, 1) > 0) { set_error("Invalid FASTA/Q record"); r = SKX_EIO; } }
                                break;
                            }
                            if (first) { first = false; if (dstp[0] != '@') { set_error("Invalid FASTA/Q record"); r = SKX_EIO; break; } }
                            x.used += got; total += got; got_file += got; last = dstp[got - 1];
                            if (x.used >= RAW_CHUNK) flush();
                        }
                        if (r != SKX_OK) break;
                        if (got_file == 0) { set_error("Invalid path/file: %s", f); r = SKX_EIO; break; }
                        if (last != '\n') {
                            if ((r = take_slot()) != SKX_OK) break;
                            ring.base[(size_t)x.slot * SLOT + x.used] = '\n'; x.used++; total++;
                            if (x.used >= RAW_CHUNK) flush();
                        }
                    }
                    sample_len = total;
                } else {
                    pk.pos = 0; pk.cap = bound[i] - 32; for (auto &c : pk.cur) c = 0;
                    pk.push = [&](const uint64_t *grp) -> int {
                        const int tr = take_slot(); if (tr != SKX_OK) return tr;
                        memcpy(ring.base + (size_t)x.slot * SLOT + x.used, grp, READ_GROUP_BYTES);
                        x.used += READ_GROUP_BYTES;
                        if (x.used == SLOT) flush();
                        return SKX_OK;
                    };
                    const std::function<int(int, const uint8_t *, size_t)> emit = [&](int which, const uint8_t *p, size_t nb) -> int { return pk.emit(which, p, nb); };
                    for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
                        if (!f) continue;
                        r = stream_fastq_file(f, emit);
                        if (r == SKF_NOT_TAKEN && !any_gz) { set_error("Invalid FASTA/Q record"); r = SKX_EIO; }      // (the first byte was '@' a moment ago)
                        if (r == SKF_NOT_TAKEN) r = SKF_OVER_BOUND;                                      // (a gzip file that is not FASTQ: the one-shot form takes the batch)
                        if (r != SKX_OK) break;
                    }
                    if (r == SKX_OK) r = pk.finish();
                    sample_len = pk.pos;
                }
                if (r == SKX_OK) flush();
                if (raw) { std::lock_guard<std::mutex> lk(ring.mu); ring.raw_active--; }
                if (r != SKX_OK) {
                    give_back();
                    if (r != SKF_ABORTED) { rcodes[i] = r; errs[i] = skx_last_error(); }      // (only the failure that started it is reported)
                    { std::lock_guard<std::mutex> lk(ring.mu); ring.abort = true; }
                    ring.cv_stream.notify_all(); ring.cv_free.notify_all(); ring.cv_ready.notify_all();
                    return;
                }
                us_files += us_since(t_files);
                { std::lock_guard<std::mutex> lk(ring.mu); smp[i].len = sample_len; smp[i].junction = junction; smp[i].read_done = true; mark_ready_locked(i); }
                ring.cv_ready.notify_all();
            }
        });
    // this thread: a sample's kernels as soon as its text is on the device
    std::vector<DevBuf<uint64_t>> wl(n), wh2(n);
    std::vector<uint64_t> cnt(n, 0);
    int done = 0, krc = SKX_OK, n_irregular = 0, n_gz_host = 0;
    double t_kernels = 0.0, t_frame = 0.0, t_inflate = 0.0;
    FastqScratch fsc;
    GzDevWork gzw2[1][2];                                               // (a sample's two files)
    // a gzip sample's two files are decoded on streams of their own, beside each other AND beside the kernels of the samples the reader threads
    // inflated (this thread goes on with those while a decode is in flight, one at a time: the inflater's buffers are one set)
    struct Aux {
        hipStream_t s[2] = {nullptr, nullptr}; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; GzDevFileInfo *fi = nullptr;      // (events, verdicts: [file]; ev[2]: a sample's text is made)
        ~Aux() { for (auto x : s) if (x) (void)hipStreamDestroy(x); for (auto e : ev) if (e) (void)hipEventDestroy(e); if (fi) (void)hipHostFree(fi); }
    } aux;
    if (G) {
        // (the reader threads are running: a failure here stops the pipeline the way a failed kernel does)
        const int ar = [&]() -> int {
            for (int f = 0; f < 2; f++) SKX_HIP(hipStreamCreateWithFlags(&aux.s[f], hipStreamNonBlocking));
            for (int e = 0; e < 4; e++) SKX_HIP(hipEventCreateWithFlags(&aux.ev[e], hipEventDisableTiming));
            SKX_HIP(hipHostMalloc((void **)&aux.fi, 4 * sizeof(GzDevFileInfo), hipHostMallocDefault));
            return SKX_OK;
        }();
        if (ar != SKX_OK) { krc = ar; { std::lock_guard<std::mutex> lk(ring.mu); ring.abort = true; } ring.cv_stream.notify_all(); ring.cv_free.notify_all(); }
    }
    // the inflater's buffers at the size of the batch's largest file, before the first decode (see gz_device_reserve)
    if (G && krc == SKX_OK) {
        for (int i = 0; i < n && krc == SKX_OK; i++)
            if (gzs[i].gz_files == gzs[i].files)
                for (int f = 0; f < gzs[i].files && krc == SKX_OK; f++) krc = gz_device_reserve(gzw2[0][f], gzs[i].comp[f], gzs[i].hint[f]);
        if (krc != SKX_OK) { { std::lock_guard<std::mutex> lk(ring.mu); ring.abort = true; } ring.cv_stream.notify_all(); ring.cv_free.notify_all(); }
    }
    int inflight = -1; const int inflight_set = 0, cur_set = 0;
    auto gz_start = [&](int i, int set) -> int {
        const uint8_t *comp = gz_slots[(size_t)smp[i].slot].p;
        for (int f = 0; f < gzs[i].files; f++) {
            GzDevWork &wk = gzw2[set][f];
            SKX_TRY(gz_device_decode(ctx, aux.s[f], comp + smp[i].coff[f], gzs[i].comp[f], gzs[i].hint[f], wk));
            SKX_HIP(hipMemcpyAsync(&aux.fi[set * 2 + f], wk.finfo.p, sizeof(GzDevFileInfo), hipMemcpyDeviceToHost, aux.s[f]));
            SKX_HIP(hipEventRecord(aux.ev[set * 2 + f], aux.s[f]));
        }
        return SKX_OK;
    };
    auto gz_decoded = [&](int i, int set) -> bool { for (int f = 0; f < gzs[i].files; f++) if (hipEventQuery(aux.ev[set * 2 + f]) != hipSuccess) return false; return true; };
    auto stop_pipeline = [&]() { { std::lock_guard<std::mutex> lk(ring.mu); ring.abort = true; } ring.cv_stream.notify_all(); ring.cv_free.notify_all(); };
    // a sample through the host reader on this thread: what the device's framing calls irregular, and gzip files the device's inflater does not
    // vouch for -- the reader accepts what is merely unusual and words the error for what is wrong
    auto host_planes = [&](int i, uint8_t *slot_p, uint64_t &positions) -> int {
        std::vector<uint64_t> hp;
        PlanePacker pk; pk.min_qual = min_qual_host; pk.gz = any_gz; pk.cap = bound[i] - 32;
        pk.push = [&](const uint64_t *grp) -> int { hp.insert(hp.end(), grp, grp + 5); return SKX_OK; };
        const std::function<int(int, const uint8_t *, size_t)> emit = [&](int which, const uint8_t *p, size_t nb) -> int { return pk.emit(which, p, nb); };
        int hr = SKX_OK;
        for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
            if (!f) continue;
            hr = stream_fastq_file(f, emit);
            if (hr == SKF_NOT_TAKEN && !any_gz) { set_error("Invalid FASTA/Q record"); hr = SKX_EIO; }
            if (hr == SKF_NOT_TAKEN || hr == SKF_OVER_BOUND) hr = SKF_NOT_TAKEN;          // (the one-shot form takes the batch)
            if (hr != SKX_OK) break;
        }
        if (hr == SKX_OK) hr = pk.finish();
        if (hr == SKX_OK && !hp.empty() && hipMemcpyAsync(slot_p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) hr = SKX_ENODEV;
        if (hr == SKX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) hr = SKX_ENODEV;      // (hp goes)
        positions = pk.pos;
        return hr;
    };
    while (done < n) {
        int i = -1; bool finish = false;
        {
            std::unique_lock<std::mutex> lk(ring.mu);
            ring.cv_ready.wait(lk, [&] { return !ring.ready.empty() || ring.abort || inflight >= 0; });      // (every sample is queued by whoever sees its last piece arrive)
            if (ring.abort) break;
            // the sample in the inflater is taken up again when its decode has ended, when nothing else waits, or when the next one needs the inflater
            if (inflight >= 0 && (ring.ready.empty() || smp[ring.ready.front()].gzdev || gz_decoded(inflight, inflight_set))) finish = true;
            else { i = ring.ready.front(); ring.ready.pop_front(); }
        }
        if (finish) { i = inflight; inflight = -1; }
        else if (smp[i].gzdev) {
            krc = gz_start(i, 0);
            if (krc != SKX_OK) { stop_pipeline(); break; }
            inflight = i;
            continue;
        }
        // the next gzip sample's decode starts as soon as this one's text is made (the inflater's buffers are free then: the streams wait for that
        // on the device), beside this one's framing and window kernels
        auto start_next_gz = [&](bool after_text) -> int {
            int j = -1;
            { std::lock_guard<std::mutex> lk(ring.mu); if (!ring.ready.empty() && smp[ring.ready.front()].gzdev) { j = ring.ready.front(); ring.ready.pop_front(); } }
            if (j < 0) return SKX_OK;
            if (after_text) {
                SKX_HIP(hipEventRecord(aux.ev[2], ctx->stream));
                for (int f = 0; f < 2; f++) SKX_HIP(hipStreamWaitEvent(aux.s[f], aux.ev[2], 0));
            }
            SKX_TRY(gz_start(j, 0));
            inflight = j;
            return SKX_OK;
        };
        const auto tk = std::chrono::steady_clock::now();
        skx_qual qs = q ? *q : skx_qual{5, 20, SKX_QUAL_STRICT};
        uint8_t *slot_p = smp[i].raw || smp[i].gzdev ? raw_planes.p : packed_pool.p + (uint64_t)smp[i].slot * pslot_bytes;      // where the sample's planes are
        uint64_t positions = smp[i].len;
        if (smp[i].gzdev) {
            // compressed bytes: both files decoded to symbols side by side, the verdicts and lengths read back, then the text of one behind the
            // other's (a '\n' behind a file that lacks its last one, as the raw form's readers put it), member CRCs checked, and the device's framing
            const int nf = gzs[i].files;
            GzDevFileInfo fi[2] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
            bool vouched = true;
            GzDevWork *gzw = gzw2[cur_set];
            for (int f = 0; f < nf && krc == SKX_OK; f++) {
                if (hipEventSynchronize(aux.ev[cur_set * 2 + f]) != hipSuccess) krc = SKX_ENODEV;
                fi[f] = aux.fi[cur_set * 2 + f];
            }
            uint64_t junction = 0, len = 0;
            if (krc == SKX_OK) {
                for (int f = 0; f < nf; f++) vouched = vouched && fi[f].status == 0 && fi[f].total > 0;
                if (vouched) {
                    junction = nf > 1 ? fi[0].total + (fi[0].last != '\n') : 0;
                    len = (nf > 1 ? junction + fi[1].total + (fi[1].last != '\n') : fi[0].total + (fi[0].last != '\n'));
                    if (len > text_bytes[i] + 2 || len + 64 > gz_text.n || fi[0].first != '@' || (nf > 1 && fi[1].first != '@')) vouched = false;      // (the host reader decides what it is)
                }
            }
            if (krc == SKX_OK && vouched) {
                hipStream_t st = ctx->stream;
                uint64_t at = 0;
                for (int f = 0; f < nf && krc == SKX_OK; f++) {
                    krc = gz_device_text(ctx, st, gzw[f], gz_text.p + at, fi[f].total, fi[f].n_members);
                    at += fi[f].total;
                    if (krc == SKX_OK && fi[f].last != '\n') { if (hipMemsetAsync(gz_text.p + at, '\n', 1, st) != hipSuccess) krc = SKX_ENODEV; at++; }
                }
                for (int f = 0; f < nf && krc == SKX_OK; f++)
                    if (hipMemcpyAsync(&fi[f], gzw[f].finfo.p, sizeof(GzDevFileInfo), hipMemcpyDeviceToHost, st) != hipSuccess) krc = SKX_ENODEV;
                if (krc == SKX_OK) krc = start_next_gz(true);
                t_inflate += std::chrono::duration<double>(std::chrono::steady_clock::now() - tk).count();
                int irregular = 0;
                if (krc == SKX_OK) krc = fastq_frame_planes(ctx, gz_text.p, len, junction, min_qual_host, (uint64_t *)slot_p, fsc, &positions, &irregular);      // (returns with the stream idle)
                for (int f = 0; f < nf; f++) vouched = vouched && fi[f].status == 0;                                                                       // (the members' CRCs)
                if (krc == SKX_OK && vouched && irregular) {
                    n_irregular++;
                    { std::lock_guard<std::mutex> lk(ring.mu); ring.prefer_packed = true; }
                    krc = host_planes(i, slot_p, positions);
                }
            }
            if (krc == SKX_OK && !vouched) {
                if (inflight < 0) krc = start_next_gz(false);
                if (knob("gz_debug"))
                    fprintf(stderr, "gz on device: sample %d (%s) not vouched for: status %u / %u, text %llu / %llu bytes, first bytes %u / %u\n", i, file1[i], fi[0].status, fi[1].status,
                            (unsigned long long)fi[0].total, (unsigned long long)fi[1].total, fi[0].first, fi[1].first);
                n_gz_host++; krc = host_planes(i, slot_p, positions);
            }
            t_frame += std::chrono::duration<double>(std::chrono::steady_clock::now() - tk).count();
        } else if (smp[i].raw) {
            // the device frames the records and makes the planes; a text it calls irregular goes through the host reader here and now (which
            // accepts what is merely unusual -- blank lines between records -- and words the error for what is wrong), and the readers pack the
            // samples that follow: files of one run tend to share their quirks
            int irregular = 0;
            krc = fastq_frame_planes(ctx, raw_slots[(size_t)smp[i].slot].p, smp[i].len, smp[i].junction, min_qual_host, (uint64_t *)slot_p, fsc, &positions, &irregular);
            if (krc == SKX_OK && irregular) {
                n_irregular++;
                { std::lock_guard<std::mutex> lk(ring.mu); ring.prefer_packed = true; }
                krc = host_planes(i, slot_p, positions);
            }
            t_frame += std::chrono::duration<double>(std::chrono::steady_clock::now() - tk).count();
        }
        // (the kernels read the packed planes themselves: the two record streams never exist in memory)
        if (krc == SKX_OK) krc = reads_sample_words(ctx, nullptr, nullptr, positions, k, rc, qs, wl[i], wh2[i], &cnt[i], (const uint64_t *)slot_p);      // (returns with the stream idle: the slot is free)
        t_kernels += std::chrono::duration<double>(std::chrono::steady_clock::now() - tk).count();
        { std::lock_guard<std::mutex> lk(ring.mu); (smp[i].gzdev ? ring.free_gz : smp[i].raw ? ring.free_raw : ring.free_stream).push_back(smp[i].slot); if (smp[i].gzdev) ring.gz_active--; if (krc != SKX_OK) ring.abort = true; }
        ring.cv_stream.notify_all();
        if (krc != SKX_OK) { ring.cv_free.notify_all(); break; }
        done++;
    }
    { std::lock_guard<std::mutex> lk(ring.mu); if (done < n) ring.abort = true; }
    ring.cv_stream.notify_all(); ring.cv_free.notify_all();
    for (auto &th : pool) th.join();
    for (auto &u : uploaders) u.join();
    phase_add("build.read_upload", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    phase_add("build.reads_kernels_overlapped", t_kernels);
    phase_add("build.reads_device_framing_overlapped", t_frame);
    phase_add("build.readers_files_thread_s", us_files.load() * 1e-6);               // parse + pack, waits for pinned slots included
    phase_add("build.readers_wait_pinned_thread_s", us_wait_ring.load() * 1e-6);
    phase_add("build.readers_wait_device_slot_thread_s", us_wait_stream.load() * 1e-6);
    phase_add("build.reads_samples_sent_raw", (double)n_raw.load());
    phase_add("build.reads_samples_sent_compressed", (double)n_gzdev.load());
    phase_add("build.reads_samples_inflated_on_host_after_all", (double)n_gz_host);
    phase_add("build.reads_device_inflate_overlapped", t_inflate);
    phase_add("build.reads_samples_irregular", (double)n_irregular);
    phase_add("build.reads_uploaded_GB", (double)bytes_up.load() * 1e-9);
    if (ring.failed) { set_error("upload of the sequence files failed"); return SKX_ENODEV; }
    // the kernels' verdict first (SKF_NOT_TAKEN included: the one-shot form takes the batch -- the readers it interrupted recorded nothing),
    // then the reader whose own failure stopped the pipeline
    if (krc != SKX_OK) return krc;
    for (int i = 0; i < n; i++) if (rcodes[i] == SKF_OVER_BOUND) return SKF_NOT_TAKEN;
    for (int i = 0; i < n; i++) if (rcodes[i] != SKX_OK) { set_error("%s", errs[i].c_str()); return rcodes[i]; }
    if (done < n) { set_error("internal: read-set pipeline stopped early"); return SKX_EUNSUP; }
    raw_alloc.join();
    packed_pool.release(); raw_planes.release(); raw_slots.clear(); gz_slots.clear(); gz_text.release();
    for (auto &st2 : gzw2) for (auto &g : st2) g = GzDevWork();
    const auto t1 = std::chrono::steady_clock::now();
    skx_dictset *d = nullptr;
    int r = reads_words_to_dictset(ctx, wl, wh2, cnt, k, rc, &d);
    if (r == SKF_NOT_TAKEN) return r;                                           // (regions beyond the LDS sort: the sort-based form, from the files)
    if (r != SKX_OK) return r;
    phase_add("build.dictionaries", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    for (int sidx2 = 0; sidx2 < n; sidx2++)
        if ((d->sorted ? d->sample_size[sidx2] : d->raw_total[sidx2]) == 0) { set_error("%s has no valid sequence", file1[sidx2]); delete d; return SKX_EEMPTY; }
    *out = d;
    return SKX_OK;
}

// Test hook (not part of the drop-in boundary): FASTQ text through the device framing (skx_fastq.hip), the planes expanded to the two record
// streams the read-set kernels would see -- sequence A C T G N '\n', quality ' ' (passes) / '!' (fails) / '\n'.  seq / qual: room for len / 2 + 64.
extern "C" int skx_debug_fastq_frame(skx_ctx *ctx, const uint8_t *text, uint64_t len, uint64_t junction, int min_qual, uint8_t *seq, uint8_t *qual,
                                     uint64_t *positions, int *irregular)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !text || !seq || !qual || !positions || !irregular) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<uint8_t> d_text, d_seq, d_qual; DevBuf<uint64_t> planes;
    SKX_TRY(d_text.alloc(len + 64)); SKX_TRY(planes.alloc((len / 2 / 64 + 4) * 5));
    SKX_HIP(hipMemsetAsync(d_text.p, 0xAA, len + 64, st));                // (what lies behind the text is not zero in the pipeline either)
    SKX_HIP(hipMemcpyAsync(d_text.p, text, len, hipMemcpyHostToDevice, st));
    FastqScratch sc;
    SKX_TRY(fastq_frame_planes(ctx, d_text.p, len, junction, min_qual, planes.p, sc, positions, irregular));
    if (*irregular || *positions == 0) { SKX_HIP(hipStreamSynchronize(st)); return SKX_OK; }
    SKX_TRY(d_seq.alloc(*positions + 64)); SKX_TRY(d_qual.alloc(*positions + 64));
    launch_expand_planes(planes.p, *positions, d_seq.p, d_qual.p, st);
    SKX_HIP(hipMemcpyAsync(seq, d_seq.p, *positions, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(qual, d_qual.p, *positions, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    return SKX_OK;
    });
}

// Test hook (not part of the drop-in boundary): a gzip file's bytes through the device inflater (skx_gzdev.hip).  status: gzd::Status -- 0: text[0..total)
// is the members' text, lengths and CRCs checked; otherwise nothing is vouched for (the engine would take the file through the reader threads'
// inflater).  ms[0..2): device time of the decode kernels and of the text + CRC kernels.
extern "C" int skx_debug_gz_inflate(skx_ctx *ctx, const uint8_t *gz, uint64_t len, uint64_t text_hint, uint8_t *text, uint64_t cap, uint64_t *total,
                                    uint32_t *status, uint32_t *n_members, double *ms)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !gz || !text || !total || !status || !n_members) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<uint8_t> d_gz, d_text;
    SKX_TRY(d_gz.alloc(len + 64));
    SKX_HIP(hipMemsetAsync(d_gz.p + (len & ~3ull), 0, 64 - (len & 3u), st));
    SKX_HIP(hipMemcpyAsync(d_gz.p, gz, len, hipMemcpyHostToDevice, st));
    GzDevWork wk;
    hipEvent_t e0, e1, e2;
    SKX_HIP(hipEventCreate(&e0)); SKX_HIP(hipEventCreate(&e1)); SKX_HIP(hipEventCreate(&e2));
    struct Ev { hipEvent_t a, b, c; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } ev{e0, e1, e2};
    int rc2 = SKX_OK;
    for (int rep = 0; rep < 2 && rc2 == SKX_OK; rep++) {                  // (the second pass is the timed one: buffers exist)
        SKX_HIP(hipEventRecord(e0, st));
        rc2 = gz_device_decode(ctx, st, d_gz.p, len, text_hint, wk);
        SKX_HIP(hipEventRecord(e1, st));
    }
    SKX_TRY(rc2);
    GzDevFileInfo fi;
    SKX_HIP(hipMemcpyAsync(&fi, wk.finfo.p, sizeof fi, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    float m0 = 0, m1 = 0;
    (void)hipEventElapsedTime(&m0, e0, e1);
    if (ms) { ms[0] = m0; ms[1] = 0; }
    *total = fi.total; *status = fi.status; *n_members = fi.n_members;
    if (fi.status == 0 && fi.total > cap) { *status = 2; return SKX_OK; }
    if (fi.status == 0 && fi.total) {
        SKX_TRY(d_text.alloc(fi.total + 64));
        SKX_HIP(hipEventRecord(e1, st));
        SKX_TRY(gz_device_text(ctx, st, wk, d_text.p, fi.total, fi.n_members));
        SKX_HIP(hipEventRecord(e2, st));
        SKX_HIP(hipMemcpyAsync(&fi, wk.finfo.p, sizeof fi, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipMemcpyAsync(text, d_text.p, fi.total, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        (void)hipEventElapsedTime(&m1, e1, e2);
        *status = fi.status;
        if (fi.status == 0 && (fi.first != text[0] || fi.last != text[fi.total - 1])) { set_error("internal: first / last byte of the inflated text"); return SKX_EUNSUP; }
    }
    if (ms) { ms[0] = m0; ms[1] = m1; }
    SKX_HIP(hipGetLastError());
    return SKX_OK;
    });
}

extern "C" int skx_dictset_build_files(skx_ctx *ctx, const char *const *file1, const char *const *file2, int n, int k, int rc,
                                       const skx_qual *q, int threads, double proportion_reads, skx_dictset **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !file1 || n <= 0 || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    if (!(proportion_reads > 0.0) && !knob("host_parse") && !knob("reads_sort")) {
        const int pr = build_reads_pipelined(ctx, file1, file2, n, k, rc, q, threads, out);
        if (pr != SKF_NOT_TAKEN) return pr;
    }
    // Reader threads.  A plain (uncompressed, single-file) FASTA sample is not parsed on the host at all: its bytes are read
    // into pinned memory and uploaded as they are, and the device strips headers and line breaks (skx_parse.hip) -- the host
    // side of an assembly is one read() and one asynchronous copy.  FASTQ, .gz and two-file samples are parsed by the host
    // reader (fastx.cpp) and uploaded as record streams.  Either way the uploads of some samples run beside the reading of
    // others, each thread on its own stream.
    std::vector<int> rcodes(n, SKX_OK);
    std::vector<std::string> errs(n);
    std::vector<DevBuf<uint8_t>> d_seq(n), d_qual(n);
    std::vector<uint64_t> raw_len(n, 0), slot_off(n, 0), slot_len(n, 0);
    std::vector<char> is_raw(n, 0);
    std::vector<skx_stream> ss(n);
    size_t step = 1;
    if (proportion_reads > 0.0) { step = (size_t)std::llround(1.0 / proportion_reads); if (step == 0) step = 1; }
    const bool device_parse = step == 1 && !knob("host_parse");
    const auto t_read0 = std::chrono::steady_clock::now();
    bool any_pair = false;
    for (int i = 0; file2 && i < n; i++) any_pair |= file2[i] != nullptr;
    const int nt = std::max(1, std::min({threads, n, any_pair ? 64 : 32, std::max(8, 2 * cpu_budget())}));      // (paired read sets are parsed on the host: CPU work, more threads pay)      // 5 GB of FASTA text: 0.43 / 0.24 / 0.24 / 0.32 s with 8 / 16 / 32 / 64 readers (tools/read_knobs.py)
    // Raw path plumbing: reader threads make no HIP calls at all (creating a stream or a pinned buffer per thread serialises in
    // the runtime: 64 threads spent 0.28 s each waiting for theirs).  They read() file pieces into the slots of ONE pinned ring;
    // a single uploader issues the copies on one stream and recycles the slots.  The ring is pinned by a helper thread while
    // this one sizes and allocates the device buffers.
    // The regions' word buffer -- the largest allocation of a build, ~10 bytes per base -- is asked for now, on a thread of its own, and put back
    // into the allocator's cache, where dictset_build_device finds it: as a box's first GPU process the allocation waits ~1 s for memory the
    // driver hands out for the first time (build.dictionaries 0.98 s of a 2.46 s `ska build`, profiles/r06f_bench_full.json), and that second
    // can pass beside the reading of the files.  The size is the one dictset_build_device will compute if the longest sample is as long as
    // the largest plain file says (headers and line ends make it ~2 % more: the cache hands out a block up to a quarter larger than asked).
    std::thread warm_thread;
    struct JoinWarm { std::thread &t; ~JoinWarm() { if (t.joinable()) t.join(); } } join_warm{warm_thread};
    if (device_parse && !any_pair && n >= 8 && !knob("no_prewarm")) {
        uint64_t maxlen = 0; bool plain = true;
        for (int i = 0; i < n && plain; i++) { struct stat sb; if (stat(file1[i], &sb) != 0 || !S_ISREG(sb.st_mode)) plain = false; else maxlen = std::max<uint64_t>(maxlen, (uint64_t)sb.st_size); }
        for (int i = 0; i < n && plain; i++) { const size_t L = strlen(file1[i]); if (L > 3 && (!strcmp(file1[i] + L - 3, ".gz") || !strcmp(file1[i] + L - 3, ".xz") || !strcmp(file1[i] + L - 4, ".bz2") || !strcmp(file1[i] + L - 4, ".zst"))) plain = false; }
        if (plain && maxlen > (1u << 20)) {
            const bool wide_w = k > 31;
            const uint64_t per_region = wide_w ? 3200 : 4900;
            const int need = std::max(0, ilog2_ceil((maxlen + per_region - 1) / per_region)), base_logB = wide_w ? 11 : 10;
            const int logB_w = std::min({need > base_logB ? std::max(base_logB, need - 5) : need, 2 * (k - 1), MAX_LOGB});
            const uint64_t mean = (maxlen >> logB_w) + 1, cap_w = ((mean + mean / 5 + 256) + 63) / 64 * 64;
            const uint64_t bytes = (((uint64_t)n << logB_w) * cap_w * (wide_w ? 2 : 1) + 2048) * 8;
            size_t free_b = 0, total_b = 0;
            (void)hipSetDevice(ctx->device);
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes < free_b / 2) {
                const int dev = ctx->device;
                warm_thread = std::thread([bytes, dev]() { (void)hipSetDevice(dev); hipError_t e = hipSuccess; if (void *p = dev_alloc(bytes, &e)) dev_free(p); });
            }
        }
    }
    constexpr size_t SLOT = 8u << 20;
    // (a thread streaming a FASTQ sample fills a sequence and a quality slot at a time: two per thread and a few in flight)
    const int n_slots = any_pair ? 2 * nt + 8 : std::max(4, std::min(2 * nt, 32));
    struct Ring {
        uint8_t *base = nullptr; std::mutex mu; std::condition_variable cv_free, cv_work;
        std::vector<int> free_slots; struct Req { int slot; uint8_t *dst; size_t bytes; }; std::deque<Req> work; int readers_left = 0; bool failed = false;
        ~Ring() { if (base) (void)hipHostFree(base); }
    } ring;
    std::thread pin_thread;
    std::atomic<int> pin_rc{-1};
    if (device_parse) pin_thread = std::thread([&]() {
        (void)hipSetDevice(ctx->device);
        pin_rc = hipHostMalloc((void **)&ring.base, (size_t)n_slots * SLOT, hipHostMallocDefault) == hipSuccess ? 0 : 1;
    });
    struct JoinPin { std::thread &t; ~JoinPin() { if (t.joinable()) t.join(); } } join_pin{pin_thread};
    // one device buffer for all raw texts and one for all record streams (a slot per single-file sample, sized from stat):
    // two allocations whatever the number of samples
    DevBuf<uint8_t> raw_all, out_all, hs_seq_all, hs_qual_all;
    std::vector<uint64_t> hs_off(n, 0), hs_len(n, 0);
    std::vector<uint8_t> hs_fq(n, 0);
    if (device_parse) {
        uint64_t tot = 0;
        for (int i = 0; i < n; i++) {
            struct stat sb; unsigned char c0 = 0;
            if (file2 && file2[i]) continue;
            // only plain FASTA text is parsed on the device: a FASTQ or gzip sample reserves nothing here (raw_upload would refuse it and
            // its two slots would stay allocated, uncounted by the batch planner)
            const int fd = ::open(file1[i], O_RDONLY);
            const bool fasta = fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 1 && ::read(fd, &c0, 1) == 1 && c0 == '>';
            if (fd >= 0) ::close(fd);
            if (!fasta) continue;
            slot_off[i] = tot; slot_len[i] = ((uint64_t)sb.st_size + 64 + 255) & ~255ull;
            tot += slot_len[i];
        }
        if (tot) { SKX_HIP(hipSetDevice(ctx->device)); SKX_TRY(raw_all.alloc(tot)); SKX_TRY(out_all.alloc(tot)); }
        // the samples the host reader will parse (two files, FASTQ): one buffer for all their record streams and one for the quality
        // streams, a slot each bounded from the file sizes -- plain FASTQ holds at most half its bytes in either stream, plain FASTA
        // all of them; gzip (size unknown) keeps an allocation of its own.  No device allocation per sample from the reader threads
        // (64 of them for 32 isolates queued behind one another: most of 5 s).
        uint64_t htot = 0; bool any_q = false;
        for (int i = 0; i < n; i++) {
            if (slot_len[i]) continue;
            uint64_t bytes = 0; bool fq = false, ok = true;
            for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
                if (!f) continue;
                struct stat sb; unsigned char c0 = 0;
                const int fd = ::open(f, O_RDONLY);
                if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || ::read(fd, &c0, 1) != 1 || (c0 != '@' && c0 != '>')) ok = false;
                else { bytes += (uint64_t)sb.st_size; fq |= c0 == '@'; }
                if (fd >= 0) ::close(fd);
            }
            if (!ok || !bytes) continue;
            hs_off[i] = htot; hs_len[i] = ((fq ? bytes / 2 : bytes) + 64 + 255) & ~255ull; hs_fq[i] = fq ? 1 : 0;
            htot += hs_len[i]; any_q |= fq;
        }
        if (htot) { SKX_HIP(hipSetDevice(ctx->device)); SKX_TRY(hs_seq_all.alloc(htot)); if (any_q) SKX_TRY(hs_qual_all.alloc(htot)); }
        pin_thread.join();
        phase_add("build.alloc_text_pin_ring", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_read0).count());
        if (pin_rc != 0) { ring.base = nullptr; }                                 // no pinned memory: every sample takes the host reader
        else for (int b = 0; b < n_slots; b++) ring.free_slots.push_back(b);
    }
    const bool ring_ok = device_parse && ring.base;                            // the pinned ring + uploader threads carry raw files and host-parsed streams alike
    const bool raw_ok = ring_ok && raw_all.p;
    ring.readers_left = nt;
    // uploader: copies queued pieces on one stream, a batch at a time, and returns their slots
    std::vector<std::thread> uploaders;
    const int n_up = 2;                                                         // two streams keep both copy engines busy
    if (ring_ok) for (int u = 0; u < n_up; u++) uploaders.emplace_back([&]() {
        (void)hipSetDevice(ctx->device);
        hipStream_t up = nullptr;
        if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) up = nullptr;
        std::vector<Ring::Req> batch;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(ring.mu);
                ring.cv_work.wait(lk, [&] { return !ring.work.empty() || ring.readers_left == 0; });
                if (ring.work.empty() && ring.readers_left == 0) break;
                const size_t take = std::max<size_t>(1, ring.work.size() / 2);          // leave work for the other uploader
                batch.assign(ring.work.begin(), ring.work.begin() + (ptrdiff_t)take); ring.work.erase(ring.work.begin(), ring.work.begin() + (ptrdiff_t)take);
            }
            bool bad = false;
            for (auto &r : batch) bad |= hipMemcpyAsync(r.dst, ring.base + (size_t)r.slot * SLOT, r.bytes, hipMemcpyHostToDevice, up) != hipSuccess;
            bad |= hipStreamSynchronize(up) != hipSuccess;
            {
                std::lock_guard<std::mutex> lk(ring.mu);
                if (bad) ring.failed = true;
                for (auto &r : batch) ring.free_slots.push_back(r.slot);
            }
            ring.cv_free.notify_all();
        }
        if (up) (void)hipStreamDestroy(up);
    });
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&]() {
            struct Leave { Ring &r; ~Leave() { { std::lock_guard<std::mutex> lk(r.mu); r.readers_left--; } r.cv_work.notify_all(); } } leave{ring};
            hipStream_t up_st = nullptr;                                             // host-reader path only, created on first use
            struct Drop { hipStream_t &s; ~Drop() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } } drop{up_st};
            // raw upload of a plain FASTA file: SKX_OK (taken), SKF_NOT_TAKEN (use the host reader), or an error
            auto raw_upload = [&](int i) -> int {
                if (!slot_len[i]) return SKF_NOT_TAKEN;
                const int fd = ::open(file1[i], O_RDONLY);
                if (fd < 0) return SKF_NOT_TAKEN;                                    // the host reader reports it
                struct Close { int fd; ~Close() { ::close(fd); } } cl{fd};
                // the pages are read once: without this hint every first access promotes its page on the kernel's LRU lists, under one lock
                // for all reader threads (1 000 fresh 5 MB files on tmpfs: 0.47 s instead of 0.23 s for the same read() calls)
                (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
                const uint64_t cap = slot_len[i] - 64;                               // the size stat reported
                uint8_t *dst = raw_all.p + slot_off[i];
                uint64_t off = 0;
                while (off < cap) {
                    int slot;
                    {
                        std::unique_lock<std::mutex> lk(ring.mu);
                        ring.cv_free.wait(lk, [&] { return !ring.free_slots.empty() || ring.failed; });
                        if (ring.failed) { set_error("upload of the sequence files failed"); return SKX_ENODEV; }
                        slot = ring.free_slots.back(); ring.free_slots.pop_back();
                    }
                    uint8_t *buf = ring.base + (size_t)slot * SLOT;
                    const size_t want = (size_t)std::min<uint64_t>(SLOT, cap - off);
                    size_t got = 0;
                    while (got < want) { const ssize_t r = read(fd, buf + got, want - got); if (r < 0 && errno == EINTR) continue; if (r <= 0) break; got += (size_t)r; }
                    const bool not_fasta = off == 0 && got && buf[0] != '>';         // FASTQ ('@'), gzip (1f 8b), anything else: the host reader's
                    if (got == 0 || not_fasta) {
                        { std::lock_guard<std::mutex> lk(ring.mu); ring.free_slots.push_back(slot); }
                        ring.cv_free.notify_one();
                        if (not_fasta) return SKF_NOT_TAKEN;
                        break;                                                         // the file shrank under us: what was read is the file
                    }
                    { std::lock_guard<std::mutex> lk(ring.mu); ring.work.push_back({slot, dst + off, got}); }
                    ring.cv_work.notify_one();
                    off += got;
                }
                if (off == 0) return SKF_NOT_TAKEN;
                raw_len[i] = off; is_raw[i] = 1;
                return SKX_OK;
            };
            // a plain FASTQ sample: its files' lines go straight from a small read buffer into pinned slots -- one filling with sequence
            // lines, one with quality lines -- which the uploaders copy to the sample's places in the two stream buffers
            auto stream_fastq = [&](int i) -> int {
                struct Out { int slot = -1; size_t used = 0; uint8_t *dst = nullptr; uint64_t off = 0; } o[2];
                o[0].dst = hs_seq_all.p + hs_off[i]; o[1].dst = hs_qual_all.p + hs_off[i];
                const uint64_t cap = hs_len[i] - 32;
                auto flush = [&](Out &x) {
                    if (x.slot < 0) return;
                    { std::lock_guard<std::mutex> lk(ring.mu); ring.work.push_back({x.slot, x.dst + x.off, x.used}); }
                    ring.cv_work.notify_one();
                    x.off += x.used; x.slot = -1; x.used = 0;
                };
                auto give_back = [&]() {
                    for (auto &x : o) if (x.slot >= 0) { { std::lock_guard<std::mutex> lk(ring.mu); ring.free_slots.push_back(x.slot); } ring.cv_free.notify_one(); x.slot = -1; }
                };
                const std::function<int(int, const uint8_t *, size_t)> emit = [&](int which, const uint8_t *p, size_t nb) -> int {
                    Out &x = o[which];
                    if (x.off + x.used + nb + 1 > cap) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }      // (more sequence than half the file: not FASTQ)
                    if (x.slot >= 0 && x.used + nb + 1 <= SLOT) {                       // the common case: the line and its terminator fit the slot being filled
                        uint8_t *d = ring.base + (size_t)x.slot * SLOT + x.used;
                        memcpy(d, p, nb); d[nb] = '\n';
                        x.used += nb + 1;
                        if (x.used == SLOT) flush(x);
                        return SKX_OK;
                    }
                    bool term = false;                                                  // the line, then its '\n', across slot ends
                    static const uint8_t nl = '\n';
                    for (;;) {
                        if (nb == 0) { if (term) break; term = true; p = &nl; nb = 1; }
                        if (x.slot < 0) {
                            std::unique_lock<std::mutex> lk(ring.mu);
                            ring.cv_free.wait(lk, [&] { return !ring.free_slots.empty() || ring.failed; });
                            if (ring.failed) { set_error("upload of the sequence files failed"); return SKX_ENODEV; }
                            x.slot = ring.free_slots.back(); ring.free_slots.pop_back(); x.used = 0;
                        }
                        const size_t take = std::min(nb, SLOT - x.used);
                        memcpy(ring.base + (size_t)x.slot * SLOT + x.used, p, take);
                        x.used += take; p += take; nb -= take;
                        if (x.used == SLOT) flush(x);
                    }
                    return SKX_OK;
                };
                for (const char *f : {file1[i], file2 ? file2[i] : nullptr}) {
                    if (!f) continue;
                    int r = stream_fastq_file(f, emit);
                    if (r == SKF_NOT_TAKEN && f != file1[i]) { set_error("Invalid FASTA/Q record"); r = SKX_EIO; }      // file 2 is parsed in file 1's mode (ska_dict.rs:356-366)
                    if (r != SKX_OK) { give_back(); return r; }              // (SKF_NOT_TAKEN: the first file, before anything was emitted)
                }
                flush(o[0]); flush(o[1]);
                if (o[0].off != o[1].off) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
                ss[i].seq = o[0].dst; ss[i].qual = o[1].dst; ss[i].len = o[0].off;
                return SKX_OK;
            };
            HostStream h;                                                            // (its buffers live across this thread's samples: no fresh pages per sample)
            for (int i; (i = next.fetch_add(1)) < n;) {
                if (raw_ok) {
                    const int r = raw_upload(i);
                    if (r == SKX_OK) continue;
                    if (r != SKF_NOT_TAKEN) { rcodes[i] = r; errs[i] = skx_last_error(); continue; }
                }
                if (ring_ok && hs_len[i] && hs_fq[i] && hs_qual_all.p) {
                    const int r = stream_fastq(i);
                    if (r == SKX_OK) continue;
                    if (r != SKF_NOT_TAKEN) { rcodes[i] = r; errs[i] = skx_last_error(); continue; }
                }
                rcodes[i] = read_sample_stream(file1[i], file2 ? file2[i] : nullptr, proportion_reads, h);
                if (rcodes[i] != SKX_OK) { errs[i] = skx_last_error(); continue; }
                const size_t len = h.seq.size();
                // the parsed streams travel through the pinned ring like the raw files (a copy from pageable memory goes through the
                // runtime's one staging path: 32 reader threads shared ~3 GB/s, 5.6 s for 32 isolates); the uploads are complete when the
                // uploader threads have been joined, which is before anything reads them
                auto up = [&](DevBuf<uint8_t> &own, uint8_t *slot_ptr, const std::vector<uint8_t> &src, uint8_t **where) -> int {
                    struct { uint8_t *p; } dst{slot_ptr};
                    if (!dst.p) { (void)hipSetDevice(ctx->device); SKX_TRY(own.alloc(len + 16)); dst.p = own.p; }      // no slot (gzip, a stream longer than its bound)
                    *where = dst.p;
                    if (!ring_ok) {
                        (void)hipSetDevice(ctx->device);
                        if (!up_st && hipStreamCreateWithFlags(&up_st, hipStreamNonBlocking) != hipSuccess) up_st = nullptr;
                        if (len) { SKX_HIP(hipMemcpyAsync(dst.p, src.data(), len, hipMemcpyHostToDevice, up_st)); SKX_HIP(hipStreamSynchronize(up_st)); }
                        return SKX_OK;
                    }
                    for (size_t off = 0; off < len; off += SLOT) {
                        int slot;
                        {
                            std::unique_lock<std::mutex> lk(ring.mu);
                            ring.cv_free.wait(lk, [&] { return !ring.free_slots.empty() || ring.failed; });
                            if (ring.failed) { set_error("upload of the sequence files failed"); return SKX_ENODEV; }
                            slot = ring.free_slots.back(); ring.free_slots.pop_back();
                        }
                        const size_t nb = std::min<size_t>(SLOT, len - off);
                        memcpy(ring.base + (size_t)slot * SLOT, src.data() + off, nb);
                        { std::lock_guard<std::mutex> lk(ring.mu); ring.work.push_back({slot, dst.p + off, nb}); }
                        ring.cv_work.notify_one();
                    }
                    return SKX_OK;
                };
                const bool fits = hs_len[i] && len + 16 <= hs_len[i] && (!h.is_fastq || hs_qual_all.p);
                uint8_t *at_seq = nullptr, *at_qual = nullptr;
                rcodes[i] = up(d_seq[i], fits ? hs_seq_all.p + hs_off[i] : nullptr, h.seq, &at_seq);
                if (rcodes[i] == SKX_OK && h.is_fastq) rcodes[i] = up(d_qual[i], fits ? hs_qual_all.p + hs_off[i] : nullptr, h.qual, &at_qual);
                if (rcodes[i] != SKX_OK) { errs[i] = skx_last_error(); continue; }
                ss[i].seq = at_seq; ss[i].qual = h.is_fastq ? at_qual : nullptr; ss[i].len = len;
            }
        });
    for (auto &th : pool) th.join();
    for (auto &u : uploaders) u.join();
    phase_add("build.read_upload", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_read0).count());
    if (ring.failed) { set_error("upload of the sequence files failed"); return SKX_ENODEV; }
    for (int i = 0; i < n; i++) if (rcodes[i] != SKX_OK) { set_error("%s", errs[i].c_str()); return rcodes[i]; }
    // the raw FASTA texts -> record streams, all files in one set of launches
    {
        PhaseTimer t_parse("build.device_fasta_parse");
        SKX_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        std::vector<int> idx;
        for (int i = 0; i < n; i++) if (is_raw[i]) idx.push_back(i);
        const int m = (int)idx.size();
        if (m) {
            std::vector<const uint8_t *> h_raw(m); std::vector<uint8_t *> h_out(m); std::vector<uint64_t> h_len(m), h_base(m + 1, 0);
            for (int j = 0; j < m; j++) {
                const int i = idx[j];
                h_raw[j] = raw_all.p + slot_off[i]; h_out[j] = out_all.p + slot_off[i]; h_len[j] = raw_len[i];
                h_base[j + 1] = h_base[j] + fasta_parse_tiles(raw_len[i]);
            }
            const uint64_t tiles = h_base[m];
            std::vector<uint32_t> h_tf(tiles);
            for (int j = 0; j < m; j++) std::fill(h_tf.begin() + (ptrdiff_t)h_base[j], h_tf.begin() + (ptrdiff_t)h_base[j + 1], (uint32_t)j);
            DevBuf<const uint8_t *> g_raw; DevBuf<uint8_t *> g_out; DevBuf<uint64_t> g_len, g_outlen, g_base, g_off, g_sum; DevBuf<uint32_t> g_tf; DevBuf<uint8_t> g_kind;
            SKX_TRY(g_raw.alloc(m)); SKX_TRY(g_out.alloc(m)); SKX_TRY(g_len.alloc(m)); SKX_TRY(g_outlen.alloc(m)); SKX_TRY(g_base.alloc(m + 1));
            SKX_TRY(g_off.alloc(tiles)); SKX_TRY(g_sum.alloc(tiles)); SKX_TRY(g_tf.alloc(tiles)); SKX_TRY(g_kind.alloc(tiles));
            SKX_HIP(hipMemcpyAsync(g_raw.p, h_raw.data(), m * sizeof(void *), hipMemcpyHostToDevice, st));
            SKX_HIP(hipMemcpyAsync(g_out.p, h_out.data(), m * sizeof(void *), hipMemcpyHostToDevice, st));
            SKX_HIP(hipMemcpyAsync(g_len.p, h_len.data(), m * 8, hipMemcpyHostToDevice, st));
            SKX_HIP(hipMemcpyAsync(g_base.p, h_base.data(), (m + 1) * 8, hipMemcpyHostToDevice, st));
            if (tiles) SKX_HIP(hipMemcpyAsync(g_tf.p, h_tf.data(), tiles * 4, hipMemcpyHostToDevice, st));
            launch_fasta_parse(g_raw.p, g_len.p, g_out.p, g_outlen.p, g_tf.p, g_base.p, tiles, g_sum.p, g_off.p, g_kind.p, m, st);
            std::vector<uint64_t> h_outlen(m);
            SKX_HIP(hipMemcpyAsync(h_outlen.data(), g_outlen.p, m * 8, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            SKX_HIP(hipGetLastError());
            for (int j = 0; j < m; j++) { const int i = idx[j]; ss[i].seq = out_all.p + slot_off[i]; ss[i].qual = nullptr; ss[i].len = h_outlen[j]; }
        }
        raw_all.release();
    }
    skx_dictset *d = nullptr;
    if (warm_thread.joinable()) { PhaseTimer t_w("build.wait_for_word_buffer"); warm_thread.join(); }
    const auto t_dev0 = std::chrono::steady_clock::now();
    int r = skx_dictset_build(ctx, ss.data(), n, 1, k, rc, q, &d);
    phase_add("build.dictionaries", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev0).count());
    if (r == SKX_EEMPTY) {      // "{file} has no valid sequence" (ska_dict.rs:374-376)
        int bad = 0; sscanf(skx_last_error(), "sample %d", &bad);
        set_error("%s has no valid sequence", file1[bad]);
    }
    if (r != SKX_OK) return r;
    *out = d;
    return SKX_OK;
    });
}

extern "C" int skx_read_records(const char *file1, const char *file2, double proportion_reads, int streaming, uint8_t **seq, uint8_t **qual, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    if (!file1 || !seq || !qual || !len) { set_error("bad arguments"); return SKX_EINVAL; }
    *seq = nullptr; *qual = nullptr; *len = 0;
    HostStream h;
    if (streaming) {
        const std::function<int(int, const uint8_t *, size_t)> emit = [&](int which, const uint8_t *p, size_t nb) -> int {
            std::vector<uint8_t> &v = which ? h.qual : h.seq;
            v.insert(v.end(), p, p + nb); v.push_back('\n');
            return SKX_OK;
        };
        for (const char *f : {file1, file2}) {
            if (!f) continue;
            const int r = stream_fastq_file(f, emit);
            if (r == SKF_NOT_TAKEN) { set_error("not a FASTQ file: %s", f); return SKX_EUNSUP; }
            if (r != SKX_OK) return r;
        }
        h.is_fastq = true;
    } else SKX_TRY(read_sample_stream(file1, file2, proportion_reads, h));
    if (h.is_fastq && h.qual.size() != h.seq.size()) { set_error("Invalid FASTA/Q record"); return SKX_EIO; }
    const size_t n = h.seq.size();
    uint8_t *s = (uint8_t *)malloc(n + 1), *q = h.is_fastq ? (uint8_t *)malloc(n + 1) : nullptr;
    if (!s || (h.is_fastq && !q)) { free(s); free(q); set_error("out of memory"); return SKX_ENOMEM; }
    memcpy(s, h.seq.data(), n);
    if (q) memcpy(q, h.qual.data(), n);
    *seq = s; *qual = q; *len = n;
    return SKX_OK;
    });
}

extern "C" int skx_dictset_size(skx_dictset *d, int sample, uint64_t *n)
{
    return skx_guarded([&]() -> int {
    if (!d || sample < 0 || sample >= d->n) { set_error("bad sample index"); return SKX_EINVAL; }
    SKX_TRY(dictset_sort(d));
    *n = d->sample_size[sample];
    return SKX_OK;
    });
}

extern "C" int skx_dictset_export(skx_dictset *d, int sample, skx_key *keys, uint8_t *bases, uint64_t cap)
{
    return skx_guarded([&]() -> int {
    if (!d || sample < 0 || sample >= d->n) { set_error("bad sample index"); return SKX_EINVAL; }
    skx_ctx *ctx = d->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(dictset_sort(d));
    const uint64_t B = 1ull << d->logB, sz = d->sample_size[sample];
    if (cap < sz) { set_error("buffer too small"); return SKX_EINVAL; }
    std::vector<uint64_t> off(B + 1); std::vector<uint32_t> uc(B);
    SKX_HIP(hipMemcpyAsync(off.data(), d->off.p + ((uint64_t)sample << d->logB), (B + 1) * 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(uc.data(), d->ucnt.p + ((uint64_t)sample << d->logB), B * 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    std::vector<skx_key> hk(sz); std::vector<uint8_t> hb(sz);
    static const char M2I[17] = "-ACMTWYHGRSVKDBN";
    const int wpk = d->wide() ? 2 : 1;
    {
        std::vector<uint64_t> raw((size_t)sz * wpk);
        uint64_t w = 0;
        for (uint64_t b = 0; b < B; b++) {
            if (uc[b]) SKX_HIP(hipMemcpyAsync(raw.data() + w * wpk, d->words.p + off[b] * wpk, (size_t)uc[b] * 8 * wpk, hipMemcpyDeviceToHost, st));
            w += uc[b];
        }
        SKX_HIP(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < sz; i++) {
            if (d->wide()) {
                const u128 word = ((u128)raw[2 * i + 1] << 64) | raw[2 * i];
                const u128 key = hunmix_w(word >> 4, d->wh);
                hk[i].lo = (uint64_t)key; hk[i].hi = (uint64_t)(key >> 64); hb[i] = (uint8_t)M2I[(unsigned)word & 15u];
            } else {
                hk[i].lo = hunmix(raw[i] >> 4, d->hp); hk[i].hi = 0; hb[i] = (uint8_t)M2I[raw[i] & 15u];
            }
        }
    }
    std::vector<uint64_t> idx(sz); std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b2) { return hk[a].hi != hk[b2].hi ? hk[a].hi < hk[b2].hi : hk[a].lo < hk[b2].lo; });
    for (uint64_t i = 0; i < sz; i++) { if (keys) keys[i] = hk[idx[i]]; if (bases) bases[i] = hb[idx[i]]; }
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ keyset
extern "C" void skx_keyset_free(skx_keyset *ks) { delete ks; }
extern "C" int skx_keyset_size(const skx_keyset *ks, uint64_t *n) { *n = ks->total; return SKX_OK; }

static int keyset_finish(skx_keyset *ks)      // scan ncnt -> roff, total, max_rows
{
    hipStream_t st = ks->ctx->stream;
    const uint64_t nsub = 1ull << ks->logN;
    SKX_TRY(ks->roff.alloc(nsub + 1));
    DevBuf<uint32_t> d_max; SKX_TRY(d_max.alloc(1));
    launch_scan_u32(ks->ncnt.p, ks->roff.p, nsub, d_max.p, st);
    SKX_HIP(hipMemcpyAsync(&ks->total, ks->roff.p + nsub, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(&ks->max_rows, d_max.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] keyset: logN=%d sub-buckets=%llu rows=%llu mean=%.1f max=%u\n", ks->logN, (unsigned long long)nsub,
                                 (unsigned long long)ks->total, (double)ks->total / (double)nsub, ks->max_rows);
    return SKX_OK;
}

// union over several dict views (one per source) into one keyset
static int keyset_union_views(skx_ctx *ctx, const DictView *views, int nviews, int k, int rc, HashParams hp, uint64_t est_hint, skx_keyset **out,
                              const skx_dictset *side_for = nullptr)
{
    const bool wide = k > 31;
    const WideHash wh = make_wide_hash(k);
    const int kbits = 2 * (k - 1);
    hipStream_t st = ctx->stream;
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1));
    int min_logN = 0;
    for (int v = 0; v < nviews; v++) min_logN = std::max(min_logN, views[v].logB);
    const uint32_t table = wide ? 4096 : 8192, stride = wide ? 2048 : 4096, target = wide ? 1200 : 2500;
    int logN = std::max(min_logN, std::min(kbits, ilog2_ceil((est_hint + target - 1) / target)));
    DevBuf<uint16_t> side_buf;                     // kept across retries at a finer split
    for (;; logN++) {
        std::unique_ptr<skx_keyset> ks(new skx_keyset());
        ks->ctx = ctx; ks->k = k; ks->rc = rc; ks->logN = logN; ks->hp = hp; ks->wh = wh; ks->wide = wide; ks->stride = stride;
        const uint64_t nsub = 1ull << logN;
        SKX_TRY(ks->stage.alloc(nsub * stride * ks->wpk())); SKX_TRY(ks->ncnt.alloc(nsub));
        SKX_TRY(d_flag.zero(st));
        if (nviews == 1) {
            // side_for: the assemble over these dictionaries follows (skx_merge): the pass also notes where every word's key went, 2 bytes
            // per word, and the matrix is then filled from the notes instead of a second read of the dictionaries
            const bool with_side = side_for && (wide || union_side_ok(views[0], logN, stride));
            if (wide && with_side) {
                if (!side_buf.p) SKX_TRY(side_buf.alloc(side_for->words.n / 2 + 64));          // one note per 16-byte word
                SKX_TRY(ks->perm.alloc(nsub * stride));
                launch_union_side_wide(views[0], logN, (u128 *)ks->stage.p, stride, ks->ncnt.p, table, d_flag.p, side_buf.p, ks->perm.p, st);
                ks->side_of = side_for;
            } else if (wide) launch_union_wide(views[0], logN, (u128 *)ks->stage.p, stride, ks->ncnt.p, table, d_flag.p, st);
            else if (with_side) {
                if (!side_buf.p) SKX_TRY(side_buf.alloc(side_for->words.n));
                SKX_TRY(ks->perm.alloc(nsub * stride));
                launch_union_side(views[0], logN, ks->stage.p, stride, ks->ncnt.p, 7168u, d_flag.p, side_buf.p, ks->perm.p, st);
                ks->side_of = side_for;
            } else launch_union(views[0], logN, ks->stage.p, stride, ks->ncnt.p, table, d_flag.p, st);
        } else {
            set_error("multi-source union goes through keyset_merge"); return SKX_EINVAL;
        }
        int overflow = 0;
        SKX_HIP(hipMemcpyAsync(&overflow, d_flag.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        SKX_HIP(hipGetLastError());
        if (overflow) {
            if (logN >= kbits) { set_error("key union overflow"); return SKX_EUNSUP; }
            continue;
        }
        SKX_TRY(keyset_finish(ks.get()));
        if (ks->side_of) ks->side = std::move(side_buf);
        *out = ks.release();
        return SKX_OK;
    }
}

static int keyset_union_dict(skx_ctx *ctx, skx_dictset *d, skx_keyset **out, bool with_side);
static int append_pass(skx_ctx *ctx, skx_dictset *d, std::unique_ptr<skx_keyset> &ks_out, std::unique_ptr<skx_pieces> &pc_out);
static int array_over_pieces(skx_ctx *ctx, const skx_dictset *d, skx_keyset *ks, skx_pieces *pc_, skx_keyset *g, const char *const *names, skx_array **out);
extern "C" int skx_keyset_union(skx_ctx *ctx, skx_dictset *d, skx_keyset **out)
{
    return skx_guarded([&]() -> int { return keyset_union_dict(ctx, d, out, false); });
}
extern "C" int skx_keyset_union_notes(skx_ctx *ctx, skx_dictset *d, skx_keyset **out)
{
    // (a sharded job hands this key set to skx_keyset_allgather, which carries the notes over to the global rows)
    return skx_guarded([&]() -> int { return keyset_union_dict(ctx, d, out, true); });
}
static int keyset_union_dict(skx_ctx *ctx, skx_dictset *d, skx_keyset **out, bool with_side)
{
    {
    if (!ctx || !d || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (with_side && !d->sorted && !d->wide()) {
        // assemblies as the extraction kernel left them: the append pass finds the rank's rows AND leaves the samples' cells as pieces, which
        // travel with the key set (to the global rows of a sharded job: skx_keyset_allgather) like the notes of the union pass
        // (128-bit keys: the exchange composes 64-bit rows only -- skx_comm.hip -- so a sharded job's ranks sort their dictionaries)
        std::unique_ptr<skx_keyset> ks; std::unique_ptr<skx_pieces> pc;
        const int r = append_pass(ctx, d, ks, pc);
        if (r == SKX_OK) { ks->pieces = pc.release(); ks->pieces_of = d; ks->pieces_of_id = d->id; *out = ks.release(); return SKX_OK; }
        if (r != SKF_NOT_TAKEN) return r;
    }
    if (!d->sorted && d->wide() && with_side) ctx->merge_path = "sorted: the key-table exchange of a sharded job composes 64-bit rows only";
    else if (!d->sorted && !with_side) ctx->merge_path = "sorted: a row set without cells was asked for (skx_keyset_union)";
    SKX_TRY(dictset_sort(d));                        // the union kernels read sorted slices
    StageTimer t(ctx, &ctx->tm.key_union);
    DictView v = d->view();
    // estimate |U| from a thin slice of the hash space (probe sub-buckets of a 2^logP split)
    uint64_t maxs = 0, sum = 0;
    for (auto s : d->sample_size) { maxs = std::max(maxs, s); sum += s; }
    uint64_t est = maxs;
    if (d->n > 1) {
        const int logP = std::min(2 * (d->k - 1), std::max(d->logB, ilog2_ceil((sum + 1023) / 1024)));
        const int probe = (int)std::min<uint64_t>(64, 1ull << logP);
        DevBuf<uint32_t> d_cnt; DevBuf<int> d_flag;
        SKX_TRY(d_cnt.alloc(1)); SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_cnt.zero(st)); SKX_TRY(d_flag.zero(st));
        if (d->wide()) launch_union_probe_wide(v, logP, probe, d_cnt.p, 4096, d_flag.p, st);
        else launch_union_probe(v, logP, probe, d_cnt.p, 8192, d_flag.p, st);
        uint32_t cnt = 0; int ov = 0;
        SKX_HIP(hipMemcpyAsync(&cnt, d_cnt.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        if (!ov) est = std::max<uint64_t>(maxs, (uint64_t)((double)cnt * (double)(1ull << logP) / probe * 1.1));
        else est = sum;
    }
    return keyset_union_views(ctx, &v, 1, d->k, d->rc, d->hp, est, out, with_side ? d : nullptr);
    }
}

int skx::keyset_flatten(skx_keyset *ks)
{
    if (ks->flat.p) return SKX_OK;
    SKX_TRY(ks->flat.alloc(ks->total * ks->wpk()));
    if (ks->wide) launch_gather_keys_wide((const u128 *)ks->stage.p, ks->stride, ks->ncnt.p, ks->roff.p, 1 << ks->logN, (u128 *)ks->flat.p, ks->ctx->stream);
    else launch_gather_keys(ks->stage.p, ks->stride, ks->ncnt.p, ks->roff.p, 1 << ks->logN, ks->flat.p, 0, ks->hp, ks->ctx->stream);
    return SKX_OK;
}

extern "C" int skx_keyset_device(skx_keyset *ks, const void **dptr, uint64_t *n_keys, int *words_per_key)
{
    return skx_guarded([&]() -> int {
    SKX_HIP(hipSetDevice(ks->ctx->device));
    SKX_TRY(keyset_flatten(ks));
    SKX_HIP(hipStreamSynchronize(ks->ctx->stream));
    *dptr = ks->flat.p; *n_keys = ks->total; if (words_per_key) *words_per_key = ks->wpk();
    return SKX_OK;
    });
}

// a flat sorted word list viewed as a one-sample, one-bucket dict
static int keyset_from_flat(skx_ctx *ctx, DevBuf<uint64_t> &&flat, uint64_t n, int k, int rc, skx_keyset **out)
{
    std::unique_ptr<skx_keyset> ks(new skx_keyset());
    ks->ctx = ctx; ks->k = k; ks->rc = rc; ks->hp = make_hash_params(std::min(k, 31)); ks->wh = make_wide_hash(k); ks->wide = k > 31; ks->logN = -1; ks->total = n;
    ks->flat = std::move(flat);
    *out = ks.release();
    return SKX_OK;
}

extern "C" int skx_keyset_from_device(skx_ctx *ctx, const void *dptr, uint64_t n_keys, int k, int rc, skx_keyset **out)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check_k(k));
    SKX_HIP(hipSetDevice(ctx->device));
    const int wpk = k > 31 ? 2 : 1;
    DevBuf<uint64_t> flat; SKX_TRY(flat.alloc(n_keys * wpk));
    SKX_HIP(hipMemcpyAsync(flat.p, dptr, n_keys * 8 * wpk, hipMemcpyDeviceToDevice, ctx->stream));
    SKX_HIP(hipStreamSynchronize(ctx->stream));          // the caller may free or reuse dptr as soon as this returns
    return keyset_from_flat(ctx, std::move(flat), n_keys, k, rc, out);
    });
}

// union of several sorted key tables lying in one device buffer: table i = n[i] keys at words + off[i] * wpk (off in keys); each
// table is a "sample" of a synthetic one-bucket dict
int skx::keyset_union_tables(skx_ctx *ctx, const uint64_t *words, const std::vector<uint64_t> &h_off, const std::vector<uint32_t> &h_cnt, int k, int rc,
                             skx_keyset **out)
{
    hipStream_t st = ctx->stream;
    const int n_sets = (int)h_cnt.size();
    StageTimer t(ctx, &ctx->tm.key_union);
    uint64_t total = 0, maxn = 0;
    for (int i = 0; i < n_sets; i++) { total += h_cnt[i]; maxn = std::max<uint64_t>(maxn, h_cnt[i]); }
    DevBuf<uint64_t> off; DevBuf<uint32_t> ucnt;
    SKX_TRY(off.alloc(n_sets + 1)); SKX_TRY(ucnt.alloc(n_sets));
    SKX_HIP(hipMemcpyAsync(off.p, h_off.data(), (n_sets + 1) * 8, hipMemcpyHostToDevice, st));
    SKX_HIP(hipMemcpyAsync(ucnt.p, h_cnt.data(), n_sets * 4, hipMemcpyHostToDevice, st));
    DictView v{words, off.p, ucnt.p, n_sets, 0, 2 * (k - 1)};
    int r = keyset_union_views(ctx, &v, 1, k, rc, make_hash_params(std::min(k, 31)), std::min(total, maxn * 2), out);
    SKX_HIP(hipStreamSynchronize(st));
    return r;
}

extern "C" int skx_keyset_merge(skx_ctx *ctx, skx_keyset *const *sets, int n_sets, skx_keyset **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !sets || n_sets <= 0 || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint64_t total = 0;
    for (int i = 0; i < n_sets; i++) {
        if (sets[i]->k != sets[0]->k) { set_error("K-mer lengths do not match: %d %d", sets[i]->k, sets[0]->k); return SKX_EINVAL; }
        if (sets[i]->rc != sets[0]->rc) { set_error("Strand use inconsistent"); return SKX_EINVAL; }
        if (sets[i]->logN >= 0) SKX_TRY(keyset_flatten(sets[i]));
        total += sets[i]->total;
    }
    DevBuf<uint64_t> words;
    const int wpk = sets[0]->wpk();
    SKX_TRY(words.alloc(total * wpk));
    std::vector<uint64_t> h_off(n_sets + 1, 0); std::vector<uint32_t> h_cnt(n_sets);
    for (int i = 0; i < n_sets; i++) {
        if (sets[i]->total > 0xFFFFFFFFull) { set_error("keyset too large"); return SKX_EUNSUP; }
        h_cnt[i] = (uint32_t)sets[i]->total; h_off[i + 1] = h_off[i] + sets[i]->total;
        SKX_HIP(hipMemcpyAsync(words.p + h_off[i] * wpk, sets[i]->flat.p, sets[i]->total * 8 * wpk, hipMemcpyDeviceToDevice, st));
    }
    return keyset_union_tables(ctx, words.p, h_off, h_cnt, sets[0]->k, sets[0]->rc, out);
    });
}

// ------------------------------------------------------------------------------------------ array
extern "C" void skx_array_free(skx_array *a) { delete a; }
extern "C" const char *skx_array_name(const skx_array *a, uint64_t i) { return i < a->names.size() ? a->names[i].c_str() : ""; }
extern "C" const char *skx_array_version(const skx_array *a) { return a->version.c_str(); }
extern "C" int skx_array_info(const skx_array *a, skx_array_info_t *info)
{
    return skx_guarded([&]() -> int {
    info->k = a->k; info->rc = a->rc; info->k_bits = a->k_bits; info->n_kmers = a->n_kmers; info->n_rows = a->n_rows;
    info->n_samples = a->names.size();
    info->total_samples = a->total_samples ? a->total_samples : a->names.size();
    return SKX_OK;
    });
}

extern "C" int skx_array_assemble(skx_ctx *ctx, skx_dictset *d, skx_keyset *rows, const char *const *names, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !d || !rows || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    if (rows->k != d->k) { set_error("K-mer lengths do not match: %d %d", d->k, rows->k); return SKX_EINVAL; }      // merge_ska_dict.rs:78-84
    if (rows->rc != d->rc) { set_error("Strand use inconsistent"); return SKX_EINVAL; }                              // :85-87
    if (d->n > 65535) { set_error("more than 65535 samples per device array"); return SKX_EUNSUP; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (rows->holds_pieces_of(d) && rows->logN >= 0) {
        // the rows came from an append pass over these samples (skx_keyset_union_notes), directly or through the key-table exchange of a sharded
        // job: the cells are there already, as pieces
        const bool global = rows->g_perm.p != nullptr;
        if (global || rows->logN == rows->pieces->logQ) {
            skx_pieces *pc = rows->pieces; rows->pieces = nullptr; rows->pieces_of = nullptr;
            return array_over_pieces(ctx, d, rows, pc, global ? rows : nullptr, names, out);
        }
    }
    SKX_TRY(dictset_sort(d));
    std::unique_ptr<skx_keyset> rebuilt;
    if (rows->logN < 0 || rows->logN < d->logB) {      // flat / too coarse: re-slab at a compatible granularity
        skx_keyset *one[1] = {rows}; skx_keyset *tmp = nullptr;
        if (rows->logN >= 0) SKX_TRY(keyset_flatten(rows));
        // reuse the merge path with a minimum logN of the dict's logB
        DevBuf<uint64_t> off; DevBuf<uint32_t> ucnt;
        SKX_TRY(off.alloc(2)); SKX_TRY(ucnt.alloc(1));
        uint64_t h_off[2] = {0, rows->total}; uint32_t h_cnt = (uint32_t)rows->total;
        SKX_HIP(hipMemcpyAsync(off.p, h_off, 16, hipMemcpyHostToDevice, st));
        SKX_HIP(hipMemcpyAsync(ucnt.p, &h_cnt, 4, hipMemcpyHostToDevice, st));
        DictView v{rows->flat.p, off.p, ucnt.p, 1, 0, 2 * (rows->k - 1)};
        // force logN >= d->logB by passing a view whose logB is the dict's (single bucket list still valid for logB 0 only)
        (void)one;
        uint64_t hint = std::max<uint64_t>(rows->total, (uint64_t)2500 << d->logB);
        SKX_TRY(keyset_union_views(ctx, &v, 1, rows->k, rows->rc, rows->hp, hint, &tmp));
        SKX_HIP(hipStreamSynchronize(st));
        rebuilt.reset(tmp);
        rows = tmp;
        if (rows->logN < d->logB) { set_error("internal: keyset granularity"); return SKX_EUNSUP; }
    }
    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx; a->k = d->k; a->rc = d->rc; a->k_bits = d->key_bits; a->hp = d->hp; a->wh = d->wh; a->version = skx_version();
    for (int i = 0; i < d->n; i++) a->names.emplace_back(names && names[i] ? names[i] : "");
    const uint64_t U = rows->total;
    a->n_rows = a->n_kmers = U; a->pitch = pitch_for(U); a->engine_order = true;
    SKX_TRY(a->matrix.alloc((uint64_t)d->n * a->pitch));
    SKX_TRY(a->present.alloc(U)); SKX_TRY(a->unambig.alloc(U)); SKX_TRY(a->mask.alloc(U)); SKX_TRY(a->keys.alloc(U * rows->wpk())); SKX_TRY(a->vcount.alloc(U));
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
    if (U) {
        AssembleArgs aa{};
        aa.d = d->view(); aa.logN = rows->logN; aa.stage = rows->stage.p; aa.stride = rows->stride; aa.ncnt = rows->ncnt.p; aa.roff = rows->roff.p;
        aa.matrix = a->matrix.p; aa.pitch = a->pitch; aa.col_present = a->present.p; aa.col_unambig = a->unambig.p; aa.col_mask = a->mask.p;
        aa.max_rows = rows->max_rows; aa.missing = d_flag.p;
        {
            StageTimer t(ctx, &ctx->tm.assemble);
            if (rows->g_perm.p && rows->side.p && rows->side_of == d && !rebuilt && !rows->wide) {
                // a sharded job: the notes were taken against this rank's own rows and composed with the global rows (skx_keyset_allgather):
                // workgroup j = own sub-bucket j, over the global rows of its hash range
                AssembleArgs ag = aa;
                ag.logN = rows->l_logN; ag.stride = rows->l_stride; ag.ncnt = rows->g_n.p; ag.roff = rows->g_base.p; ag.max_rows = rows->g_max;
                launch_assemble_side(ag, rows->side.p, rows->g_perm.p, st, false);
            }
            else if (rows->side.p && !rows->g_perm.p && rows->side_of == d && !rebuilt) launch_assemble_side(aa, rows->side.p, rows->perm.p, st, rows->wide);      // the union over d left its notes
            else if (rows->wide) launch_assemble_wide(aa, st);
            else launch_assemble(aa, st);
        }
        if (rows->wide) launch_gather_keys_wide((const u128 *)rows->stage.p, rows->stride, rows->ncnt.p, rows->roff.p, 1 << rows->logN, (u128 *)a->keys.p, st);
        else launch_gather_keys(rows->stage.p, rows->stride, rows->ncnt.p, rows->roff.p, 1 << rows->logN, a->keys.p, 0, rows->hp, st);
        SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, U * 4, hipMemcpyDeviceToDevice, st));     // merge_ska_array.rs:172
    }
    int missing = 0;
    SKX_HIP(hipMemcpyAsync(&missing, d_flag.p, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    if (missing) { set_error("row keyset does not contain every split k-mer of the samples"); return SKX_EINVAL; }
    *out = a.release();
    return SKX_OK;
    });
}

// the pieces of an array merged by the append pass (skx_append.hip), as the kernel that turns them into rows x samples cells takes them
static PiecesRowsArgs pieces_args(skx_array *a)
{
    PiecesRowsArgs pa{};
    const skx_pieces *pc = a->pieces; const skx_keyset *rows = a->lazy_rows;
    pa.pieces = pc->data.p; pa.plen = pc->plen.p; pa.perm = pc->perm.p; pa.nrank = pc->nrank.p; pa.cap = pc->cap; pa.n_samples = (int)a->names.size();
    pa.ncnt = rows->ncnt.p; pa.roff = rows->roff.p;
    return pa;
}
// ---- lazily held arrays -------------------------------------------------------------------------------------------------
// the assemble kernel of the array's key width
static void launch_assemble_lazy(skx_array *a, const AssembleArgs &aa, hipStream_t st, int mode, uint32_t n_blocks = 0)
{
    if (a->lazy_rows->wide) launch_assemble_wide(aa, st, mode, n_blocks); else launch_assemble(aa, st, mode, n_blocks);
}
static AssembleArgs lazy_args(skx_array *a, int *d_flag)
{
    AssembleArgs aa{};
    skx_keyset *rows = a->lazy_rows;
    aa.d = a->lazy_dict->view(); aa.logN = rows->logN; aa.stage = rows->stage.p; aa.stride = rows->stride; aa.ncnt = rows->ncnt.p; aa.roff = rows->roff.p;
    aa.matrix = nullptr; aa.pitch = 0; aa.col_present = a->present.p; aa.col_unambig = a->unambig.p; aa.col_mask = a->mask.p;
    aa.max_rows = rows->max_rows; aa.missing = d_flag;
    return aa;
}
// takes ownership of d and rows (also on failure)
static int array_make_lazy(skx_ctx *ctx, skx_dictset *d, skx_keyset *rows, const char *const *names, skx_array **out)
{
    std::unique_ptr<skx_array> a(new skx_array());
    a->lazy_dict = d; a->lazy_rows = rows;
    // the union's notes serve the eager assemble only (skx_merge, skx_array_assemble): a lazily held array fills its windows from the
    // dictionaries, so the 2 bytes per word are not kept for its lifetime
    rows->side.release(); rows->perm.release(); rows->g_perm.release(); rows->g_n.release(); rows->g_base.release(); rows->side_of = nullptr;
    hipStream_t st = ctx->stream;
    a->ctx = ctx; a->k = d->k; a->rc = d->rc; a->k_bits = d->key_bits; a->hp = d->hp; a->wh = d->wh; a->version = skx_version();
    for (int i = 0; i < d->n; i++) a->names.emplace_back(names && names[i] ? names[i] : "");
    const uint64_t U = rows->total;
    a->n_rows = a->n_kmers = U; a->pitch = 0; a->engine_order = true; a->stats_ready = false;
    SKX_TRY(a->present.alloc(U)); SKX_TRY(a->unambig.alloc(U)); SKX_TRY(a->mask.alloc(U)); SKX_TRY(a->keys.alloc(U * rows->wpk())); SKX_TRY(a->vcount.alloc(U));
    if (U) {
        if (rows->wide) launch_gather_keys_wide((const u128 *)rows->stage.p, rows->stride, rows->ncnt.p, rows->roff.p, 1 << rows->logN, (u128 *)a->keys.p, st);
        else launch_gather_keys(rows->stage.p, rows->stride, rows->ncnt.p, rows->roff.p, 1 << rows->logN, a->keys.p, 0, rows->hp, st);
    }
    SKX_HIP(hipStreamSynchronize(st));
    *out = a.release();
    return SKX_OK;
}
static int lazy_check_missing(skx_array *a, DevBuf<int> &d_flag)
{
    int missing = 0;
    SKX_HIP(hipMemcpyAsync(&missing, d_flag.p, 4, hipMemcpyDeviceToHost, a->ctx->stream));
    SKX_HIP(hipStreamSynchronize(a->ctx->stream));
    SKX_HIP(hipGetLastError());
    if (missing) { set_error("row keyset does not contain every split k-mer of the samples"); return SKX_EINVAL; }
    return SKX_OK;
}
extern "C" int skx_array_assemble_lazy(skx_ctx *ctx, skx_dictset *d, skx_keyset *rows, const char *const *names, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !d || !rows || !out) { skx_dictset_free(d); skx_keyset_free(rows); set_error("bad arguments"); return SKX_EINVAL; }
    if (rows->holds_pieces_of(d)) {                                    // (an append pass's rows: the array over its pieces is the lazy form)
        const int r = skx_array_assemble(ctx, d, rows, names, out);
        skx_dictset_free(d); skx_keyset_free(rows);
        return r;
    }
    // the lazy form needs a row set slabbed at least as finely as the dictionaries' buckets; anything else is assembled at once
    if (rows->wide != d->wide() || rows->k != d->k || rows->rc != d->rc || rows->logN < 0 || rows->logN < d->logB || d->n > 65535 || knob("eager_array")) {
        const int r = skx_array_assemble(ctx, d, rows, names, out);
        skx_dictset_free(d); skx_keyset_free(rows);
        return r;
    }
    SKX_HIP(hipSetDevice(ctx->device));
    { const int rs = dictset_sort(d); if (rs != SKX_OK) { skx_dictset_free(d); skx_keyset_free(rows); return rs; } }
    return array_make_lazy(ctx, d, rows, names, out);
    });
}
int skx::array_lazy_stats(skx_array *a)
{
    if (!a->lazy() || a->stats_ready) return SKX_OK;
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
    if (a->n_rows) {
        AssembleArgs aa = lazy_args(a, d_flag.p);
        { StageTimer t(ctx, &ctx->tm.assemble); launch_assemble_lazy(a, aa, st, 1); }
        SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, a->n_rows * 4, hipMemcpyDeviceToDevice, st));     // merge_ska_array.rs:172
    }
    SKX_TRY(lazy_check_missing(a, d_flag));
    a->stats_ready = true;
    return SKX_OK;
}
int skx::array_materialize(skx_array *a)
{
    if (!a->lazy()) return SKX_OK;
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const uint64_t U = a->n_rows; const size_t S = a->names.size();
    a->pitch = pitch_for(U);
    SKX_TRY(a->matrix.alloc((uint64_t)S * a->pitch));
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
    if (a->pieces) {
        if (U) {
            PiecesRowsArgs pa = pieces_args(a);
            pa.out = a->matrix.p; pa.pitch = a->pitch;
            { StageTimer t(ctx, &ctx->tm.assemble); launch_pieces_rows(pa, 1u << a->pieces->logQ, st); }
        }
        SKX_HIP(hipStreamSynchronize(st));
        SKX_HIP(hipGetLastError());
        a->drop_lazy();
        return SKX_OK;
    }
    if (U) {
        AssembleArgs aa = lazy_args(a, d_flag.p);
        aa.matrix = a->matrix.p; aa.pitch = a->pitch;
        { StageTimer t(ctx, &ctx->tm.assemble); launch_assemble_lazy(a, aa, st, 0); }
        if (!a->stats_ready) SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, U * 4, hipMemcpyDeviceToDevice, st));
    }
    SKX_TRY(lazy_check_missing(a, d_flag));
    a->stats_ready = true;
    a->drop_lazy();
    return SKX_OK;
}
int skx::array_lazy_window(skx_array *a, uint64_t r0, uint64_t nr, DevBuf<uint8_t> &buf, const uint8_t **win, uint64_t *wpitch)
{
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    skx_keyset *rows = a->lazy_rows;
    if (rows->h_roff.empty()) {
        rows->h_roff.resize((1ull << rows->logN) + 1);
        SKX_HIP(hipMemcpy(rows->h_roff.data(), rows->roff.p, rows->h_roff.size() * 8, hipMemcpyDeviceToHost));
    }
    const auto &ro = rows->h_roff;
    const uint64_t j0 = (uint64_t)(std::upper_bound(ro.begin(), ro.end(), r0) - ro.begin()) - 1;                // sub-bucket holding row r0
    const uint64_t j1 = (uint64_t)(std::lower_bound(ro.begin(), ro.end(), r0 + nr) - ro.begin());                // first sub-bucket past the last row
    const uint64_t c0 = ro[j0] & ~15ull, span = ro[j1] - c0;                                                     // 16-B aligned window start
    const uint64_t wp = pitch_for(span);
    const size_t S = a->names.size();
    if (buf.n < S * wp) SKX_TRY(buf.alloc(S * wp));
    if (a->pieces) {
        PiecesRowsArgs pa = pieces_args(a);
        pa.out = buf.p; pa.pitch = wp; pa.j_base = (uint32_t)j0; pa.col_base = c0;
        { StageTimer t(ctx, &ctx->tm.assemble); launch_pieces_rows(pa, (uint32_t)(j1 - j0), st); }
        SKX_HIP(hipGetLastError());
        *win = buf.p + (r0 - c0); *wpitch = wp;
        return SKX_OK;
    }
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
    AssembleArgs aa = lazy_args(a, d_flag.p);
    aa.matrix = buf.p; aa.pitch = wp; aa.j_base = (uint32_t)j0; aa.col_base = c0;
    { StageTimer t(ctx, &ctx->tm.assemble); launch_assemble_lazy(a, aa, st, 0, (uint32_t)(j1 - j0)); }
    SKX_TRY(lazy_check_missing(a, d_flag));
    *win = buf.p + (r0 - c0); *wpitch = wp;
    return SKX_OK;
}

// ---- MergeSkaDict::append from the raw regions (skx_append.hip) ----------------------------------------------------------------------
// rows, row statistics and the cells (as pieces) of all samples in one pass over the words the extraction kernel scattered; no sample is
// sorted.  Both key widths (append_kernel; append_wide_kernel for k > 31).  SKF_NOT_TAKEN: not this kind of dictset (sorted already, regions
// beyond the kernel's load rounds, a row-block split that would read every region too often), or blocks that kept overflowing -- the caller
// sorts the dictionaries and takes the union / assemble kernels.  ctx->last_merge_path says which way a merge went (skx_debug_last_merge_path).
// the pass itself: the row blocks' keys (ks: stage / ncnt / roff, stride = cap) and the samples' pieces
static int append_pass(skx_ctx *ctx, skx_dictset *d, std::unique_ptr<skx_keyset> &ks_out, std::unique_ptr<skx_pieces> &pc_out)
{
    auto not_taken = [&](const char *why) -> int {
        ctx->merge_path = std::string("sorted: ") + why;
        if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] merge: %s\n", ctx->merge_path.c_str());
        return SKF_NOT_TAKEN;
    };
    if (d->sorted) return not_taken("the dictionaries are sorted (read sets, oversize or repeat-rich samples, SKX_KNOBS=sorted_dicts)");
    if (d->n > 65535 || d->n < 1) return not_taken("more than 65535 samples");
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool wide = d->wide();                                       // 16-byte words, 16-byte table entries: skx_append_wide.inc
    const int S = d->n, bits = wide ? d->wh.bits : d->hp.bits, logB = d->logB, wpk = wide ? 2 : 1;
    const uint32_t region_cap = d->region_cap;
    const uint32_t max_cap = wide ? APPEND_WIDE_MAX_CAP : APPEND_MAX_CAP, max_slots = wide ? APPEND_WIDE_MAX_SLOTS : APPEND_MAX_SLOTS;
    uint64_t raw_sum = 0, raw_max = 0;
    for (auto v : d->raw_total) { raw_sum += v; raw_max = std::max(raw_max, v); }
    const int min_logQ = std::max(logB, bits - (wide ? 113 : 50));
    if (min_logQ > bits) return not_taken("no row-block split with room for a rank in a table entry");
    // (regions of any size: a wave streams a region in chunks of 1 024 words -- the samples above ~5 Mbp of round 6 keep their bucket count and
    // grow their regions)
    auto ranks_for = [&](double mean) -> uint32_t {
        const double c = mean + 6.0 * std::sqrt(mean + 1.0) + 32.0;
        return (uint32_t)std::min<double>(max_cap, std::ceil(c / 128.0) * 128.0);
    };
    auto slots_for = [&](uint32_t cap) -> uint32_t {                  // a power of two (the home slot is a shift), a third more than the ranks at least
        uint32_t n = 256u;
        while (n < cap + cap / 3 && n < max_slots) n <<= 1;
        return n;
    };
    auto pass_ok = [&](int lq, uint32_t nslots, uint32_t cap) { return wide ? append_wide_ok(bits, logB, lq, nslots, cap) : append_ok(bits, logB, lq, region_cap, nslots, cap); };
    DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1));
    DevBuf<unsigned long long> d_probe; SKX_TRY(d_probe.alloc(2));
    AppendArgs aa{};
    aa.words = d->words.p; aa.off = d->off.p; aa.raw = d->raw.p; aa.n_samples = S; aa.logB = logB; aa.bits = bits; aa.overflow = d_flag.p; aa.probe = d_probe.p;
    StageTimer t(ctx, &ctx->tm.key_union);
    // |U| from a thin slice of the hash space: the first row blocks of a 2^logP split, rows counted only
    double u_est = (double)raw_max;
    const double target = wide ? 2600.0 : 5400.0;    // mean rows of a probe block (the final split: the coarsest whose blocks fit their ranks)
    int logQ = min_logQ;
    if (S > 1) {
        int logP = std::min(bits, std::max(min_logQ, ilog2_ceil((uint64_t)((double)raw_max * 3.0 / target) + 1)));
        for (int attempt = 0;; attempt++) {
            const unsigned blocks = (unsigned)std::min<uint64_t>(64, 1ull << logP);
            aa.logQ = logP; aa.nslots = max_slots; aa.cap = max_cap; aa.rounds = 1; aa.bar = nullptr;
            if (!pass_ok(logP, aa.nslots, aa.cap)) return not_taken("no probe split with room for a rank in a table entry (very short or very long hash)");
            SKX_TRY(d_flag.zero(st)); SKX_TRY(d_probe.zero(st));
            { KernelTimer kt(ctx, &ctx->tm.append_probe); if (wide) launch_append_wide_probe(aa, blocks, st); else launch_append_probe(aa, region_cap, blocks, st); }
            unsigned long long pr[2] = {0, 0}; int ov = 0;
            SKX_HIP(hipMemcpyAsync(pr, d_probe.p, 16, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            SKX_HIP(hipGetLastError());
            if (!ov) { u_est = std::max((double)raw_max * 0.5, (double)pr[0] / blocks * (double)(1ull << logP)); break; }
            if (ov & 4) return not_taken("the table's bytes are not at LDS address 0");
            if (attempt >= 4 || logP + 2 > bits) return not_taken("the probe blocks kept overflowing");
            logP += 2;
        }
    }
    // the coarsest split whose blocks hold their rows with six sigma to spare (a finer one reads every region more often)
    for (logQ = min_logQ; logQ < bits && u_est / (double)(1ull << logQ) + 6.0 * std::sqrt(u_est / (double)(1ull << logQ) + 1.0) + 32.0 > (double)max_cap; logQ++) { }
    for (int attempt = 0;; attempt++, logQ++) {
        if (logQ > bits || attempt > 3) return not_taken("row blocks kept overflowing (a sample that is one repeat, or far more rows than the probe saw)");
        const uint64_t nsub = 1ull << logQ;
        const double mean = u_est / (double)nsub;
        const uint32_t cap = ranks_for(mean), nslots = slots_for(cap);
        const uint64_t amp = 1ull << (logQ - logB);
        // every region is read by `amp` workgroups that keep a share each: 8 at most where the regions are the 5 Mbp shape's (beyond that the
        // samples are unrelated and the sorted path reads less), 32 where the regions have grown with the samples (one XCD's workgroups: 256 / 8)
        const uint64_t amp_max = region_cap > (wide ? LDS_SORT_MAX_WIDE : LDS_SORT_MAX) ? 32 : 8;
        if (amp > amp_max && (double)raw_sum * (double)amp > 4e8) return not_taken("too many row blocks per region: every region would be read too often");
        if (!pass_ok(logQ, nslots, cap)) return not_taken("no row-block split that fits the LDS");
        std::unique_ptr<skx_keyset> ks(new skx_keyset());
        ks->ctx = ctx; ks->k = d->k; ks->rc = d->rc; ks->logN = logQ; ks->hp = d->hp; ks->wh = d->wh; ks->wide = wide; ks->stride = cap;
        std::unique_ptr<skx_pieces> pc(new skx_pieces());
        pc->cap = cap; pc->logQ = logQ;
        SKX_TRY(ks->stage.alloc(nsub * cap * wpk)); SKX_TRY(ks->ncnt.alloc(nsub));
        if (pc->data.alloc(nsub * (uint64_t)S * (cap / 2)) != SKX_OK) return not_taken("no room for the pieces");      // (the sorted path holds rows + dictionaries instead)
        SKX_TRY(pc->plen.alloc(nsub * (uint64_t)S + 2)); SKX_TRY(pc->perm.alloc(nsub * cap)); SKX_TRY(pc->nrank.alloc(nsub));
        SKX_TRY(d_flag.zero(st));
        aa.logQ = logQ; aa.nslots = nslots; aa.cap = cap;
        // persistent launch: one workgroup per CU, whole groups of a region's readers (8 XCDs x A) -- when the blocks divide that way
        aa.rounds = 1; aa.bar = nullptr;
        DevBuf<int> d_bar;
        {
            int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            uint64_t g = nsub;
            while (g > (uint64_t)std::max(cus, 1) && g % 2 == 0) g /= 2;
            if (g <= (uint64_t)std::max(cus, 1) && nsub % g == 0 && g % (8 * amp) == 0 && logB >= 3) {
                aa.rounds = (uint32_t)(nsub / g);
                SKX_TRY(d_bar.alloc(2ull << logB)); SKX_TRY(d_bar.zero(st)); aa.bar = d_bar.p;      // (second half: the readers' meetings inside a round)
            }
        }
        aa.pieces = pc->data.p; aa.plen = pc->plen.p; aa.perm = pc->perm.p; aa.nrank = pc->nrank.p;
        aa.stage = ks->stage.p; aa.stride = cap; aa.ncnt = ks->ncnt.p;
        { KernelTimer kt(ctx, &ctx->tm.append); if (wide) launch_append_wide(aa, st); else launch_append(aa, region_cap, st); }
        int ov = 0;
        SKX_HIP(hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        SKX_HIP(hipGetLastError());
        if (getenv("SKX_DEBUG")) fprintf(stderr, "[skx] append%s: logQ=%d (x%llu per region) cap=%u slots=%u rows~%.0f -> %s\n", wide ? " (128-bit keys)" : "", logQ, (unsigned long long)amp, cap, nslots, u_est, ov ? "overflow" : "ok");
        if (ov & 4) return not_taken("the table's bytes are not at LDS address 0");
        if (ov) continue;
        SKX_TRY(keyset_finish(ks.get()));
        ctx->merge_path = wide ? "append128" : "append64";
        ks_out = std::move(ks); pc_out = std::move(pc);
        return SKX_OK;
    }
}
// An array over the pieces of an append pass.  own rows: the pass's own row blocks (ks: ncnt / roff / stage, the pieces' perm); global rows
// (a sharded job, `g` = the rows skx_keyset_allgather returned): the rank's columns over all ranks' rows -- block j covers rows
// [g_base[j], g_base[j] + g_n[j]) and g_perm maps its first-seen ranks to places in that range; rows the rank does not hold have no cell.
static int array_over_pieces(skx_ctx *ctx, const skx_dictset *d, skx_keyset *ks, skx_pieces *pc_, skx_keyset *g, const char *const *names, skx_array **out)
{
    std::unique_ptr<skx_pieces> pc(pc_);
    hipStream_t st = ctx->stream;
    const int S = d->n, logQ = pc->logQ;
    const uint64_t nsub = 1ull << logQ;
    std::unique_ptr<skx_keyset> blk(new skx_keyset());                 // what the array keeps of the row blocks: ncnt / roff
    blk->ctx = ctx; blk->k = d->k; blk->rc = d->rc; blk->logN = logQ; blk->hp = d->hp; blk->wh = d->wh; blk->wide = d->wide(); blk->stride = pc->cap;
    uint64_t U;
    if (g) {
        U = g->total;
        pc->perm = std::move(g->g_perm);
        blk->ncnt = std::move(g->g_n);
        SKX_TRY(blk->roff.alloc(nsub + 1));
        SKX_HIP(hipMemcpyAsync(blk->roff.p, g->g_base.p, nsub * 8, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipMemcpyAsync(blk->roff.p + nsub, &U, 8, hipMemcpyHostToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));                             // (&U)
        g->g_base.release();
        blk->total = U; blk->max_rows = g->g_max;
    } else {
        // (copied, not taken: the caller's key set stays whole -- skx_keyset_allgather, keyset_flatten or a second assemble may follow)
        U = ks->total;
        SKX_TRY(blk->ncnt.alloc(nsub)); SKX_TRY(blk->roff.alloc(nsub + 1));
        SKX_HIP(hipMemcpyAsync(blk->ncnt.p, ks->ncnt.p, nsub * 4, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipMemcpyAsync(blk->roff.p, ks->roff.p, (nsub + 1) * 8, hipMemcpyDeviceToDevice, st));
        blk->total = U; blk->max_rows = ks->max_rows;
    }
    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx; a->k = d->k; a->rc = d->rc; a->k_bits = d->key_bits; a->hp = d->hp; a->wh = d->wh; a->version = skx_version();
    for (int i = 0; i < S; i++) a->names.emplace_back(names && names[i] ? names[i] : "");
    a->n_rows = a->n_kmers = U; a->pitch = 0; a->engine_order = true; a->stats_ready = true;
    SKX_TRY(a->present.alloc(U)); SKX_TRY(a->unambig.alloc(U)); SKX_TRY(a->mask.alloc(U)); SKX_TRY(a->keys.alloc(U * (d->wide() ? 2 : 1))); SKX_TRY(a->vcount.alloc(U));
    if (U) {
        skx_keyset *src = g ? g : ks;                                  // the rows' keys
        if (src->wide) launch_gather_keys_wide((const u128 *)src->stage.p, src->stride, g ? g->ncnt.p : blk->ncnt.p, g ? g->roff.p : blk->roff.p, 1 << src->logN, (u128 *)a->keys.p, st);
        else launch_gather_keys(src->stage.p, src->stride, g ? g->ncnt.p : blk->ncnt.p, g ? g->roff.p : blk->roff.p, 1 << src->logN, a->keys.p, 0, src->hp, st);
        if (g) { SKX_TRY(a->present.zero(st)); SKX_TRY(a->unambig.zero(st)); SKX_TRY(a->mask.zero(st)); SKX_TRY(a->vcount.zero(st)); }
        // the rows' statistics, counted from the pieces
        StageTimer t(ctx, &ctx->tm.assemble);
        KernelTimer kt(ctx, &ctx->tm.pieces_stats);
        launch_pieces_stats(pc->data.p, pc->plen.p, pc->perm.p, pc->nrank.p, blk->ncnt.p, blk->roff.p, pc->cap, S, 1 << logQ, a->present.p, a->unambig.p, a->mask.p, a->vcount.p, st);
    }
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    a->pieces = pc.release(); a->lazy_rows = blk.release();
    *out = a.release();
    return SKX_OK;
}
static int merge_append(skx_ctx *ctx, skx_dictset *d, const char *const *names, skx_array **out)
{
    std::unique_ptr<skx_keyset> ks; std::unique_ptr<skx_pieces> pc;
    const int r = append_pass(ctx, d, ks, pc);
    if (r != SKX_OK) return r;
    return array_over_pieces(ctx, d, ks.get(), pc.release(), nullptr, names, out);
}

extern "C" int skx_merge(skx_ctx *ctx, skx_dictset *d, const char *const *names, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !d || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    { const int ra = merge_append(ctx, d, names, out); if (ra != SKF_NOT_TAKEN) return ra; }
    skx_keyset *ks = nullptr;
    SKX_TRY(keyset_union_dict(ctx, d, &ks, true));
    int r = skx_array_assemble(ctx, d, ks, names, out);
    skx_keyset_free(ks);
    return r;
    });
}

// ---- `ska build` on more samples than fit in HBM at once ---------------------------------------------------------------
// The per-sample dictionaries are the large transient (8-byte packed words, raw and deduplicated, per input base); the
// array that survives is 1 byte per (row, sample).  A batch of samples whose dictionaries fit is built and merged into
// an array, the batch arrays are then joined by the `ska merge` row-set path (skx_array_merge): the result has the same
// rows and columns as one big batch (merge_ska_dict.rs:354-417 builds the same union through its tree of appends).
// resident: what a sample holds on the device while its batch is built; transient: what building it needs for a moment (read sets are
// counted one sample at a time).  An assembly's dictionary is a packed word per base, raw and deduplicated, with slack.  A read set
// (first byte '@', or gzip: taken as reads) keeps its two record streams -- about the size of its files -- and a dictionary of at most
// one word per min_count windows; the windows of ONE sample with their partition copies are the transient.  (Pricing a 50x isolate as
// an assembly -- 22 GB instead of ~1.5 -- cut 12 isolates into three batches whose arrays were then joined on the host: 6 s instead of 2.5.)
struct SampleNeed { uint64_t resident = 0, transient = 0; };
static SampleNeed sample_device_bytes(const char *f1, const char *f2, bool wide, unsigned min_count)
{
    uint64_t bytes = 0; bool reads = false;
    for (const char *f : {f1, f2}) {
        if (!f) continue;
        struct stat sb;
        if (stat(f, &sb) != 0) continue;                       // the reader reports a missing file
        unsigned char c0 = 0;
        const int fd = ::open(f, O_RDONLY);
        if (fd >= 0) { if (::read(fd, &c0, 1) != 1) c0 = 0; ::close(fd); }
        const bool gz = c0 == 0x1f;
        reads |= gz || c0 == '@';
        bytes += (uint64_t)sb.st_size * (gz ? 5u : 1u);        // upper estimate of the text
    }
    SampleNeed n;
    if (reads) {
        const uint64_t bases = bytes / 2;                      // sequence and quality lines
        n.resident = bytes + bases / std::max(1u, min_count) * (wide ? 20u : 12u) + (8u << 20);
        n.transient = bases * (wide ? 56u : 28u);
    } else
        n.resident = bytes * (wide ? 44u : 24u) + (8u << 20);  // sequence + raw regions (with slack) + deduplicated words
    return n;
}
static uint64_t cached_device_bytes()        // of the current device
{
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_cache.mu);
    uint64_t n = 0;
    auto it = g_cache.free_blocks.find(dev);
    if (it != g_cache.free_blocks.end()) for (auto &kv : it->second) n += kv.first;
    return n;
}
static uint64_t build_budget_bytes(skx_ctx *ctx)
{
    if (const char *e = getenv("SKX_BUILD_BATCH_MB")) return (uint64_t)std::max(1.0, atof(e)) << 20;
    size_t fr = 0, tot = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return ~0ull;
    return (uint64_t)(fr + cached_device_bytes()) / 10 * 6;
}
static int build_range(skx_ctx *ctx, const char *const *names, const char *const *file1, const char *const *file2, int lo, int hi,
                       int k, int rc, const skx_qual *q, int threads, double proportion_reads, std::vector<skx_array *> &parts)
{
    skx_dictset *d = nullptr;
    skx_array *a = nullptr;
    int r = skx_dictset_build_files(ctx, file1 + lo, file2 ? file2 + lo : nullptr, hi - lo, k, rc, q, threads, proportion_reads, &d);
    if (r == SKX_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        int ra = merge_append(ctx, d, names + lo, &a);           // assemblies: straight from the extraction kernel's regions; the array holds the pieces
        if (ra != SKF_NOT_TAKEN) r = ra;
        else if (!knob("eager_array")) {
            // rows now, cells on demand: the array keeps the dictionaries (see skx_array::lazy_dict)
            skx_keyset *ks = nullptr;
            r = skx_keyset_union(ctx, d, &ks);                                         // (without notes: a lazily held array does not use them)
            if (r == SKX_OK) r = array_make_lazy(ctx, d, ks, names + lo, &a); else skx_dictset_free(d);
            d = nullptr;
        } else
            r = skx_merge(ctx, d, names + lo, &a);
        const auto t1 = std::chrono::steady_clock::now();
        skx_dictset_free(d);
        phase_add("build.merge", std::chrono::duration<double>(t1 - t0).count());
        phase_add("build.release_dictionaries", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    }
    if (r == SKX_ENOMEM && hi - lo > 1) {                      // the estimate was too low: halve the batch
        dev_trim();
        const int mid = lo + (hi - lo) / 2;
        SKX_TRY(build_range(ctx, names, file1, file2, lo, mid, k, rc, q, threads, proportion_reads, parts));
        return build_range(ctx, names, file1, file2, mid, hi, k, rc, q, threads, proportion_reads, parts);
    }
    if (r != SKX_OK) return r;
    parts.push_back(a);
    return SKX_OK;
}

extern "C" int skx_build_and_merge(skx_ctx *ctx, const char *const *names, const char *const *file1, const char *const *file2, int n,
                                   int k, int rc, const skx_qual *q, int threads, double proportion_reads, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !names || !file1 || n <= 0 || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    const uint64_t budget = build_budget_bytes(ctx);
    std::vector<skx_array *> parts;
    struct Drop { std::vector<skx_array *> &v; ~Drop() { for (auto *a : v) skx_array_free(a); } } drop{parts};
    int r = SKX_OK, lo = 0;
    while (lo < n && r == SKX_OK) {
        uint64_t need = 0, transient = 0;
        int hi = lo;
        while (hi < n) {
            const SampleNeed s = sample_device_bytes(file1[hi], file2 ? file2[hi] : nullptr, k > 31, q ? q->min_count : 1);
            if (hi > lo && need + s.resident + std::max(transient, s.transient) > budget) break;
            need += s.resident; transient = std::max(transient, s.transient); hi++;
        }
        need += transient;
        if (getenv("SKX_DEBUG") && (lo || hi < n)) fprintf(stderr, "[skx] build: samples %d..%d as one batch (estimate %.1f MB of %.1f MB)\n", lo, hi - 1, need / 1048576.0, budget / 1048576.0);
        r = build_range(ctx, names, file1, file2, lo, hi, k, rc, q, threads, proportion_reads, parts);
        lo = hi;
    }
    if (r != SKX_OK) return r;
    if (parts.size() == 1) { *out = parts[0]; parts.clear(); return SKX_OK; }
    return skx_array_merge(ctx, parts.data(), (int)parts.size(), out);
    });
}

extern "C" int skx_array_device_matrix(skx_array *a, const uint8_t **dptr, uint64_t *pitch, uint64_t *n_rows)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(array_materialize(a));
    *dptr = a->matrix.p; *pitch = a->pitch; *n_rows = a->n_rows;
    return SKX_OK;
    });
}

extern "C" int skx_array_device_stats(skx_array *a, uint32_t **present, uint32_t **unambig, uint32_t **mask, uint32_t **variant_count)
{
    { const int r = skx_guarded([&]() -> int { return a->lazy() ? array_lazy_stats(a) : SKX_OK; }); if (r != SKX_OK) return r; }      // a lazily held array stays one: statistics only
    if (present) *present = a->present.p;
    if (unambig) *unambig = a->unambig.p;
    if (mask) *mask = a->mask.p;
    if (variant_count) *variant_count = a->vcount.p;
    return SKX_OK;
}
extern "C" int skx_array_set_total_samples(skx_array *a, uint64_t total) { a->total_samples = total; return SKX_OK; }

extern "C" int skx_array_from_host(skx_ctx *ctx, int k, int rc, const char *const *names, int n_samples, const skx_key *keys,
                                   const uint8_t *variants, const uint64_t *variant_count, uint64_t n_rows, const char *version, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !out || n_samples <= 0) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx; a->k = k; a->rc = rc; a->k_bits = k <= 31 ? 64 : 128; a->hp = make_hash_params(std::min(k, 31)); a->wh = make_wide_hash(k);
    a->version = version ? version : skx_version();
    for (int i = 0; i < n_samples; i++) a->names.emplace_back(names[i]);
    a->n_rows = a->n_kmers = n_rows; a->pitch = pitch_for(n_rows); a->engine_order = false;
    SKX_TRY(a->matrix.alloc((uint64_t)n_samples * a->pitch));
    SKX_TRY(a->present.alloc(n_rows)); SKX_TRY(a->unambig.alloc(n_rows)); SKX_TRY(a->mask.alloc(n_rows)); SKX_TRY(a->vcount.alloc(n_rows));
    if (k <= 31) {
        SKX_TRY(a->keys.alloc(n_rows));
        std::vector<uint64_t> lo(n_rows);
        for (uint64_t i = 0; i < n_rows; i++) lo[i] = keys[i].lo;
        DevBuf<uint64_t> tmp; SKX_TRY(tmp.alloc(n_rows));
        SKX_HIP(hipMemcpyAsync(tmp.p, lo.data(), n_rows * 8, hipMemcpyHostToDevice, st));
        launch_hash_keys(tmp.p, a->keys.p, n_rows, a->hp, st);
        SKX_HIP(hipStreamSynchronize(st));
    } else {
        a->host_keys.assign(keys, keys + n_rows);
    }
    DevBuf<int> d_bad; SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st));
    if (n_rows) {
        DevBuf<uint8_t> rm; SKX_TRY(rm.alloc(n_rows * (uint64_t)n_samples));
        SKX_HIP(hipMemcpyAsync(rm.p, variants, n_rows * (uint64_t)n_samples, hipMemcpyHostToDevice, st));
        SKX_HIP(hipMemsetAsync(a->matrix.p, '-', (uint64_t)n_samples * a->pitch, st));
        launch_transpose(rm.p, (uint64_t)n_samples, n_rows, (uint64_t)n_samples, a->matrix.p, a->pitch, st);
        launch_col_stats(a->matrix.p, a->pitch, n_samples, n_rows, a->present.p, a->unambig.p, a->mask.p, d_bad.p, st);
        if (variant_count) {
            std::vector<uint32_t> vc(n_rows);
            for (uint64_t i = 0; i < n_rows; i++) vc[i] = (uint32_t)variant_count[i];
            SKX_HIP(hipMemcpyAsync(a->vcount.p, vc.data(), n_rows * 4, hipMemcpyHostToDevice, st));
            SKX_HIP(hipStreamSynchronize(st));
        } else SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, n_rows * 4, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));
    }
    int bad = 0;
    SKX_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    SKX_HIP(hipGetLastError());
    if (bad) { set_error("variants contain a byte outside -ACGTMRWSYKVHDBN (not supported on the device path)"); return SKX_EUNSUP; }
    *out = a.release();
    return SKX_OK;
    });
}

// split k-mers of an array as the reference stores them (hash undone), in the array's row order
int skx::array_host_keys(skx_array *a, std::vector<skx_key> &hk)
{
    if (a->keys_absent) { set_error("this array was loaded without its split k-mers (skx_array_load_filtered)"); return SKX_EINVAL; }
    const uint64_t K = a->n_kmers;
    hk.assign(K, skx_key{0, 0});
    if (a->k <= 31) {
        std::vector<uint64_t> w(K);
        if (K) SKX_HIP(hipMemcpy(w.data(), a->keys.p, K * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < K; i++) { hk[i].lo = hunmix(w[i] >> 4, a->hp); hk[i].hi = 0; }
    } else if (!a->host_keys.empty() || K == 0) hk = a->host_keys;
    else {
        std::vector<uint64_t> w(2 * K);
        SKX_HIP(hipMemcpy(w.data(), a->keys.p, K * 16, hipMemcpyDeviceToHost));
        // (the hash undone on the host: three rounds of 64-bit multiplies a key -- by a team, 0.16 s of `ska merge`'s save on one thread)
        const int team = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min(16, cpu_budget()), K >> 16));
        std::vector<std::thread> th;
        auto part = [&](int t) {
            for (uint64_t i = K * (uint64_t)t / (uint64_t)team, e = K * (uint64_t)(t + 1) / (uint64_t)team; i < e; i++) {
                const u128 key = hunmix_w((((u128)w[2 * i + 1] << 64) | w[2 * i]) >> 4, a->wh);
                hk[i].lo = (uint64_t)key; hk[i].hi = (uint64_t)(key >> 64);
            }
        };
        for (int t = 1; t < team; t++) th.emplace_back(part, t);
        part(0);
        for (auto &x : th) x.join();
    }
    return SKX_OK;
}

int skx::array_wide_words(skx_array *a, DevBuf<uint64_t> &tmp, const u128 **words)
{
    if (a->keys_absent) { set_error("this array was loaded without its split k-mers (skx_array_load_filtered)"); return SKX_EINVAL; }
    const uint64_t K = a->n_kmers;
    if (a->host_keys.empty()) { *words = (const u128 *)a->keys.p; return SKX_OK; }           // built here: already packed words
    hipStream_t st = a->ctx->stream;
    DevBuf<uint64_t> raw; SKX_TRY(raw.alloc(2 * K)); SKX_TRY(tmp.alloc(2 * K));
    static_assert(sizeof(skx_key) == 16 && offsetof(skx_key, lo) == 0, "skx_key is a little-endian u128");
    SKX_HIP(hipMemcpyAsync(raw.p, a->host_keys.data(), K * 16, hipMemcpyHostToDevice, st));
    launch_hash_keys_wide((const u128 *)raw.p, (u128 *)tmp.p, K, a->wh, st);
    SKX_HIP(hipStreamSynchronize(st));
    *words = (const u128 *)tmp.p;
    return SKX_OK;
}

extern "C" int skx_array_export(skx_array *a, skx_key *keys, uint8_t *variants, uint64_t *counts)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    if (variants) SKX_TRY(array_materialize(a)); else SKX_TRY(array_lazy_stats(a));
    const uint64_t U = a->n_rows, S = a->names.size(), K = a->n_kmers;
    std::vector<skx_key> hk;
    SKX_TRY(array_host_keys(a, hk));
    std::vector<uint8_t> rm(variants ? U * S : 0);              // keys / counts alone (variants == NULL) never move the matrix
    std::vector<uint32_t> pres(U);
    if (U) {
        if (variants) {
            DevBuf<uint8_t> d_rm; SKX_TRY(d_rm.alloc(U * S));
            launch_transpose(a->matrix.p, a->pitch, S, U, d_rm.p, S, st);
            SKX_HIP(hipMemcpyAsync(rm.data(), d_rm.p, U * S, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
        }
        SKX_HIP(hipMemcpyAsync(pres.data(), a->vcount.p, U * 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
    }
    // rows sorted by key when keys and rows are in step; otherwise stored order
    std::vector<uint64_t> idx(U); std::iota(idx.begin(), idx.end(), 0);
    const bool in_step = K == U;
    if (in_step) std::sort(idx.begin(), idx.end(), [&](uint64_t x, uint64_t y) { return hk[x].hi != hk[y].hi ? hk[x].hi < hk[y].hi : hk[x].lo < hk[y].lo; });
    if (keys) { if (in_step) for (uint64_t i = 0; i < K; i++) keys[i] = hk[idx[i]]; else std::copy(hk.begin(), hk.end(), keys); }
    if (variants) for (uint64_t i = 0; i < U; i++) memcpy(variants + i * S, rm.data() + idx[i] * S, S);
    if (counts) for (uint64_t i = 0; i < U; i++) counts[i] = pres[idx[i]];
    return SKX_OK;
    });
}

extern "C" int skx_array_sample_kmers(skx_array *a, int64_t *out)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const size_t S = a->names.size();
    if (a->pieces) {                                                  // the cells of the sample's pieces that are not empty (counted when first asked for)
        skx_pieces *pc = a->pieces;
        if (pc->sample_cells.size() != S) {
            DevBuf<unsigned long long> d; SKX_TRY(d.alloc(S)); SKX_TRY(d.zero(st));
            launch_pieces_cells(pc->data.p, pc->plen.p, pc->cap, (int)S, 1 << pc->logQ, d.p, st);
            std::vector<unsigned long long> h(S);
            SKX_HIP(hipMemcpyAsync(h.data(), d.p, S * 8, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            pc->sample_cells.assign(h.begin(), h.end());
        }
        for (size_t i = 0; i < S; i++) out[i] = (int64_t)pc->sample_cells[i];
        return SKX_OK;
    }
    if (a->lazy()) { for (size_t i = 0; i < S; i++) out[i] = (int64_t)a->lazy_dict->sample_size[i]; return SKX_OK; }      // a sample's cells = its dictionary
    DevBuf<unsigned long long> d; SKX_TRY(d.alloc(S)); SKX_TRY(d.zero(st));
    launch_row_nonmissing(a->matrix.p, a->pitch, (int)S, a->n_rows, d.p, st);
    std::vector<unsigned long long> h(S);
    SKX_HIP(hipMemcpyAsync(h.data(), d.p, S * 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < S; i++) out[i] = (int64_t)h[i];
    return SKX_OK;
    });
}

extern "C" int skx_array_pieces_info(skx_array *a, uint64_t *piece_bytes, uint64_t *row_blocks, uint32_t *ranks_per_block)
{
    return skx_guarded([&]() -> int {
    if (!a) { set_error("bad arguments"); return SKX_EINVAL; }
    if (piece_bytes) *piece_bytes = 0;
    if (row_blocks) *row_blocks = 0;
    if (ranks_per_block) *ranks_per_block = 0;
    const skx_pieces *pc = a->pieces;
    if (!pc) return SKX_OK;
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const uint64_t n = ((uint64_t)1 << pc->logQ) * a->names.size();
    std::vector<uint16_t> h(n);
    SKX_HIP(hipMemcpyAsync(h.data(), pc->plen.p, n * 2, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    uint64_t b = 0;
    for (uint16_t v : h) b += ((uint64_t)v + 31) / 32 * 16;           // append_kernel stores a piece 16 bytes (32 ranks) at a time
    if (piece_bytes) *piece_bytes = b;
    if (row_blocks) *row_blocks = (uint64_t)1 << pc->logQ;
    if (ranks_per_block) *ranks_per_block = pc->cap;
    return SKX_OK;
    });
}

// shared tail of filter / weed / delete_samples: rows with keep == 1 survive (pos = exclusive scan of keep)
static int array_compact(skx_array *a, DevBuf<uint8_t> &keep, DevBuf<uint64_t> &pos, uint64_t kept, int mask_ambig, bool vcount_from_unambig,
                         bool keys_follow)
{
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    const uint64_t U = a->n_rows; const size_t S = a->names.size();
    const bool filter_ambig_as_missing = vcount_from_unambig;
    if (kept == U && !mask_ambig) {                 // nothing goes: no 2 x U x S bytes of traffic, no second matrix
        if (filter_ambig_as_missing && U) SKX_HIP(hipMemcpyAsync(a->vcount.p, a->unambig.p, U * 4, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));
        return SKX_OK;
    }
    {
        StageTimer t(ctx, &ctx->tm.compact);
        const uint64_t np = pitch_for(kept);
        DevBuf<uint8_t> nm; DevBuf<uint32_t> p2, u2, m2, v2;
        SKX_TRY(nm.alloc((uint64_t)S * np)); SKX_TRY(p2.alloc(kept)); SKX_TRY(u2.alloc(kept)); SKX_TRY(m2.alloc(kept)); SKX_TRY(v2.alloc(kept));
        if (a->pieces) {
            // the kept rows straight from the pieces: the unfiltered rows x samples matrix is never written
            PiecesRowsArgs pa = pieces_args(a);
            pa.out = nm.p; pa.pitch = np; pa.keep = keep.p; pa.kpos = pos.p; pa.mask_ambig = mask_ambig;
            KernelTimer kt(ctx, &ctx->tm.pieces_rows);
            launch_pieces_rows(pa, 1u << a->pieces->logQ, st);
        } else if (a->lazy()) {
            // the kept rows are assembled straight from the dictionaries: the unfiltered matrix is never written
            DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
            AssembleArgs aa = lazy_args(a, d_flag.p);
            aa.matrix = nm.p; aa.pitch = np; aa.keep = keep.p; aa.kpos = pos.p; aa.mask_ambig = mask_ambig;
            { StageTimer t2(ctx, &ctx->tm.assemble); launch_assemble_lazy(a, aa, st, 2); }
            SKX_TRY(lazy_check_missing(a, d_flag));
        } else
            launch_compact_matrix(a->matrix.p, a->pitch, nm.p, np, (int)S, U, keep.p, pos.p, mask_ambig, st);
        launch_compact_u32(a->present.p, p2.p, U, keep.p, pos.p, st);
        launch_compact_u32(a->unambig.p, u2.p, U, keep.p, pos.p, st);
        launch_compact_u32(a->mask.p, m2.p, U, keep.p, pos.p, st);
        // update_counts(true) rewrites variant_count with the unambiguous counts (merge_ska_array.rs:139-163)
        launch_compact_u32(filter_ambig_as_missing ? a->unambig.p : a->vcount.p, v2.p, U, keep.p, pos.p, st);
        if (mask_ambig) launch_mask_ambig_stats(m2.p, kept, st);
        // update_counts(true) (merge_ska_array.rs:139-163) rewrites counts AND split_kmers whenever it ran
        if (a->n_kmers == U && keys_follow) {
            if (a->k <= 31) {
                DevBuf<uint64_t> k2; SKX_TRY(k2.alloc(kept));
                launch_compact_u64(a->keys.p, k2.p, U, keep.p, pos.p, st);
                a->keys = std::move(k2);
            } else if (a->host_keys.empty()) {
                DevBuf<uint64_t> k2; SKX_TRY(k2.alloc(kept * 2));
                launch_compact_u128(a->keys.p, k2.p, U, keep.p, pos.p, st);
                a->keys = std::move(k2);
            } else {
                std::vector<uint8_t> hkeep(U);
                SKX_HIP(hipMemcpyAsync(hkeep.data(), keep.p, U, hipMemcpyDeviceToHost, st));
                SKX_HIP(hipStreamSynchronize(st));
                std::vector<skx_key> nk; nk.reserve(kept);
                for (uint64_t i = 0; i < U; i++) if (hkeep[i] == 1) nk.push_back(a->host_keys[i]);
                a->host_keys.swap(nk);
            }
            a->n_kmers = kept;
        }
        SKX_HIP(hipStreamSynchronize(st));
        a->matrix = std::move(nm); a->present = std::move(p2); a->unambig = std::move(u2); a->mask = std::move(m2); a->vcount = std::move(v2);
        a->pitch = np; a->n_rows = kept;
        a->drop_lazy();
    }
    return SKX_OK;
}

extern "C" int skx_array_filter(skx_array *a, uint64_t min_count, int filter_ambig_as_missing, int filter_type, int mask_ambig,
                                int ignore_const_gaps, int update_kmers, int32_t *removed)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    const uint64_t U = a->n_rows; const size_t S = a->names.size();
    SKX_TRY(array_lazy_stats(a));
    DevBuf<uint8_t> keep; DevBuf<uint64_t> pos;
    SKX_TRY(keep.alloc(U)); SKX_TRY(pos.alloc(U + 1));
    uint64_t kept = 0, silent = 0;
    {
        StageTimer t(ctx, &ctx->tm.filter);
        FilterArgs fa{a->vcount.p, a->present.p, a->unambig.p, a->mask.p, U, (uint32_t)(a->total_samples ? a->total_samples : S), min_count, filter_ambig_as_missing, filter_type, ignore_const_gaps, keep.p};
        launch_filter_flags(fa, st);
        DevBuf<uint32_t> sc_sums; DevBuf<uint64_t> sc_offs;
        SKX_TRY(sc_sums.alloc(scan_u8_blocks(U))); SKX_TRY(sc_offs.alloc(scan_u8_blocks(U) + 1));
        launch_scan_u8(keep.p, pos.p, U, sc_sums.p, sc_offs.p, st);
        SKX_HIP(hipMemcpyAsync(&kept, pos.p + U, 8, hipMemcpyDeviceToHost, st));
        if (filter_ambig_as_missing) {
            DevBuf<unsigned long long> d_sil; SKX_TRY(d_sil.alloc(1)); SKX_TRY(d_sil.zero(st));
            launch_count_u8(keep.p, U, 2, d_sil.p, st);
            unsigned long long s2 = 0;
            SKX_HIP(hipMemcpyAsync(&s2, d_sil.p, 8, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            silent = s2;
        }
        SKX_HIP(hipStreamSynchronize(st));
    }
    SKX_TRY(array_compact(a, keep, pos, kept, mask_ambig, filter_ambig_as_missing != 0, update_kmers || filter_ambig_as_missing));
    SKX_HIP(hipGetLastError());
    if (removed) *removed = (int32_t)(U - kept - silent);
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ skf life-cycle (N1)

// flags -> scan -> compaction with the keys following the rows
static int array_keep_rows(skx_array *a, DevBuf<uint8_t> &keep, uint64_t *removed)
{
    hipStream_t st = a->ctx->stream;
    const uint64_t U = a->n_rows;
    DevBuf<uint64_t> pos; SKX_TRY(pos.alloc(U + 1));
    uint64_t kept = 0;
    DevBuf<uint32_t> sc_sums; DevBuf<uint64_t> sc_offs;
    SKX_TRY(sc_sums.alloc(scan_u8_blocks(U))); SKX_TRY(sc_offs.alloc(scan_u8_blocks(U) + 1));
    launch_scan_u8(keep.p, pos.p, U, sc_sums.p, sc_offs.p, st);
    SKX_HIP(hipMemcpyAsync(&kept, pos.p + U, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_TRY(array_compact(a, keep, pos, kept, 0, false, true));
    SKX_HIP(hipGetLastError());
    if (removed) *removed = U - kept;
    return SKX_OK;
}

extern "C" skx_ctx *skx_array_ctx(const skx_array *a) { return a->ctx; }
extern "C" void skx_set_last_error(const char *msg) { set_error("%s", msg ? msg : ""); }

// RefSka::new(k, file, rc, ..) + kmer_iter (ska_ref.rs:189-262,541) as `ska weed` uses it: the canonical split k-mers of every
// record of a FASTA file, whatever their middle bases = the key set of the file's dictionary
extern "C" int skx_keyset_from_fasta(skx_ctx *ctx, const char *path, int k, int rc, skx_keyset **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !path || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_TRY(check_k(k));
    HostStream hs;
    SKX_TRY(read_sample_stream(path, nullptr, 0.0, hs));
    if (hs.is_fastq) { set_error("Cannot create reference from FASTQ files"); return SKX_EINVAL; }                                // ska_ref.rs:206-208
    skx_stream ss; ss.seq = hs.seq.data(); ss.qual = nullptr; ss.len = hs.seq.size();
    skx_dictset *d = nullptr;
    int r = skx_dictset_build(ctx, &ss, 1, 0, k, rc, nullptr, &d);
    if (r == SKX_EEMPTY) set_error("%s has no valid sequence", path);                                                             // ska_ref.rs:255-257
    if (r != SKX_OK) return r;
    r = skx_keyset_union(ctx, d, out);
    skx_dictset_free(d);
    return r;
    });
}

extern "C" int skx_array_merge(skx_ctx *ctx, skx_array *const *in, int n, skx_array **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !in || n <= 0 || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint64_t tot = 0; size_t S = 0;
    for (int i = 0; i < n; i++) SKX_TRY(array_materialize(in[i]));
    for (int i = 0; i < n; i++) {
        if (in[i]->k != in[0]->k) { set_error("K-mer lengths do not match: %d %d", in[i]->k, in[0]->k); return SKX_EINVAL; }     // merge_ska_dict.rs:169-171
        if (in[i]->rc != in[0]->rc) { set_error("Strand use inconsistent"); return SKX_EINVAL; }                                  // :172-174
        if (in[i]->n_kmers != in[i]->n_rows || in[i]->keys_absent) { set_error("split k-mers and variants are out of step (filtered without update_kmers)"); return SKX_EINVAL; }
        tot += in[i]->n_rows; S += in[i]->names.size();
    }
    if (S > 65535) { set_error("more than 65535 samples per device array"); return SKX_EUNSUP; }
    std::unique_ptr<skx_array> a(new skx_array());
    a->ctx = ctx; a->k = in[0]->k; a->rc = in[0]->rc; a->k_bits = in[0]->k_bits; a->hp = in[0]->hp; a->wh = in[0]->wh; a->version = skx_version();
    for (int i = 0; i < n; i++) for (auto &nm : in[i]->names) a->names.push_back(nm);                                             // :177
    // rows of the result = distinct split k-mers over all inputs; idx[i][r] = result row of row r of input i
    std::vector<DevBuf<uint32_t>> idx(n);
    uint64_t U = 0;
    if (a->k <= 31) {
        DevBuf<uint64_t> all, rows; SKX_TRY(all.alloc(tot));
        uint64_t o = 0;
        for (int i = 0; i < n; i++) { if (in[i]->n_rows) SKX_HIP(hipMemcpyAsync(all.p + o, in[i]->keys.p, in[i]->n_rows * 8, hipMemcpyDeviceToDevice, st)); o += in[i]->n_rows; }
        SKX_TRY(sort_unique_words(all.p, tot, rows, &U, st));
        for (int i = 0; i < n; i++) { SKX_TRY(idx[i].alloc(in[i]->n_rows)); launch_lookup_rows(in[i]->keys.p, in[i]->n_rows, rows.p, U, idx[i].p, st); }
        SKX_TRY(a->keys.alloc(U));
        if (U) SKX_HIP(hipMemcpyAsync(a->keys.p, rows.p, U * 8, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));
        a->engine_order = true;
    } else {
        // 128-bit keys: the same on 2-word packed words (arrays from files keep the reference's keys on the host; they are
        // packed on the device for the occasion)
        std::vector<DevBuf<uint64_t>> tmp(n); std::vector<const u128 *> w(n, nullptr);
        DevBuf<uint64_t> all, rows; SKX_TRY(all.alloc(2 * tot));
        uint64_t o = 0;
        for (int i = 0; i < n; i++) {
            if (!in[i]->n_rows) continue;
            SKX_TRY(array_wide_words(in[i], tmp[i], &w[i]));
            SKX_HIP(hipMemcpyAsync(all.p + 2 * o, w[i], in[i]->n_rows * 16, hipMemcpyDeviceToDevice, st));
            o += in[i]->n_rows;
        }
        SKX_TRY(sort_unique_wide((const u128 *)all.p, tot, rows, &U, st));
        for (int i = 0; i < n; i++) { SKX_TRY(idx[i].alloc(in[i]->n_rows)); launch_lookup_rows_wide(w[i], in[i]->n_rows, (const u128 *)rows.p, U, idx[i].p, st); }
        SKX_TRY(a->keys.alloc(2 * U));
        if (U) SKX_HIP(hipMemcpyAsync(a->keys.p, rows.p, U * 16, hipMemcpyDeviceToDevice, st));
        SKX_HIP(hipStreamSynchronize(st));
        a->engine_order = true;
    }
    if (U > 0xFFFFFFF0ull) { set_error("too many rows"); return SKX_EUNSUP; }
    a->n_rows = a->n_kmers = U; a->pitch = pitch_for(U);
    SKX_TRY(a->matrix.alloc((uint64_t)S * a->pitch));
    SKX_TRY(a->present.alloc(U)); SKX_TRY(a->unambig.alloc(U)); SKX_TRY(a->mask.alloc(U)); SKX_TRY(a->vcount.alloc(U));
    SKX_HIP(hipMemsetAsync(a->matrix.p, '-', (uint64_t)S * a->pitch, st));                                                        // absent = 0 -> '-' (merge_ska_array.rs:175)
    size_t c0 = 0;
    for (int i = 0; i < n; i++) {
        launch_scatter_rows(in[i]->matrix.p, in[i]->pitch, (int)in[i]->names.size(), a->matrix.p + c0 * a->pitch, a->pitch, idx[i].p, in[i]->n_rows, st);
        c0 += in[i]->names.size();
    }
    DevBuf<int> d_bad; SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st));
    if (U) {
        launch_col_stats(a->matrix.p, a->pitch, (int)S, U, a->present.p, a->unambig.p, a->mask.p, d_bad.p, st);
        SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, U * 4, hipMemcpyDeviceToDevice, st));                                   // :172
    }
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    *out = a.release();
    return SKX_OK;
    });
}

extern "C" int skx_array_delete_samples(skx_array *a, const char *const *del_names, int n_del)
{
    return skx_guarded([&]() -> int {
    if (!a) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const size_t S = a->names.size();
    if (n_del <= 0 || (size_t)n_del == S) { set_error("Invalid number of samples to remove"); return SKX_EINVAL; }                // merge_ska_array.rs:232-234
    if (a->n_kmers != a->n_rows || a->keys_absent) { set_error("split k-mers and variants are out of step (filtered without update_kmers)"); return SKX_EINVAL; }
    // the request is a set of names; each must match a column (first match wins, :243-249)
    std::vector<std::string> want;
    for (int d = 0; d < n_del; d++) if (std::find(want.begin(), want.end(), del_names[d]) == want.end()) want.emplace_back(del_names[d]);
    std::vector<char> drop(S, 0);
    for (auto &w : want) {
        bool found = false;
        for (size_t s = 0; s < S && !found; s++) if (!drop[s] && a->names[s] == w) { drop[s] = 1; found = true; }
        if (!found) { set_error("Could not find sample(s): {\"%s\"}", w.c_str()); return SKX_EINVAL; }                           // :252-254
    }
    std::vector<std::string> names2;
    for (size_t s = 0; s < S; s++) if (!drop[s]) names2.push_back(a->names[s]);
    const size_t S2 = names2.size();
    if (S2 == 0) { set_error("Invalid number of samples to remove"); return SKX_EINVAL; }
    const uint64_t U = a->n_rows;
    DevBuf<uint8_t> nm; SKX_TRY(nm.alloc((uint64_t)S2 * a->pitch));
    size_t c = 0;
    for (size_t s = 0; s < S; s++) if (!drop[s]) { SKX_HIP(hipMemcpyAsync(nm.p + c * a->pitch, a->matrix.p + s * a->pitch, a->pitch, hipMemcpyDeviceToDevice, st)); c++; }
    SKX_HIP(hipStreamSynchronize(st));
    a->matrix = std::move(nm); a->names.swap(names2); a->total_samples = 0;
    // update_counts(false) (:139-163,270): recount, drop the rows no remaining sample has
    DevBuf<int> d_bad; SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st));
    DevBuf<uint8_t> keep; SKX_TRY(keep.alloc(U));
    if (U) {
        launch_col_stats(a->matrix.p, a->pitch, (int)S2, U, a->present.p, a->unambig.p, a->mask.p, d_bad.p, st);
        SKX_HIP(hipMemcpyAsync(a->vcount.p, a->present.p, U * 4, hipMemcpyDeviceToDevice, st));
        launch_nonzero_flags(a->present.p, U, keep.p, st);
    }
    return array_keep_rows(a, keep, nullptr);
    });
}

extern "C" int skx_array_weed(skx_array *a, skx_keyset *weed, int reverse, uint64_t *removed)
{
    return skx_guarded([&]() -> int {
    if (!a || !weed) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    if (weed->k != a->k) { set_error("K-mer lengths do not match: %d %d", weed->k, a->k); return SKX_EINVAL; }
    if (weed->rc != a->rc) { set_error("Strand use inconsistent"); return SKX_EINVAL; }
    if (a->n_kmers != a->n_rows || a->keys_absent) { set_error("split k-mers and variants are out of step (filtered without update_kmers)"); return SKX_EINVAL; }
    const uint64_t U = a->n_rows;
    if (weed->logN >= 0) SKX_TRY(keyset_flatten(weed));
    DevBuf<uint8_t> keep; SKX_TRY(keep.alloc(U));
    if (a->k <= 31) {
        DevBuf<uint32_t> idx; SKX_TRY(idx.alloc(U));
        launch_lookup_rows(a->keys.p, U, weed->flat.p, weed->total, idx.p, st);
        launch_member_flags(idx.p, U, reverse, keep.p, st);                                                                       // merge_ska_array.rs:468
        SKX_HIP(hipStreamSynchronize(st));
    } else {
        DevBuf<uint64_t> tmp; const u128 *w = nullptr;
        if (U) SKX_TRY(array_wide_words(a, tmp, &w));
        DevBuf<uint32_t> idx; SKX_TRY(idx.alloc(U));
        launch_lookup_rows_wide(w, U, (const u128 *)weed->flat.p, weed->total, idx.p, st);
        launch_member_flags(idx.p, U, reverse, keep.p, st);
        SKX_HIP(hipStreamSynchronize(st));
    }
    return array_keep_rows(a, keep, removed);
    });
}

extern "C" int skx_array_fasta(skx_array *a, char **buf, uint64_t *len)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const size_t S = a->names.size(); const uint64_t U = a->n_rows;
    uint64_t tot = 0;
    for (auto &nm : a->names) tot += nm.size() + U + 3;
    char *out = (char *)malloc(tot + 1), *p = out;
    if (!out) { set_error("out of host memory"); return SKX_ENOMEM; }
    for (size_t s = 0; s < S; s++) {
        *p++ = '>'; memcpy(p, a->names[s].data(), a->names[s].size()); p += a->names[s].size(); *p++ = '\n';
        if (U) { hipError_t e = hipMemcpy(p, a->matrix.p + s * a->pitch, U, hipMemcpyDeviceToHost); if (e != hipSuccess) { free(out); return hip_fail(e, "hipMemcpy"); } }
        p += U; *p++ = '\n';
    }
    *p = 0; *buf = out; *len = (uint64_t)(p - out);
    return SKX_OK;
    });
}
// write_fasta (merge_ska_array.rs:507-520) streamed to a file descriptor: batches of samples (header + row + newline) come off
// the device into pinned buffers; no copy of the alignment is held on the host.
//  * regular file opened read-write: the file is sized up front and mapped, and every batch is copied into the mapping by
//    several threads at once (page-cache pages of one file are filled in parallel; write()/pwrite() on one inode are serialised
//    by the kernel, which is what bounded the first version: 4.9 GB in 1.0 s);
//  * regular file, write-only: several batches in flight with pwrite at known offsets;
//  * pipe, or a descriptor opened with O_APPEND (`ska align x.skf >> out.aln`: Linux pwrite ignores the offset there and
//    appends): batches in order through write().
extern "C" int skx_array_write_fasta(skx_array *a, int fd)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const size_t S = a->names.size(); const uint64_t U = a->n_rows;
    size_t max_rec = 0; uint64_t total = 0;
    for (auto &nm : a->names) { max_rec = std::max<size_t>(max_rec, nm.size() + U + 3); total += nm.size() + U + 3; }
    struct stat sb;
    const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    const int fl = fcntl(fd, F_GETFL);
    const bool append = fl >= 0 && (fl & O_APPEND);
    off_t pos = regular ? lseek(fd, 0, SEEK_CUR) : 0;
    // ---- mapped output
    uint8_t *map = nullptr; size_t map_len = 0, map_skew = 0;
    if (regular && !append && pos >= 0 && fl >= 0 && (fl & O_ACCMODE) == O_RDWR && total) {
        const long pg = sysconf(_SC_PAGESIZE);
        map_skew = (size_t)(pos % pg);
        // the file is only ever grown: several writers (one per GPU) may each hold a window of the same file
        if ((off_t)sb.st_size >= pos + (off_t)total || ftruncate(fd, pos + (off_t)total) == 0) {
            map_len = total + map_skew;
            void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, pos - (off_t)map_skew);
            if (m != MAP_FAILED) map = (uint8_t *)m; else map_len = 0;
        }
    }
    struct Unmap { uint8_t *&p; size_t &n; ~Unmap() { if (p) munmap(p, n); } } unmap{map, map_len};
    // the file's pages are allocated by fallocate before the copies start (allocating a page-cache page inside a page fault
    // costs ~1 us and does not scale over the threads of one file; fallocate does ~12 GB/s on tmpfs, and copies that run beside
    // it contend with it: 4.9 GB in 0.8 s this way, 1.0 s with the copies chasing the allocation, 1.0-1.4 s through
    // write()/pwrite() -- tools/fasta_knobs.py); the first batches come off the device meanwhile
    std::shared_ptr<Preallocator> pre;
    std::atomic<bool> ok{true};                 // declared before the threads and slots that store to it
    std::atomic<bool> backed{false};
    struct Joiner { std::thread th; ~Joiner() { if (th.joinable()) th.join(); } } falloc;
    if (map) {
        if (a->prealloc && a->prealloc->fd == fd && a->prealloc->base == pos) pre = a->prealloc;      // started while the rows were being read
        else pre = std::make_shared<Preallocator>(fd, pos);
        a->prealloc.reset();
        // no copier touches the mapping before its pages exist; when they cannot be had the call fails with SKX_EIO as the write() path does
        falloc.th = std::thread([&backed, &ok, pre, total]() { pre->finish(total); if (pre->failed.load(std::memory_order_acquire)) ok = false; backed.store(true, std::memory_order_release); });
    }
    const size_t cap = std::max<size_t>(max_rec, map ? (32u << 20) : (64u << 20));
    const int NB = map ? 4 : (regular && !append && pos >= 0 ? 6 : 2);
    const int KT = map ? 4 : 1;                   // copier threads per batch (mapped output)

    struct Slot { char *p = nullptr; std::vector<std::thread> th; void join() { for (auto &t : th) if (t.joinable()) t.join(); th.clear(); }
                  ~Slot() { join(); if (p) (void)hipHostFree(p); } } slot[6];
    { PhaseTimer t_pin("fasta.pinned_buffers");
      for (int b = 0; b < NB; b++) if (hipHostMalloc((void **)&slot[b].p, cap, hipHostMallocDefault) != hipSuccess) { slot[b].p = nullptr; set_error("out of host memory"); return SKX_ENOMEM; } }
    int cur = 0, last = -1;
    double t_join = 0, t_d2h = 0;
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    uint64_t done = 0;                          // bytes of the alignment handed to writers so far
    for (size_t s = 0; s < S && ok;) {
        Slot &sl = slot[cur];
        { const auto t0 = clk(); sl.join(); t_join += sec(t0, clk()); }
        char *buf = sl.p; size_t used = 0;
        const auto t1 = clk();
        while (s < S && used + a->names[s].size() + U + 3 <= cap) {
            const std::string &nm = a->names[s];
            buf[used++] = '>'; memcpy(buf + used, nm.data(), nm.size()); used += nm.size(); buf[used++] = '\n';
            if (U) SKX_HIP(hipMemcpyAsync(buf + used, a->matrix.p + s * a->pitch, U, hipMemcpyDeviceToHost, st));
            used += U; buf[used++] = '\n';
            s++;
        }
        SKX_HIP(hipStreamSynchronize(st));
        t_d2h += sec(t1, clk());
        if (map) {
            uint8_t *dst = map + map_skew + done;
            for (int t = 0; t < KT; t++) {
                const size_t lo = used * (size_t)t / KT, hi = used * (size_t)(t + 1) / KT;
                sl.th.emplace_back([dst, buf, lo, hi, &backed, &ok]() {
                    while (!backed.load(std::memory_order_acquire)) usleep(100);
                    if (!ok.load()) return;
                    memcpy(dst + lo, buf + lo, hi - lo);
                });
            }
        } else {
            if (NB == 2 && last >= 0) slot[last].join();                                     // keep the order on a pipe
            const off_t at = pos + (off_t)done;
            const bool positioned = NB > 2;
            sl.th.emplace_back([&ok, buf, used, fd, at, positioned]() {
                size_t w = 0;
                while (w < used) {
                    const ssize_t r = positioned ? pwrite(fd, buf + w, used - w, at + (off_t)w) : write(fd, buf + w, used - w);
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { ok = false; return; }
                    w += (size_t)r;
                }
            });
        }
        done += used; last = cur; cur = (cur + 1) % NB;
    }
    { const auto t0 = clk(); for (int b = 0; b < NB; b++) slot[b].join(); t_join += sec(t0, clk()); }
    phase_add("fasta.device_to_pinned", t_d2h); phase_add("fasta.wait_for_writers", t_join);
    if (!ok) { set_error("write failed"); return SKX_EIO; }
    if (map || NB > 2) (void)lseek(fd, pos + (off_t)done, SEEK_SET);
    return SKX_OK;
    });
}

// numerators over 36 of |S1 n S2| / (|S1||S2|) per pair class [2..11] of pair_counts_kernel<false>
static const int PAIR_CLASS_NUM[10] = {36, 18, 12, 9, 18, 6, 12, 4, 8, 12};

// pair-class counts -> VariantDist (merge_ska_array.rs:596-631), pairs (i in [i_lo, i_hi), j > i) row-major; h rows are relative to i_lo
static void finish_pairs(const unsigned long long *h, int S, int i_lo, int i_hi, double constant, int filt_ambig, skx_dist *out)
{
    uint64_t n = 0;
    for (int i = i_lo; i < i_hi; i++)
        for (int j = i + 1; j < S; j++, n++) {
            const unsigned long long *c = &h[((uint64_t)(i - i_lo) * S + j) * DIST_NCOUNT];
            double mismatches = (double)c[0], matches = constant, distance;
            if (filt_ambig) { matches += (double)c[2]; distance = (double)(c[2] - c[3]); }
            else {
                unsigned long long m = 0, num = 0;
                for (int q = 0; q < 10; q++) { m += c[2 + q]; num += c[2 + q] * (unsigned long long)PAIR_CLASS_NUM[q]; }
                matches += (double)m;
                distance = (double)(36ull * c[1] - num) / 36.0;
            }
            out[n].distance = distance;
            out[n].mismatch_prop = (matches + mismatches) == 0.0 ? 0.0 : mismatches / (matches + mismatches);
            out[n].match_count = (uint64_t)matches; out[n].mismatch_count = (uint64_t)mismatches;
        }
}
// bit planes of the rows flagged 1 in keep: scan, keep words, planes (every word written).  rows = how many
int skx::planes_of_kept_rows(skx_array *a, const uint8_t *keep, int filt, DevBuf<uint64_t> &planes, uint64_t &wpr, uint64_t &rows)
{
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    const int S = (int)a->names.size(); const uint64_t U = a->n_rows;
    DevBuf<uint64_t> pos, sc_offs, kb, gp; DevBuf<uint32_t> sc_sums, fg;
    SKX_TRY(pos.alloc(U + 1)); SKX_TRY(sc_sums.alloc(scan_u8_blocks(U))); SKX_TRY(sc_offs.alloc(scan_u8_blocks(U) + 1));
    SKX_TRY(kb.alloc((U + 63) / 64)); SKX_TRY(gp.alloc((U + 63) / 64));
    launch_scan_u8(keep, pos.p, U, sc_sums.p, sc_offs.p, st);
    SKX_HIP(hipMemcpyAsync(&rows, pos.p + U, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    wpr = std::max<uint64_t>((rows + 63) / 64, 1);
    SKX_TRY(planes.alloc((filt ? 4 : 8) * (uint64_t)S * wpr));
    if (!rows) { SKX_TRY(planes.zero(st)); return SKX_OK; }
    launch_keep_bits(keep, pos.p, U, kb.p, gp.p, st);
    SKX_TRY(fg.alloc(rows / 4096 + 2));
    launch_build_planes_keep(a->matrix.p, a->pitch, S, U, kb.p, gp.p, planes.p, wpr, filt, st, fg.p, rows);
    SKX_HIP(hipStreamSynchronize(st));            // pos / kb / gp / fg go out of scope
    return SKX_OK;
}
// --allow-ambiguous over the rows flagged in keep (nullptr: all): the twelve pair classes differ from the three of the default sweep only on rows
// that hold an ambiguous cell (the row statistics say which), so the rows without one go through the 4-plane sweep, their counts filed as classes
// 0-2, and only the others through the 8-plane, twelve-class one (merge_ska_array.rs:587-632 sums per row: any split of the rows gives the sums)
static int distance_ambiguous_split(skx_array *a, const uint8_t *keep, double constant, skx_dist *out)
{
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    const int S = (int)a->names.size(); const uint64_t U = a->n_rows;
    DevBuf<uint8_t> clean, dirty;
    SKX_TRY(clean.alloc(U)); SKX_TRY(dirty.alloc(U));
    launch_split_keep(keep, a->mask.p, U, clean.p, dirty.p, st, knob("stale_row_mask") ? 2 : 0);
    DevBuf<uint64_t> pc, pd; uint64_t wc = 1, wd = 1, nc = 0, nd = 0;
    SKX_TRY(planes_of_kept_rows(a, clean.p, 1, pc, wc, nc));
    if (nc) {
        // the split rests on the row statistics: on a clean row every present cell is one base, i.e. plane 0 (present) == plane 1 (unambiguous).
        // Statistics that missed a code (none of the engine's operations leaves such, but the array is the caller's) show up here: all rows
        // go through the twelve-class sweep then
        DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
        launch_differ_u32((const uint32_t *)pc.p, (const uint32_t *)(pc.p + (uint64_t)S * wc), (uint64_t)S * wc * 2, d_flag.p, st);
        int differ = 0;
        SKX_HIP(hipMemcpyAsync(&differ, d_flag.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        if (differ) { launch_split_keep(keep, a->mask.p, U, clean.p, dirty.p, st, 1); nc = 0; pc.release(); }
    }
    SKX_TRY(planes_of_kept_rows(a, dirty.p, 0, pd, wd, nd));
    return planes_distance_split(ctx, pc.p, wc, nc, pd.p, wd, nd, S, constant, 0, S, out);
}
int skx::planes_distance_split(skx_ctx *ctx, const uint64_t *planes_clean, uint64_t wpr_clean, uint64_t rows_clean, const uint64_t *planes_dirty, uint64_t wpr_dirty,
                               uint64_t rows_dirty, int S, double constant, int i_lo, int i_hi, skx_dist *out)
{
    hipStream_t st = ctx->stream;
    if (S < 2 || i_lo >= i_hi) return SKX_OK;
    const uint64_t rows = (uint64_t)(i_hi - i_lo);
    DevBuf<unsigned long long> cnt;
    SKX_TRY(cnt.alloc(rows * S * DIST_NCOUNT)); SKX_TRY(cnt.zero(st));
    if (rows_clean && planes_clean) SKX_HIP((hipError_t)launch_pair_counts(planes_clean, S, wpr_clean, 2, cnt.p, st, i_lo, i_hi));
    if (rows_dirty && planes_dirty) SKX_HIP((hipError_t)launch_pair_counts(planes_dirty, S, wpr_dirty, 0, cnt.p, st, i_lo, i_hi));
    std::vector<unsigned long long> h(rows * S * DIST_NCOUNT);
    SKX_HIP(hipMemcpyAsync(h.data(), cnt.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    finish_pairs(h.data(), S, i_lo, i_hi, constant, 0, out);
    return SKX_OK;
}
int skx::planes_distance(skx_ctx *ctx, const uint64_t *planes, int S, uint64_t wpr, int filt_ambig, double constant, int i_lo, int i_hi, skx_dist *out)
{
    hipStream_t st = ctx->stream;
    if (S < 2 || i_lo >= i_hi) return SKX_OK;
    const uint64_t rows = (uint64_t)(i_hi - i_lo);
    DevBuf<unsigned long long> cnt;
    SKX_TRY(cnt.alloc(rows * S * DIST_NCOUNT)); SKX_TRY(cnt.zero(st));
    SKX_HIP((hipError_t)launch_pair_counts(planes, S, wpr, filt_ambig, cnt.p, st, i_lo, i_hi));
    std::vector<unsigned long long> h(rows * S * DIST_NCOUNT);
    SKX_HIP(hipMemcpyAsync(h.data(), cnt.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    finish_pairs(h.data(), S, i_lo, i_hi, constant, filt_ambig, out);
    return SKX_OK;
}

// generic_modes::distance (generic_modes.rs:136-189) on an array in memory without touching it: the two filters decide per row, the
// bit planes are built over the rows that stay, the pair sweep runs on those -- no compaction of the rows x samples matrix.
extern "C" int skx_array_distance_filtered(skx_array *a, double min_freq, int filt_ambig, skx_dist *out, int64_t *constant, uint64_t *rows_used)
{
    return skx_guarded([&]() -> int {
    if (!a || !out) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const int S = (int)a->names.size(); const uint64_t U = a->n_rows;
    const uint64_t S_total = a->total_samples ? a->total_samples : (uint64_t)S;
    const uint64_t thr = min_freq * (double)S_total >= 1.0 ? (uint64_t)std::ceil((double)S_total * min_freq) : 0;       // generic_modes.rs:149-159
    if (constant) *constant = 0;
    if (rows_used) *rows_used = 0;
    if (S < 2) return SKX_OK;
    StageTimer t(ctx, &ctx->tm.distance);
    uint64_t kept = 0; unsigned long long n_const = 0;
    DevBuf<uint64_t> planes; uint64_t wpr = 1;
    if (U) {
        DevBuf<uint8_t> keep; DevBuf<uint64_t> pos, sc_offs, kb, gp; DevBuf<uint32_t> sc_sums; DevBuf<unsigned long long> d_c;
        SKX_TRY(keep.alloc(U)); SKX_TRY(pos.alloc(U + 1)); SKX_TRY(sc_sums.alloc(scan_u8_blocks(U))); SKX_TRY(sc_offs.alloc(scan_u8_blocks(U) + 1));
        SKX_TRY(kb.alloc((U + 63) / 64)); SKX_TRY(gp.alloc((U + 63) / 64)); SKX_TRY(d_c.alloc(1)); SKX_TRY(d_c.zero(st));
        FilterArgs fa{a->vcount.p, a->present.p, a->unambig.p, a->mask.p, U, (uint32_t)S_total, thr, 0, SKX_FILTER_NO_CONST, 0, keep.p, 1};
        launch_filter_flags(fa, st);
        launch_scan_u8(keep.p, pos.p, U, sc_sums.p, sc_offs.p, st);
        launch_count_u8(keep.p, U, 3, d_c.p, st);
        SKX_HIP(hipMemcpyAsync(&kept, pos.p + U, 8, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipMemcpyAsync(&n_const, d_c.p, 8, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        if (!filt_ambig && kept) {
            if (constant) *constant = (int64_t)n_const;
            if (rows_used) *rows_used = kept;
            return distance_ambiguous_split(a, keep.p, (double)n_const, out);
        }
        wpr = std::max<uint64_t>((kept + 63) / 64, 1);
        SKX_TRY(planes.alloc((filt_ambig ? 4 : 8) * (uint64_t)S * wpr));
        if (!kept) SKX_TRY(planes.zero(st));                                 // (otherwise every word is written by the plane kernel)
        launch_keep_bits(keep.p, pos.p, U, kb.p, gp.p, st);
        DevBuf<uint32_t> fg; SKX_TRY(fg.alloc(kept / 4096 + 2));
        if (kept) launch_build_planes_keep(a->matrix.p, a->pitch, S, U, kb.p, gp.p, planes.p, wpr, filt_ambig, st, fg.p, kept);
        SKX_HIP(hipStreamSynchronize(st));            // keep / pos / kb / gp go out of scope
    } else { SKX_TRY(planes.alloc((filt_ambig ? 4 : 8) * (uint64_t)S)); SKX_TRY(planes.zero(st)); }
    if (constant) *constant = (int64_t)n_const;
    if (rows_used) *rows_used = kept;
    return planes_distance(ctx, planes.p, S, wpr, filt_ambig, (double)n_const, 0, S, out);
    });
}

extern "C" int skx_array_distance_planes(skx_array *a, int filt_ambig, const void **planes, uint64_t *words_per_row, int *n_planes)
{
    return skx_guarded([&]() -> int {
    if (!a || !planes || !words_per_row) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const int S = (int)a->names.size(); const uint64_t U = a->n_rows;
    const uint64_t wpr = std::max<uint64_t>((U + 63) / 64, 1);
    const int np = filt_ambig ? 4 : 8;
    SKX_TRY(a->planes.alloc((uint64_t)np * S * wpr));
    if (U == 0) SKX_TRY(a->planes.zero(st));
    launch_build_planes(a->matrix.p, a->pitch, S, U, a->planes.p, wpr, filt_ambig, st);
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    *planes = a->planes.p; *words_per_row = wpr; if (n_planes) *n_planes = np;
    return SKX_OK;
    });
}
extern "C" int skx_planes_distance(skx_ctx *ctx, const void *planes, int n_samples, uint64_t words_per_row, int filt_ambig, double constant,
                                   int i_lo, int i_hi, skx_dist *out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !planes || !out || n_samples < 0 || i_lo < 0 || i_hi > n_samples) { set_error("bad arguments"); return SKX_EINVAL; }
    SKX_HIP(hipSetDevice(ctx->device));
    StageTimer t(ctx, &ctx->tm.distance);
    return planes_distance(ctx, (const uint64_t *)planes, n_samples, words_per_row, filt_ambig, constant, i_lo, i_hi, out);
    });
}

extern "C" int skx_array_distance(skx_array *a, double constant, int filt_ambig, skx_dist *out)
{
    return skx_guarded([&]() -> int {
    skx_ctx *ctx = a->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    SKX_TRY(array_materialize(a));
    const int S = (int)a->names.size(); const uint64_t U = a->n_rows;
    if (S < 2) return SKX_OK;
    StageTimer t(ctx, &ctx->tm.distance);
    if (!filt_ambig && U) return distance_ambiguous_split(a, nullptr, constant, out);
    const uint64_t wpr = (U + 63) / 64;
    DevBuf<uint64_t> planes;
    SKX_TRY(planes.alloc((filt_ambig ? 4 : 8) * (uint64_t)S * std::max<uint64_t>(wpr, 1)));
    launch_build_planes(a->matrix.p, a->pitch, S, U, planes.p, wpr, filt_ambig, st);
    SKX_TRY(planes_distance(ctx, planes.p, S, wpr, filt_ambig, constant, 0, S, out));
    return SKX_OK;
    });
}

