// skx_comm.hip -- the exchanges of the multi-GPU path behind the C ABI (include/skx.h "collectives").
//
// One process per GPU; samples shard contiguously by rank and nothing on the data path is collective.  What ranks exchange:
//   1. the per-rank key tables            -> skx_keyset_allgather      (one ncclAllGather; every rank derives the same rows)
//   2. the per-row filter statistics      -> skx_array_reduce_stats    (one ncclAllReduce of the packed counts + one ncclAllGather of
//                                                                       the 16-bit code sets, OR-ed on the device)
//   3. `distance`: the per-rank bit planes -> skx_array_distance_sharded (one ncclAllGather; the pair matrix is tiled over ranks by
//                                                                       bands of first samples, finished pairs are sent to rank 0)
// This replaces the thread tree of build_and_merge (merge_ska_dict.rs:354-417: per-thread dictionaries merged pairwise) for a job
// whose "threads" are GPUs: the partitioning of samples keeps input order as the reference's offsets do (merge_ska_dict.rs:243-253).
//
// Transports: RCCL over xGMI (librccl is opened at run time, so the library also loads where RCCL is absent and shares the copy a
// host such as PyTorch has already mapped), and a host-staged one through a directory on tmpfs for ranks that share one device
// (RCCL refuses two ranks on one GPU; the tests and single-GPU emulations of an N-rank job use it).  Same entry points either way.
#include "skx_internal.h"
#include <rccl/rccl.h>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <dlfcn.h>
#include <link.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace skx;

namespace {
// ---- RCCL, bound at run time ----
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;             // (optional: skx_comm_transport, reports)
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
};
Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // one copy of RCCL per process: when the host (PyTorch ships its own librccl.so) has already mapped one, that copy is used --
        // two copies each tear down the same runtime state at exit
        std::string loaded;
        dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *out) -> int {
            const char *n = info->dlpi_name;
            if (n && strstr(n, "librccl.so")) { *(std::string *)out = n; return 1; }
            return 0;
        }, &loaded);
        if (!loaded.empty()) r.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
        if (!r.lib)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) return;
#define SKX_SYM(f) r.f = (decltype(r.f))dlsym(r.lib, "nccl" #f)
        SKX_SYM(GetUniqueId); SKX_SYM(CommInitRank); SKX_SYM(CommDestroy); SKX_SYM(AllGather); SKX_SYM(AllReduce); SKX_SYM(Send); SKX_SYM(Recv);
        SKX_SYM(GroupStart); SKX_SYM(GroupEnd); SKX_SYM(GetErrorString); SKX_SYM(CommCount); SKX_SYM(CommCuDevice);
#undef SKX_SYM
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) r.lib = nullptr;
    });
    return r.lib ? &r : nullptr;
}
int nccl_fail(ncclResult_t e, const char *what)
{
    Rccl *r = rccl();
    set_error("RCCL: %s failed: %s", what, r && r->GetErrorString ? r->GetErrorString(e) : "?");
    return SKX_ENODEV;
}
#define SKX_NCCL(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return nccl_fail(e_, #call); } while (0)

// ---- host-staged transport: a directory on tmpfs, one control page (sense-reversing barrier), one file per (operation, rank) ----
struct LocalCtrl { std::atomic<uint32_t> arrived, generation; };
}  // namespace

struct skx_comm {
    skx_ctx *ctx = nullptr;          // may be null for the host-staged transport (host buffers only)
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;       // RCCL transport
    std::string dir;                 // host-staged transport
    bool own_dir = false;            // a single rank's private directory (created here, removed with the communicator)
    LocalCtrl *ctrl = nullptr;
    uint64_t seq = 0;
    double timeout_s = 600.0;
    uint64_t bytes_moved = 0;        // received by this rank since creation (reports)
    bool local() const { return nccl == nullptr; }
};

namespace {
int local_barrier(skx_comm *c)
{
    if (c->world == 1) return SKX_OK;
    const uint32_t g = c->ctrl->generation.load(std::memory_order_acquire);
    if (c->ctrl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        c->ctrl->arrived.store(0, std::memory_order_relaxed);
        c->ctrl->generation.store(g + 1, std::memory_order_release);
        return SKX_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; c->ctrl->generation.load(std::memory_order_acquire) == g; spin++) {
        if (spin > 200) usleep(spin > 2000 ? 1000 : 50);
        if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            set_error("rank %d: a peer did not reach the barrier within %.0f s", c->rank, c->timeout_s); return SKX_EIO;
        }
    }
    return SKX_OK;
}
std::string local_path(const skx_comm *c, uint64_t seq, int rank) { return c->dir + "/x" + std::to_string(seq) + "." + std::to_string(rank); }
int write_all(const std::string &path, const void *p, size_t n)
{
    int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) { set_error("cannot create %s: %s", path.c_str(), strerror(errno)); return SKX_EIO; }
    const char *b = (const char *)p;
    while (n) { ssize_t w = write(fd, b, n); if (w < 0) { if (errno == EINTR) continue; close(fd); set_error("write %s: %s", path.c_str(), strerror(errno)); return SKX_EIO; } b += w; n -= (size_t)w; }
    close(fd);
    return SKX_OK;
}
int read_all(const std::string &path, void *p, size_t n)
{
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { set_error("cannot open %s: %s", path.c_str(), strerror(errno)); return SKX_EIO; }
    char *b = (char *)p;
    while (n) { ssize_t r = read(fd, b, n); if (r < 0 && errno == EINTR) continue; if (r <= 0) { close(fd); set_error("short read from %s", path.c_str()); return SKX_EIO; } b += r; n -= (size_t)r; }
    close(fd);
    return SKX_OK;
}
// every rank contributes n_mine bytes (host memory); rank r's lie at recv + roff[r] (ranks in `want` only; others skipped)
int local_exchange(skx_comm *c, const void *send, size_t n_mine, void *recv, const std::vector<size_t> &rsize, const std::vector<size_t> &roff, bool root_only)
{
    const uint64_t seq = c->seq++;
    SKX_TRY(write_all(local_path(c, seq, c->rank), send, n_mine));
    SKX_TRY(local_barrier(c));
    int rc = SKX_OK;
    if (!root_only || c->rank == 0)
        for (int r = 0; r < c->world && rc == SKX_OK; r++) {
            if (r == c->rank) { if ((const char *)recv + roff[r] != (const char *)send) memcpy((char *)recv + roff[r], send, n_mine); }
            else { rc = read_all(local_path(c, seq, r), (char *)recv + roff[r], rsize[r]); c->bytes_moved += rsize[r]; }
        }
    const int rb = local_barrier(c);
    unlink(local_path(c, seq, c->rank).c_str());
    return rc != SKX_OK ? rc : rb;
}

// ---- primitives over either transport ----
// all-gather of `bytes` per rank; device (or, host-staged transport only, host) memory; recv may not alias send
int comm_allgather(skx_comm *c, const void *send, void *recv, size_t bytes, bool on_device)
{
    if (!bytes) return SKX_OK;
    if (!c->local()) {
        if (!on_device) { set_error("the RCCL transport moves device memory"); return SKX_EINVAL; }
        SKX_NCCL(rccl()->AllGather(send, recv, bytes, ncclInt8, c->nccl, c->ctx->stream));
        c->bytes_moved += bytes * (size_t)(c->world - 1);
        return SKX_OK;
    }
    std::vector<size_t> rsize(c->world, bytes), roff(c->world);
    for (int r = 0; r < c->world; r++) roff[r] = (size_t)r * bytes;
    if (!on_device) return local_exchange(c, send, bytes, recv, rsize, roff, false);
    if (!c->ctx) { set_error("this communicator has no device context"); return SKX_EINVAL; }
    std::vector<char> hs(bytes), hr(bytes * (size_t)c->world);
    SKX_HIP(hipMemcpyAsync(hs.data(), send, bytes, hipMemcpyDeviceToHost, c->ctx->stream));
    SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    SKX_TRY(local_exchange(c, hs.data(), bytes, hr.data(), rsize, roff, false));
    SKX_HIP(hipMemcpyAsync(recv, hr.data(), hr.size(), hipMemcpyHostToDevice, c->ctx->stream));
    SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    return SKX_OK;
}
// in-place sum over ranks of n 32-bit counters
int comm_allreduce_u32(skx_comm *c, uint32_t *buf, size_t n, bool on_device)
{
    if (!n || c->world == 1) return SKX_OK;
    if (!c->local()) {
        if (!on_device) { set_error("the RCCL transport moves device memory"); return SKX_EINVAL; }
        SKX_NCCL(rccl()->AllReduce(buf, buf, n, ncclUint32, ncclSum, c->nccl, c->ctx->stream));
        c->bytes_moved += 2 * n * 4;
        return SKX_OK;
    }
    std::vector<uint32_t> hs(n), hr(n * (size_t)c->world);
    if (on_device) {
        if (!c->ctx) { set_error("this communicator has no device context"); return SKX_EINVAL; }
        SKX_HIP(hipMemcpyAsync(hs.data(), buf, n * 4, hipMemcpyDeviceToHost, c->ctx->stream));
        SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    } else memcpy(hs.data(), buf, n * 4);
    std::vector<size_t> rsize(c->world, n * 4), roff(c->world);
    for (int r = 0; r < c->world; r++) roff[r] = (size_t)r * n * 4;
    SKX_TRY(local_exchange(c, hs.data(), n * 4, hr.data(), rsize, roff, false));
    for (size_t i = 0; i < n; i++) { uint32_t s = 0; for (int r = 0; r < c->world; r++) s += hr[(size_t)r * n + i]; hs[i] = s; }
    if (on_device) {
        SKX_HIP(hipMemcpyAsync(buf, hs.data(), n * 4, hipMemcpyHostToDevice, c->ctx->stream));
        SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    } else memcpy(buf, hs.data(), n * 4);
    return SKX_OK;
}
// a few bytes of host metadata from every rank (sizes): RCCL stages them through a scratch buffer on the device
int comm_allgather_host(skx_comm *c, const void *send, void *recv, size_t bytes)
{
    if (c->world == 1) { memcpy(recv, send, bytes); return SKX_OK; }
    if (c->local()) return comm_allgather(c, send, recv, bytes, false);
    DevBuf<uint8_t> s, r; SKX_TRY(s.alloc(bytes)); SKX_TRY(r.alloc(bytes * (size_t)c->world));
    SKX_HIP(hipMemcpyAsync(s.p, send, bytes, hipMemcpyHostToDevice, c->ctx->stream));
    SKX_TRY(comm_allgather(c, s.p, r.p, bytes, true));
    SKX_HIP(hipMemcpyAsync(recv, r.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->ctx->stream));
    SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    return SKX_OK;
}
// host blocks of different sizes from every rank to rank 0 (sizes known everywhere): recv (rank 0) = the blocks in rank order
int comm_gather_root_host(skx_comm *c, const void *send, const std::vector<size_t> &sizes, void *recv)
{
    std::vector<size_t> roff(c->world, 0);
    for (int r = 1; r < c->world; r++) roff[r] = roff[r - 1] + sizes[r - 1];
    if (c->world == 1) { if (recv != send) memcpy(recv, send, sizes[0]); return SKX_OK; }
    if (c->local()) return local_exchange(c, send, sizes[c->rank], recv, sizes, roff, true);
    hipStream_t st = c->ctx->stream;
    if (c->rank != 0) {
        if (!sizes[c->rank]) return SKX_OK;
        DevBuf<uint8_t> s; SKX_TRY(s.alloc(sizes[c->rank]));
        SKX_HIP(hipMemcpyAsync(s.p, send, sizes[c->rank], hipMemcpyHostToDevice, st));
        SKX_NCCL(rccl()->Send(s.p, sizes[c->rank], ncclInt8, 0, c->nccl, st));
        SKX_HIP(hipStreamSynchronize(st));
        return SKX_OK;
    }
    const size_t total = roff[c->world - 1] + sizes[c->world - 1];
    DevBuf<uint8_t> r; SKX_TRY(r.alloc(total));
    SKX_NCCL(rccl()->GroupStart());
    for (int p = 1; p < c->world; p++)
        if (sizes[p]) SKX_NCCL(rccl()->Recv(r.p + roff[p], sizes[p], ncclInt8, p, c->nccl, st));
    SKX_NCCL(rccl()->GroupEnd());
    if (total > sizes[0]) SKX_HIP(hipMemcpyAsync((char *)recv + sizes[0], r.p + sizes[0], total - sizes[0], hipMemcpyDeviceToHost, st));
    if (recv != send) memcpy(recv, send, sizes[0]);
    SKX_HIP(hipStreamSynchronize(st));
    c->bytes_moved += total - sizes[0];
    return SKX_OK;
}
int comm_barrier(skx_comm *c)
{
    if (c->world == 1) return SKX_OK;
    if (c->local()) return local_barrier(c);
    DevBuf<uint32_t> one; SKX_TRY(one.alloc(1)); SKX_TRY(one.zero(c->ctx->stream));
    SKX_NCCL(rccl()->AllReduce(one.p, one.p, 1, ncclUint32, ncclSum, c->nccl, c->ctx->stream));
    SKX_HIP(hipStreamSynchronize(c->ctx->stream));
    return SKX_OK;
}
int check(const skx_comm *c) { if (!c) { set_error("no communicator"); return SKX_EINVAL; } return SKX_OK; }
int check_dev(const skx_comm *c)
{
    SKX_TRY(check(c));
    if (!c->ctx) { set_error("this communicator has no device context"); return SKX_EINVAL; }
    return SKX_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------ communicators
extern "C" int skx_comm_unique_id(uint8_t *id)
{
    return skx_guarded([&]() -> int {
    if (!id) { set_error("bad arguments"); return SKX_EINVAL; }
    Rccl *r = rccl();
    if (!r) { set_error("librccl.so not found: %s", dlerror() ? dlerror() : "no loader message"); return SKX_ENODEV; }
    static_assert(sizeof(ncclUniqueId) == SKX_COMM_ID_BYTES, "skx.h and rccl.h disagree on the id size");
    ncclUniqueId u;
    SKX_NCCL(r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return SKX_OK;
    });
}

extern "C" int skx_comm_create(skx_ctx *ctx, int rank, int world, const uint8_t *id, skx_comm **out)
{
    return skx_guarded([&]() -> int {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad arguments"); return SKX_EINVAL; }
    Rccl *r = rccl();
    if (!r) { set_error("librccl.so not found"); return SKX_ENODEV; }
    SKX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<skx_comm> c(new skx_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    SKX_NCCL(r->CommInitRank(&c->nccl, world, u, rank));
    *out = c.release();
    return SKX_OK;
    });
}

extern "C" int skx_comm_create_local(skx_ctx *ctx, int rank, int world, const char *dir, skx_comm **out)
{
    return skx_guarded([&]() -> int {
    if ((!dir && world != 1) || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad arguments"); return SKX_EINVAL; }
    std::unique_ptr<skx_comm> c(new skx_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    if (dir) c->dir = dir;
    else {                                                         // a single rank exchanges with nobody: a directory of its own, gone with the communicator
        char tmpl[] = "/tmp/skx_one_rank_XXXXXX";
        if (!mkdtemp(tmpl)) { set_error("cannot create a directory for the single rank's communicator: %s", strerror(errno)); return SKX_EIO; }
        c->dir = tmpl; c->own_dir = true;
    }
    if (const char *t = getenv("SKX_COMM_TIMEOUT_S")) c->timeout_s = atof(t) > 0 ? atof(t) : c->timeout_s;
    const std::string path = c->dir + "/ctrl";
    int fd = open(path.c_str(), O_RDWR | O_CREAT, 0600);          // every rank creates-or-opens; a fresh page of zeros is the initial state
    if (fd < 0) { set_error("cannot open %s: %s", path.c_str(), strerror(errno)); return SKX_EIO; }
    if (ftruncate(fd, 4096) != 0) { close(fd); set_error("cannot size %s", path.c_str()); return SKX_EIO; }
    void *m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { set_error("cannot map %s", path.c_str()); return SKX_EIO; }
    c->ctrl = (LocalCtrl *)m;
    SKX_TRY(local_barrier(c.get()));                               // everybody is here (and the directory is not a stale one)
    *out = c.release();
    return SKX_OK;
    });
}

extern "C" void skx_comm_destroy(skx_comm *c)
{
    if (!c) return;
    if (c->nccl) { if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); } (void)rccl()->CommDestroy(c->nccl); }
    if (c->ctrl) munmap(c->ctrl, 4096);
    if (c->own_dir) { unlink((c->dir + "/ctrl").c_str()); rmdir(c->dir.c_str()); }
    delete c;
}
extern "C" int skx_comm_rank(const skx_comm *c) { return c ? c->rank : 0; }
extern "C" int skx_comm_world(const skx_comm *c) { return c ? c->world : 1; }
extern "C" uint64_t skx_comm_bytes_received(const skx_comm *c) { return c ? c->bytes_moved : 0; }
extern "C" int skx_comm_transport(const skx_comm *c, int *rccl_ranks, int *rccl_device, int *ctx_device)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check(c));
    if (rccl_ranks) *rccl_ranks = 0;
    if (rccl_device) *rccl_device = -1;
    if (ctx_device) *ctx_device = c->ctx ? c->ctx->device : -1;
    if (c->local()) return SKX_OK;
    Rccl *r = rccl();
    if (r && r->CommCount && rccl_ranks) SKX_NCCL(r->CommCount(c->nccl, rccl_ranks));
    if (r && r->CommCuDevice && rccl_device) SKX_NCCL(r->CommCuDevice(c->nccl, rccl_device));
    return SKX_OK;
    });
}
extern "C" int skx_comm_barrier(skx_comm *c)
{
    return skx_guarded([&]() -> int { SKX_TRY(check(c)); if (c->ctx) SKX_HIP(hipSetDevice(c->ctx->device)); return comm_barrier(c); });
}
extern "C" int skx_comm_allgather(skx_comm *c, const void *send, void *recv, uint64_t bytes, int on_device)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check(c));
    if ((!send || !recv) && bytes) { set_error("bad arguments"); return SKX_EINVAL; }
    if (c->ctx) SKX_HIP(hipSetDevice(c->ctx->device));
    if (c->world == 1) {
        if (!bytes) return SKX_OK;
        if (!on_device) { memcpy(recv, send, bytes); return SKX_OK; }
        SKX_TRY(check_dev(c));
        SKX_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
        if (!c->local()) return SKX_OK;
        SKX_HIP(hipStreamSynchronize(c->ctx->stream)); return SKX_OK;
    }
    return comm_allgather(c, send, recv, bytes, on_device != 0);
    });
}
extern "C" int skx_comm_allreduce_u32(skx_comm *c, uint32_t *buf, uint64_t n, int on_device)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check(c));
    if (!buf && n) { set_error("bad arguments"); return SKX_EINVAL; }
    if (c->ctx) SKX_HIP(hipSetDevice(c->ctx->device));
    return comm_allreduce_u32(c, buf, n, on_device != 0);
    });
}

extern "C" int skx_comm_gather_root(skx_comm *c, const void *send, const uint64_t *sizes, void *recv)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check(c));
    if (!sizes || (!send && sizes[c->rank]) || (c->rank == 0 && !recv)) { set_error("bad arguments"); return SKX_EINVAL; }
    if (!c->local()) { SKX_TRY(check_dev(c)); SKX_HIP(hipSetDevice(c->ctx->device)); }
    std::vector<size_t> sz(sizes, sizes + c->world);
    return comm_gather_root_host(c, send, sz, recv);
    });
}

// ------------------------------------------------------------------------------------------ partitioning (pure host arithmetic)
// contiguous shard of rank: input order is kept, so names come out in CLI order (cf. the offsets of merge_ska_dict.rs:243-253,277-291)
extern "C" int skx_shard_range(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi)
{
    if (world < 1 || rank < 0 || rank >= world || !lo || !hi) { set_error("bad arguments"); return SKX_EINVAL; }
    const uint64_t base = n_items / world, rem = n_items % world;
    *lo = (uint64_t)rank * base + std::min<uint64_t>(rank, rem);
    *hi = *lo + base + ((uint64_t)rank < rem ? 1 : 0);
    return SKX_OK;
}
// rows of the pair matrix dealt to ranks: contiguous bands [lo, hi) of first samples, starts on multiples of `align`, about the same
// number of pairs (i, j > i) each (row i holds n - 1 - i of them); the (i < j) row-major order of merge_ska_array.rs:416-438 is kept
extern "C" int skx_pair_bands(int n_samples, int world, int align, int *lo_hi)
{
    if (n_samples < 0 || world < 1 || align < 1 || !lo_hi) { set_error("bad arguments"); return SKX_EINVAL; }
    const int64_t n = n_samples, total = n * (n - 1) / 2;
    auto before = [&](int64_t h) { return h * (n - 1) - h * (h - 1) / 2; };          // pairs in rows [0, h)
    int64_t lo = 0;
    for (int r = 0; r < world; r++) {
        int64_t hi;
        if (r == world - 1) hi = n;
        else {
            const int64_t want = total * (r + 1) / world;
            int64_t h = lo;
            while (h < n && before(h) < want) h++;
            const int64_t down = h / align * align, up = std::min<int64_t>(n, (h + align - 1) / align * align);
            hi = (down >= lo && std::llabs(before(down) - want) <= std::llabs(before(up) - want)) ? down : up;
        }
        hi = std::max(hi, lo);
        lo_hi[2 * r] = (int)lo; lo_hi[2 * r + 1] = (int)hi;
        lo = hi;
    }
    return SKX_OK;
}

// ------------------------------------------------------------------------------------------ exchange 1: key tables -> global rows
extern "C" int skx_keyset_allgather(skx_comm *c, skx_keyset *local, skx_keyset **rows)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check_dev(c));
    if (!local || !rows) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = c->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    if (local->logN >= 0) SKX_TRY(keyset_flatten(local));
    const int wpk = local->wpk();
    // sizes first (and k / strand use: the reference's merge panics on a mismatch, merge_ska_dict.rs:166-173)
    const uint64_t mine[3] = {local->total, (uint64_t)local->k, (uint64_t)local->rc};
    std::vector<uint64_t> all(3 * (size_t)c->world);
    SKX_TRY(comm_allgather_host(c, mine, all.data(), sizeof mine));
    uint64_t mx = 1;
    std::vector<uint64_t> h_off(c->world + 1, 0); std::vector<uint32_t> h_cnt(c->world);
    for (int r = 0; r < c->world; r++) {
        if (all[3 * r + 1] != (uint64_t)local->k) { set_error("K-mer lengths do not match: %d %d", (int)all[3 * r + 1], local->k); return SKX_EINVAL; }
        if (all[3 * r + 2] != (uint64_t)local->rc) { set_error("Strand use inconsistent"); return SKX_EINVAL; }
        if (all[3 * r] > 0xFFFFFFFFull) { set_error("keyset too large"); return SKX_EUNSUP; }
        mx = std::max(mx, all[3 * r]); h_cnt[r] = (uint32_t)all[3 * r];
    }
    for (int r = 0; r <= c->world; r++) h_off[r] = (uint64_t)r * mx;             // tables padded to the longest: one collective
    const uint64_t slot = mx * (uint64_t)wpk;                                      // 64-bit words per rank
    DevBuf<uint64_t> gathered; SKX_TRY(gathered.alloc(slot * (uint64_t)c->world));
    if (c->world == 1) {
        SKX_HIP(hipMemcpyAsync(gathered.p, local->flat.p, local->total * 8 * wpk, hipMemcpyDeviceToDevice, st));
    } else {
        DevBuf<uint64_t> padded; SKX_TRY(padded.alloc(slot)); SKX_TRY(padded.zero(st));
        SKX_HIP(hipMemcpyAsync(padded.p, local->flat.p, local->total * 8 * wpk, hipMemcpyDeviceToDevice, st));
        PhaseTimer pt("comm.key_table_allgather");
        SKX_TRY(comm_allgather(c, padded.p, gathered.p, slot * 8, true));
        SKX_HIP(hipStreamSynchronize(st));
    }
    SKX_TRY(keyset_union_tables(ctx, gathered.p, h_off, h_cnt, local->k, local->rc, rows));
    // the notes of the rank's own union pass (skx_keyset_union took them) move to the global rows: own row -> global row per own sub-bucket
    skx_keyset *g = *rows;
    const uint16_t *l_perm = local->pieces ? local->pieces->perm.p : local->perm.p;      // notes of the union pass, or pieces of the append pass
    if (((local->side.p && local->side_of) || local->pieces) && l_perm && !local->wide && g->logN >= local->logN && local->logN >= 0) {
        const uint64_t nsub = 1ull << local->logN;
        DevBuf<int> d_bad; DevBuf<uint32_t> d_max;
        SKX_TRY(d_bad.alloc(1)); SKX_TRY(d_bad.zero(st)); SKX_TRY(d_max.alloc(1)); SKX_TRY(d_max.zero(st));
        SKX_TRY(g->g_perm.alloc(nsub * local->stride)); SKX_TRY(g->g_n.alloc(nsub)); SKX_TRY(g->g_base.alloc(nsub));
        launch_compose_perm(local->stage.p, local->stride, local->ncnt.p, l_perm, local->logN, g->stage.p, g->stride, g->ncnt.p, g->roff.p, g->logN,
                            2 * (local->k - 1), g->g_perm.p, g->g_n.p, g->g_base.p, d_max.p, d_bad.p, st);
        int bad = 0; uint32_t gmax = 0;
        SKX_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipMemcpyAsync(&gmax, d_max.p, 4, hipMemcpyDeviceToHost, st));
        SKX_HIP(hipStreamSynchronize(st));
        SKX_HIP(hipGetLastError());
        if (!bad && gmax > 0 && local->pieces) {
            g->pieces = local->pieces; g->pieces_of = local->pieces_of; g->pieces_of_id = local->pieces_of_id; local->pieces = nullptr; local->pieces_of = nullptr;
            g->l_logN = local->logN; g->l_stride = local->stride; g->g_max = gmax;
        } else if (!bad && gmax > 0) {
            g->side = std::move(local->side); g->side_of = local->side_of; local->side_of = nullptr;
            g->l_logN = local->logN; g->l_stride = local->stride; g->g_max = gmax;
        } else { g->g_perm.release(); g->g_n.release(); g->g_base.release(); }      // (the two-read assemble takes it)
    }
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ exchange 2: per-row filter statistics
namespace {
__global__ void pack_counts_kernel(const uint32_t *present, const uint32_t *unambig, uint32_t *packed, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) packed[i] = present[i] + (unambig[i] << 16);
}
__global__ void unpack_counts_kernel(const uint32_t *packed, uint32_t *present, uint32_t *unambig, uint32_t *vcount, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t p = packed[i] & 0xFFFFu; present[i] = p; unambig[i] = packed[i] >> 16; vcount[i] = p; }
}
__global__ void copy_u32_kernel(const uint32_t *src, uint32_t *dst, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void narrow_mask_kernel(const uint32_t *mask, uint16_t *m16, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m16[i] = (uint16_t)mask[i];
}
__global__ void or_masks_kernel(const uint16_t *parts, uint32_t *mask, uint64_t n, int world)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m = 0;
    for (int r = 0; r < world; r++) m |= parts[(uint64_t)r * n + i];
    mask[i] = m;
}
inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace

// Global per-row statistics of a column slab: the counts add, the code sets OR (RCCL has no bitwise reduction, so the 16-bit sets are
// all-gathered -- two bytes per row and rank -- and OR-ed locally).  While the whole job has fewer than 32 768 samples both counts
// travel in one all-reduce, 15 bits each (no reliance on wrap-around).  variant_count becomes the global count of cells != '-'
// (merge_ska_array.rs:172) and the array learns the job's sample count for the filter's thresholds and gap test.
extern "C" int skx_array_reduce_stats(skx_comm *c, skx_array *a, uint64_t total_samples)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check_dev(c));
    if (!a || a->ctx != c->ctx) { set_error("bad arguments"); return SKX_EINVAL; }
    skx_ctx *ctx = c->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    uint32_t *pp, *pu, *pm, *pv;
    SKX_TRY(skx_array_device_stats(a, &pp, &pu, &pm, &pv));           // a lazily held array stays one: statistics-only pass
    const uint64_t U = a->n_rows;
    a->total_samples = total_samples;
    if (!U) return SKX_OK;
    PhaseTimer pt("comm.row_stats");
    if (total_samples > 0 && total_samples <= 0x7FFF) {
        DevBuf<uint32_t> packed; SKX_TRY(packed.alloc(U));
        hipLaunchKernelGGL(pack_counts_kernel, dim3(grid_for(U)), dim3(256), 0, st, pp, pu, packed.p, U);
        SKX_TRY(comm_allreduce_u32(c, packed.p, U, true));
        hipLaunchKernelGGL(unpack_counts_kernel, dim3(grid_for(U)), dim3(256), 0, st, packed.p, pp, pu, pv, U);
        SKX_HIP(hipStreamSynchronize(st));                              // `packed` goes out of scope
    } else {
        SKX_TRY(comm_allreduce_u32(c, pp, U, true));
        SKX_TRY(comm_allreduce_u32(c, pu, U, true));
        hipLaunchKernelGGL(copy_u32_kernel, dim3(grid_for(U)), dim3(256), 0, st, pp, pv, U);
    }
    if (c->world > 1) {
        DevBuf<uint16_t> m16, parts; SKX_TRY(m16.alloc(U)); SKX_TRY(parts.alloc(U * (uint64_t)c->world));
        hipLaunchKernelGGL(narrow_mask_kernel, dim3(grid_for(U)), dim3(256), 0, st, pm, m16.p, U);
        SKX_TRY(comm_allgather(c, m16.p, parts.p, U * 2, true));
        hipLaunchKernelGGL(or_masks_kernel, dim3(grid_for(U)), dim3(256), 0, st, parts.p, pm, U, c->world);
        SKX_HIP(hipStreamSynchronize(st));
    }
    SKX_HIP(hipStreamSynchronize(st));
    SKX_HIP(hipGetLastError());
    return SKX_OK;
    });
}

// ------------------------------------------------------------------------------------------ exchange 3: bit planes -> pair bands
// All-vs-all distances of a job whose samples are sharded over ranks (MergeSkaArray::distance, merge_ska_array.rs:416-438,587-632).
// `a` = this rank's column slab over the globally filtered rows (skx_array_reduce_stats + skx_array_filter on every rank: same rows
// everywhere).  Every rank builds the planes of its own samples, one all-gather replicates them, each rank finishes a band of first
// samples against every later sample, rank 0 receives the bands: out[n_out] there (n_out >= S (S - 1) / 2, pairs (i < j) row-major).
extern "C" int skx_array_distance_sharded(skx_comm *c, skx_array *a, int filt_ambig, double constant, skx_dist *out, uint64_t n_out)
{
    return skx_guarded([&]() -> int {
    SKX_TRY(check_dev(c));
    if (!a || a->ctx != c->ctx) { set_error("bad arguments"); return SKX_EINVAL; }
    if (c->world > 255) { set_error("more than 255 ranks: the per-row agreement of the sharded distance sums one byte per row over the ranks"); return SKX_EUNSUP; }
    skx_ctx *ctx = c->ctx; hipStream_t st = ctx->stream;
    SKX_HIP(hipSetDevice(ctx->device));
    // --allow-ambiguous: the rows without an ambiguous cell in ANY rank's samples (the ranks agree on them through one all-reduce of a byte
    // per row) travel as 4 planes and go through the three-count sweep, only the others as 8 planes through the twelve-class one
    const bool split = !filt_ambig;
    const void *lp = nullptr; uint64_t wpr = 0; int np = 0;
    DevBuf<uint64_t> lp_clean, lp_dirty; uint64_t wpr_c = 1, wpr_d = 1, rows_c = 0, rows_d = 0;
    if (!split) SKX_TRY(skx_array_distance_planes(a, filt_ambig, &lp, &wpr, &np));
    else {
        SKX_TRY(array_materialize(a));
        const uint64_t U = a->n_rows, U4 = (U + 3) / 4 * 4;
        DevBuf<uint8_t> clean, dirty;
        SKX_TRY(clean.alloc(U4 + 4)); SKX_TRY(dirty.alloc(U4 + 4)); SKX_TRY(dirty.zero(st));
        launch_split_keep(nullptr, a->mask.p, U, clean.p, dirty.p, st);
        if (c->world > 1 && U) SKX_TRY(comm_allreduce_u32(c, reinterpret_cast<uint32_t *>(dirty.p), U4 / 4, true));       // a byte per row, summed: < 256 ranks
        launch_split_keep_from_dirty(dirty.p, U, clean.p, st);                     // dirty: != 0 -> 1; clean = !dirty
        SKX_TRY(planes_of_kept_rows(a, clean.p, 1, lp_clean, wpr_c, rows_c));
        // what the split rests on, checked as the single process checks it (present == unambiguous on the rows passed off as clean), by all ranks
        // together: statistics that missed a code on any of them send every row through the twelve-class sweep on all of them
        uint64_t differ = 0;
        if (rows_c) {
            DevBuf<int> d_flag; SKX_TRY(d_flag.alloc(1)); SKX_TRY(d_flag.zero(st));
            launch_differ_u32((const uint32_t *)lp_clean.p, (const uint32_t *)(lp_clean.p + (uint64_t)a->names.size() * wpr_c), (uint64_t)a->names.size() * wpr_c * 2, d_flag.p, st);
            int df = 0;
            SKX_HIP(hipMemcpyAsync(&df, d_flag.p, 4, hipMemcpyDeviceToHost, st));
            SKX_HIP(hipStreamSynchronize(st));
            differ = df ? 1 : 0;
        }
        std::vector<uint64_t> differs((size_t)c->world);
        SKX_TRY(comm_allgather_host(c, &differ, differs.data(), sizeof differ));
        for (uint64_t d : differs) differ |= d;
        if (differ || knob("stale_row_mask")) {
            launch_split_keep(nullptr, a->mask.p, U, clean.p, dirty.p, st, 1);
            lp_clean.release(); rows_c = 0; wpr_c = 1;
        }
        SKX_TRY(planes_of_kept_rows(a, dirty.p, 0, lp_dirty, wpr_d, rows_d));
        SKX_HIP(hipStreamSynchronize(st));
    }
    const uint64_t s_loc = a->names.size();
    const uint64_t mine[3] = {s_loc, split ? wpr_c : wpr, split ? wpr_d : 0};
    std::vector<uint64_t> all(3 * (size_t)c->world);
    SKX_TRY(comm_allgather_host(c, mine, all.data(), sizeof mine));
    uint64_t S = 0, mx = 1;
    for (int r = 0; r < c->world; r++) {
        if (all[3 * r + 1] != mine[1] || all[3 * r + 2] != mine[2]) { set_error("ranks disagree on the filtered rows (%llu vs %llu words per sample)", (unsigned long long)all[3 * r + 1], (unsigned long long)mine[1]); return SKX_EINVAL; }
        S += all[3 * r]; mx = std::max(mx, all[3 * r]);
    }
    if (S > 0x7FFFFFFF) { set_error("too many samples"); return SKX_EUNSUP; }
    // planes[p][sample][word] over all samples, samples in rank order
    auto gather = [&](const void *lpl, int npl, uint64_t w, DevBuf<uint64_t> &planes) -> int {
        SKX_TRY(planes.alloc((uint64_t)npl * S * w));
        if (c->world == 1) { SKX_HIP(hipMemcpyAsync(planes.p, lpl, (uint64_t)npl * S * w * 8, hipMemcpyDeviceToDevice, st)); return SKX_OK; }
        PhaseTimer pt("comm.planes_allgather");
        const uint64_t slot = (uint64_t)npl * mx * w;                // ranks may hold different numbers of samples: padded to the largest
        DevBuf<uint64_t> padded, gathered; SKX_TRY(padded.alloc(slot)); SKX_TRY(gathered.alloc(slot * (uint64_t)c->world)); SKX_TRY(padded.zero(st));
        SKX_HIP(hipMemcpy2DAsync(padded.p, mx * w * 8, lpl, s_loc * w * 8, s_loc * w * 8, npl, hipMemcpyDeviceToDevice, st));
        SKX_TRY(comm_allgather(c, padded.p, gathered.p, slot * 8, true));
        uint64_t s0 = 0;
        for (int r = 0; r < c->world; r++) {
            const uint64_t sr = all[3 * r];
            if (sr) SKX_HIP(hipMemcpy2DAsync(planes.p + s0 * w, S * w * 8, gathered.p + (uint64_t)r * slot, mx * w * 8, sr * w * 8, npl, hipMemcpyDeviceToDevice, st));
            s0 += sr;
        }
        SKX_HIP(hipStreamSynchronize(st));
        return SKX_OK;
    };
    DevBuf<uint64_t> planes, planes_c, planes_d;
    if (!split) SKX_TRY(gather(lp, np, wpr, planes));
    else {
        if (rows_c && lp_clean.p) SKX_TRY(gather(lp_clean.p, 4, wpr_c, planes_c));
        if (rows_d) SKX_TRY(gather(lp_dirty.p, 8, wpr_d, planes_d));
        lp_clean.release(); lp_dirty.release();
    }
    std::vector<int> bands(2 * (size_t)c->world);
    SKX_TRY(skx_pair_bands((int)S, c->world, 8, bands.data()));
    auto pairs_in = [&](int lo, int hi) { uint64_t n = 0; for (int i = lo; i < hi; i++) n += S - 1 - (uint64_t)i; return n; };
    std::vector<size_t> sizes(c->world);
    uint64_t total_pairs = 0;
    for (int r = 0; r < c->world; r++) { const uint64_t n = pairs_in(bands[2 * r], bands[2 * r + 1]); sizes[r] = n * sizeof(skx_dist); total_pairs += n; }
    if (c->rank == 0 && (!out || n_out < total_pairs)) { set_error("distance table too small"); return SKX_EINVAL; }
    const int lo = bands[2 * c->rank], hi = bands[2 * c->rank + 1];
    std::vector<skx_dist> band;
    skx_dist *mine_out = c->rank == 0 ? out : (band.resize(sizes[c->rank] / sizeof(skx_dist) + 1), band.data());
    {
        PhaseTimer pt("distance.pair_sweep");
        if (hi > lo && !split) SKX_TRY(planes_distance(ctx, planes.p, (int)S, wpr, filt_ambig, constant, lo, hi, mine_out));
        if (hi > lo && split) SKX_TRY(planes_distance_split(ctx, planes_c.p, wpr_c, rows_c, planes_d.p, wpr_d, rows_d, (int)S, constant, lo, hi, mine_out));
    }
    PhaseTimer pt("comm.pairs_to_rank0");
    return comm_gather_root_host(c, mine_out, sizes, out);
    });
}
