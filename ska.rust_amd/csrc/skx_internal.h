// skx_internal.h -- private host-side types shared by the C-ABI translation units.
#pragma once
#include "../../include/skx.h"
#include "skx_device.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <sys/types.h>
#include <stdexcept>
#include <string>
#include <vector>

namespace skx {

void set_error(const char *fmt, ...);

int hip_fail(hipError_t e, const char *what);       // sets the error, returns SKX_ENODEV / SKX_ENOMEM

#define SKX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return skx::hip_fail(e_, #call); } while (0)
#define SKX_TRY(call) do { int r_ = (call); if (r_ != SKX_OK) return r_; } while (0)

// wall-clock phases of the host path (file reading, uploads, codec, writers): accumulated per name in first-use order, printed
// as they are recorded when SKX_DEBUG is set, returned as JSON by skx_phases_json (bench.py's end_to_end leg reads them)
void phase_add(const char *name, double secs);
struct PhaseTimer {
    const char *name; std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    double stop() { double s = 0; if (name) { s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); phase_add(name, s); name = nullptr; } return s; }
    ~PhaseTimer() { stop(); }
};

// caching device allocator (per process): freed blocks are kept and reused for later requests of a similar size,
// so steady-state batches do not go through hipMalloc/hipFree (which map/unmap tens of GB and cost seconds)
void *dev_alloc(size_t bytes, hipError_t *err);
void dev_free(void *p);
void dev_trim();                                   // release every cached block

template <typename T>
struct DevBuf {
    T *p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) dev_free(p); p = nullptr; n = 0; }
    int alloc(size_t count) {
        release();
        if (!count) count = 1;
        hipError_t e = hipSuccess;
        p = (T *)dev_alloc(count * sizeof(T), &e);
        if (!p) return hip_fail(e, "hipMalloc");
        n = count; return SKX_OK;
    }
    int zero(hipStream_t st) { return p ? (hipMemsetAsync(p, 0, n * sizeof(T), st) == hipSuccess ? SKX_OK : SKX_ENODEV) : SKX_OK; }
};

}  // namespace skx

namespace skx {
// allocates the page-cache pages of a file range ahead of the writer (posix_fallocate in steps on its own thread): the target can
// be raised while the data that will fill the range is still being produced
struct Preallocator {
    int fd; off_t base;
    std::mutex mu; std::condition_variable cv; uint64_t target = 0, done = 0; bool stop = false;
    std::atomic<bool> failed{false};       // posix_fallocate refused (ENOSPC, quota): the range must not be written through a mapping
    std::thread th;
    Preallocator(int fd_, off_t base_);
    ~Preallocator();
    void raise(uint64_t bytes);            // pages of [base, base + bytes) wanted
    void finish(uint64_t bytes);           // raise, then wait until they are there
};
}  // namespace skx

struct skx_ctx {
    int expect_fd = -1;              // skx_ctx_expect_output: where the next alignment goes (pages allocated while its rows are read)
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};      // [0..1] a stage, [2..3] one kernel inside it
    skx_timings tm{};
    bool timing = true;
    std::string merge_path;          // which kernels the last merge on this context went through, and why (skx_ctx_merge_path)
};

namespace skx { uint64_t next_object_id(); }
struct skx_dictset {
    const uint64_t id = skx::next_object_id();   // what a key set's pieces / notes are matched to their dictset by (an address can be reused)
    skx_ctx *ctx = nullptr;
    int n = 0, k = 0, rc = 0, logB = 0, key_bits = 64;
    skx::HashParams hp{};
    skx::WideHash wh{};              // k > 31
    bool wide() const { return key_bits == 128; }
    skx::DevBuf<uint64_t> words;     // all (sample,bucket) regions (2 x u64 per word when wide)
    skx::DevBuf<uint64_t> off;       // [n<<logB + 1]
    skx::DevBuf<uint32_t> raw;       // [n<<logB] windows per region
    skx::DevBuf<uint32_t> ucnt;      // [n<<logB] distinct split k-mers per region
    skx::DevBuf<uint16_t> sidx;      // [n<<logB][16] sub-index of every region (narrow keys, device-built dictionaries)
    int sb = 0;
    std::vector<uint64_t> sample_size;   // SkaDict::ksize per sample (sorted dictionaries)
    // Assemblies are kept as the extraction kernel scattered them (regions unsorted, repeats not folded) until something needs a sample's
    // SkaDict as such: MergeSkaDict::append reads the raw regions (skx_append.hip), skx::dictset_sort makes the sorted form in place.
    bool sorted = true;
    uint32_t region_cap = 0; uint64_t maxlen = 0;        // what the later sort needs: words per region, the longest record stream
    std::vector<uint64_t> raw_total;     // windows per sample (raw form): 0 = the sample has no valid sequence
    skx::DictView view() const { return skx::DictView{words.p, off.p, ucnt.p, n, logB, wide() ? wh.bits : hp.bits, sidx.p, sb}; }
};

// An array as MergeSkaDict::append leaves it (skx_append.hip): per row block and sample a piece of 4-bit base sets indexed by first-seen
// rank; the rows x samples cells in the order of H are produced from the pieces on demand (all rows, a window, or the rows a filter keeps).
struct skx_pieces {
    skx::DevBuf<uint8_t> data;       // [1 << logQ][S][cap / 2]
    skx::DevBuf<uint16_t> plen;      // [1 << logQ][S] ranks a piece holds
    skx::DevBuf<uint16_t> perm;      // [1 << logQ][cap] first-seen rank -> row of the block (0xFFFF: none)
    skx::DevBuf<uint32_t> nrank;     // [1 << logQ] ranks handed out
    uint32_t cap = 0; int logQ = 0;
    std::vector<uint64_t> sample_cells;   // SkaDict::ksize per sample
};

struct skx_keyset {
    skx_ctx *ctx = nullptr;
    int k = 0, rc = 0, logN = 0;
    skx::HashParams hp{};
    skx::WideHash wh{};
    bool wide = false;               // k > 31: 2 x u64 per key
    int wpk() const { return wide ? 2 : 1; }
    uint32_t stride = 0, max_rows = 0;
    uint64_t total = 0;
    skx::DevBuf<uint64_t> stage;     // slab j at stage + j*stride (sorted, engine order)
    skx::DevBuf<uint32_t> ncnt;      // [1<<logN]
    skx::DevBuf<uint64_t> roff;      // [1<<logN + 1]
    skx::DevBuf<uint64_t> flat;      // lazily built compact copy (engine order words)
    std::vector<uint64_t> h_roff;    // host copy of roff (windows of a lazily held array)
    // notes of the union pass for the assemble that follows it on the same dictionaries (skx_merge): where every word's key went
    skx::DevBuf<uint16_t> side;      // [side_of->words.n] (first-seen rank in the sub-bucket << 4) | base set
    skx::DevBuf<uint16_t> perm;      // [1 << logN][stride] first-seen rank -> row of the slab
    const skx_dictset *side_of = nullptr;
    // the same notes carried over to the rows of a sharded job (skx_keyset_allgather): they were taken against the rank's OWN rows; per own
    // sub-bucket, g_perm maps a first-seen rank to its row among the GLOBAL rows of that sub-bucket's hash range, which start at row g_base
    // and number g_n (skx_array_assemble then fills the rank's columns over the global rows without reading its dictionaries again)
    skx::DevBuf<uint16_t> g_perm;    // [1 << l_logN][l_stride]
    skx::DevBuf<uint32_t> g_n;       // [1 << l_logN]
    skx::DevBuf<uint64_t> g_base;    // [1 << l_logN]
    int l_logN = -1; uint32_t l_stride = 0, g_max = 0;
    // or the pieces of an append pass over `pieces_of` (skx_keyset_union_notes on assemblies kept as extracted): they travel like the notes
    skx_pieces *pieces = nullptr; const skx_dictset *pieces_of = nullptr; uint64_t pieces_of_id = 0;
    bool holds_pieces_of(const skx_dictset *d) const { return pieces && pieces_of == d && d && pieces_of_id == d->id; }
    skx_keyset() = default;
    skx_keyset(const skx_keyset &) = delete; skx_keyset &operator=(const skx_keyset &) = delete;
    ~skx_keyset() { delete pieces; }
};

struct skx_array {
    skx_ctx *ctx = nullptr;
    int k = 0, rc = 0, k_bits = 64;
    skx::HashParams hp{};
    std::vector<std::string> names;
    std::string version;
    uint64_t n_kmers = 0;            // split_kmers.len()
    uint64_t n_rows = 0;             // variants.nrows()
    uint64_t pitch = 0;
    uint64_t total_samples = 0;      // samples over all ranks when this array is one column slab (0: names.size())
    bool engine_order = false;       // rows sorted by H(key)
    skx::WideHash wh{};
    skx::DevBuf<uint64_t> keys;      // [n_kmers] packed words (H(key)<<4 | 1); 2 x u64 per key when k > 31 and built on the device
    skx::DevBuf<uint8_t> matrix;     // [n_samples][pitch], sample-major
    skx::DevBuf<uint32_t> present, unambig, mask;   // per row statistics of the matrix
    skx::DevBuf<uint32_t> vcount;                   // variant_count as the reference stores it (merge_ska_array.rs:121)
    // 128-bit keys of arrays loaded from k>31 files are kept on the host (filter/align/distance never touch them)
    std::vector<skx_key> host_keys;
    skx::DevBuf<uint64_t> planes;    // bit planes of the distance sweep (skx_array_distance_planes)
    bool keys_absent = false;        // loaded through skx_array_load_filtered: the split k-mer list was stepped over
    std::shared_ptr<skx::Preallocator> prealloc;   // output pages being allocated for skx_array_write_fasta (skx_ctx_expect_output)
    // Lazily held (the build path): rows, keys and names are known but the rows x samples matrix has not been
    // assembled; the dictionaries and the row keyset are kept instead.  `ska build` streams such an array into its .skf window
    // by window and `ska align *.fa` filters it before any cell is written, so neither ever holds the unfiltered matrix;
    // every other operation assembles it first (skx::array_materialize).
    // Or (arrays merged by the append pass): the pieces and the row blocks (lazy_rows: ncnt / roff), statistics ready.
    skx_dictset *lazy_dict = nullptr; skx_keyset *lazy_rows = nullptr; skx_pieces *pieces = nullptr;
    bool stats_ready = true;         // present / unambig / mask / vcount filled (lazy arrays get them from a statistics-only pass)
    bool lazy() const { return lazy_dict != nullptr || pieces != nullptr; }
    void drop_lazy() { delete lazy_dict; delete lazy_rows; delete pieces; lazy_dict = nullptr; lazy_rows = nullptr; pieces = nullptr; }
    ~skx_array() { drop_lazy(); }
};

namespace skx {
// host reader (fastx.cpp): one sample -> record stream; returns SKX_* and sets the error
struct HostStream { std::vector<uint8_t> seq, qual; bool is_fastq = false; std::vector<std::string> ids; };   // ids: FASTA record ids (header up to white space)
int read_sample_stream(const char *file1, const char *file2, double proportion_reads, HostStream &out);
// a plain FASTQ file line by line: emit(0, sequence line) / emit(1, quality line), without terminators -- the sink adds the '\n' that ends a
// record of the stream (SKF_NOT_TAKEN: not plain FASTQ)
// FASTQ text -> the read-set kernels' bit planes on the device (skx_fastq.hip); scratch that lives across a batch's samples
struct FastqScratch { DevBuf<uint32_t> tile, info, line_end, rec_seq, rec_qual, rec_len, rec_end; };
int fastq_frame_planes(skx_ctx *ctx, const uint8_t *raw, uint64_t len, uint64_t junction, int min_qual, uint64_t *planes, FastqScratch &sc, uint64_t *positions, int *irregular);
// `.fastq.gz` inflated on the device (skx_gzdev.hip; logic: gz_device.h): one file's buffers, kept across a batch's samples
struct GzDevFileInfo { uint64_t total; uint32_t status, n_members, first, last; };      // status: gzd::Status (0: the text is vouched for)
struct GzDevWork {
    DevBuf<uint64_t> sync, base, m_end; DevBuf<uint32_t> m_crc, m_acc; DevBuf<uint8_t> cinfo, members, finfo; DevBuf<uint16_t> sym, maps, gwin;
    const uint8_t *src = nullptr; uint64_t src_bytes = 0; uint32_t n_chunks = 0, n_groups = 0, chunk_bytes = 0, ratio = 0, group = 0, max_members = 0;
};
int gz_device_reserve(GzDevWork &wk, uint64_t bytes, uint64_t text_hint);
int gz_device_decode(skx_ctx *ctx, hipStream_t st, const uint8_t *src, uint64_t bytes, uint64_t text_hint, GzDevWork &wk);
int gz_device_text(skx_ctx *ctx, hipStream_t st, GzDevWork &wk, uint8_t *dst, uint64_t total, uint32_t n_members);
int stream_fastq_file(const char *path, const std::function<int(int which, const uint8_t *p, size_t n)> &emit);
// gzip members inflated piece by piece into a window of the reader's own (gz_inflate.cpp: the reader threads' inflater).  next(): more text,
// in place -- the `keep` bytes in front of the last call's end stay in front of it (the caller's unfinished line); *n == 0 at the end of the
// last member; -1: not gzip, damaged, truncated (CRC-32 and length of every member are checked) or a read error
struct GzReader {
    struct Impl; Impl *impl;
    GzReader(); ~GzReader();
    GzReader(const GzReader &) = delete; GzReader &operator=(const GzReader &) = delete;
    void open(int fd);                                        // (the descriptor stays the caller's)
    int next(const uint8_t **p, size_t *n, size_t keep);
    static constexpr size_t KEEP_MAX = 128u << 10;            // callers put longer unfinished lines aside themselves
};
uint32_t gz_crc32(uint32_t crc, const uint8_t *p, size_t n);  // zlib's crc32() convention
// a sequence / quality line of a read as bit planes, 64 positions per word, bits beyond the line zero (fastx.cpp; READ_GROUP_BYTES: five
// 64-position words -- lo, hi, bad, newline, quality verdict -- make one group of the packed stream, launch_expand_planes takes it apart)
void pack_bases_planes(const uint8_t *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *bad);
void pack_qual_plane(const uint8_t *q, size_t n, int min_qual, uint64_t *qb);
constexpr size_t READ_GROUP_BYTES = 40;
// FASTQ sample -> sorted unique packed words (skx_reads.hip)
int reads_sample_dict(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q,
                      DevBuf<uint64_t> &out_words, uint64_t *n_out);
// the same through the engine's own partition kernels (skx_reads2.hip): the packed words of the windows that enter the dictionary,
// unsorted and with duplicates; SKF_NOT_TAKEN leaves the sample to reads_sample_dict
int reads_windows(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q, DevBuf<uint64_t> &hash,
                  DevBuf<uint64_t> &wlo, DevBuf<uint64_t> &whi, DevBuf<uint8_t> &flag, unsigned long long *d_n_valid, bool want_words, const uint64_t *planes,
                  DevBuf<uint16_t> &rec_t, DevBuf<uint32_t> &tile_cnt);
int reads_tile();                      // positions a workgroup of the window pass takes (= the first partition pass's tile)
void launch_words_rebuild(const uint32_t *pos, uint64_t n, const uint8_t *seq, uint64_t len, int k, int rc, uint64_t *out_lo, uint64_t *out_hi, hipStream_t st);
void launch_words_rebuild_planes(const uint32_t *pos, uint64_t n, const uint64_t *planes, int k, int rc, uint64_t *out_lo, uint64_t *out_hi, hipStream_t st);
int reads_sample_words(skx_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t len, int k, int rc, const skx_qual &q,
                       DevBuf<uint64_t> &out_lo, DevBuf<uint64_t> &out_hi, uint64_t *n_out, const uint64_t *planes = nullptr);
void launch_words_regions(bool count, const uint64_t *in_lo, const uint64_t *in_hi, uint64_t n, int bits, int logB, uint64_t region0, uint32_t *raw,
                          const uint64_t *off, uint32_t *cursor, uint64_t *words, hipStream_t st);
// `ska cov`: occurrence-count histogram of the split k-mers of a read stream (skx_reads.hip); d_hist[1000] zeroed by the caller
int cov_histogram(skx_ctx *ctx, const uint8_t *d_seq, uint64_t len, int k, int rc, uint32_t *d_hist);
// `ska map` helpers (skx_reads.hip)
int ref_windows(skx_ctx *ctx, const uint8_t *d_seq, uint64_t len, int k, int rc, DevBuf<uint64_t> &wlo, DevBuf<uint64_t> &whi, DevBuf<uint8_t> &flag);
int sort_fold_words(skx_ctx *ctx, const uint64_t *alo, const uint64_t *ahi, uint64_t m, DevBuf<uint64_t> &out_words, uint64_t *n_out);   // ahi: nullptr for 64-bit keys
void launch_gather_regions(const uint64_t *words, const uint64_t *off, const uint32_t *raw, const uint64_t *dst_off, uint64_t r0, uint64_t n_regions, int wpk,
                           uint64_t *lo, uint64_t *hi, hipStream_t st);
int ref_repeat_flags(skx_ctx *ctx, const uint64_t *wlo, const uint64_t *whi, const uint8_t *flag, uint64_t len, DevBuf<uint8_t> &rep);   // whi: nullptr for k <= 31
int select_mapped(const uint32_t *row, uint64_t len, DevBuf<uint32_t> &mapped, uint64_t *m, hipStream_t st);
int sort_words_perm(const uint64_t *words, uint64_t n, DevBuf<uint64_t> &sorted, DevBuf<uint32_t> &perm, hipStream_t st);
int sort_unique_wide(const u128 *in, uint64_t n, DevBuf<uint64_t> &out, uint64_t *n_out, hipStream_t st);
int sort_wide_perm(const u128 *words, uint64_t n, DevBuf<uint64_t> &sorted, DevBuf<uint32_t> &perm, hipStream_t st);
// the engine's own device primitives (skx_prims.hip): stable radix sorts (`bits` significant key bits, 8 a pass; the input is left as it is),
// inclusive scans, order-keeping selections, unique on sorted keys, exclusive OR-scan within runs of equal keys.  Each returns with the stream idle.
int prim_sort_keys_u64(const uint64_t *in, uint64_t *out, uint64_t n, int bits, hipStream_t st);
int prim_sort_pairs_u64(const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int bits, hipStream_t st);
int prim_sort_keys_u128(const u128 *in, u128 *out, uint64_t n, int bits, hipStream_t st);
int prim_sort_pairs_u128(const u128 *kin, u128 *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int bits, hipStream_t st);
int prim_scan_add_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st);
int prim_scan_max_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st);
int prim_select_index_u8(const uint8_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st);       // the positions of the flagged items
int prim_select_index_u32(const uint32_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st);
int prim_select_u32(const uint32_t *in, const uint8_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st);
int prim_unique_u64(const uint64_t *sorted, uint64_t *out, uint64_t n, uint64_t *count, hipStream_t st);
int prim_unique_keys_u128(const u128 *sorted, u128 *out, uint64_t n, uint64_t *count, hipStream_t st);             // equal = equal above the 4 base-set bits
int prim_seg_exscan_or_u64(const uint32_t *keys, const uint64_t *vals, uint64_t *out, uint64_t n, hipStream_t st);
// sorted duplicate-free copy of packed words (skx_setops.hip)
int sort_unique_words(const uint64_t *in, uint64_t n, DevBuf<uint64_t> &out, uint64_t *n_out, hipStream_t st);
// .skf codec (skf_codec.cpp)
struct SkfData {
    int k = 0, rc = 0, k_bits = 64; std::vector<std::string> names; std::string version;
    std::vector<skx_key> keys; uint64_t n_rows = 0; std::vector<uint8_t> variants; std::vector<uint64_t> counts;
};
int skf_peek_k(const char *path);
int skf_read(const char *path, SkfData &out);
int skf_write(const char *path, const SkfData &in);
// streaming forms: the matrix moves in row blocks (row-major [nrows][n_samples]), the snappy chunks are
// (de)compressed by `threads` host threads (0 = all cores)
struct SkfMeta { int k = 0, rc = 0, k_bits = 64; std::vector<std::string> names; std::string version; uint64_t n_rows = 0; };
typedef std::function<int(uint64_t row0, uint64_t nrows, uint8_t *dst)> RowFetch;
typedef std::function<int(uint64_t row0, uint64_t nrows, const uint8_t *src)> RowSink;
// Optional device hooks for the `variants` data section (2 bytes per cell; skx_snappy.hip).  Decode: the compressed file image,
// its chunk directory (uoff = offset of the chunk's output in the uncompressed stream) and the section's start; SKX_OK = the
// matrix is filled, SKF_NOT_TAKEN = use the host path.  Encode: write the finished frame chunks covering stream bytes
// [uoff0, uoff0 + n_chunks * 64 KB) to f.
struct SkfChunk { size_t off, len; uint32_t ulen, crc; bool compressed; uint64_t uoff; };
constexpr int SKF_NOT_TAKEN = -1000;
constexpr int SKF_ABORTED = -1001;      // a reader of the read-set pipeline that was stopped by somebody else's failure
typedef std::function<int(const uint8_t *file, const SkfChunk *chunks, size_t n_chunks, uint64_t upos, uint64_t n_rows, uint64_t n_samples)> DevDecode;
typedef std::function<int(FILE *f, uint64_t upos, uint64_t uoff0, uint64_t n_chunks)> DevEncode;
// layout-only view of a .skf for the streaming load (`ska align x.skf`, `ska distance x.skf`): header parsed, the split k-mer
// list stepped over, the data section located; SKF_NOT_TAKEN when the file does not have the usual shape (the general reader's)
struct SkfFile {
    struct Impl; Impl *impl;
    SkfMeta m; uint64_t n_keys = 0, upos_keys = 0, upos_data = 0;
    SkfFile(); ~SkfFile();
    SkfFile(const SkfFile &) = delete; SkfFile &operator=(const SkfFile &) = delete;
    int open(const char *path);
    int read_tail(std::vector<uint32_t> &counts);           // variant_count + ska_version + k_bits (fills m.version / m.k_bits)
    const uint8_t *file() const; const SkfChunk *chunks() const;
    // the chunk directory is filled in by a walker thread: entries below n_chunks() are final and never move
    size_t n_chunks() const; bool wait_chunk(size_t idx); size_t chunk_of(uint64_t stream_offset);
    int walk_result();                                      // waits for the walk: SKX_OK, or the file's framing error
};
// optional shortcuts of skf_write_stream: the split k-mer list already as CBOR bytes (the device writes 9-byte uints), the counts
// as 32-bit values (encoded by the thread team); when set, `keys` / `counts` are ignored
struct SkfFastSections { const uint8_t *keys_cbor = nullptr; uint64_t keys_cbor_len = 0, n_keys = 0; const uint32_t *counts32 = nullptr; uint64_t n_counts = 0; };
int skf_write_stream(const char *path, const SkfMeta &m, const std::vector<skx_key> &keys, const std::vector<uint64_t> &counts,
                     const RowFetch &fetch, int threads, const DevEncode *dev = nullptr, const SkfFastSections *fast = nullptr);
int skf_read_stream(const char *path, SkfMeta &m, std::vector<skx_key> &keys, std::vector<uint64_t> &counts,
                    const std::function<int(uint64_t n_rows, uint64_t n_samples)> &begin_rows, const RowSink &sink, int threads,
                    const DevDecode *dev = nullptr);
}  // namespace skx

// helpers shared by the ABI translation units (skx_api.cpp, skx_api_io.cpp)
namespace skx {
// pieces of the single-GPU path the collective layer (skx_comm.hip) composes
int dictset_sort(skx_dictset *d);                                    // raw regions -> sorted, folded regions + sub-index (no-op when they are)
int keyset_flatten(skx_keyset *ks);                                  // ks->flat = the rows as one compact list of packed words (engine order)
int keyset_union_tables(skx_ctx *ctx, const uint64_t *words, const std::vector<uint64_t> &h_off, const std::vector<uint32_t> &h_cnt, int k, int rc, skx_keyset **out);
int planes_distance(skx_ctx *ctx, const uint64_t *planes, int S, uint64_t wpr, int filt_ambig, double constant, int i_lo, int i_hi, skx_dist *out);
// --allow-ambiguous with the rows split by whether one of their cells is ambiguous: planes_clean = 4 planes (FILT) of the rows without such a cell
// (counts filed as classes 0-2), planes_dirty = 8 planes of the others; either may be absent (nullptr / 0 rows)
int planes_distance_split(skx_ctx *ctx, const uint64_t *planes_clean, uint64_t wpr_clean, uint64_t rows_clean, const uint64_t *planes_dirty, uint64_t wpr_dirty,
                          uint64_t rows_dirty, int S, double constant, int i_lo, int i_hi, skx_dist *out);
// bit planes of the rows flagged 1 in keep (4 planes with filt, else 8): every word written; rows = how many
int planes_of_kept_rows(skx_array *a, const uint8_t *keep, int filt, DevBuf<uint64_t> &planes, uint64_t &wpr, uint64_t &rows);
int cpu_budget();                                                    // CPUs the process may keep busy (hardware threads, or the control group's quota)
int check_k(int k);                                                  // "Invalid k-mer length" (ska_dict.rs:342-344)
bool mappable_output_fd(int fd, off_t *pos);                         // regular file, read-write, not O_APPEND: can be written through a mapping
int array_wide_words(skx_array *a, DevBuf<uint64_t> &tmp, const u128 **words);   // k > 31: the rows' packed 128-bit words on the device (tmp backs them for loaded arrays)
int array_host_keys(skx_array *a, std::vector<skx_key> &hk);         // the array's split k-mers as the reference stores them, in row order
int array_materialize(skx_array *a);                                 // a lazily held array gets its matrix (no-op otherwise)
int array_lazy_stats(skx_array *a);                                  // a lazily held array gets its per-row statistics
// rows [r0, r0 + nr) of a lazily held array as a sample-major window: cell (s, r) at win + s * wpitch + (r - r0)
int array_lazy_window(skx_array *a, uint64_t r0, uint64_t nr, DevBuf<uint8_t> &buf, const uint8_t **win, uint64_t *wpitch);
inline uint64_t pitch_for(uint64_t cols) { return ((cols + 255) / 256) * 256 + 256; }
inline bool key_less(const skx_key &x, const skx_key &y) { return x.hi != y.hi ? x.hi < y.hi : x.lo < y.lo; }
inline bool key_eq(const skx_key &x, const skx_key &y) { return x.hi == y.hi && x.lo == y.lo; }
}  // namespace skx
// nothing may unwind across the C boundary
template <typename F>
inline int skx_guarded(F &&f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc &) { skx::set_error("out of host memory"); return SKX_ENOMEM; }
    catch (const std::exception &e) { skx::set_error("internal error: %s", e.what()); return SKX_EINVAL; }
    catch (...) { skx::set_error("internal error"); return SKX_EINVAL; }
}

