/*
 * skx.h -- C ABI of the MI355X-native split-k-mer engine (libskx.so).
 *
 * Drop-in boundary for the build -> merge -> align/distance path of
 * bacpop/ska.rust v0.5.2.  The reference has no FFI layer; the seam is the
 * generic Rust API that lib.rs::main calls.  Every entry
 * point below cites the reference interface it replaces.  Plain pointers and
 * sizes only; no C++/torch types.  All functions return 0 on success and a
 * non-zero SKX_E* code on failure (skx_last_error() gives the message, which
 * matches the reference's panic text where one exists); nothing throws or
 * longjmps across the boundary.  Handles own host + device memory and are
 * released by the caller.  There is NO CPU fallback: every compute entry
 * point fails with SKX_ENODEV when no gfx950 device is usable.
 */
#ifndef SKX_H
#define SKX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKX_OK        0
#define SKX_EINVAL    1   /* bad argument (bad k, mismatched k/rc, ...)          */
#define SKX_EIO       2   /* unreadable / malformed input file                   */
#define SKX_ENODEV    3   /* no usable HIP device / HIP runtime error            */
#define SKX_ENOMEM    4
#define SKX_EEMPTY    5   /* "<file> has no valid sequence" (ska_dict.rs:374)    */
#define SKX_EUNSUP    6   /* feature not available on the device path            */
#define SKX_EFORMAT   7   /* .skf decode error (lets a u64 load fall through to u128, lib.rs:635-661) */

/* src/lib.rs QualFilter; src/cli.rs:27-29 defaults (min_count 5, min_qual 20, strict) */
enum { SKX_QUAL_NOFILTER = 0, SKX_QUAL_MIDDLE = 1, SKX_QUAL_STRICT = 2 };
/* src/cli.rs FilterType */
enum { SKX_FILTER_NONE = 0, SKX_FILTER_NO_CONST = 1, SKX_FILTER_NO_AMBIG = 2, SKX_FILTER_NO_AMBIG_OR_CONST = 3 };

/* QualOpts (src/lib.rs; used by merge_ska_dict.rs:354-361, ska_dict.rs:333-341) */
typedef struct {
    uint16_t min_count;
    uint8_t  min_qual;
    int32_t  qual_filter;
} skx_qual;

typedef struct { uint64_t lo, hi; } skx_key;   /* split k-mer as the reference encodes it; hi == 0 for k <= 31 */

typedef struct skx_ctx     skx_ctx;      /* one per GPU / process                                   */
typedef struct skx_dictset skx_dictset;  /* a batch of per-sample SkaDicts, device resident         */
typedef struct skx_keyset  skx_keyset;   /* sorted unique split k-mers (rows of a merged array)     */
typedef struct skx_array   skx_array;    /* MergeSkaArray: keys + samples x k-mers middle bases     */

const char *skx_last_error(void);
const char *skx_version(void);            /* "0.5.2": the ska_version written into .skf files      */

int  skx_ctx_create(int device, skx_ctx **out);
void skx_ctx_destroy(skx_ctx *ctx);
int  skx_ctx_sync(skx_ctx *ctx);          /* drain the context's HIP stream                         */
void *skx_ctx_stream(skx_ctx *ctx);       /* the hipStream_t all kernels of this ctx are launched on */

/* ------------------------------------------------------------------------------------------
 * Record stream: what the host reader hands to the device in place of needletail's record
 * iterator (ska_dict.rs:131-153).  Each record's bases (line breaks removed) followed by one
 * '\n'; `qual` (FASTQ only) has the same layout.  Pointers are host memory unless on_device.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *seq;
    const uint8_t *qual;      /* NULL for FASTA input */
    uint64_t       len;       /* bytes in seq (and qual) */
} skx_stream;

/* SkaDict::new for a batch of samples (ska_dict.rs:333-378; one GPU unit of work per sample,
 * merge_ska_dict.rs:244,404).  k odd in 5..=63 else SKX_EINVAL "Invalid k-mer length".
 * A sample without any split k-mer -> SKX_EEMPTY.  FASTQ streams (qual != NULL) apply the
 * quality rules of split_kmer.rs:66-71,328-339 and the KmerFilter of bloom_filter.rs:116-148. */
int skx_dictset_build(skx_ctx *ctx, const skx_stream *samples, int n_samples, int on_device,
                      int k, int rc, const skx_qual *qual, skx_dictset **out);
/* same, reading FASTA/FASTQ(.gz) files with the library's own reader (needletail replacement).
 * file2[i] may be NULL.  proportion_reads 0 == None (ska_dict.rs:125-141). */
int skx_dictset_build_files(skx_ctx *ctx, const char *const *file1, const char *const *file2, int n_samples,
                            int k, int rc, const skx_qual *qual, int threads, double proportion_reads,
                            skx_dictset **out);
/* One sample's files as the record stream skx_dictset_build takes: what needletail's parse_fastx_file hands SkaDict::new record by
 * record (ska_dict.rs:131-153, 356-366) -- each record's bases (and qualities) followed by '\n'.  Host only, no device: plain, gzip
 * (the library's own inflater; members' CRC-32 and length checked: a damaged or truncated file is SKX_EIO), bzip2 / xz / zstd; FASTA or
 * FASTQ.  streaming != 0: through the line-by-line FASTQ reader of the read-set pipeline (SKX_EUNSUP: not FASTQ).  seq / qual:
 * malloc'd (skx_free); *qual = NULL for FASTA. */
int skx_read_records(const char *file1, const char *file2, double proportion_reads, int streaming, uint8_t **seq, uint8_t **qual, uint64_t *len);
void skx_dictset_free(skx_dictset *d);
int  skx_dictset_nsamples(const skx_dictset *d);
int  skx_dictset_key_bits(const skx_dictset *d);      /* 64 | 128 (lib.rs:592) */
/* SkaDict::ksize / kmers() (ska_dict.rs:499-512): entry count, then copy-out sorted by key */
int  skx_dictset_size(skx_dictset *d, int sample, uint64_t *n);
int  skx_dictset_export(skx_dictset *d, int sample, skx_key *keys, uint8_t *bases, uint64_t cap);

/* ------------------------------------------------------------------------------------------
 * MergeSkaDict::append/merge + MergeSkaArray::new (merge_ska_dict.rs:77-151,
 * merge_ska_array.rs:166-186), split so that a multi-GPU host can all-gather key tables
 * between the two halves:
 *   skx_keyset_union   : distinct split k-mers over all samples of the dictset
 *   skx_keyset_device / skx_keyset_from_device / skx_keyset_merge : exchange + combine tables
 *   skx_array_assemble : rows = keyset, columns = the dictset's samples ('-' where absent)
 * skx_merge() is the single-GPU composition of the three.
 * ------------------------------------------------------------------------------------------ */
int  skx_keyset_union(skx_ctx *ctx, skx_dictset *d, skx_keyset **out);
/* The same pass, which also keeps its notes (2 bytes per dictionary word: which row of its sub-bucket the word's key became, which bases)
 * for the skx_array_assemble over the same dictset that follows -- directly, or on the global rows skx_keyset_allgather returns for this
 * key set (the notes travel with it): the matrix is then filled without a second read of the dictionaries, as skx_merge does on one GPU
 * (merge_ska_dict.rs:77-109 appends a sample in one pass over its dictionary).  Costs the notes' memory until the key set / the rows are freed. */
int  skx_keyset_union_notes(skx_ctx *ctx, skx_dictset *d, skx_keyset **out);
int  skx_keyset_size(const skx_keyset *ks, uint64_t *n);
/* device pointer to n packed 64-bit words per key (1 for k<=31, 2 above), engine order */
int  skx_keyset_device(skx_keyset *ks, const void **dptr, uint64_t *n_keys, int *words_per_key);
int  skx_keyset_from_device(skx_ctx *ctx, const void *dptr, uint64_t n_keys, int k, int rc, skx_keyset **out);
int  skx_keyset_merge(skx_ctx *ctx, skx_keyset *const *sets, int n_sets, skx_keyset **out);
void skx_keyset_free(skx_keyset *ks);
int  skx_array_assemble(skx_ctx *ctx, skx_dictset *d, skx_keyset *rows, const char *const *names, skx_array **out);
/* The same array held as rows + dictionaries (no rows x samples matrix until an operation needs one: skx_array_device_stats gathers
 * the row statistics alone, skx_array_filter then writes only the kept rows, skx_array_save streams windows).  TAKES OWNERSHIP of d and
 * rows, also on failure: a multi-GPU rank's column slab over the global rows of 8 000 samples is 100+ GB it need not allocate. */
int  skx_array_assemble_lazy(skx_ctx *ctx, skx_dictset *d, skx_keyset *rows, const char *const *names, skx_array **out);
int  skx_merge(skx_ctx *ctx, skx_dictset *d, const char *const *names, skx_array **out);

/* build_and_merge (merge_ska_dict.rs:354-417) + MergeSkaArray::new: the `ska build` body.
 * When the samples' dictionaries do not fit in free device memory together (estimate from the file sizes, 60 % of free
 * HBM; SKX_BUILD_BATCH_MB overrides the budget), consecutive batches of samples are built and joined by the
 * skx_array_merge row-set path; a batch that still runs out of memory is halved and retried.  Same rows, columns and
 * sample order as a single batch. */
int skx_build_and_merge(skx_ctx *ctx, const char *const *names, const char *const *file1, const char *const *file2,
                        int n_samples, int k, int rc, const skx_qual *qual, int threads, double proportion_reads,
                        skx_array **out);

/* ------------------------------------------------------------------------------------------ MergeSkaArray */
void skx_array_free(skx_array *a);
/* save / load (merge_ska_array.rs:191-204): snappy-frame(CBOR), want_bits 64|128|0 */
int  skx_array_save(skx_array *a, const char *path);
int  skx_array_load(skx_ctx *ctx, const char *path, int want_bits, skx_array **out);
/* load + filter in one pass over the file (the body of `ska align x.skf` and `ska distance x.skf`): the same array as
 * skx_array_load followed by generic_modes::apply_filters (generic_modes.rs:112-131: threshold ceil(n_samples * min_freq),
 * update_kmers = false), or -- two_stage != 0 -- by the two filters of generic_modes::distance (generic_modes.rs:149-168: NoFilter
 * at ceil(n_samples * min_freq) when min_freq * n_samples >= 1, then NoConst at 0; *constant = rows the second one removed).
 * Rows are filtered as they come off the decoder, so the unfiltered rows x samples matrix never exists on the device and the
 * split k-mer list is not decoded (the resulting array has no keys: save / merge / weed / map on it fail with SKX_EINVAL).
 * Files the one-pass reader does not take (k > 31, unusual field order, stored counts that differ from the rows) go through
 * skx_array_load + skx_array_filter inside the call. */
typedef struct { double min_freq; int32_t filter_ambig_as_missing, filter_type, mask_ambig, ignore_const_gaps, two_stage; } skx_filter_spec;
int  skx_array_load_filtered(skx_ctx *ctx, const char *path, const skx_filter_spec *f, skx_array **out, int64_t *removed, int64_t *constant);
/* construct from host data (row-major [n_rows, n_samples] as in the .skf) */
int  skx_array_from_host(skx_ctx *ctx, int k, int rc, const char *const *names, int n_samples,
                         const skx_key *keys, const uint8_t *variants, const uint64_t *variant_count /* NULL: recount */,
                         uint64_t n_rows, const char *version, skx_array **out);
typedef struct {
    int32_t  k, rc, k_bits;
    uint64_t n_kmers;      /* split_kmers.len() */
    uint64_t n_rows;       /* variants.nrows()  */
    uint64_t n_samples;    /* columns held by this array                                  */
    uint64_t total_samples;/* samples over all ranks when this array is one column slab (== n_samples otherwise) */
} skx_array_info_t;
int  skx_array_info(const skx_array *a, skx_array_info_t *info);
const char *skx_array_name(const skx_array *a, uint64_t i);
const char *skx_array_version(const skx_array *a);
/* copy-out, rows sorted by key: keys[n_kmers], variants row-major [n_rows, n_samples], counts[n_rows] */
int  skx_array_export(skx_array *a, skx_key *keys, uint8_t *variants, uint64_t *counts);
/* n_sample_kmers (merge_ska_array.rs:554-559) */
int  skx_array_sample_kmers(skx_array *a, int64_t *out);
/* an array still held as the append pass left it (pieces of 4-bit base sets by first-seen rank: csrc/skx_append.hip): bytes of pieces written,
 * row blocks, ranks per block; *piece_bytes = 0 when the array holds a matrix instead.  No counterpart in the reference (its merged dictionary
 * holds a Vec<u8> per row, merge_ska_dict.rs:41-58); bench.py prices the append kernel's needed traffic with it. */
int  skx_array_pieces_info(skx_array *a, uint64_t *piece_bytes, uint64_t *row_blocks, uint32_t *ranks_per_block);
/* MergeSkaArray::filter (merge_ska_array.rs:289-402); *removed = its i32 return value */
int  skx_array_filter(skx_array *a, uint64_t min_count, int filter_ambig_as_missing, int filter_type,
                      int mask_ambig, int ignore_const_gaps, int update_kmers, int32_t *removed);
/* Optional hint before skx_array_load_filtered: the array about to be loaded will be written with skx_array_write_fasta to `fd` (a
 * regular file opened read-write, at its current offset).  The engine then allocates the file's pages while the rows are still
 * being read -- page allocation, not copying, is what bounds a 5 GB alignment on tmpfs.  No effect on the bytes written. */
int  skx_ctx_expect_output(skx_ctx *ctx, int fd);
/* write_fasta (merge_ska_array.rs:499-517): ">name\nSEQ\n" per sample to fd */
int  skx_array_write_fasta(skx_array *a, int fd);
/* same into a malloc'd buffer (free with skx_free) */
int  skx_array_fasta(skx_array *a, char **buf, uint64_t *len);
/* device view of the sample-major middle-base matrix: row s = sample s, pitch bytes apart */
int  skx_array_device_matrix(skx_array *a, const uint8_t **dptr, uint64_t *pitch, uint64_t *n_rows);

/* multi-GPU column slabs: device views of the per-row statistics the filter reads
 * (cells != '-', cells in ACGT, set of IUPAC codes present, stored variant_count; u32 each, n_rows long) so that
 * the host can reduce them across ranks, and the global sample count the gap test must use */
int  skx_array_device_stats(skx_array *a, uint32_t **present, uint32_t **unambig, uint32_t **mask, uint32_t **variant_count);
int  skx_array_set_total_samples(skx_array *a, uint64_t total_samples);

typedef struct { double distance, mismatch_prop; uint64_t match_count, mismatch_count; } skx_dist;
/* MergeSkaArray::distance (merge_ska_array.rs:416-438,587-632): upper triangle, pairs (i<j) row-major.  filt_ambig = 0 (--allow-ambiguous):
 * rows in which no cell is ambiguous -- the array's row statistics say which -- are counted by the 4-plane sweep, the others by the
 * twelve-class one; the sums are the reference's per-row sums either way */
int  skx_array_distance(skx_array *a, double constant, int filt_ambig, skx_dist *out);
/* generic_modes::distance (generic_modes.rs:136-189) in one call that leaves the array as it is: rows below ceil(n_samples *
 * min_freq) (when min_freq * n_samples >= 1) and constant rows are skipped while the bit planes are built (*constant = rows the
 * NoConst stage removes, added to every pair's matches), instead of being filtered out of the matrix first.  Same numbers as
 * skx_array_filter x 2 + skx_array_distance. */
int  skx_array_distance_filtered(skx_array *a, double min_freq, int filt_ambig, skx_dist *out, int64_t *constant, uint64_t *rows_used);
/* The two halves of skx_array_distance, so that a multi-GPU host can exchange the bit planes between them (SURVEY.md 8e:
 * "tile the pair matrix over ranks"): every rank builds the planes of its own samples over the (globally filtered) rows, the
 * planes are all-gathered (plane-major: planes[p][sample][word], 4 planes with filt_ambig, 8 without), and each rank
 * finishes a band of first samples [i_lo, i_hi) against every later sample -- `out` = those pairs, (i, j > i) row-major, in the
 * arithmetic of merge_ska_array.rs:596-631.  Any 0 <= i_lo < i_hi <= n_samples (pair tiles start at i_lo).  The planes pointer stays valid while the array lives. */
int  skx_array_distance_planes(skx_array *a, int filt_ambig, const void **planes, uint64_t *words_per_row, int *n_planes);
int  skx_planes_distance(skx_ctx *ctx, const void *planes, int n_samples, uint64_t words_per_row, int filt_ambig, double constant,
                         int i_lo, int i_hi, skx_dist *out);
void skx_free(void *p);

/* ------------------------------------------------------------------------------------------
 * Collectives: the exchanges of a job whose samples are sharded over the GPUs of a node, one process per
 * GPU.  They stand where the reference's build_and_merge joins the dictionaries of its worker threads (merge_ska_dict.rs:354-417:
 * samples dealt to threads in input order, per-thread dictionaries merged pairwise) -- here the "threads" are ranks, each rank keeps
 * the columns of its own samples, and what is exchanged is the row set (one all-gather of the per-rank key tables), the per-row
 * filter statistics, and for `ska distance` the bit planes.  Transport: RCCL over xGMI on the context's stream
 * (skx_comm_create; librccl is opened at run time), or a host-staged exchange through a directory on tmpfs for ranks that share one
 * device, which RCCL refuses (skx_comm_create_local; tests and single-GPU emulation of an N-rank job).  The host launches the
 * processes and carries the 128-byte id from rank 0 to the others (a file, an environment variable, MPI, a torch store ...).
 * Every call below is collective: all ranks of the communicator make it, in the same order.
 * ------------------------------------------------------------------------------------------ */
typedef struct skx_comm skx_comm;
#define SKX_COMM_ID_BYTES 128
int  skx_comm_unique_id(uint8_t id[SKX_COMM_ID_BYTES]);                        /* ncclGetUniqueId; rank 0 only */
int  skx_comm_create(skx_ctx *ctx, int rank, int world, const uint8_t id[SKX_COMM_ID_BYTES], skx_comm **out);   /* ncclCommInitRank on ctx's device */
/* host-staged transport: `dir` is a fresh directory every rank can reach (NULL at world 1: the library makes its own and removes it with the
 * communicator); ctx may be NULL when only host buffers are exchanged */
int  skx_comm_create_local(skx_ctx *ctx, int rank, int world, const char *dir, skx_comm **out);
void skx_comm_destroy(skx_comm *c);
int  skx_comm_rank(const skx_comm *c);
int  skx_comm_world(const skx_comm *c);
uint64_t skx_comm_bytes_received(const skx_comm *c);                           /* payload this rank has received so far (reports) */
/* what the communicator runs on: *rccl_ranks = ncclCommCount of the RCCL communicator (0: the host-staged transport), *rccl_device = the
 * device RCCL bound it to (ncclCommCuDevice; -1: none), *ctx_device = the engine context's device.  A report for a job's first contact with
 * a node (bench.py prints them per rank); stands where merge_ska_dict.rs:354-417 logs its thread count. */
int  skx_comm_transport(const skx_comm *c, int *rccl_ranks, int *rccl_device, int *ctx_device);
int  skx_comm_barrier(skx_comm *c);
/* the two primitives the exchanges are made of, for hosts with exchanges of their own: `bytes` from every rank, rank r's at
 * recv + r * bytes (recv must not overlap send); in-place sum of n 32-bit counters.  on_device: the pointers are device memory
 * (required by the RCCL transport). */
int  skx_comm_allgather(skx_comm *c, const void *send, void *recv, uint64_t bytes, int on_device);
int  skx_comm_allreduce_u32(skx_comm *c, uint32_t *buf, uint64_t n, int on_device);
/* host blocks of different sizes (sizes[world], known to every rank) to rank 0: recv there = the blocks in rank order */
int  skx_comm_gather_root(skx_comm *c, const void *send, const uint64_t *sizes, void *recv);
/* partitioning, pure arithmetic: the contiguous shard [lo, hi) of `rank` (input order kept, so names stay in CLI order, cf. the
 * offsets of merge_ska_dict.rs:243-253,277-291), and the bands of first samples of the pair matrix (lo_hi[2 r], lo_hi[2 r + 1];
 * starts on multiples of `align`, about the same number of pairs each, merge_ska_array.rs:416-438 order kept) */
int  skx_shard_range(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi);
int  skx_pair_bands(int n_samples, int world, int align, int *lo_hi);
/* exchange 1: one all-gather of the per-rank key tables (padded to the longest) + their union: every rank gets the same global
 * row set, which skx_array_assemble / skx_array_assemble_lazy then fills with the rank's own columns.  SKX_EINVAL with the
 * reference's "K-mer lengths do not match" / "Strand use inconsistent" when ranks disagree. */
int  skx_keyset_allgather(skx_comm *c, skx_keyset *local, skx_keyset **rows);
/* exchange 2: the per-row statistics the filter reads, over all ranks (counts summed in one all-reduce, 16-bit code sets
 * all-gathered and OR-ed); variant_count becomes the global count, the array's total_samples the job's sample count */
int  skx_array_reduce_stats(skx_comm *c, skx_array *a, uint64_t total_samples);
/* exchange 3: MergeSkaArray::distance (merge_ska_array.rs:416-438,587-632) for a sharded job: `a` = this rank's samples over the
 * globally filtered rows; one all-gather of the bit planes (--allow-ambiguous: one all-reduce of a byte per row first, so that the ranks
 * agree on the rows without an ambiguous cell, which travel as 4 planes; the others as 8), each rank finishes a band of the pair
 * matrix, rank 0 receives all pairs: out[n_out], n_out >= S (S - 1) / 2 there (ignored elsewhere) */
int  skx_array_distance_sharded(skx_comm *c, skx_array *a, int filt_ambig, double constant, skx_dist *out, uint64_t n_out);


/* ---- .skf life-cycle: `ska merge`, `ska delete`, `ska weed` ---- */
/* generic_modes::merge (generic_modes.rs:90-106) = MergeSkaArray::to_dict (merge_ska_array.rs:208-222) +
 * MergeSkaDict::extend (merge_ska_dict.rs:160-193) folded over the inputs + MergeSkaArray::new (:166-186): rows = union of
 * the inputs' split k-mers, columns = their samples in order, absent cells '-'.  SKX_EINVAL with the reference's panic
 * texts "K-mer lengths do not match: a b" / "Strand use inconsistent". */
int  skx_array_merge(skx_ctx *ctx, skx_array *const *arrays, int n_arrays, skx_array **out);
/* MergeSkaArray::delete_samples (merge_ska_array.rs:231-271) incl. update_counts(false): rows no remaining sample has are
 * dropped.  SKX_EINVAL "Invalid number of samples to remove" / "Could not find sample(s): {..}" where it panics. */
int  skx_array_delete_samples(skx_array *a, const char *const *names, int n_names);
/* MergeSkaArray::weed (merge_ska_array.rs:452-487): rows whose split k-mer is (reverse: is not) in `weed` are removed;
 * `weed` = the split k-mers of the weed FASTA (RefSka::new + kmer_iter, ska_ref.rs:189-262,541), i.e. the key set of its
 * dictionary: skx_dictset_build_files (1 sample) -> skx_keyset_union. */
int  skx_array_weed(skx_array *a, skx_keyset *weed, int reverse, uint64_t *removed);
/* RefSka::new + kmer_iter (ska_ref.rs:189-262,541) for `ska weed`: the split k-mers of a FASTA file as a key set;
 * SKX_EINVAL "Cannot create reference from FASTQ files" (ska_ref.rs:206-208), SKX_EEMPTY "<file> has no valid sequence" */
int  skx_keyset_from_fasta(skx_ctx *ctx, const char *path, int k, int rc, skx_keyset **out);
/* ---- `ska cov` ----
 * CoverageHistogram::new (coverage.rs:70-148) + the histogram step of fit_histogram (:158-163): occurrence counts of the
 * split k-mers of a FASTQ pair (qualities ignored): hist[c - 1] = #split k-mers seen c times, c <= 1000 (hist has 1000 entries).
 * SKX_EINVAL "<file> appears to be FASTA.\nCoverage can only be used with FASTQ files, not FASTA." */
int  skx_cov_histogram(skx_ctx *ctx, const char *fastq_fwd, const char *fastq_rev, int k, int rc, uint32_t *hist);

/* ---- `ska map` ----
 * generic_modes::map (generic_modes.rs:56-84) = RefSka::new(k, reference, rc, ambig_mask, repeat_mask) (ska_ref.rs:189-311)
 * + RefSka::map (:508-533) + write_aln (format 0, :622-645, AlnWriter aln_writer.rs) | write_vcf (format 1, :648-765).
 * The text is malloc'd (skx_free).  Errors where the reference panics: "Cannot create reference from FASTQ files",
 * "<file> has no valid sequence" (SKX_EEMPTY), "No split k-mers mapped to reference". */
int  skx_array_map(skx_array *a, const char *reference_fasta, int ambig_mask, int repeat_mask, int format, int threads, char **buf, uint64_t *len);
/* the `k` a .skf file states in its first fields, without loading it (0: not found there, or the file cannot be read): lets a caller that
 * would try 64-bit keys first and 128-bit keys next (lib.rs:635-661) go to the right width at once */
int  skx_skf_peek_k(const char *path);
/* the context an array lives on; and a way for host glue above the ABI to leave its message in skx_last_error() */
skx_ctx *skx_array_ctx(const skx_array *a);
void skx_set_last_error(const char *msg);

/* wall-clock phases of the host-side path (file reading + upload, .skf codec, FASTA writer ...), accumulated per name since the
 * last reset: a JSON object {"phase": seconds, ...} in first-use order (malloc'd, skx_free).  The reference has no counterpart;
 * bench.py's end_to_end leg and SKX_DEBUG read them.  skx_phase_add lets host glue above the ABI record its own phases. */
int  skx_phases_json(char **buf, uint64_t *len, int reset);
void skx_phase_add(const char *name, double seconds);

/* per-stage device timings of the last call on this ctx (ms; HIP events on the ctx stream).  The last four are single kernels of the
 * append pass, measured inside their stages (probe + append inside key_union, pieces_stats inside assemble, the kept rows' pieces_rows
 * inside compact): what bench.py's `roofline.dominant` is computed from. */
typedef struct { double hist, scatter, dedupe, key_union, assemble, filter, compact, distance, append_probe, append, pieces_stats, pieces_rows; } skx_timings;
int  skx_ctx_timings(skx_ctx *ctx, skx_timings *t, int reset);
/* which kernels the last merge on this context went through: "append64" / "append128" (MergeSkaDict::append in one pass over the
 * unsorted regions, merge_ska_dict.rs:77-109, for u64 / u128 keys) or "sorted: <why the pass was not taken>" (per-sample sort +
 * union + assemble); "" before the first merge.  The reference has no counterpart: its append is one code path. */
const char *skx_ctx_merge_path(skx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
