/*
 * ska_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the split-k-mer build -> merge -> align/distance
 * path of bacpop/ska.rust v0.5.2.  Every function cites the reference
 * file:line whose behaviour it restates.  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time
 * the reference algorithm; nothing under ska.rust_amd/ may include, link or
 * execute anything in this directory.
 *
 * Parity pinning: the real `ska` binary cannot be built in this image (no
 * Rust toolchain, unvendored deps), so the restatement is pinned against the
 * reference's own committed fixtures (tests/golden/): the four .skf files
 * written by the Rust binary, every *.dist.stdout / nk / align golden, the
 * inline expectations of tests/{align,fasta_input,fastq_input,distance}.rs,
 * the skf_ops.rs goldens (merge / delete / weed) and the 19 map_* goldens of
 * tests/map.rs.  See tests/test_oracle_golden.py.
 * One part is pinned only by a single known answer: the mixture fit of
 * `ska cov` (ora_cov.c) restates the argmin crate's optimiser, which is not in
 * /root/reference; the reference's unit-test vector (coverage.rs:369-385 ->
 * cutoff 9) is reproduced, nothing else constrains its iterates.
 */
#ifndef SKA_ORACLE_H
#define SKA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t lo, hi; } ora_key;          /* u64 keys have hi == 0 */

/* QualFilter (src/lib.rs QualFilter enum; cli.rs:27-29) */
enum { ORA_QUAL_NOFILTER = 0, ORA_QUAL_MIDDLE = 1, ORA_QUAL_STRICT = 2 };
/* FilterType (src/cli.rs FilterType) */
enum { ORA_FILTER_NONE = 0, ORA_FILTER_NO_CONST = 1, ORA_FILTER_NO_AMBIG = 2, ORA_FILTER_NO_AMBIG_OR_CONST = 3 };

typedef struct {
    uint16_t min_count;   /* QualOpts.min_count */
    uint8_t  min_qual;    /* QualOpts.min_qual  */
    int      qual_filter; /* ORA_QUAL_*         */
} ora_qual;

/* flags emitted per window by ora_extract_record */
#define ORA_F_IS_RC   1u  /* canonical form is the reverse complement     */
#define ORA_F_PALIN   2u  /* self_palindrome()                            */
#define ORA_F_MIDQ_OK 4u  /* middle_base_qual()                           */

const char *ora_last_error(void);
void ora_free(void *p);

/* ---- L1: one record through SplitKmer (split_kmer.rs:63-340) ---------- */
/* Returns the number of windows the reference iterator yields, in order.
 * Arrays may be NULL; at most cap entries are written. */
size_t ora_extract_record(const uint8_t *seq, size_t len, const uint8_t *qual,
                          int k, int rc, int min_qual, int qual_filter, int is_reads,
                          ora_key *keys, uint8_t *mid, uint8_t *flags, uint64_t *hashes,
                          size_t cap);

/* ---- L2: SkaDict (ska_dict.rs:56-378) --------------------------------- */
typedef struct ora_dict ora_dict;
ora_dict *ora_dict_new(int k, int rc, const ora_qual *q);
/* add_file_kmers' inner body for one record (ska_dict.rs:143-177) */
void ora_dict_add_record(ora_dict *d, const uint8_t *seq, size_t len, const uint8_t *qual, int is_reads);
/* SkaDict::new: file1 (+file2), FASTA/FASTQ decided by first record of file1 */
ora_dict *ora_dict_from_files(int k, int rc, const char *file1, const char *file2,
                              const ora_qual *q, double proportion_reads /*0 = None*/);
size_t ora_dict_size(const ora_dict *d);
int ora_dict_key_bits(const ora_dict *d);
/* entries sorted ascending by key (the reference's order is unspecified) */
void ora_dict_export_sorted(const ora_dict *d, ora_key *keys, uint8_t *bases);
void ora_dict_free(ora_dict *d);

/* ---- L3: MergeSkaDict + MergeSkaArray --------------------------------- */
typedef struct ora_array ora_array;
/* build_and_merge (merge_ska_dict.rs:354-417) + MergeSkaArray::new (merge_ska_array.rs:166-186).
 * names/file1/file2 arrays of length n (file2[i] may be NULL). Rows come out sorted by key. */
ora_array *ora_build_and_merge(const char *const *names, const char *const *file1, const char *const *file2,
                               int n, int k, int rc, const ora_qual *q, int threads, double proportion_reads);
/* same, but from already-built per-sample dicts (append in index order) */
ora_array *ora_array_from_dicts(ora_dict *const *dicts, const char *const *names, int n);
ora_array *ora_array_load(const char *path, int want_bits /*64|128|0=either*/);
int  ora_array_save(const ora_array *a, const char *path);
void ora_array_free(ora_array *a);

int    ora_array_k(const ora_array *a);
int    ora_array_rc(const ora_array *a);
int    ora_array_k_bits(const ora_array *a);
size_t ora_array_nrows(const ora_array *a);      /* variants.nrows()   */
size_t ora_array_nkmers(const ora_array *a);     /* split_kmers.len()  */
size_t ora_array_nsamples(const ora_array *a);
const char *ora_array_name(const ora_array *a, size_t i);
const char *ora_array_version(const ora_array *a);
/* copies: keys[nkmers], variants[nrows*nsamples] row-major, counts[nrows] */
void ora_array_export(const ora_array *a, ora_key *keys, uint8_t *variants, uint64_t *counts);
/* rows re-sorted by key (used to order-normalise reference-written fixtures) */
void ora_array_sort_rows(ora_array *a);

/* MergeSkaArray::filter (merge_ska_array.rs:289-402); returns #removed */
int32_t ora_array_filter(ora_array *a, size_t min_count, int filter_ambig_as_missing, int filter_type,
                         int mask_ambig, int ignore_const_gaps, int update_kmers);
/* generic_modes::apply_filters (generic_modes.rs:112-131) */
int32_t ora_apply_filters(ora_array *a, double min_freq, int filter_ambig_as_missing, int filter_type,
                          int ambig_mask, int ignore_const_gaps);
/* write_fasta (merge_ska_array.rs:499-517) into a malloc'd buffer */
char *ora_array_fasta(const ora_array *a, size_t *len);
/* Display / Debug (`ska nk`, merge_ska_array.rs:649-698) exactly as main prints them */
char *ora_array_nk(const ora_array *a, int full_info, size_t *len);

/* ---- skf life-cycle ---- */
/* generic_modes::merge (generic_modes.rs:90-106) = to_dict + MergeSkaDict::extend (merge_ska_dict.rs:160-193) + ::new */
ora_array *ora_array_merge(const ora_array *const *in, int n);
/* MergeSkaArray::delete_samples (merge_ska_array.rs:231-271); -1 + ora_last_error() where the reference panics */
int ora_array_delete_samples(ora_array *a, const char *const *del_names, int n_del);
/* MergeSkaArray::weed (merge_ska_array.rs:452-487) against a key list (RefSka::kmer_iter, ska_ref.rs:541) */
int ora_array_weed(ora_array *a, const ora_key *weed_keys, size_t n_weed, int reverse);
/* generic_modes::weed (generic_modes.rs:207-262): optional weed FASTA, then the filter with a floor() threshold */
int ora_weed(ora_array *a, const char *weed_fasta, int reverse, double min_freq, int filter_ambig_as_missing, int filter_type,
             int ambig_mask, int ignore_const_gaps);

/* ---- `ska map` ---- */
typedef struct ora_ref ora_ref;
/* RefSka::new (ska_ref.rs:189-311): split k-mers + middle positions of a reference FASTA, optional repeat coordinates */
ora_ref *ora_ref_new(int k, const char *fasta, int rc, int ambig_mask, int repeat_mask);
/* generic_modes::map's to_dict + RefSka::map (generic_modes.rs:56-67, ska_ref.rs:508-533) */
int ora_ref_map(ora_ref *r, const ora_array *a);
/* write_aln / write_vcf (ska_ref.rs:622-765, aln_writer.rs, idx_check.rs) into malloc'd text */
char *ora_ref_write_aln(ora_ref *r, size_t *len);
char *ora_ref_write_vcf(ora_ref *r, size_t *len);
void ora_ref_free(ora_ref *r);

/* ---- `ska cov` / --min-count auto (coverage.rs) ---- */
/* occurrence-count histogram of the split k-mers of a FASTQ pair: hist[c-1] = #split k-mers seen c times (c <= 1000) */
int ora_cov_histogram(const char *fq1, const char *fq2, int k, int rc, uint32_t hist[1000]);
/* two-component Poisson mixture fit (argmin BFGS restated) + find_cutoff on a truncated histogram; -1 if not converged */
int ora_cov_fit(const double *counts, size_t n, double *w0, double *c, size_t *cutoff);
/* the whole subcommand: plot_hist text (malloc'd) and the cutoff */
char *ora_cov(const char *fq1, const char *fq2, int k, int rc, size_t *len, size_t *cutoff);

typedef struct { double distance, mismatch_prop; uint64_t match_count, mismatch_count; } ora_dist;
/* MergeSkaArray::distance (merge_ska_array.rs:416-438,587-632): upper triangle, (i<j) row-major */
void ora_array_distance(const ora_array *a, double constant, int filt_ambig, ora_dist *out);
/* generic_modes::distance (generic_modes.rs:136-189): filters then the long-form TSV */
char *ora_distance_tsv(ora_array *a, double min_freq, int filt_ambig, size_t *len);
/* generic_modes::align (generic_modes.rs:22-50) */
char *ora_align_fasta(ora_array *a, int filter_type, int mask_ambig, int ignore_const_gaps,
                      double min_freq, int filter_ambig_as_missing, size_t *len);

/* ---- phase timers for the cpu_baseline leg (seconds, cumulative) ------ */
typedef struct { double read_parse, dict, append, merge, to_array, filter, fasta; } ora_timers;
void ora_timers_get(ora_timers *t, int reset);

/* io_utils::read_input_fastas name rule (io_utils.rs:31-46); returns malloc'd */
char *ora_sample_name(const char *path);

#ifdef __cplusplus
}
#endif
#endif
