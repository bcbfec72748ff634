/*
 * skx_host.h -- host-side mirror of the reference's mode glue for the hot path, above the skx.h C ABI.
 * Same names, argument meaning and error behaviour as the Rust functions they replace (the reference's
 * toolchain is absent from this image, so the host side is C++; a Rust host would call skx.h directly and
 * keep its own generic_modes.rs / io_utils.rs).  Text results are malloc'd; free with skx_free().
 */
#ifndef SKX_HOST_H
#define SKX_HOST_H
#include "skx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* generic_modes::apply_filters (generic_modes.rs:112-131): threshold = ceil(n_samples * min_freq), update_kmers = false */
int skh_apply_filters(skx_array *a, double min_freq, int filter_ambig_as_missing, int filter_type, int ambig_mask,
                      int ignore_const_gaps, int32_t *removed);
/* generic_modes::align (generic_modes.rs:22-50): filters then the FASTA alignment */
int skh_align(skx_array *a, int filter_type, int mask_ambig, int ignore_const_gaps, double min_freq,
              int filter_ambig_as_missing, char **buf, uint64_t *len);
int skh_align_fd(skx_array *a, int filter_type, int mask_ambig, int ignore_const_gaps, double min_freq,
                 int filter_ambig_as_missing, int fd);   /* the same, streamed to a file descriptor */
/* `ska align <inputs>` (lib.rs:617-662 = io_utils::load_array + generic_modes::align): one .skf input goes through the engine's
 * one-pass load + filter (skx_array_load_filtered), several sequence files through build_and_merge with the CLI defaults */
int skh_align_inputs_fd(skx_ctx *ctx, const char *const *inputs, int n_inputs, int threads, int filter_type, int mask_ambig, int ignore_const_gaps,
                        double min_freq, int filter_ambig_as_missing, int fd);
/* `ska distance <skf>` (lib.rs:710-727 = load + generic_modes::distance), same one-pass load */
int skh_distance_skf_tsv(skx_ctx *ctx, const char *skf_file, double min_freq, int filt_ambig, char **buf, uint64_t *len);
/* generic_modes::distance (generic_modes.rs:136-189): two-stage filter, then the long-form TSV with the
 * VariantDist Display format "{:.2}\t{:.5}\t{}\t{}" (merge_ska_array.rs:57-65) */
int skh_distance_tsv(skx_array *a, double min_freq, int filt_ambig, char **buf, uint64_t *len);
/* Display / Debug of MergeSkaArray as `ska nk [--full-info]` prints them (merge_ska_array.rs:649-698, lib.rs:808-827) */
int skh_nk(skx_array *a, int full_info, char **buf, uint64_t *len);
/* generic_modes::save_skf (generic_modes.rs:270-283): appends ".skf" unless already there */
int skh_save_skf(skx_array *a, const char *out_prefix);
/* io_utils::load_array (io_utils.rs:60-93) + the u64-then-u128 retry of lib.rs:635-661 */
int skh_load_array(skx_ctx *ctx, const char *const *inputs, int n_inputs, int threads, skx_array **out);
/* generic_modes::merge (generic_modes.rs:90-106): first file decides u64/u128 (lib.rs:728-741), the others must load as the
 * same type ("Failed to load input file (inconsistent k-mer lengths?)"); saved through save_skf (".skf" appended) */
int skh_merge(skx_ctx *ctx, const char *const *skf_files, int n_files, const char *out_prefix);
/* generic_modes::delete (generic_modes.rs:192-210): delete_samples then save (".skf" appended unless present) */
int skh_delete(skx_array *a, const char *const *names, int n_names, const char *out_file);
/* generic_modes::weed (generic_modes.rs:213-267): optional weed FASTA (FASTQ refused, ska_ref.rs:206-208), then the filter
 * with threshold floor(n_samples * min_freq) and update_kmers = true when anything is asked for; out_file NULL = no save */
int skh_weed(skx_array *a, const char *weed_file, int reverse, double min_freq, int filter_ambig_as_missing, int filter_type,
             int ambig_mask, int ignore_const_gaps, const char *out_file);
/* CoverageHistogram::fit_histogram + plot_hist (coverage.rs:151-250) on the device-built histogram: two-component Poisson
 * mixture by maximum likelihood (argmin's BFGS + back-tracking line search restated), cutoff = first count at which the
 * coverage component is the likelier one.  text = plot_hist's table (malloc'd; NULL to skip). */
int skh_cov(skx_ctx *ctx, const char *fastq_fwd, const char *fastq_rev, int k, int rc, char **text, uint64_t *len, uint64_t *cutoff);
/* the fit alone on an already truncated histogram (the reference's unit test drives exactly this, coverage.rs:369-385) */
int skh_cov_fit(const double *counts, uint64_t n, double *w0, double *c, uint64_t *cutoff);
/* io_utils::read_input_fastas sample-name rule (io_utils.rs:31-46) */
char *skh_sample_name(const char *path);
/* ---- the same modes over the GPUs of one node, one process per GPU; each function is the body of ONE rank
 * and is collective over `comm` (include/skx.h "Collectives").  The job's samples (all of them, in input order) are dealt to the
 * ranks in contiguous shards; a rank builds the dictionaries of its shard (skx_dictset_build_files), the key tables are all-gathered
 * (skx_keyset_allgather) and the rank holds its own columns over the global rows (skx_array_assemble_lazy).  This is what
 * build_and_merge's thread tree becomes when the workers are GPUs (merge_ska_dict.rs:354-417). */
typedef struct {
    const char *const *names, *const *file1, *const *file2;   /* the whole job; file2[i] may be NULL (file2 itself too) */
    int n_samples;
    int k, rc;
    skx_qual qual;
    int threads;
    double proportion_reads;                                   /* 0 == None */
    const char *output;                                        /* build: prefix of the .skf; align / distance: the file (NULL = stdout, rank 0) */
    int merge_parts;                                           /* build: rank 0 joins the per-rank parts into <output>.skf */
    double min_freq;
    int filter_type, mask_ambig, ignore_const_gaps, filter_ambig_as_missing;    /* align (generic_modes.rs:112-131) */
    int filt_ambig;                                            /* distance: !--allow-ambiguous */
} skh_job;
/* `ska build`: one .skf per rank, <output>.part<r>of<N>.skf = the global rows x that rank's samples (each a valid MergeSkaArray;
 * `ska merge` joins them), or with merge_parts the one file generic_modes::save_skf would write */
int skh_build_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job);
/* `ska align`: row statistics reduced over ranks, the filter decided identically everywhere, every rank writes its own samples'
 * records at their offsets of the one output file (write_fasta's order, merge_ska_array.rs:499-517) */
int skh_align_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job);
/* `ska distance`: generic_modes::distance's two filters on the reduced statistics, then skx_array_distance_sharded; rank 0 writes the table */
int skh_distance_sharded(skx_ctx *ctx, skx_comm *comm, const skh_job *job);
/* the `ska` command line (build | align | map | distance | nk | merge | delete | weed | cov); returns the process exit code.
 * `--gpus N` on build / align / distance starts one process per GPU (this executable again, SKX_RANK / SKX_WORLD / SKX_COMM_ID_FILE in
 * their environment) and runs the sharded bodies above; a launcher of one's own sets the same variables. */
int skh_main(int argc, char **argv);
/* `ska --help | -h | help [cmd] | <cmd> --help | --version | -V` (cli.rs:154 `#[command(author, version, about)]`, propagate_version; per-flag
 * help cli.rs:168-459): 1 = answered on stdout (exit code 0), 0 = not a help / version request, 2 = `ska help <unknown>`.  skh_main calls it first. */
int skh_help(int argc, char **argv);
/* a line of the reference's logger (simple_logger: lib.rs:559-563, Warn by default, Info with -v) on stderr: level 1 = WARN, 2 = INFO;
 * target = the Rust module the reference logs it from ("ska::io_utils" ...) */
void skh_log(int level, const char *target, const char *message);

#ifdef __cplusplus
}
#endif
#endif
