// skx_device.h -- launch wrappers of the gfx950 kernels (skx_device.hip) used by the C-ABI layer.
//
// Data layout in HBM:
//   record stream   : bytes, each record's bases followed by '\n' (host reader output)
//   packed word     : (H(split k-mer) << 4) | base-set mask   [u64 for k<=31]
//                     H = bijective mix of the 2(k-1)-bit canonical split k-mer; "engine order" == order of H
//   dictset         : per (sample, bucket) a region of packed words, bucket = top logB bits of H;
//                     after dedupe the first ucnt words of each region are sorted unique
//   keyset          : per sub-bucket (top logN bits of H) a slab of sorted unique words (mask bits 0)
//   array           : sample-major matrix [S][pitch] of ASCII middle bases, '-' == absent
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skx {

// test / measurement knobs: SKX_KNOBS="name=value,name,..." (ONE environment variable; a bare name is 1); absent = the product path
long knob(const char *name, long absent = 0);
int extract_tile_bases(int logB);         // window end positions per workgroup of the extraction kernels (16 per thread)
constexpr int MAX_LOGB = 13;              // buckets per sample <= 8192 (LDS histogram)
constexpr uint64_t EMPTY64 = ~0ull;

// H: a 2-round Feistel bijection on the 2(k-1)-bit split k-mer (upper arm | lower arm), 32-bit arithmetic only.
// Its top bits are uniform whatever the genome's composition, so hash buckets are balanced; "engine order" of
// keys is the numeric order of H(key).
struct HashParams {
    int bits;          // 2(k-1)
    int hb;            // bits / 2 = k-1 (<= 30 on the 64-bit path)
    uint32_t hmask;    // (1<<hb)-1
    uint32_t c[4];     // odd round multipliers
};
HashParams make_hash_params(int k);
__host__ __device__ inline uint32_t hround(uint32_t v, uint32_t c, int hb) { return (v * c) >> (32 - hb); }
__host__ __device__ inline void hmix_halves(uint32_t &L, uint32_t &R, const HashParams &p)
{
    L ^= hround(R, p.c[0], p.hb); R ^= hround(L, p.c[1], p.hb);
}
__host__ __device__ inline uint64_t hmix(uint64_t x, const HashParams &p)
{
    uint32_t L = (uint32_t)(x >> p.hb), R = (uint32_t)x & p.hmask;
    hmix_halves(L, R, p);
    return ((uint64_t)L << p.hb) | R;
}
__host__ __device__ inline uint64_t hunmix(uint64_t x, const HashParams &p)
{
    uint32_t L = (uint32_t)(x >> p.hb), R = (uint32_t)x & p.hmask;
    R ^= hround(L, p.c[1], p.hb); L ^= hround(R, p.c[0], p.hb);
    return ((uint64_t)L << p.hb) | R;
}

// ---- 128-bit path (k > 31): arms of up to 62 bits, packed word = (H(key) << 4) | mask in 128 bits ----
typedef unsigned __int128 u128;
struct WideHash { int bits, hb; uint64_t hmask; uint64_t c[3]; };
WideHash make_wide_hash(int k);
__host__ __device__ inline uint64_t hround64(uint64_t v, uint64_t c, int hb) { return (v * c) >> (64 - hb); }
__host__ __device__ inline void hmix_halves_w(uint64_t &L, uint64_t &R, const WideHash &p)
{
    L ^= hround64(R, p.c[0], p.hb); R ^= hround64(L, p.c[1], p.hb); L ^= hround64(R, p.c[2], p.hb);
}
__host__ __device__ inline u128 hunmix_w(u128 x, const WideHash &p)
{
    uint64_t L = (uint64_t)(x >> p.hb), R = (uint64_t)x & p.hmask;
    L ^= hround64(R, p.c[2], p.hb); R ^= hround64(L, p.c[1], p.hb); L ^= hround64(R, p.c[0], p.hb);
    return ((u128)L << p.hb) | R;
}
__host__ __device__ inline u128 hmix_w(u128 x, const WideHash &p)
{
    uint64_t L = (uint64_t)(x >> p.hb), R = (uint64_t)x & p.hmask;
    hmix_halves_w(L, R, p);
    return ((u128)L << p.hb) | R;
}

struct ExtractArgs {
    const uint8_t *const *seqs;   // [n] device pointers to record streams (16-B aligned)
    const uint8_t *const *quals;  // [n] or nullptr
    const uint64_t *lens;         // [n]
    int n_samples;
    int tiles_max;                // ceil(max len / TILE_BASES)
    int parts = 0, tiles_part = 0; // fewer than 8 samples: a sample's tiles in `parts` contiguous ranges of tiles_part, a range per XCD (set by the launcher)
    int k, rc;
    int min_qual, qual_filter;    // FASTQ only
    int logB;
    HashParams hp;
    WideHash wh;                  // k > 31
    uint32_t *hist;               // [n << logB] raw window counts (pass 1 out / pass 2 cursors)
    const uint64_t *off;          // [n << logB] region offsets (pass 2)
    uint64_t *words;              // dict storage (pass 2)
    uint32_t capacity;            // words per region (pass 2): ranks beyond it are dropped and *overflow is set
    int *overflow;
};

void launch_hist(const ExtractArgs &a, hipStream_t st);
void launch_scatter(const ExtractArgs &a, hipStream_t st);

// exclusive scan of u32 counts into u64 offsets (+ total at out[n]); also max of the counts
// off[r] = r * capacity (fixed-capacity regions of the single-pass path)
void launch_fill_offsets(uint64_t *off, uint64_t n, uint32_t capacity, hipStream_t st);
void launch_scan_u32(const uint32_t *in, uint64_t *out, uint64_t n, uint32_t *max_out, hipStream_t st);

constexpr int SUBIDX = 16;     // sub-ranges indexed per (sample, bucket) region
// in-place sort + dedupe (OR of masks) of every (sample,bucket) region through an order-preserving LDS table
void launch_dedupe(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_regions,
                   uint32_t table_slots, int rem_bits, int *overflow, uint32_t min_n, uint16_t *sidx, int sb, hipStream_t st);

// typical / big_list / big_from: the two-shape launch (skx_device.hip); *overflow bit 4 = relaunch with big_from += dedupe_spill_grid()
uint32_t dedupe_spill_grid();      // workgroups of one second-stage launch (16 384; SKX_DEDUPE_SPILL_GRID overrides, for tests)
void launch_dedupe_mb(uint64_t *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_regions,
                      uint32_t cap, int rem_bits, int *overflow, uint16_t *sidx, int sb, hipStream_t st,
                      uint32_t typical = 0, uint32_t *big_list = nullptr, uint32_t big_from = 0);

struct DictView {
    const uint64_t *words; const uint64_t *off; const uint32_t *ucnt;   // wide: words are u128 (2 x u64), off in elements
    int n_samples, logB, bits;
    // optional sub-index written by the dedupe kernels: sidx[region * 16 + s] = first word of the region whose next
    // `sb` hash bits (below the bucket bits) are >= s.  Turns the per-(sample, sub-bucket) slice search of the union and
    // assemble kernels into two independent loads (nullptr: binary search).
    const uint16_t *sidx = nullptr; int sb = 0;
};

// distinct keys per sub-bucket (logN >= logB): slab j at stage + j*stride, count in ncnt[j]
void launch_union(const DictView &d, int logN, uint64_t *stage, uint32_t stride, uint32_t *ncnt,
                  uint32_t table_slots, int *overflow, hipStream_t st);
// estimate |U|: union restricted to `probe` sub-buckets of a 2^logP split; returns distinct count in *cnt
void launch_union_probe(const DictView &d, int logP, int probe, uint32_t *cnt, uint32_t table_slots, int *overflow,
                        hipStream_t st);

struct AssembleArgs {
    DictView d;
    int logN;                  // sub-buckets of the row keyset
    const uint64_t *stage;     // row keys: slab j at stage + j*stride
    uint32_t stride;
    const uint32_t *ncnt;      // rows per sub-bucket
    const uint64_t *roff;      // exclusive scan of ncnt (row offset of sub-bucket j)
    uint8_t *matrix;           // [n_samples][pitch]
    uint64_t pitch;
    uint32_t *col_present;     // [U] cells != '-'
    uint32_t *col_unambig;     // [U] cells in ACGT
    uint32_t *col_mask;        // [U] bit c set iff IUPAC set-code c (1..15) occurs in the column
    uint32_t max_rows;         // max ncnt (LDS sizing)
    int *missing;              // set if a dict key is not among the rows
    uint32_t j_base = 0;       // first sub-bucket of this launch
    uint64_t col_base = 0;     // mode 0: output column of row r is r - col_base (a window of rows in a small buffer)
    const uint8_t *keep = nullptr; const uint64_t *kpos = nullptr;    // mode 2: row flags (1 = kept) and their exclusive scan [U + 1]
    int mask_ambig = 0;        // mode 2: ambiguous cells are written as 'N' (MergeSkaArray::filter's ambig_mask)
};
// mode 0: matrix + statistics; 1: statistics only; 2: kept rows only
void launch_assemble(const AssembleArgs &a, hipStream_t st, int mode = 0, uint32_t n_blocks = 0);
// the union that also notes, per word of the dictionaries, where its key went (side: one u16 per element of d.words; perm: [2^logN][stride]
// u16), and the assemble that fills the matrix from those notes instead of the words (64-bit keys, whole array, mode 0)
bool union_side_ok(const DictView &d, int logN, uint32_t stride);
void launch_union_side(const DictView &d, int logN, uint64_t *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots, int *overflow,
                       uint16_t *side, uint16_t *perm, hipStream_t st);
void launch_assemble_side(const AssembleArgs &a, const uint16_t *side, const uint16_t *perm, hipStream_t st, bool wide = false);
void launch_compose_perm(const uint64_t *l_stage, uint32_t l_stride, const uint32_t *l_ncnt, const uint16_t *l_perm, int l_logN, const uint64_t *g_stage,
                         uint32_t g_stride, const uint32_t *g_ncnt, const uint64_t *g_roff, int g_logN, int bits, uint16_t *g_perm, uint32_t *g_n,
                         uint64_t *g_base, uint32_t *g_max, int *bad, hipStream_t st);
void launch_union_side_wide(const DictView &d, int logN, u128 *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots, int *overflow,
                            uint16_t *side, uint16_t *perm, hipStream_t st);

// ---- MergeSkaDict::append straight from the extraction kernel's regions (skx_append.hip): no per-sample sorted dictionaries ----
// A *row block* j holds the rows whose top logQ hash bits are j (logQ >= logB); its workgroup appends the samples in order and leaves, per
// sample, a *piece*: the sample's cells of the block as 4-bit base sets indexed by first-seen rank, plen[j * S + s] of them, at
// pieces + (j * S + s) * (cap / 2).  perm[j * cap + rank] = row of the block (key order) or 0xFFFF; nrank[j] = ranks handed out.
constexpr uint32_t APPEND_MAX_CAP = 6016, APPEND_MAX_SLOTS = 8192;    // what sixteen row buffers, the queues and the table leave of 160 KB
struct AppendArgs {
    const uint64_t *words; const uint64_t *off; const uint32_t *raw;      // regions as extract_kernel left them (off in words, raw = fill)
    int n_samples, logB, bits, logQ;
    uint32_t nslots, cap;                                                // table slots; ranks a block may hand out (multiple of 32)
    uint8_t *pieces; uint16_t *plen; uint16_t *perm; uint32_t *nrank;
    uint64_t *stage; uint32_t stride; uint32_t *ncnt;                    // row keys of block j: sorted slab at stage + j * stride
    unsigned long long *probe;                                           // count-only launches: [0] += rows, [1] = max rows of a block
    int *overflow;                                                       // |= 1 table / ranks / slab full, |= 2 queue full
    uint32_t rounds = 1;                                                 // row blocks per workgroup: the launch has (1 << logQ) / rounds workgroups (launch_append)
    int *bar = nullptr;                                                  // [1 << logB] zeroed: where the readers of a region meet before a round
};
bool append_ok(int bits, int logB, int logQ, uint32_t region_cap, uint32_t nslots, uint32_t cap);
void launch_append(const AppendArgs &a, uint32_t region_cap, hipStream_t st);
void launch_append_probe(const AppendArgs &a, uint32_t region_cap, unsigned blocks, hipStream_t st);     // the first `blocks` row blocks, rows counted only
// the same pass over 16-byte words (k > 31; skx_append_wide.inc): words / stage address u128 elements, off counts them; 16-byte table entries
constexpr uint32_t APPEND_WIDE_MAX_CAP = 3072, APPEND_WIDE_MAX_SLOTS = 4096;
bool append_wide_ok(int bits, int logB, int logQ, uint32_t nslots, uint32_t cap);
void launch_append_wide(const AppendArgs &a, hipStream_t st);
void launch_append_wide_probe(const AppendArgs &a, unsigned blocks, hipStream_t st);
// row statistics from the pieces (present, unambiguous, code set, variant_count), written to the rows in the order of H
void launch_pieces_stats(const uint8_t *pieces, const uint16_t *plen, const uint16_t *perm, const uint32_t *nrank, const uint32_t *ncnt, const uint64_t *roff, uint32_t cap,
                         int n_samples, int n_blocks, uint32_t *present, uint32_t *unambig, uint32_t *mask, uint32_t *vcount, hipStream_t st);
void launch_pieces_cells(const uint8_t *pieces, const uint16_t *plen, uint32_t cap, int n_samples, int n_blocks, unsigned long long *out, hipStream_t st);
void launch_region_totals(const uint32_t *raw, int n_samples, int logB, unsigned long long *out, hipStream_t st);
struct PiecesRowsArgs {
    const uint8_t *pieces; const uint16_t *plen; const uint16_t *perm; const uint32_t *nrank; uint32_t cap; int n_samples;
    const uint32_t *ncnt; const uint64_t *roff;                          // rows per block and their exclusive scan
    uint8_t *out; uint64_t pitch;                                        // cell (s, c) at out + s * pitch + c
    uint32_t j_base = 0; uint64_t col_base = 0;                          // all rows: block j's rows land at columns roff[j] - col_base
    const uint8_t *keep = nullptr; const uint64_t *kpos = nullptr;        // kept rows only: row r (keep[r] == 1) lands at column kpos[r]
    int mask_ambig = 0;                                                  // ambiguous cells are written as 'N'
    int samples_per_wg = 0;                                              // (set by the launcher)
};
void launch_pieces_rows(const PiecesRowsArgs &a, uint32_t n_blocks, hipStream_t st);      // blocks [j_base, j_base + n_blocks)

// compact slabs into one array; unhash=1 converts engine-order words back to reference keys
void launch_gather_keys(const uint64_t *stage, uint32_t stride, const uint32_t *ncnt, const uint64_t *roff, int n_sub,
                        uint64_t *out, int unhash, HashParams hp, hipStream_t st);
void launch_hash_keys(const uint64_t *keys, uint64_t *words, uint64_t n, HashParams hp, hipStream_t st);
// packed words -> the .skf's split k-mer list (9 bytes of CBOR per key, out padded to 9 n + 16); *short_key = 1 if a key needs fewer
void launch_keys_cbor(const uint64_t *words, uint64_t n, HashParams hp, uint8_t *out, int *short_key, hipStream_t st);
void launch_unhash_dict(const uint64_t *words, uint64_t n, uint64_t *keys, uint8_t *bases, HashParams hp, hipStream_t st);

// column statistics of a sample-major matrix (for arrays that did not come from assemble)
void launch_col_stats(const uint8_t *matrix, uint64_t pitch, int n_samples, uint64_t n_cols, uint32_t *present,
                      uint32_t *unambig, uint32_t *mask, int *bad_byte, hipStream_t st);
// MergeSkaArray::filter row rule -> keep flags (u8)
struct FilterArgs {
    const uint32_t *vcount, *present, *unambig, *mask; uint64_t n_cols; uint32_t n_samples;
    uint64_t min_count; int ambig_as_missing, filter_type, ignore_const_gaps;
    uint8_t *keep;
    int two_stage = 0;         // generic_modes::distance's pair of filters in one pass: keep = 3 marks a row its second stage removes
};
void launch_filter_flags(const FilterArgs &a, hipStream_t st);
// streaming load: statistics of row-major rows (S cells each), and their kept rows into the sample-major matrix at col0 + pos[r]
void launch_row_stats_rm(const uint8_t *cells, uint64_t S, uint64_t nr, uint32_t *present, uint32_t *unambig, uint32_t *mask, int *bad_byte, hipStream_t st);
void launch_compact_rm(const uint8_t *cells, uint64_t S, uint64_t nr, const uint8_t *keep, const uint64_t *pos, uint8_t *out, uint64_t pitch,
                       uint64_t col0, int mask_ambig, hipStream_t st);
uint64_t scan_u8_blocks(uint64_t n);      // scratch of launch_scan_u8: sums[blocks] u32 + offs[blocks + 1] u64, owned by the caller
void launch_scan_u8(const uint8_t *flags, uint64_t *pos, uint64_t n, uint32_t *sums, uint64_t *offs, hipStream_t st);   // pos[n] = total
// out[s][pos[c]] = in[s][c] for kept c (optionally ambiguous -> 'N'); also compacts the stat/key arrays
void launch_compact_matrix(const uint8_t *in, uint64_t in_pitch, uint8_t *out, uint64_t out_pitch, int n_samples,
                           uint64_t n_cols, const uint8_t *keep, const uint64_t *pos, int mask_ambig, hipStream_t st);
void launch_compact_u32(const uint32_t *in, uint32_t *out, uint64_t n, const uint8_t *keep, const uint64_t *pos, hipStream_t st);
void launch_compact_u64(const uint64_t *in, uint64_t *out, uint64_t n, const uint8_t *keep, const uint64_t *pos, hipStream_t st);
void launch_compact_u128(const uint64_t *in, uint64_t *out, uint64_t n, const uint8_t *keep, const uint64_t *pos, hipStream_t st);
void launch_mask_ambig_stats(uint32_t *mask, uint64_t n, hipStream_t st);
void launch_count_u8(const uint8_t *v, uint64_t n, uint8_t value, unsigned long long *out, hipStream_t st);
void launch_differ_u32(const uint32_t *a, const uint32_t *b, uint64_t n, int *flag, hipStream_t st);      // *flag = 1 if a[i] != b[i] anywhere
// tiled transpose of a byte matrix: in [rows][in_pitch] -> out [cols][out_pitch]
void launch_transpose(const uint8_t *in, uint64_t in_pitch, uint64_t rows, uint64_t cols, uint8_t *out, uint64_t out_pitch,
                      hipStream_t st);
// .skf data section on the device (skx_snappy.hip): one snappy-frame chunk per wavefront
struct SnapChunk { uint64_t src_off, uoff; uint32_t src_len, ulen, crc, compressed; };   // where its bytes are, where its output belongs
constexpr uint32_t SKF_SLOT = 8 + 65536;                                                  // frame header + largest payload
int launch_skf_decode_cells(int device, const uint8_t *src, const SnapChunk *chunks, uint32_t n_chunks, uint64_t upos, uint64_t uend,
                            uint8_t *scratch /* n_chunks x 64 KB */, uint8_t *cells, uint64_t base_cell, int *status, hipStream_t st);
int launch_skf_encode_cells(int device, const uint8_t *cells, uint64_t base_cell, uint64_t upos, uint64_t uoff0, uint32_t n_chunks,
                            uint8_t *slots, uint32_t *sizes, hipStream_t st);
void launch_skf_gather(const uint8_t *slots, const uint32_t *sizes, const uint64_t *off, uint32_t n_chunks, uint8_t *dense, hipStream_t st);
// per-sample count of cells != '-'
void launch_row_nonmissing(const uint8_t *matrix, uint64_t pitch, int n_samples, uint64_t n_cols, unsigned long long *out,
                           hipStream_t st);

// ---- distance: bit planes + pair popcounts ----
// planes[p][s][w]: p in {present, A, C, G, T, sz1, sz2, sz3}, w = 64 columns per word
void launch_build_planes(const uint8_t *matrix, uint64_t pitch, int n_samples, uint64_t n_cols, uint64_t *planes, uint64_t wpr, int filt,
                         hipStream_t st);
// the same over the rows with keep == 1 only (pos = exclusive scan of the flags): keep_bits / gpos hold one word per 64 rows; planes zeroed by the caller
void launch_keep_bits(const uint8_t *keep, const uint64_t *pos, uint64_t n, uint64_t *keep_bits, uint64_t *gpos, hipStream_t st);
void launch_split_keep(const uint8_t *keep, const uint32_t *col_mask, uint64_t n, uint8_t *clean, uint8_t *dirty, hipStream_t st, int mode = 0);
void launch_split_keep_from_dirty(uint8_t *dirty, uint64_t n, uint8_t *clean, hipStream_t st);
void launch_build_planes_keep(const uint8_t *matrix, uint64_t pitch, int n_samples, uint64_t n_cols, const uint64_t *keep_bits, const uint64_t *gpos,
                              uint64_t *planes, uint64_t wpr, int filt, hipStream_t st, uint32_t *first_group, uint64_t kept);
// (first_group: [kept / 4096 + 2] words of scratch; kept = rows that stay: every word [0, wpr) of every plane and sample is written)
// out[pair][c]: integer pair-class counts, see skx_device.hip
constexpr int DIST_NCOUNT = 16;
int launch_pair_counts(const uint64_t *planes, int n_samples, uint64_t words_per_row, int filt_ambig,
                       unsigned long long *out, hipStream_t st, int i_lo = 0, int i_hi = 0);        // rows [i_lo, i_hi) of the pair matrix (0, 0: all); 0 or a hipError_t

// ---- wide (k > 31) launchers: same roles as their 64-bit counterparts; word pointers address u128 elements ----
void launch_hist_wide(const ExtractArgs &a, hipStream_t st);
void launch_scatter_wide(const ExtractArgs &a, hipStream_t st);
int extract_tile_bases_wide();
void launch_dedupe_wide(u128 *words, const uint64_t *off, const uint32_t *raw, uint32_t *ucnt, uint64_t n_regions, uint32_t cap,
                        int rem_bits, int *overflow, uint16_t *sidx, int sb, hipStream_t st);
void launch_union_wide(const DictView &d, int logN, u128 *stage, uint32_t stride, uint32_t *ncnt, uint32_t table_slots, int *overflow,
                       hipStream_t st);
void launch_union_probe_wide(const DictView &d, int logP, int probe, uint32_t *cnt, uint32_t table_slots, int *overflow, hipStream_t st);
void launch_assemble_wide(const AssembleArgs &a, hipStream_t st, int mode = 0, uint32_t n_blocks = 0);      // a.stage addresses u128 slabs; modes as launch_assemble
void launch_gather_keys_wide(const u128 *stage, uint32_t stride, const uint32_t *ncnt, const uint64_t *roff, int n_sub, u128 *out,
                             hipStream_t st);

// ---- FASTA text -> record stream (skx_parse.hip): tiles of 16 KB; scratch per tile: 8 B summary, 8 B offset, 1 B kind
uint64_t fasta_parse_tiles(uint64_t len);
// Packed read sets (fastx.cpp pack_*_planes: groups of 64 positions, five words each -- two code bits, the bytes valid_base rejects, line ends,
// quality verdicts).  planes_bytes16: sixteen positions from position 16 t as the bytes the read-set kernels look at: sequence A C T G (any
// byte of that code), N (rejected), '\n'; quality ' ' (passes every min_qual below 255: (32 - 33) & 255 = 255), '!' (fails every one), '\n';
// positions from `len` on are line ends.
#ifdef __HIPCC__
__device__ static inline void planes_bytes16(const uint64_t *groups, uint64_t t, uint64_t len, uint32_t sw[4], uint32_t qw[4])
{
    const uint64_t p0 = t * 16;
    const uint64_t *g = groups + (t >> 2) * 5;
    const int sh = (int)(t & 3) * 16;
    const uint32_t lo = (uint32_t)(g[0] >> sh) & 0xFFFFu, hi = (uint32_t)(g[1] >> sh) & 0xFFFFu, bad = (uint32_t)(g[2] >> sh) & 0xFFFFu,
                   nl = (uint32_t)(g[3] >> sh) & 0xFFFFu, qb = (uint32_t)(g[4] >> sh) & 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = 0, y = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = 4 * i + b;
            const uint32_t code = ((lo >> j) & 1u) | (((hi >> j) & 1u) << 1);
            const bool isnl = ((nl >> j) & 1u) || p0 + (uint64_t)j >= len;
            const uint32_t c = isnl ? 10u : ((bad >> j) & 1u) ? 78u : (0x47544341u >> (8 * code)) & 0xFFu;      // "ACTG"
            const uint32_t q = isnl ? 10u : ((qb >> j) & 1u) ? 33u : 32u;
            x |= c << (8 * b); y |= q << (8 * b);
        }
        sw[i] = x; qw[i] = y;
    }
}
#endif
// the same as two record streams in device memory (16 bytes of slack past len): the forms that read streams (sort-based read sets, tests)
void launch_expand_planes(const uint64_t *groups, uint64_t len, uint8_t *seq, uint8_t *qual, hipStream_t st);
void launch_fasta_parse(const uint8_t *const *raw, const uint64_t *rawlen, uint8_t *const *out, uint64_t *outlen, const uint32_t *tile_file,
                        const uint64_t *tile_base, uint64_t n_tiles, void *summary, uint64_t *tile_off, uint8_t *tile_kind, int n, hipStream_t st);

// ---- row-set operations for `ska merge` / `ska weed` / `ska delete` (skx_setops.hip)
void launch_lookup_rows(const uint64_t *words, uint64_t n, const uint64_t *sorted, uint64_t m, uint32_t *idx, hipStream_t st);
void launch_lookup_rows_wide(const u128 *words, uint64_t n, const u128 *sorted, uint64_t m, uint32_t *idx, hipStream_t st);
void launch_hash_keys_wide(const u128 *keys, u128 *words, uint64_t n, const WideHash &wh, hipStream_t st);
void launch_map_lookup_wide(const uint64_t *wlo, const uint64_t *whi, const uint8_t *flag, const uint8_t *seq, uint64_t len, int h, const u128 *sorted,
                            const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc, hipStream_t st);
void launch_member_flags(const uint32_t *idx, uint64_t n, int reverse, uint8_t *keep, hipStream_t st);
void launch_scatter_rows(const uint8_t *src, uint64_t src_pitch, int n_samples, uint8_t *dst, uint64_t dst_pitch, const uint32_t *idx,
                         uint64_t n, hipStream_t st);
void launch_nonzero_flags(const uint32_t *v, uint64_t n, uint8_t *keep, hipStream_t st);

// ---- `ska map` (skx_reads.hip)
struct MapWriteArgs {
    const uint8_t *mv; uint64_t mpitch, M;            // mapped variants, sample-major [n_samples][mpitch]
    const uint32_t *m_pos, *m_chrom;                  // middle position inside its chromosome / chromosome of mapped window m
    const uint64_t *mlo, *mhi;                        // per chromosome: its range in the mapped list
    const uint8_t *refcat; uint64_t total;            // reference bases, chromosomes concatenated (= output coordinates)
    const uint64_t *clen, *coff; int n_chrom;         // per chromosome: length, offset in the output
    uint64_t half; int ambig_mask;
    const uint64_t *repeat; uint64_t n_repeat;        // output coordinates to mask (repeat_coors)
    uint8_t *out; uint64_t opitch; int n_samples;     // pseudoalignments [n_samples][opitch], pre-filled with '-'
    uint32_t *pres; uint64_t ppitch;                  // presence bits of the middle bases [n_samples][ppitch] (zeroed, >= total/32 + 3 words)
    uint32_t *first, *last;                           // [n_samples][n_chrom] scratch
};
void launch_map_lookup(const uint64_t *wlo, const uint8_t *flag, const uint8_t *seq, uint64_t len, int h, const uint64_t *sorted,
                       const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc, hipStream_t st);
void launch_gather_mapped(const uint8_t *matrix, uint64_t pitch, int n_samples, const uint32_t *mapped, const uint32_t *row, const uint8_t *is_rc,
                          uint64_t M, uint8_t *mv, uint64_t mpitch, hipStream_t st);
void launch_aln_write(const MapWriteArgs &a, hipStream_t st);

}  // namespace skx
