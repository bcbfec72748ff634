// skx_setops.hip -- row-set operations on MergeSkaArrays for the .skf life-cycle:
//   `ska merge`  : to_dict + MergeSkaDict::extend + MergeSkaArray::new   (generic_modes.rs:90-106, merge_ska_dict.rs:160-193)
//   `ska weed`   : MergeSkaArray::weed                                   (merge_ska_array.rs:452-487)
// Rows are identified by their packed word (H(key) << 4 | 1), so "same split k-mer" is one 64-bit compare and every
// set operation is a sort / search in the order of H.  The sort and the unique are the engine's own primitives
// (skx_prims.hip: stable LSD radix sort, compaction; rocPRIM calls until round 5); the look-ups and the column scatter are below.
#include <algorithm>
#include <cstring>
#include "skx_internal.h"

namespace skx {


// sorted, duplicate-free copy of `n` packed words (any order, duplicates allowed) -> out[0 .. *n_out)
int sort_unique_words(const uint64_t *in, uint64_t n, DevBuf<uint64_t> &out, uint64_t *n_out, hipStream_t st)
{
    *n_out = 0;
    if (!n) return SKX_OK;
    if (n > 0x7FFFFFFFull) { set_error("more than 2^31 rows in one row-set operation"); return SKX_EUNSUP; }
    DevBuf<uint64_t> sorted;
    SKX_TRY(sorted.alloc(n)); SKX_TRY(out.alloc(n));
    SKX_TRY(prim_sort_keys_u64(in, sorted.p, n, 64, st));
    return prim_unique_u64(sorted.p, out.p, n, n_out, st);
}

// position of every word in a sorted, duplicate-free word list (compare on word >> 4); 0xFFFFFFFF when absent
__global__ __launch_bounds__(256) void lookup_rows_kernel(const uint64_t *words, uint64_t n, const uint64_t *sorted, uint64_t m, uint32_t *idx)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = words[i] >> 4;
    uint64_t lo = 0, hi = m;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((sorted[mid] >> 4) < key) lo = mid + 1; else hi = mid; }
    idx[i] = (lo < m && (sorted[lo] >> 4) == key) ? (uint32_t)lo : 0xFFFFFFFFu;
}
void launch_lookup_rows(const uint64_t *words, uint64_t n, const uint64_t *sorted, uint64_t m, uint32_t *idx, hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(lookup_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, n, sorted, m, idx);
}

// the same for 128-bit keys (k > 31): both lists hold (H(key) << 4 | code bits) as u128, compared on the key bits
__global__ __launch_bounds__(256) void lookup_rows_wide_kernel(const u128 *words, uint64_t n, const u128 *sorted, uint64_t m, uint32_t *idx)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const u128 key = words[i] >> 4;
    uint64_t lo = 0, hi = m;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((sorted[mid] >> 4) < key) lo = mid + 1; else hi = mid; }
    idx[i] = (lo < m && (sorted[lo] >> 4) == key) ? (uint32_t)lo : 0xFFFFFFFFu;
}
void launch_lookup_rows_wide(const u128 *words, uint64_t n, const u128 *sorted, uint64_t m, uint32_t *idx, hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(lookup_rows_wide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, n, sorted, m, idx);
}

// 128-bit row keys of arrays that came from files or from the caller (stored as the reference stores them) -> packed words
__global__ __launch_bounds__(256) void hash_keys_wide_kernel(const u128 *keys, u128 *words, uint64_t n, WideHash wh)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) words[i] = (hmix_w(keys[i], wh) << 4) | (u128)1;
}
void launch_hash_keys_wide(const u128 *keys, u128 *words, uint64_t n, const WideHash &wh, hipStream_t st)
{
    if (!n) return;
    const unsigned g = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(hash_keys_wide_kernel, dim3(g), dim3(256), 0, st, keys, words, n, wh);
}
// sort_unique_words for 128-bit words (out holds 2 x u64 per word)
int sort_unique_wide(const u128 *in, uint64_t n, DevBuf<uint64_t> &out, uint64_t *n_out, hipStream_t st)
{
    *n_out = 0;
    if (!n) return SKX_OK;
    if (n > 0x7FFFFFFFull) { set_error("more than 2^31 rows in one row-set operation"); return SKX_EUNSUP; }
    DevBuf<uint64_t> sorted;
    SKX_TRY(sorted.alloc(2 * n)); SKX_TRY(out.alloc(2 * n));
    SKX_TRY(prim_sort_keys_u128(in, (u128 *)sorted.p, n, 128, st));
    return prim_unique_keys_u128((const u128 *)sorted.p, (u128 *)out.p, n, n_out, st);      // one word per key: equal above the base-set bits
}
__global__ __launch_bounds__(256) void iota_rows_kernel(uint32_t *v, uint64_t n)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
// sort_words_perm for 128-bit words: sorted copy + the row each sorted word came from
int sort_wide_perm(const u128 *words, uint64_t n, DevBuf<uint64_t> &sorted, DevBuf<uint32_t> &perm, hipStream_t st)
{
    DevBuf<uint32_t> iota;
    SKX_TRY(sorted.alloc(2 * n)); SKX_TRY(perm.alloc(n)); SKX_TRY(iota.alloc(n));
    if (!n) return SKX_OK;
    if (n > 0x7FFFFFFFull) { set_error("more than 2^31 rows in one row-set operation"); return SKX_EUNSUP; }
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, iota.p, n);
    return prim_sort_pairs_u128(words, (u128 *)sorted.p, iota.p, perm.p, n, 128, st);
}
// map_lookup for 128-bit words: the window words arrive as two 64-bit halves
__global__ __launch_bounds__(256) void map_lookup_wide_kernel(const uint64_t *wlo, const uint64_t *whi, const uint8_t *flag, const uint8_t *seq, uint64_t len,
                                                             int h, const u128 *sorted, const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc)
{
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= len) return;
    uint32_t r = 0xFFFFFFFFu; uint8_t rcf = 0;
    if (flag[p]) {
        const uint64_t w = wlo[p];
        const u128 key = (((u128)whi[p] << 64) | w) >> 4;
        uint64_t lo = 0, hi = U;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((sorted[mid] >> 4) < key) lo = mid + 1; else hi = mid; }
        if (lo < U && (sorted[lo] >> 4) == key) r = perm ? perm[lo] : (uint32_t)lo;
        const uint32_t mid_code = (seq[p - h] >> 1) & 3u;
        rcf = ((uint32_t)w & 15u) == (1u << (mid_code ^ 2u));
    }
    row[p] = r; is_rc[p] = rcf;
}
void launch_map_lookup_wide(const uint64_t *wlo, const uint64_t *whi, const uint8_t *flag, const uint8_t *seq, uint64_t len, int h, const u128 *sorted,
                            const uint32_t *perm, uint64_t U, uint32_t *row, uint8_t *is_rc, hipStream_t st)
{
    if (!len) return;
    hipLaunchKernelGGL(map_lookup_wide_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, wlo, whi, flag, seq, len, h, sorted, perm, U, row, is_rc);
}

// weed: keep[i] = 1 when the row survives ((!reverse && !found) || (reverse && found), merge_ska_array.rs:468), else 0
__global__ __launch_bounds__(256) void member_flags_kernel(const uint32_t *idx, uint64_t n, int reverse, uint8_t *keep)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const bool found = idx[i] != 0xFFFFFFFFu;
    keep[i] = (uint8_t)(reverse ? found : !found);
}
void launch_member_flags(const uint32_t *idx, uint64_t n, int reverse, uint8_t *keep, hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(member_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, n, reverse, keep);
}

// merge: the columns (samples) of one input array land in the merged matrix, row r of the input at row idx[r] of the
// result.  Sample-major on both sides: reads are coalesced, the byte writes follow idx (monotone when both arrays are in
// the order of H, so mostly sequential too).
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint8_t *src, uint64_t src_pitch, uint8_t *dst, uint64_t dst_pitch,
                                                          const uint32_t *idx, uint64_t n)
{
    const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
    if (r >= n) return;
    const uint64_t s = blockIdx.y;
    dst[s * dst_pitch + idx[r]] = src[s * src_pitch + r];
}
void launch_scatter_rows(const uint8_t *src, uint64_t src_pitch, int n_samples, uint8_t *dst, uint64_t dst_pitch, const uint32_t *idx,
                         uint64_t n, hipStream_t st)
{
    if (!n || !n_samples) return;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n_samples), dim3(256), 0, st, src, src_pitch, dst, dst_pitch, idx, n);
}

// keep[i] = present[i] > 0  (update_counts(false) after delete_samples, merge_ska_array.rs:139-163,270)
__global__ __launch_bounds__(256) void nonzero_flags_kernel(const uint32_t *v, uint64_t n, uint8_t *keep)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i < n) keep[i] = v[i] != 0u;
}
void launch_nonzero_flags(const uint32_t *v, uint64_t n, uint8_t *keep, hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(nonzero_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, n, keep);
}

}  // namespace skx
