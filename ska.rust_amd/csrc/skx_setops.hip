// skx_setops.hip -- row-set operations on MergeSkaArrays for the .skf life-cycle:
//   `ska merge`  : to_dict + MergeSkaDict::extend + MergeSkaArray::new   (generic_modes.rs:90-106, merge_ska_dict.rs:160-193)
//   `ska weed`   : MergeSkaArray::weed                                   (merge_ska_array.rs:452-487)
// Rows are identified by their packed word (H(key) << 4 | 1), so "same split k-mer" is one 64-bit compare and every
// set operation is a sort / search in the order of H.  The sort and the run-length unique are rocPRIM device
// primitives (plain library calls); the look-ups and the column scatter are hand-written below.
#include <cstring>
#include "skx_internal.h"
#include <rocprim/rocprim.hpp>

namespace skx {

#define RPS(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return SKX_ENODEV; } while (0)

// sorted, duplicate-free copy of `n` packed words (any order, duplicates allowed) -> out[0 .. *n_out)
int sort_unique_words(const uint64_t *in, uint64_t n, DevBuf<uint64_t> &out, uint64_t *n_out, hipStream_t st)
{
    *n_out = 0;
    if (!n) return SKX_OK;
    if (n > 0x7FFFFFFFull) return SKX_EUNSUP;
    DevBuf<uint64_t> sorted; DevBuf<unsigned char> tmp; DevBuf<unsigned int> d_cnt;
    SKX_TRY(sorted.alloc(n)); SKX_TRY(out.alloc(n)); SKX_TRY(d_cnt.alloc(1));
    size_t bytes = 0;
    RPS(rocprim::radix_sort_keys(nullptr, bytes, in, sorted.p, (unsigned int)n, 0, 64, st));
    SKX_TRY(tmp.alloc(bytes ? bytes : 1));
    RPS(rocprim::radix_sort_keys(tmp.p, bytes, in, sorted.p, (unsigned int)n, 0, 64, st));
    size_t bytes2 = 0;
    RPS(rocprim::unique(nullptr, bytes2, sorted.p, out.p, d_cnt.p, (unsigned int)n, rocprim::equal_to<uint64_t>(), st));
    DevBuf<unsigned char> tmp2; SKX_TRY(tmp2.alloc(bytes2 ? bytes2 : 1));
    RPS(rocprim::unique(tmp2.p, bytes2, sorted.p, out.p, d_cnt.p, (unsigned int)n, rocprim::equal_to<uint64_t>(), st));
    unsigned int cnt = 0;
    RPS(hipMemcpyAsync(&cnt, d_cnt.p, 4, hipMemcpyDeviceToHost, st));
    RPS(hipStreamSynchronize(st));
    *n_out = cnt;
    return SKX_OK;
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
