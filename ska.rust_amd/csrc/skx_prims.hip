// skx_prims.hip -- the engine's own device primitives for its sort-based forms (gfx950, wave64): stable LSD radix sort of 64- and 128-bit
// keys (with or without 32-bit values), inclusive scans, compaction (select / unique / run starts) and a segmented OR-scan.  Until round 5
// these were rocPRIM calls (`ska merge` / `weed` key sorts, `ska map`'s row order, `ska cov`, the one-shot read-set form); they are not on
// the headline path, so the kernels are plain: three launches per sort pass (per-wave digit histograms, one scan, a stable scatter in which
// a wave ranks its 64 keys per digit with eight ballots and no barrier), three per scan or compaction (block sums, their scan, the blocks
// again with their carries).  Everything is stable and deterministic: the callers rely on "equal keys stay in stream order".
#include "skx_internal.h"
#include <algorithm>
#include <vector>

namespace skx {

namespace {
constexpr int PR_NT = 256, PR_ITEMS = 16, PR_TILE = PR_NT * PR_ITEMS;       // scans / compactions: a block owns 4 096 consecutive items
constexpr int RS_WAVE_ITEMS = 16, RS_WTILE = 64 * RS_WAVE_ITEMS;            // sorts: a WAVE owns 1 024 consecutive keys (sixteen rounds of 64)
#define PRH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ inline uint32_t mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// ---------------------------------------------------------------------------------------------------------------- scans
struct OpAdd { __device__ static uint32_t id() { return 0u; } __device__ static uint32_t f(uint32_t a, uint32_t b) { return a + b; } };
struct OpMax { __device__ static uint32_t id() { return 0u; } __device__ static uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; } };

template <typename Op> __device__ inline uint32_t wave_incl(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if ((int)lane_id() >= d) v = Op::f(t, v); }
    return v;
}
// inclusive scan of one value per thread over the block; *total = the block's aggregate.  s_tmp: 8 words
template <typename Op> __device__ inline uint32_t block_incl(uint32_t v, uint32_t *s_tmp, uint32_t *total)
{
    const int wv = threadIdx.x >> 6;
    const uint32_t inc = wave_incl<Op>(v);
    if (lane_id() == 63) s_tmp[wv] = inc;
    __syncthreads();
    uint32_t carry = Op::id();
    for (int w = 0; w < wv; w++) carry = Op::f(carry, s_tmp[w]);
    uint32_t tot = Op::id();
    for (int w = 0; w < PR_NT / 64; w++) tot = Op::f(tot, s_tmp[w]);
    *total = tot;
    __syncthreads();
    return Op::f(carry, inc);
}
template <typename Op>
__global__ __launch_bounds__(PR_NT) void scan_sums_kernel(const uint32_t *in, uint64_t n, uint32_t *sums)
{
    __shared__ uint32_t s_tmp[8];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    uint32_t acc = Op::id();
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (base + j < n) acc = Op::f(acc, in[base + j]);
    uint32_t tot; (void)block_incl<Op>(acc, s_tmp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
// what lies before a thread in the block's order, from the block-level inclusive values: the inclusive value of the thread before it (max has no
// inverse, so "inclusive less own" will not do).  s_last: one word per wave
template <typename Op> __device__ inline uint32_t block_before(uint32_t inc, uint32_t *s_last)
{
    const uint32_t prev = __shfl_up(inc, 1, 64);
    if (lane_id() == 63) s_last[threadIdx.x >> 6] = inc;
    __syncthreads();
    const uint32_t pre = lane_id() ? prev : ((threadIdx.x >> 6) ? s_last[(threadIdx.x >> 6) - 1] : Op::id());
    __syncthreads();
    return pre;
}
// the block sums -> their exclusive scan, in place (one block walks them 4 096 at a time)
template <typename Op>
__global__ __launch_bounds__(PR_NT) void scan_carries_kernel(uint32_t *sums, uint64_t nb)
{
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_last[PR_NT / 64];
    uint32_t run = Op::id();
    for (uint64_t c0 = 0; c0 < nb; c0 += PR_TILE) {
        const uint64_t base = c0 + (uint64_t)threadIdx.x * PR_ITEMS;
        uint32_t v[PR_ITEMS], acc = Op::id();
#pragma unroll
        for (int j = 0; j < PR_ITEMS; j++) { v[j] = base + j < nb ? sums[base + j] : Op::id(); acc = Op::f(acc, v[j]); }
        uint32_t tot;
        const uint32_t inc = block_incl<Op>(acc, s_tmp, &tot);
        uint32_t before = Op::f(run, block_before<Op>(inc, s_last));
#pragma unroll
        for (int j = 0; j < PR_ITEMS; j++) if (base + j < nb) { sums[base + j] = before; before = Op::f(before, v[j]); }
        run = Op::f(run, tot);
    }
}
template <typename Op, bool EXCLUSIVE>
__global__ __launch_bounds__(PR_NT) void scan_apply_kernel(const uint32_t *in, uint32_t *out, uint64_t n, const uint32_t *carries)
{
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_last[PR_NT / 64];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    uint32_t v[PR_ITEMS], acc = Op::id();
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) { v[j] = base + j < n ? in[base + j] : Op::id(); acc = Op::f(acc, v[j]); }
    uint32_t tot;
    const uint32_t inc = block_incl<Op>(acc, s_tmp, &tot);
    uint32_t run = Op::f(carries[blockIdx.x], block_before<Op>(inc, s_last));
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (base + j < n) { const uint32_t nx = Op::f(run, v[j]); out[base + j] = EXCLUSIVE ? run : nx; run = nx; }
}
template <typename Op, bool EXCLUSIVE>
int scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st)
{
    if (!n) return SKX_OK;
    const uint64_t nb = (n + PR_TILE - 1) / PR_TILE;
    DevBuf<uint32_t> sums; SKX_TRY(sums.alloc(nb));
    hipLaunchKernelGGL(scan_sums_kernel<Op>, dim3((unsigned)nb), dim3(PR_NT), 0, st, in, n, sums.p);
    hipLaunchKernelGGL(scan_carries_kernel<Op>, dim3(1), dim3(PR_NT), 0, st, sums.p, nb);
    hipLaunchKernelGGL((scan_apply_kernel<Op, EXCLUSIVE>), dim3((unsigned)nb), dim3(PR_NT), 0, st, in, out, n, sums.p);
    PRH(hipStreamSynchronize(st));                       // (sums goes out of scope)
    return SKX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- compaction
// Flag functors say which items stay; emit functors store item i at place pos.  Order is kept.
struct FlagU8 { const uint8_t *f; __device__ bool operator()(uint64_t i) const { return f[i] != 0; } };
struct FlagU32 { const uint32_t *f; __device__ bool operator()(uint64_t i) const { return f[i] != 0u; } };
struct FlagNewU64 { const uint64_t *k; __device__ bool operator()(uint64_t i) const { return i == 0 || k[i] != k[i - 1]; } };
struct FlagNewKeyU128 { const u128 *k; __device__ bool operator()(uint64_t i) const { return i == 0 || (k[i] >> 4) != (k[i - 1] >> 4); } };      // equal above the base-set bits
struct EmitIndex { uint32_t *out; __device__ void operator()(uint64_t i, uint64_t pos) const { out[pos] = (uint32_t)i; } };
struct EmitU32 { const uint32_t *in; uint32_t *out; __device__ void operator()(uint64_t i, uint64_t pos) const { out[pos] = in[i]; } };
struct EmitU64 { const uint64_t *in; uint64_t *out; __device__ void operator()(uint64_t i, uint64_t pos) const { out[pos] = in[i]; } };
struct EmitU128 { const u128 *in; u128 *out; __device__ void operator()(uint64_t i, uint64_t pos) const { out[pos] = in[i]; } };

template <typename Flag>
__global__ __launch_bounds__(PR_NT) void compact_count_kernel(Flag flag, uint64_t n, uint32_t *sums)
{
    __shared__ uint32_t s_tmp[8];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (base + j < n) c += flag(base + j) ? 1u : 0u;
    uint32_t tot; (void)block_incl<OpAdd>(c, s_tmp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
// block carries are 32-bit within a launch's range; `offs` (64-bit) holds the carries' own carries for inputs beyond 2^32 kept items -- not
// needed: every caller's n is below 2^32
template <typename Flag, typename Emit>
__global__ __launch_bounds__(PR_NT) void compact_emit_kernel(Flag flag, Emit emit, uint64_t n, const uint32_t *carries, unsigned long long *total)
{
    __shared__ uint32_t s_tmp[8];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    uint32_t keep = 0, c = 0;
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (base + j < n && flag(base + j)) { keep |= 1u << j; c++; }
    uint32_t tot;
    const uint32_t inc = block_incl<OpAdd>(c, s_tmp, &tot);
    uint64_t pos = (uint64_t)carries[blockIdx.x] + (inc - c);
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (keep & (1u << j)) emit(base + j, pos++);
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == PR_NT - 1) *total = (unsigned long long)carries[blockIdx.x] + tot;
}
template <typename Flag, typename Emit>
int compact(Flag flag, Emit emit, uint64_t n, uint64_t *count, hipStream_t st)
{
    *count = 0;
    if (!n) return SKX_OK;
    if (n > 0xFFFFFFF0ull) { set_error("more than 2^32 items in one compaction"); return SKX_EUNSUP; }
    const uint64_t nb = (n + PR_TILE - 1) / PR_TILE;
    DevBuf<uint32_t> sums; DevBuf<unsigned long long> d_total;
    SKX_TRY(sums.alloc(nb)); SKX_TRY(d_total.alloc(1));
    hipLaunchKernelGGL(compact_count_kernel<Flag>, dim3((unsigned)nb), dim3(PR_NT), 0, st, flag, n, sums.p);
    hipLaunchKernelGGL(scan_carries_kernel<OpAdd>, dim3(1), dim3(PR_NT), 0, st, sums.p, nb);
    hipLaunchKernelGGL((compact_emit_kernel<Flag, Emit>), dim3((unsigned)nb), dim3(PR_NT), 0, st, flag, emit, n, sums.p, d_total.p);
    unsigned long long t = 0;
    PRH(hipMemcpyAsync(&t, d_total.p, 8, hipMemcpyDeviceToHost, st));
    PRH(hipStreamSynchronize(st));
    *count = t;
    return SKX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- radix sort
template <typename K> __device__ inline uint32_t digit_of(K k, int shift) { return (uint32_t)(k >> shift) & 255u; }

// per wave tile (1 024 consecutive keys) the number of keys of every digit: counts[digit * ntiles + tile]
template <typename K>
__global__ __launch_bounds__(PR_NT) void rs_hist_kernel(const K *keys, uint64_t n, int shift, uint64_t ntiles, uint32_t *counts)
{
    __shared__ uint32_t s_h[PR_NT / 64][256];
    const int wv = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * (PR_NT / 64) + wv;
    for (int i = lane_id(); i < 256; i += 64) s_h[wv][i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < ntiles) {
        const uint64_t base = tile * RS_WTILE;
#pragma unroll 4
        for (int r = 0; r < RS_WAVE_ITEMS; r++) {
            const uint64_t i = base + (uint64_t)r * 64 + lane_id();
            if (i < n) atomicAdd(&s_h[wv][digit_of(keys[i], shift)], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        for (int d = lane_id(); d < 256; d += 64) counts[(uint64_t)d * ntiles + tile] = s_h[wv][d];
    }
}
// stable scatter: a wave walks its tile 64 keys at a time; the lanes holding the same digit find one another with eight ballots, the lowest of
// them takes the digit's running place (kept per wave in LDS: no other wave touches it, no barrier), every lane adds its rank among them
template <typename K, bool HASV>
__global__ __launch_bounds__(PR_NT) void rs_scatter_kernel(const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int shift, uint64_t ntiles,
                                                          const uint32_t *offs)
{
    __shared__ uint32_t s_run[PR_NT / 64][256];
    const int wv = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * (PR_NT / 64) + wv;
    if (tile >= ntiles) return;
    for (int d = lane_id(); d < 256; d += 64) s_run[wv][d] = offs[(uint64_t)d * ntiles + tile];
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = tile * RS_WTILE;
    for (int r = 0; r < RS_WAVE_ITEMS; r++) {
        const uint64_t i = base + (uint64_t)r * 64 + lane_id();
        const bool valid = i < n;
        K key = 0; uint32_t val = 0;
        if (valid) { key = kin[i]; if (HASV) val = vin[i]; }
        const uint32_t dg = digit_of(key, shift);
        unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
            same &= bit ? bal : ~bal;
        }
        // (an invalid lane's mask is empty of valid lanes only if it is invalid itself: it does nothing below)
        // (an invalid lane matches nobody that is valid and does nothing)
        const uint32_t rank = mbcnt64(same), cnt = (uint32_t)__popcll(same);
        const int leader = valid ? __ffsll((long long)same) - 1 : (int)lane_id();
        uint32_t start = 0;
        if (valid && (int)lane_id() == leader) { start = s_run[wv][dg]; s_run[wv][dg] = start + cnt; }
        const uint32_t place = __shfl(start, leader, 64) + rank;      // the leader's start to its group
        if (valid) { kout[place] = key; if (HASV) vout[place] = val; }
        __builtin_amdgcn_wave_barrier();
    }
}
template <typename K, bool HASV>
int radix_sort(const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int bits, hipStream_t st)
{
    if (!n) return SKX_OK;
    if (n > 0xFFFFFFF0ull) { set_error("more than 2^32 items in one sort"); return SKX_EUNSUP; }
    const int npass = (std::max(bits, 1) + 7) / 8;
    const uint64_t ntiles = (n + RS_WTILE - 1) / RS_WTILE, nblocks = (ntiles + PR_NT / 64 - 1) / (PR_NT / 64);
    DevBuf<uint32_t> counts, vtmp; DevBuf<K> ktmp;
    SKX_TRY(counts.alloc(256 * ntiles));
    if (npass > 1) { SKX_TRY(ktmp.alloc(n)); if (HASV) SKX_TRY(vtmp.alloc(n)); }
    // ping-pong so that the last pass lands in kout: with an even number of passes the first goes to kout... the input is never written
    const K *src = kin; const uint32_t *vsrc = vin;
    for (int p = 0; p < npass; p++) {
        const bool to_out = ((npass - 1 - p) % 2) == 0;
        K *dst = to_out ? kout : ktmp.p; uint32_t *vdst = to_out ? vout : vtmp.p;
        hipLaunchKernelGGL(rs_hist_kernel<K>, dim3((unsigned)nblocks), dim3(PR_NT), 0, st, src, n, 8 * p, ntiles, counts.p);
        SKX_TRY((scan_u32<OpAdd, true>(counts.p, counts.p, 256 * ntiles, st)));
        hipLaunchKernelGGL((rs_scatter_kernel<K, HASV>), dim3((unsigned)nblocks), dim3(PR_NT), 0, st, src, dst, vsrc, vdst, n, 8 * p, ntiles, counts.p);
        src = dst; vsrc = vdst;
    }
    PRH(hipStreamSynchronize(st));
    PRH(hipGetLastError());
    return SKX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- segmented OR-scan
// out[i] = OR of vals[j] over the items j < i of i's segment (a segment = a run of equal keys); 0 at a segment's first item.
// As an inclusive scan of the sequence shifted by one -- (head_i, head_i ? 0 : vals[i - 1]) -- under (f1, v1) + (f2, v2) = (f1 | f2, f2 ? v2 : v1 | v2).
struct Seg { uint32_t f; uint64_t v; };
__device__ inline Seg seg_op(Seg a, Seg b) { Seg r; r.f = a.f | b.f; r.v = b.f ? b.v : (a.v | b.v); return r; }
__device__ inline Seg seg_item(const uint32_t *keys, const uint64_t *vals, uint64_t i, uint64_t n)
{
    Seg s; s.f = 0; s.v = 0;
    if (i < n) { s.f = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; s.v = s.f ? 0ull : vals[i - 1]; }
    return s;
}
__device__ inline Seg seg_wave_incl(Seg s)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Seg t; t.f = __shfl_up(s.f, d, 64); t.v = __shfl_up(s.v, d, 64);
        if ((int)lane_id() >= d) s = seg_op(t, s);
    }
    return s;
}
__global__ __launch_bounds__(PR_NT) void seg_sums_kernel(const uint32_t *keys, const uint64_t *vals, uint64_t n, Seg *sums)
{
    __shared__ Seg s_w[PR_NT / 64];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    Seg acc; acc.f = 0; acc.v = 0;
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) acc = seg_op(acc, seg_item(keys, vals, base + j, n));
    const Seg inc = seg_wave_incl(acc);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { Seg t = s_w[0]; for (int w = 1; w < PR_NT / 64; w++) t = seg_op(t, s_w[w]); sums[blockIdx.x] = t; }
}
__global__ void seg_carries_kernel(Seg *sums, uint64_t nb)       // one thread: the block aggregates -> their exclusive scan (nb = n / 4 096)
{
    if (threadIdx.x || blockIdx.x) return;
    Seg run; run.f = 0; run.v = 0;
    for (uint64_t b = 0; b < nb; b++) { const Seg t = sums[b]; sums[b] = run; run = seg_op(run, t); }
}
__global__ __launch_bounds__(PR_NT) void seg_apply_kernel(const uint32_t *keys, const uint64_t *vals, uint64_t n, const Seg *carries, uint64_t *out)
{
    __shared__ Seg s_w[PR_NT / 64];
    const uint64_t base = (uint64_t)blockIdx.x * PR_TILE + (uint64_t)threadIdx.x * PR_ITEMS;
    Seg it[PR_ITEMS], acc; acc.f = 0; acc.v = 0;
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) { it[j] = seg_item(keys, vals, base + j, n); acc = seg_op(acc, it[j]); }
    const Seg inc = seg_wave_incl(acc);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    Seg run = carries[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run = seg_op(run, s_w[w]);
    Seg prev; prev.f = __shfl_up(inc.f, 1, 64); prev.v = __shfl_up(inc.v, 1, 64);
    if (lane_id()) run = seg_op(run, prev);
#pragma unroll
    for (int j = 0; j < PR_ITEMS; j++) if (base + j < n) { run = seg_op(run, it[j]); out[base + j] = run.v; }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- entry points (skx_internal.h)
int prim_sort_keys_u64(const uint64_t *in, uint64_t *out, uint64_t n, int bits, hipStream_t st) { return radix_sort<uint64_t, false>(in, out, nullptr, nullptr, n, bits, st); }
int prim_sort_pairs_u64(const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int bits, hipStream_t st) { return radix_sort<uint64_t, true>(kin, kout, vin, vout, n, bits, st); }
int prim_sort_keys_u128(const u128 *in, u128 *out, uint64_t n, int bits, hipStream_t st) { return radix_sort<u128, false>(in, out, nullptr, nullptr, n, bits, st); }
int prim_sort_pairs_u128(const u128 *kin, u128 *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int bits, hipStream_t st) { return radix_sort<u128, true>(kin, kout, vin, vout, n, bits, st); }
int prim_scan_add_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st) { return scan_u32<OpAdd, false>(in, out, n, st); }
int prim_scan_max_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t st) { return scan_u32<OpMax, false>(in, out, n, st); }
int prim_select_index_u8(const uint8_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st) { return compact(FlagU8{flags}, EmitIndex{out}, n, count, st); }
int prim_select_index_u32(const uint32_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st) { return compact(FlagU32{flags}, EmitIndex{out}, n, count, st); }
int prim_select_u32(const uint32_t *in, const uint8_t *flags, uint32_t *out, uint64_t n, uint64_t *count, hipStream_t st) { return compact(FlagU8{flags}, EmitU32{in, out}, n, count, st); }
int prim_unique_u64(const uint64_t *sorted, uint64_t *out, uint64_t n, uint64_t *count, hipStream_t st) { return compact(FlagNewU64{sorted}, EmitU64{sorted, out}, n, count, st); }
int prim_unique_keys_u128(const u128 *sorted, u128 *out, uint64_t n, uint64_t *count, hipStream_t st) { return compact(FlagNewKeyU128{sorted}, EmitU128{sorted, out}, n, count, st); }
int prim_seg_exscan_or_u64(const uint32_t *keys, const uint64_t *vals, uint64_t *out, uint64_t n, hipStream_t st)
{
    if (!n) return SKX_OK;
    const uint64_t nb = (n + PR_TILE - 1) / PR_TILE;
    DevBuf<Seg> sums; SKX_TRY(sums.alloc(nb));
    hipLaunchKernelGGL(seg_sums_kernel, dim3((unsigned)nb), dim3(PR_NT), 0, st, keys, vals, n, sums.p);
    hipLaunchKernelGGL(seg_carries_kernel, dim3(1), dim3(64), 0, st, sums.p, nb);
    hipLaunchKernelGGL(seg_apply_kernel, dim3((unsigned)nb), dim3(PR_NT), 0, st, keys, vals, n, sums.p, out);
    PRH(hipStreamSynchronize(st));
    return SKX_OK;
}

// Self-check of the primitives against the host's own sort / scan / filter on seeded data (tests/test_gpu_prims.py calls it through
// the library: the primitives have no entry of their own in the C ABI).  0 = all equal; otherwise the number of the first check that failed.
extern "C" int skx_debug_prims_selftest(int device, uint64_t n, uint64_t seed)
{
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t st = nullptr;
    if (hipStreamCreate(&st) != hipSuccess) return -1;
    auto rnd = [&seed]() { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    int fail = 0;
    auto up = [&](auto &dev, const auto &host) { return dev.alloc(host.size() ? host.size() : 1) == SKX_OK && hipMemcpy(dev.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice) == hipSuccess; };
    auto down = [&](auto &host, const auto &dev, size_t cnt) { host.resize(cnt); return hipMemcpy(host.data(), dev.p, cnt * sizeof(host[0]), hipMemcpyDeviceToHost) == hipSuccess; };
    {   // 1: pairs of 64-bit keys (few distinct values: stability shows) and their stream positions
        std::vector<uint64_t> k(n), ks; std::vector<uint32_t> v(n), vs;
        for (uint64_t i = 0; i < n; i++) { k[i] = (rnd() % (n / 3 + 1)) * 0x0101010101010101ull ^ (rnd() & 0xFF00000000000000ull); v[i] = (uint32_t)i; }
        DevBuf<uint64_t> dk, dko; DevBuf<uint32_t> dv, dvo;
        if (!up(dk, k) || !up(dv, v) || dko.alloc(n ? n : 1) != SKX_OK || dvo.alloc(n ? n : 1) != SKX_OK) return -1;
        if (prim_sort_pairs_u64(dk.p, dko.p, dv.p, dvo.p, n, 64, st) != SKX_OK || !down(ks, dko, n) || !down(vs, dvo, n)) return -1;
        std::vector<uint32_t> idx(n); for (uint64_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
        for (uint64_t i = 0; i < n && !fail; i++) if (vs[i] != idx[i] || ks[i] != k[idx[i]]) fail = 1;
        // 2: keys only, 40 significant bits; 3: unique
        std::vector<uint64_t> k2(n), s2, u2;
        for (uint64_t i = 0; i < n; i++) k2[i] = rnd() % (n / 2 + 1) * 7919ull & ((1ull << 40) - 1);
        DevBuf<uint64_t> d2, d2o, d2u;
        if (!up(d2, k2) || d2o.alloc(n ? n : 1) != SKX_OK || d2u.alloc(n ? n : 1) != SKX_OK) return -1;
        uint64_t cnt = 0;
        if (prim_sort_keys_u64(d2.p, d2o.p, n, 40, st) != SKX_OK || !down(s2, d2o, n) || prim_unique_u64(d2o.p, d2u.p, n, &cnt, st) != SKX_OK || !down(u2, d2u, cnt)) return -1;
        std::sort(k2.begin(), k2.end());
        if (!fail && s2 != k2) fail = 2;
        k2.erase(std::unique(k2.begin(), k2.end()), k2.end());
        if (!fail && u2 != k2) fail = 3;
    }
    {   // 4: 128-bit pairs, 5: unique on the key bits
        std::vector<u128> k(n), ks, ku; std::vector<uint32_t> v(n), vs;
        for (uint64_t i = 0; i < n; i++) { k[i] = (((u128)(rnd() % 5) << 64) | (rnd() % (n / 2 + 1))) << 4 | (rnd() & 15u); v[i] = (uint32_t)i; }
        DevBuf<u128> dk, dko, dku; DevBuf<uint32_t> dv, dvo;
        if (!up(dk, k) || !up(dv, v) || dko.alloc(n ? n : 1) != SKX_OK || dvo.alloc(n ? n : 1) != SKX_OK || dku.alloc(n ? n : 1) != SKX_OK) return -1;
        uint64_t cnt = 0;
        if (prim_sort_pairs_u128(dk.p, dko.p, dv.p, dvo.p, n, 128, st) != SKX_OK || !down(ks, dko, n) || !down(vs, dvo, n)) return -1;
        std::vector<uint32_t> idx(n); for (uint64_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
        for (uint64_t i = 0; i < n && !fail; i++) if (vs[i] != idx[i] || ks[i] != k[idx[i]]) fail = 4;
        if (prim_unique_keys_u128(dko.p, dku.p, n, &cnt, st) != SKX_OK || !down(ku, dku, cnt)) return -1;
        std::vector<u128> want;
        for (uint64_t i = 0; i < n; i++) if (i == 0 || (ks[i] >> 4) != (ks[i - 1] >> 4)) want.push_back(ks[i]);
        if (!fail && ku != want) fail = 5;
    }
    {   // 6-7: scans; 8-9: selections; 10: segmented OR-scan
        std::vector<uint32_t> a(n), sa, sm, si, sv; std::vector<uint8_t> f(n); std::vector<uint32_t> keys(n); std::vector<uint64_t> vals(n), so;
        for (uint64_t i = 0; i < n; i++) { a[i] = (uint32_t)(rnd() % 1000); f[i] = (rnd() % 3) == 0; vals[i] = rnd(); }
        uint32_t kk = 0; for (uint64_t i = 0; i < n; i++) { if (rnd() % 4 == 0) kk++; keys[i] = kk; }
        DevBuf<uint32_t> da, dsa, dsm, dsi, dsv, dkeys; DevBuf<uint8_t> df; DevBuf<uint64_t> dvals, dso;
        if (!up(da, a) || !up(df, f) || !up(dkeys, keys) || !up(dvals, vals)) return -1;
        const uint64_t n1 = n ? n : 1;
        if (dsa.alloc(n1) != SKX_OK || dsm.alloc(n1) != SKX_OK || dsi.alloc(n1) != SKX_OK || dsv.alloc(n1) != SKX_OK || dso.alloc(n1) != SKX_OK) return -1;
        uint64_t c1 = 0, c2 = 0;
        if (prim_scan_add_u32(da.p, dsa.p, n, st) != SKX_OK || prim_scan_max_u32(da.p, dsm.p, n, st) != SKX_OK || prim_select_index_u8(df.p, dsi.p, n, &c1, st) != SKX_OK ||
            prim_select_u32(da.p, df.p, dsv.p, n, &c2, st) != SKX_OK || prim_seg_exscan_or_u64(dkeys.p, dvals.p, dso.p, n, st) != SKX_OK) return -1;
        if (!down(sa, dsa, n) || !down(sm, dsm, n) || !down(si, dsi, c1) || !down(sv, dsv, c2) || !down(so, dso, n)) return -1;
        uint32_t run = 0, mx = 0; std::vector<uint32_t> wi, wv2;
        for (uint64_t i = 0; i < n; i++) {
            run += a[i]; mx = std::max(mx, a[i]);
            if (!fail && sa[i] != run) fail = 6;
            if (!fail && sm[i] != mx) fail = 7;
            if (f[i]) { wi.push_back((uint32_t)i); wv2.push_back(a[i]); }
        }
        if (!fail && si != wi) fail = 8;
        if (!fail && sv != wv2) fail = 9;
        uint64_t acc = 0;
        for (uint64_t i = 0; i < n; i++) { if (i == 0 || keys[i] != keys[i - 1]) acc = 0; if (!fail && so[i] != acc) fail = 10; acc |= vals[i]; }
    }
    (void)hipStreamDestroy(st);
    return fail;
}

}  // namespace skx
