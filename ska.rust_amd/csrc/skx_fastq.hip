// skx_fastq.hip -- FASTQ text -> the read-set kernels' bit planes, on the device (the needletail record iterator of
// ska_dict.rs:131-153,356-366 for read sets; the host form of the same is fastx.cpp stream_fastq_file + pack_*_planes).
// The host only read()s file bytes into the pinned ring; what crosses PCIe is the file as it is, and the four-line framing, the checks
// a record must pass and the 5-bit planes (two code bits, the bytes valid_base rejects, line ends, quality verdicts: skx_device.h
// planes_bytes16) are made here:
//   fq_count_kernel   : line ends per 16 KB tile
//   fq_offsets_kernel : one workgroup -- exclusive scan of the tile counts, total number of lines
//   fq_emit_kernel    : the byte offset of every line end, in order (line_end[i])
//   fq_records_kernel : a thread per record (lines 4 r .. 4 r + 3): '@' / '+' at the starts of lines 0 and 2, |sequence| == |quality| with a
//                       trailing '\r' dropped from either -- the rules of fastx.cpp take_line --, where its sequence and quality lines
//                       start and the positions it gives (bases + 1); anything else (a blank line, a truncated record, a line count
//                       that is not a multiple of four, file 1 ending inside a record) raises `irregular`: the caller then takes the
//                       sample through the host reader, which accepts what is merely unusual and words the error for what is wrong
//   (scan of the records' positions: skx_prims)
//   fq_pack_kernel    : a workgroup per 4 096 output positions -- the records that overlap them into LDS, a position finds its record by
//                       binary search there, loads its base and quality byte, a wave's five ballots are a group's five plane words;
//                       the tile's 64 groups leave as one coalesced store
// HBM traffic per 50x isolate (0.55 GB of text): 3 reads of the text's bytes + 27 MB of line ends written and read + 157 MB of planes.
#include "skx_internal.h"
#include "skx_device.h"

namespace skx {

constexpr int FQ_TILE = 16384, FQ_NT = 256;                          // 64 bytes per thread
constexpr uint32_t FQ_IRREGULAR = 1u;

// the line ends among my 64 bytes as a 64-bit mask
__device__ static inline uint64_t fq_newlines64(const uint8_t *raw, uint64_t len, uint64_t p0)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint64_t m = 0;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        u32x4 x = {0, 0, 0, 0};
        if (p0 + 16u * v < len) x = *reinterpret_cast<const u32x4 *>(raw + p0 + 16u * v);     // (buffers are padded to 16 bytes)
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t y = w[i] ^ 0x0A0A0A0Au;
            const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);       // bit 7 of every byte that is 0 (exact, no borrow between bytes)
            const uint32_t nib = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
            m |= (uint64_t)nib << (16 * v + 4 * i);
        }
    }
    const uint64_t left = p0 >= len ? 0 : len - p0;
    if (left < 64) m &= left ? ((1ull << left) - 1ull) : 0ull;
    return m;
}

__global__ __launch_bounds__(FQ_NT) void fq_count_kernel(const uint8_t *raw, uint64_t len, uint32_t *tile_cnt)
{
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * FQ_TILE + (uint64_t)threadIdx.x * 64;
    uint32_t c = (uint32_t)__popcll(fq_newlines64(raw, len, p0));
#pragma unroll
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_sum, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_sum;
}

// tile_cnt -> exclusive offsets in place; info[0] = number of lines
__global__ __launch_bounds__(1024) void fq_offsets_kernel(uint32_t *tile_cnt, uint64_t ntiles, uint32_t *info)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t b = 0; b < ntiles; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const uint32_t v = i < ntiles ? tile_cnt[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t before = s_carry, all = 0;
        for (int w = 0; w < 16; w++) { const uint32_t c = s_w[w]; if (w < wv) before += c; all += c; }
        if (i < ntiles) tile_cnt[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) info[0] = s_carry;
}

__global__ __launch_bounds__(FQ_NT) void fq_emit_kernel(const uint8_t *raw, uint64_t len, const uint32_t *tile_off, uint32_t *line_end, uint64_t cap)
{
    __shared__ uint32_t s_w[FQ_NT / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * FQ_TILE + (uint64_t)threadIdx.x * 64;
    uint64_t m = fq_newlines64(raw, len, p0);
    const uint32_t n = (uint32_t)__popcll(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t before = tile_off[blockIdx.x];
    for (int w = 0; w < wv; w++) before += s_w[w];
    uint64_t at = (uint64_t)before + inc - n;
    while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        if (at < cap) line_end[at] = (uint32_t)(p0 + (uint64_t)b);
        at++;
    }
}

// records: rec_seq / rec_qual = where the two lines start, rec_len = positions (bases + the line end)
__global__ __launch_bounds__(256) void fq_records_kernel(const uint8_t *raw, const uint32_t *line_end, uint64_t n_lines, uint64_t n_rec, uint64_t junction,
                                                         uint32_t *rec_seq, uint32_t *rec_qual, uint32_t *rec_len, uint32_t *info)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r == 0 && (n_lines & 3u)) atomicOr(&info[1], FQ_IRREGULAR);                        // a record cut short (or blank lines)
    if (r >= n_rec) return;
    const uint64_t s0 = r ? (uint64_t)line_end[4 * r - 1] + 1 : 0;
    const uint64_t e0 = line_end[4 * r], e1 = line_end[4 * r + 1], e2 = line_end[4 * r + 2], e3 = line_end[4 * r + 3];
    const uint64_t s1 = e0 + 1, s2 = e1 + 1, s3 = e2 + 1;
    uint64_t slen = e1 - s1, qlen = e3 - s3;
    if (slen && raw[e1 - 1] == '\r') slen--;
    if (qlen && raw[e3 - 1] == '\r') qlen--;
    // (a header of '\r' alone is a blank line to the host reader, and so is an empty one: both are irregular here)
    bool ok = e0 > s0 && raw[s0] == '@' && e2 > s2 && raw[s2] == '+' && slen == qlen;
    // file 2 begins where a record begins: no record straddles the junction
    if (junction && s0 < junction && e3 >= junction) ok = false;
    if (!ok) atomicOr(&info[1], FQ_IRREGULAR);
    rec_seq[r] = (uint32_t)s1; rec_qual[r] = (uint32_t)s3; rec_len[r] = (uint32_t)slen + 1u;
}

// rec_end: inclusive scan of rec_len (rec_end[r] = first position behind record r)
constexpr int FQP_TILE = 4096, FQP_NT = 256, FQP_REC = FQP_TILE + 2;
__global__ __launch_bounds__(FQP_NT) void fq_pack_kernel(const uint8_t *raw, const uint32_t *rec_seq, const uint32_t *rec_qual, const uint32_t *rec_end, uint64_t n_rec,
                                                        uint32_t min_qual, uint64_t *planes)
{
    __shared__ uint32_t s_end[FQP_REC], s_seq[FQP_REC], s_qual[FQP_REC];
    __shared__ uint64_t s_out[(FQP_TILE / 64) * 5];
    __shared__ uint32_t s_first, s_count;
    const uint64_t total = n_rec ? rec_end[n_rec - 1] : 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * FQP_TILE;
    if (t0 >= total) return;
    const uint64_t t1 = t0 + FQP_TILE < total ? t0 + FQP_TILE : total;
    if (threadIdx.x < 2) {
        // the first record that ends behind `want` positions: [0] the record of the tile's first position, [1] of its last
        const uint64_t want = threadIdx.x == 0 ? t0 : t1 - 1;
        uint64_t lo = 0, hi = n_rec;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)rec_end[mid] <= want) lo = mid + 1; else hi = mid; }
        if (threadIdx.x == 0) s_first = (uint32_t)lo; else s_count = (uint32_t)lo;
    }
    __syncthreads();
    const uint32_t first = s_first, cnt = s_count - s_first + 1;      // (<= FQP_TILE + 1: every record holds a position)
    for (uint32_t i = threadIdx.x; i < cnt; i += FQP_NT) { s_end[i] = rec_end[first + i]; s_seq[i] = rec_seq[first + i]; s_qual[i] = rec_qual[first + i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t before_first = first ? (uint64_t)rec_end[first - 1] : 0;
    for (int g = wv; g < FQP_TILE / 64; g += FQP_NT / 64) {
        const uint64_t p = t0 + (uint64_t)g * 64 + lane;
        uint32_t code = 0, bad = 0, nl = 0, qv = 0;
        if (p < t1) {
            uint32_t lo = 0, hi = cnt - 1;                             // the record whose end lies behind p
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)s_end[mid] <= p) lo = mid + 1; else hi = mid; }
            const uint64_t start = lo ? (uint64_t)s_end[lo - 1] : before_first;
            const uint32_t off = (uint32_t)(p - start);
            if (p + 1 == (uint64_t)s_end[lo]) nl = 1;
            else {
                const uint32_t c = raw[(uint64_t)s_seq[lo] + off], q = raw[(uint64_t)s_qual[lo] + off];
                bad = (c & 15u) == 14u;                                 // valid_base (bit_encoding.rs:52-54)
                code = bad ? 0u : (c >> 1) & 3u;                        // encode_base (bit_encoding.rs:42-51)
                qv = ((q - 33u) & 0xFFu) <= min_qual;                   // !(q - 33 > min_qual) in u8 arithmetic (split_kmer.rs:98-101)
            }
        }
        const uint64_t b0 = __ballot(code & 1u), b1 = __ballot(code & 2u), b2 = __ballot(bad), b3 = __ballot(nl), b4 = __ballot(qv);
        if (lane < 5) s_out[g * 5 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : lane == 3 ? b3 : b4;
    }
    __syncthreads();
    const uint32_t groups = (uint32_t)((t1 - t0 + 63) / 64);
    uint64_t *dst = planes + (t0 / 64) * 5;
    for (uint32_t i = threadIdx.x; i < groups * 5u; i += FQP_NT) dst[i] = s_out[i];
}

// raw: the sample's file(s) as read, file 2 behind file 1 (a '\n' between them and at the end where the files lack it: the caller's), 16 readable
// bytes behind len; junction: where file 2 starts (0: one file).  planes: room for len / 2 / 64 + 2 groups (a record's positions are at most half
// its bytes).  *irregular: the text is not plain four-line FASTQ all the way -- nothing was written that the caller may use.
int fastq_frame_planes(skx_ctx *ctx, const uint8_t *raw, uint64_t len, uint64_t junction, int min_qual, uint64_t *planes, FastqScratch &sc, uint64_t *positions, int *irregular)
{
    hipStream_t st = ctx->stream;
    *positions = 0; *irregular = 0;
    if (len == 0) return SKX_OK;
    if (len > 0xFFFFFFF0ull) { *irregular = 1; return SKX_OK; }
    const uint64_t ntiles = (len + FQ_TILE - 1) / FQ_TILE;
    if (sc.tile.n < ntiles) SKX_TRY(sc.tile.alloc(ntiles + ntiles / 4 + 64));
    if (!sc.info.p) SKX_TRY(sc.info.alloc(4));
    SKX_HIP(hipMemsetAsync(sc.info.p, 0, 16, st));
    hipLaunchKernelGGL(fq_count_kernel, dim3((unsigned)ntiles), dim3(FQ_NT), 0, st, raw, len, sc.tile.p);
    hipLaunchKernelGGL(fq_offsets_kernel, dim3(1), dim3(1024), 0, st, sc.tile.p, ntiles, sc.info.p);
    uint32_t h_info[2] = {0, 0};
    SKX_HIP(hipMemcpyAsync(h_info, sc.info.p, 8, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    const uint64_t n_lines = h_info[0], n_rec = n_lines / 4;
    if ((n_lines & 3u) || n_rec == 0) { *irregular = 1; return SKX_OK; }
    if (sc.line_end.n < n_lines) SKX_TRY(sc.line_end.alloc(n_lines + n_lines / 4 + 1024));
    if (sc.rec_seq.n < n_rec) { const uint64_t c = n_rec + n_rec / 4 + 1024; SKX_TRY(sc.rec_seq.alloc(c)); SKX_TRY(sc.rec_qual.alloc(c)); SKX_TRY(sc.rec_len.alloc(c)); SKX_TRY(sc.rec_end.alloc(c)); }
    hipLaunchKernelGGL(fq_emit_kernel, dim3((unsigned)ntiles), dim3(FQ_NT), 0, st, raw, len, (const uint32_t *)sc.tile.p, sc.line_end.p, n_lines);
    hipLaunchKernelGGL(fq_records_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, raw, (const uint32_t *)sc.line_end.p, n_lines, n_rec, junction,
                       sc.rec_seq.p, sc.rec_qual.p, sc.rec_len.p, sc.info.p);
    SKX_TRY(prim_scan_add_u32(sc.rec_len.p, sc.rec_end.p, n_rec, st));                     // (returns with the stream idle)
    uint32_t h_tot = 0;
    SKX_HIP(hipMemcpyAsync(&h_info[1], sc.info.p + 1, 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipMemcpyAsync(&h_tot, sc.rec_end.p + (n_rec - 1), 4, hipMemcpyDeviceToHost, st));
    SKX_HIP(hipStreamSynchronize(st));
    if (h_info[1]) { *irregular = 1; return SKX_OK; }
    const uint64_t total = h_tot;
    // (n_rec <= len / 8 and every record's positions are at most half its bytes: the sum cannot have wrapped 32 bits while len < 4 GB)
    hipLaunchKernelGGL(fq_pack_kernel, dim3((unsigned)((total + FQP_TILE - 1) / FQP_TILE)), dim3(FQP_NT), 0, st, raw, (const uint32_t *)sc.rec_seq.p,
                       (const uint32_t *)sc.rec_qual.p, (const uint32_t *)sc.rec_end.p, n_rec, (uint32_t)(min_qual & 0xFF), planes);
    SKX_HIP(hipGetLastError());
    *positions = total;
    return SKX_OK;
}

}  // namespace skx
