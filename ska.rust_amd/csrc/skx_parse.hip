// skx_parse.hip -- FASTA text -> record stream on the device (the needletail record iterator of ska_dict.rs:131-153 for
// assemblies): the host only reads file bytes into pinned memory and uploads them; headers, line breaks and carriage returns
// are stripped here.  Same record rules as the host reader (fastx.cpp parse_fasta): a line that starts with '>' is a header
// and starts a record, every other line is sequence, '\r' is dropped, each record's bases are followed by one '\n'.
//
// Three launches over tiles of 16 KB per workgroup:
//   fasta_tile_summary : per tile -- does a line start inside it, the kind of line its last byte belongs to, and how many
//                        bytes it emits before / from its first line start (the count before depends on the kind of line the
//                        tile starts in, which only the scan knows)
//   fasta_tile_scan    : per file -- kind of line and output offset at the start of every tile (one wave walks the file's
//                        summaries in LDS), total output length (+ the final terminator)
//   fasta_tile_emit    : per tile -- classification again with the known start state, kept bytes staged in LDS in order and
//                        written as aligned dwords
// HBM traffic: 2 reads + 1 write of the text (15 GB per 1 000 x 5 Mbp assemblies, a few ms).
#include "skx_device.h"

namespace skx {

constexpr int PT_TILE = 16384, PT_NT = 256, PT_PER = PT_TILE / PT_NT;      // 64 bytes per thread

struct ParseArgs {
    const uint8_t *const *raw;        // [n] file bytes (16-B aligned buffers)
    const uint64_t *rawlen;           // [n]
    uint8_t *const *out;              // [n] record streams (>= rawlen + 16 bytes each)
    uint64_t *outlen;                 // [n]
    const uint32_t *tile_file;        // [tiles] file of a tile
    const uint64_t *tile_base;        // [n + 1] first tile of a file
    uint2 *summary;                   // [tiles] x: emitted before the first line start (if the tile starts in a sequence line) | has_ls << 30 | kind_out << 31; y: emitted from there on
    uint64_t *tile_off;               // [tiles] output offset of a tile
    uint8_t *tile_kind;               // [tiles] kind of line at the start of a tile (1 = header)
    int n;
};

// wave-wide then block-wide inclusive "last defined value" scan: (has, val) o (has', val') = has' ? (1, val') : (has, val)
__device__ static inline void carry_scan(bool &has, uint32_t &val, uint32_t *s_tmp)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = (has ? 2u : 0u) | (val & 1u);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d && !(x & 2u)) x = y; }
    if (lane == 63) s_tmp[wv] = x;
    __syncthreads();
    uint32_t pre = 0;                                                    // carry from the waves before mine
    for (int w = 0; w < wv; w++) { const uint32_t y = s_tmp[w]; if (y & 2u) pre = y; }
    __syncthreads();
    // exclusive form: state BEFORE my bytes = inclusive state of the previous lane (or the carry of the previous waves)
    uint32_t prev = __shfl_up(x, 1, 64);
    if (lane == 0) prev = 0;
    if (!(prev & 2u)) prev = pre;
    has = prev & 2u; val = prev & 1u;
}

// classification of my 64 bytes; `kind` = kind of line my first byte continues (ignored if that byte starts a line);
// returns the number of bytes emitted; EMIT: writes them to dst
template <bool EMIT>
__device__ static inline uint32_t walk64(const uint32_t w[16], uint32_t nvalid, bool first_is_ls, bool file_start, uint32_t &kind, bool &saw_ls,
                                         uint32_t &emitted_before_ls, unsigned char *dst)
{
    uint32_t n = 0;
    bool ls = first_is_ls;
    saw_ls = false; emitted_before_ls = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t p = 4 * i + b;
            const uint32_t c = (w[i] >> (8 * b)) & 0xFFu;
            const bool in = p < nvalid;
            if (in && ls) {
                if (!saw_ls) { saw_ls = true; emitted_before_ls = n; }
                kind = c == '>';
                if (kind && !(file_start && p == 0)) { if (EMIT) dst[n] = '\n'; n++; }          // terminator of the record before this header
            }
            if (in && !kind && c != '\n' && c != '\r') { if (EMIT) dst[n] = (unsigned char)c; n++; }
            ls = c == '\n';
        }
    }
    if (!saw_ls) emitted_before_ls = n;
    return n;
}

__device__ static inline void load64(const uint8_t *raw, uint64_t len, uint64_t p0, uint32_t w[16], uint32_t &nvalid, bool &first_is_ls)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    nvalid = p0 >= len ? 0u : (uint32_t)(len - p0 < 64 ? len - p0 : 64);
#pragma unroll
    for (int v = 0; v < 4; v++) {
        u32x4 x = {0, 0, 0, 0};
        if (p0 + 16u * v < len) x = *reinterpret_cast<const u32x4 *>(raw + p0 + 16u * v);     // buffers are padded to 16 bytes
        w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
    }
    first_is_ls = p0 == 0 || (p0 < len && raw[p0 - 1] == '\n');
}

__global__ __launch_bounds__(PT_NT) void fasta_tile_summary(ParseArgs a)
{
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_red[3];                       // [0] emitted whatever the start kind, [1] last line start (thread, kind), [2] emitted only if the tile starts in a sequence line
    const uint64_t t = blockIdx.x;
    const uint32_t f = a.tile_file[t];
    const uint64_t len = a.rawlen[f], tl = t - a.tile_base[f];
    const uint8_t *raw = a.raw[f];
    const uint64_t p0 = tl * PT_TILE + (uint64_t)threadIdx.x * PT_PER;
    uint32_t w[16], nvalid; bool fls;
    load64(raw, len, p0, w, nvalid, fls);
    if (threadIdx.x < 3) s_red[threadIdx.x] = 0u;
    // my bytes as if they continued a sequence line: what they emit, how much of it before my first line start, the kind that leaves them
    uint32_t kind = 0, before; bool saw;
    const uint32_t n_seq = walk64<false>(w, nvalid, fls, p0 == 0, kind, saw, before, nullptr);
    const bool my_saw = saw; const uint32_t my_kind = kind;
    bool has = saw; uint32_t val = kind;
    carry_scan(has, val, s_tmp);                        // (has, val): a line started in this tile before my bytes, and the kind of the latest one
    uint32_t known, inherit = 0;
    if (has) known = (val ? 0u : before) + (n_seq - before);
    else { inherit = before; known = n_seq - before; }  // my head continues whatever line the tile started in
    for (int d = 32; d; d >>= 1) { known += __shfl_down(known, d, 64); inherit += __shfl_down(inherit, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_red[0], known); atomicAdd(&s_red[2], inherit); }
    if (my_saw) atomicMax(&s_red[1], ((uint32_t)threadIdx.x + 1u) * 2u + my_kind);      // ordered by thread, kind in the low bit
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t l = s_red[1];
        a.summary[t] = make_uint2(s_red[2] | ((l != 0u) << 30) | ((l & 1u) << 31), s_red[0]);
    }
}

__global__ __launch_bounds__(64) void fasta_tile_scan(ParseArgs a)
{
    const int f = blockIdx.x;
    if (f >= a.n) return;
    if (threadIdx.x != 0) return;
    const uint64_t t0 = a.tile_base[f], t1 = a.tile_base[f + 1];
    uint64_t off = 0; uint32_t kind = 1;                                 // byte 0 starts a line, so the start kind of tile 0 is never used
    for (uint64_t t = t0; t < t1; t++) {
        const uint2 s = a.summary[t];
        a.tile_off[t] = off; a.tile_kind[t] = (uint8_t)kind;
        off += s.y + (kind ? 0u : (s.x & 0x3FFFFFFFu));
        if (s.x & (1u << 30)) kind = s.x >> 31;
    }
    if (t1 > t0) { a.out[f][off] = '\n'; off++; }                        // terminator of the last record
    a.outlen[f] = off;
}

__global__ __launch_bounds__(PT_NT) void fasta_tile_emit(ParseArgs a)
{
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_scan[PT_NT / 64];
    __shared__ unsigned char s_priv[PT_NT * (PT_PER + 1)];                                 // thread i emits into [65 i, 65 i + 65)
    __shared__ __attribute__((aligned(16))) unsigned char s_out[PT_TILE + PT_NT + 16];     // the tile's output in order
    const uint64_t t = blockIdx.x;
    const uint32_t f = a.tile_file[t];
    const uint64_t len = a.rawlen[f], tl = t - a.tile_base[f];
    const uint8_t *raw = a.raw[f];
    const uint64_t p0 = tl * PT_TILE + (uint64_t)threadIdx.x * PT_PER;
    uint32_t w[16], nvalid; bool fls;
    load64(raw, len, p0, w, nvalid, fls);
    // state before my bytes: last line start of the tile before me, else the tile's start kind (from the scan)
    uint32_t kind = 0, before; bool saw;
    (void)walk64<false>(w, nvalid, fls, p0 == 0, kind, saw, before, nullptr);
    bool has = saw; uint32_t val = kind;
    carry_scan(has, val, s_tmp);
    uint32_t k2 = has ? val : (uint32_t)a.tile_kind[t];
    unsigned char *mine = s_priv + (uint32_t)threadIdx.x * (PT_PER + 1);
    const uint32_t n = walk64<true>(w, nvalid, fls, p0 == 0, k2, saw, before, mine);
    // exclusive scan of n over the block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_scan[wv] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int q = 0; q < PT_NT / 64; q++) { const uint32_t c = s_scan[q]; if (q < wv) base += c; total += c; }
    const uint32_t o = base + inc - n;
    for (uint32_t i = 0; i < n; i++) s_out[o + i] = mine[i];
    __syncthreads();
    // copy out: dword body aligned to the destination, bytes at both ends
    uint8_t *dst = a.out[f] + a.tile_off[t];
    const uint32_t mis = (uint32_t)((uintptr_t)dst & 3u);
    const uint32_t head = mis ? (4u - mis < total ? 4u - mis : total) : 0u;
    if (threadIdx.x < head) dst[threadIdx.x] = s_out[threadIdx.x];
    const uint32_t body = (total - head) / 4u;
    for (uint32_t v = threadIdx.x; v < body; v += PT_NT) {
        const unsigned char *q = s_out + head + 4u * v;
        *reinterpret_cast<uint32_t *>(dst + head + 4u * v) = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    }
    const uint32_t done = head + 4u * body;
    if (done + threadIdx.x < total) dst[done + threadIdx.x] = s_out[done + threadIdx.x];
}

uint64_t fasta_parse_tiles(uint64_t len) { return (len + PT_TILE - 1) / PT_TILE; }

// ---- read sets packed by the reader threads (fastx.cpp pack_*_planes): groups of 64 positions, five words each -> the record streams
// the read-set kernels take: sequence bytes A C T G (any byte of that code), N (a byte valid_base rejects), '\n'; quality bytes ' ' (passes
// every min_qual below 255: (32 - 33) & 255 = 255), '!' (fails every one), '\n'.  One thread = 16 positions = one 16-byte store per stream.
__global__ __launch_bounds__(256) void expand_planes_kernel(const uint64_t *groups, uint64_t len, uint8_t *seq, uint8_t *qual)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t * 16 >= len) return;
    uint32_t sw[4], qw[4];
    planes_bytes16(groups, t, len, sw, qw);
    *reinterpret_cast<uint4 *>(seq + t * 16) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    *reinterpret_cast<uint4 *>(qual + t * 16) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
}
void launch_expand_planes(const uint64_t *groups, uint64_t len, uint8_t *seq, uint8_t *qual, hipStream_t st)
{
    if (!len) return;
    const uint64_t threads = (len + 15) / 16;
    hipLaunchKernelGGL(expand_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, groups, len, seq, qual);
}

void launch_fasta_parse(const uint8_t *const *raw, const uint64_t *rawlen, uint8_t *const *out, uint64_t *outlen, const uint32_t *tile_file,
                        const uint64_t *tile_base, uint64_t n_tiles, void *summary /* 8 B per tile */, uint64_t *tile_off, uint8_t *tile_kind, int n,
                        hipStream_t st)
{
    if (!n) return;
    ParseArgs a{raw, rawlen, out, outlen, tile_file, tile_base, (uint2 *)summary, tile_off, tile_kind, n};
    if (n_tiles) hipLaunchKernelGGL(fasta_tile_summary, dim3((unsigned)n_tiles), dim3(PT_NT), 0, st, a);
    hipLaunchKernelGGL(fasta_tile_scan, dim3((unsigned)n), dim3(64), 0, st, a);
    if (n_tiles) hipLaunchKernelGGL(fasta_tile_emit, dim3((unsigned)n_tiles), dim3(PT_NT), 0, st, a);
}

}  // namespace skx
