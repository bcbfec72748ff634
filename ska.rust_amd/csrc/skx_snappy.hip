// skx_snappy.hip -- the bulk section of a .skf on the device.
//
// A .skf is snappy-frame(CBOR(MergeSkaArray)) (merge_ska_array.rs:191-204).  All but a sliver of it is the `variants`
// matrix: U x S cells, each a CBOR uint of two bytes (0x18, base letter), cut by the snappy frame format into independent
// 64 KB chunks.  That section never has to exist on the host:
//   load: the compressed chunks go to the device as they are in the file; one wavefront per chunk decodes the snappy
//         block (any valid element stream: literals, 1/2/4-byte-offset copies, overlapping copies) -- a sequential walk,
//         so it runs with many wavefronts per CU and no LDS -- then a workgroup per chunk checks the chunk's CRC-32C and
//         the 0x18 prefixes and writes the cell bytes row-major for the transpose into the sample-major matrix;
//   save: one wavefront per chunk builds the chunk's CBOR bytes in LDS from row-major cells, encodes them as snappy
//         elements (16-byte granules that repeat the bytes two back become copies, 64-byte runs one copy; the rest
//         literals -- a valid stream for any decoder, ~20x on real arrays), computes the CRC-32C and leaves a finished
//         frame chunk (header + payload) that a gather kernel packs into the file image.
// Everything outside that section (names, split k-mers, counts, the ragged ends of the section) stays with the host
// codec (skf_codec.cpp), which also remains the path for files whose cells are not all two-byte uints.
#include "skx_internal.h"
#include <algorithm>
#include <mutex>

namespace skx {

namespace {
constexpr uint32_t SNAP_CHUNK = 65536;
constexpr uint32_t CRC_POLY = 0x82F63B78u;                 // CRC-32C (Castagnoli), reflected
constexpr uint32_t CRC_SEG = 1028;                         // bytes per lane: 257 dwords, so the 64 lanes start in 64 different LDS banks

__device__ uint32_t g_crc_tab[1024];                       // slice-by-4 tables
__device__ uint32_t g_crc_shift[64];                       // x^(8 * bytes after lane i's segment) for a full 64 KB chunk
constexpr uint32_t CRC_SEG4 = 260;                         // the same for a 256-thread workgroup: 65 dwords per thread
__device__ uint32_t g_crc_shift4[256];

inline uint32_t h_gf_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1; }
    return p;
}
inline uint32_t h_xpow_bytes(uint64_t n)                   // x^(8n) mod P
{
    uint32_t p = 0x80000000u, base = 0x00800000u;         // 1 and x^8 in the reflected representation
    for (; n; n >>= 1) { if (n & 1) p = h_gf_mul(p, base); base = h_gf_mul(base, base); }
    return p;
}
int upload_tables(int device)
{
    static std::mutex mu; static bool done[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device < 64 && done[device]) return SKX_OK;
    uint32_t tab[1024], shift[64];
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1; tab[i] = c; }
    for (int t = 1; t < 4; t++) for (uint32_t i = 0; i < 256; i++) tab[t * 256 + i] = (tab[(t - 1) * 256 + i] >> 8) ^ tab[tab[(t - 1) * 256 + i] & 0xFF];
    for (uint32_t i = 0; i < 64; i++) { const uint64_t end = std::min<uint64_t>((uint64_t)(i + 1) * CRC_SEG, SNAP_CHUNK); shift[i] = h_xpow_bytes(SNAP_CHUNK - end); }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_crc_tab), tab, sizeof tab) != hipSuccess) return SKX_ENODEV;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_crc_shift), shift, sizeof shift) != hipSuccess) return SKX_ENODEV;
    uint32_t shift4[256];
    for (uint32_t i = 0; i < 256; i++) { const uint64_t end = std::min<uint64_t>((uint64_t)(i + 1) * CRC_SEG4, SNAP_CHUNK); shift4[i] = h_xpow_bytes(SNAP_CHUNK - end); }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_crc_shift4), shift4, sizeof shift4) != hipSuccess) return SKX_ENODEV;
    if (device >= 0 && device < 64) done[device] = true;
    return SKX_OK;
}

__device__ inline uint32_t gf_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1; }
    return p;
}
__device__ inline uint32_t xpow_bytes(uint32_t n)
{
    uint32_t p = 0x80000000u, base = 0x00800000u;
    for (; n; n >>= 1) { if (n & 1u) p = gf_mul(p, base); base = gf_mul(base, base); }
    return p;
}

// CRC-32C of s_buf[0 .. n) (n <= 65536) by one wavefront: every lane takes a contiguous 1 028-byte segment (slice-by-4 from
// LDS tables), the 64 partial CRCs are shifted to the end of the message in GF(2)[x] / P and XOR-ed.  Returns the CRC in
// every lane.
__device__ inline uint32_t wave_crc32c(const uint8_t *s_buf, const uint32_t *s_tab, uint32_t n, uint32_t lane)
{
    const uint32_t a = min(lane * CRC_SEG, n), b = min(a + CRC_SEG, n);
    uint32_t c = 0xFFFFFFFFu;
    uint32_t i = a;
    for (; i + 4 <= b; i += 4) {
        c ^= *(const uint32_t *)(s_buf + i);
        c = s_tab[768 + (c & 0xFF)] ^ s_tab[512 + ((c >> 8) & 0xFF)] ^ s_tab[256 + ((c >> 16) & 0xFF)] ^ s_tab[c >> 24];
    }
    for (; i < b; i++) c = s_tab[(c ^ s_buf[i]) & 0xFF] ^ (c >> 8);
    c = ~c;
    if (a == b) c = 0;                                     // empty segment
    uint32_t w = n == SNAP_CHUNK ? g_crc_shift[lane] : xpow_bytes(n - b);
    c = gf_mul(w, c);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c ^= __shfl_xor(c, d);
    return c;
}
// the same by a 256-thread workgroup (260-byte segments); s_red: 4 words of LDS.  Contains two barriers.
__device__ inline uint32_t block_crc32c(const uint8_t *s_buf, const uint32_t *s_tab, uint32_t n, uint32_t tid, uint32_t *s_red)
{
    const uint32_t a = min(tid * CRC_SEG4, n), b = min(a + CRC_SEG4, n);
    uint32_t c = 0xFFFFFFFFu;
    uint32_t i = a;
    for (; i + 4 <= b; i += 4) {
        c ^= *(const uint32_t *)(s_buf + i);
        c = s_tab[768 + (c & 0xFF)] ^ s_tab[512 + ((c >> 8) & 0xFF)] ^ s_tab[256 + ((c >> 16) & 0xFF)] ^ s_tab[c >> 24];
    }
    for (; i < b; i++) c = s_tab[(c ^ s_buf[i]) & 0xFF] ^ (c >> 8);
    c = ~c;
    if (a == b) c = 0;
    const uint32_t w = n == SNAP_CHUNK ? g_crc_shift4[tid] : xpow_bytes(n - b);
    c = gf_mul(w, c);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c ^= __shfl_xor(c, d);
    if ((tid & 63u) == 0) s_red[tid >> 6] = c;
    __syncthreads();
    c = s_red[0] ^ s_red[1] ^ s_red[2] ^ s_red[3];
    __syncthreads();
    return c;
}
__device__ inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }
}  // namespace

// One snappy block -> out (global memory), by one wavefront.  The element stream is walked from registers: the wavefront holds
// a 256-byte window of the source, one dword per lane, and a tag with its (up to four) trailing bytes is two v_readlane away.
// A single wavefront issues an instruction every few cycles at best, so the walk is made cheap per element and run by many
// wavefronts at once (no LDS, few registers):
//   * the copy this data is full of -- offset 2, the (0x18, base) pair repeated -- reads nothing: the last two output bytes
//     are carried in a scalar and the lanes write them out;
//   * literals come from the source;
//   * any other copy reads output this wavefront wrote earlier through memory: an agent-scope fence first, then one lane per
//     byte (a snappy copy is at most 64 bytes long; one that overlaps its own output repeats its last `off` bytes).
__device__ __forceinline__ int snappy_block_to_global(const uint8_t *__restrict__ in, uint32_t n_in, uint32_t ulen, uint8_t *out, uint32_t lane)
{
    const uint32_t a0 = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t *__restrict__ src4 = (const uint32_t *)(in - a0);
    uint32_t wbase = 0, win = src4[lane];
    auto fetch = [&](uint32_t ip) -> uint64_t {                        // source bytes ip .. ip+4 in the low 40 bits
        const uint32_t abs = a0 + ip;
        if (abs - wbase > 244u) { wbase = abs & ~3u; win = src4[(wbase >> 2) + lane]; }
        const uint32_t idx = abs - wbase, l = idx >> 2;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)l), hi = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)l + 1);
        return (((uint64_t)hi << 32) | lo) >> (8u * (idx & 3u));
    };
    uint32_t ip = 0, op = 0;
    {   // preamble: uncompressed length as a varint
        uint32_t v = 0, shift = 0;
        for (;;) {
            if (ip >= n_in || shift > 28) return 1;
            const uint32_t b = (uint32_t)fetch(ip++) & 0xFFu;
            v |= (b & 0x7Fu) << shift; shift += 7;
            if (!(b & 0x80u)) break;
        }
        if (v != ulen) return 1;
    }
    uint32_t pair = 0;                                                  // out[op-2] | out[op-1] << 8 when pair_ok
    bool pair_ok = false;
    volatile uint8_t *vout = out;
    while (ip < n_in) {
        const uint64_t x = fetch(ip);
        const uint32_t tag = (uint32_t)x & 0xFFu, type = tag & 3u;
        const uint32_t ext = (uint32_t)(x >> 8);                      // the four bytes after the tag
        if (((uint32_t)x & 0x00FFFF03u) == 0x00000202u && op >= 2) {   // copy, 2-byte offset == 2: the hot element
            const uint32_t len = (tag >> 2) + 1;
            if (ip + 3 > n_in || len > ulen - op) return 1;
            if (!pair_ok) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                const uint32_t t = lane < 2 ? vout[op - 2 + lane] : 0u;
                pair = (uint32_t)__builtin_amdgcn_readlane((int)t, 0) | ((uint32_t)__builtin_amdgcn_readlane((int)t, 1) << 8);
                pair_ok = true;
            }
            if (lane < len) out[op + lane] = (uint8_t)(pair >> (8u * (lane & 1u)));
            if (len & 1u) pair = (pair >> 8) | ((pair & 0xFFu) << 8);
            op += len; ip += 3;
            continue;
        }
        if (type == 0) {                                               // literal
            uint32_t len = (tag >> 2) + 1;
            ip += 1;
            if (len > 60) {
                const uint32_t nb = len - 60;
                if (ip + nb > n_in) return 1;
                len = nb == 4 ? ext : (ext & ((1u << (8 * nb)) - 1u));
                ip += nb;
                if (len >= SNAP_CHUNK) return 1;
                len += 1;
            }
            if (len > n_in - ip || len > ulen - op) return 1;
            for (uint32_t j = lane; j < len; j += 64) out[op + j] = in[ip + j];
            ip += len; op += len;
            if (len >= 2) { pair = (uint32_t)fetch(ip - 2) & 0xFFFFu; pair_ok = true; }
            else if (pair_ok) pair = (pair >> 8) | (((uint32_t)fetch(ip - 1) & 0xFFu) << 8);
            continue;
        }
        uint32_t len, off;
        if (type == 1) {
            if (ip + 2 > n_in) return 1;
            len = 4 + ((tag >> 2) & 7u); off = ((tag >> 5) << 8) | (ext & 0xFFu); ip += 2;
        } else if (type == 2) {
            if (ip + 3 > n_in) return 1;
            len = (tag >> 2) + 1; off = ext & 0xFFFFu; ip += 3;
        } else {
            if (ip + 5 > n_in) return 1;
            len = (tag >> 2) + 1; off = ext; ip += 5;
        }
        if (off == 0 || off > op || len > ulen - op) return 1;
        uint32_t back = lane;
        if (off < len) back = (off & (off - 1u)) ? lane % off : (lane & (off - 1u));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");             // this wavefront's earlier stores, visible to its loads
        uint32_t v = 0;
        if (lane < len) v = vout[op - off + back];
        if (lane < len) out[op + lane] = (uint8_t)v;
        op += len;
        if (len >= 2) { pair = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)len - 2) | ((uint32_t)__builtin_amdgcn_readlane((int)v, (int)len - 1) << 8); pair_ok = true; }
        else if (pair_ok) pair = (pair >> 8) | ((uint32_t)__builtin_amdgcn_readlane((int)v, 0) << 8);
    }
    return op == ulen ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------------------- load
// status: 0 ok, 1 corrupt snappy block, 2 checksum mismatch, 3 a cell that is not (0x18, byte)
// Step 1: chunk -> scratch[chunk * 64 KB ..): four chunks per workgroup, one per wavefront.
__global__ __launch_bounds__(256) void skf_decode_kernel(const uint8_t *__restrict__ src, const SnapChunk *__restrict__ chunks, uint32_t n_chunks,
                                                        uint8_t *__restrict__ scratch, int *status)
{
    const uint32_t lane = threadIdx.x & 63u, ci = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (ci >= n_chunks) return;
    const SnapChunk c = chunks[ci];
    const uint8_t *in = src + c.src_off;
    uint8_t *out = scratch + (uint64_t)ci * SNAP_CHUNK;
    int err = 0;
    if (c.ulen > SNAP_CHUNK) err = 1;
    else if (!c.compressed) {
        if (c.src_len != c.ulen) err = 1;
        else for (uint32_t i = lane; i < c.ulen; i += 64) out[i] = in[i];
    } else err = snappy_block_to_global(in, c.src_len, c.ulen, out, lane);
    if (err && lane == 0) atomicMax(status, err);
}

// Step 2: a workgroup per chunk: CRC-32C, 0x18 prefixes, cell bytes to their row-major place.
__global__ __launch_bounds__(256) void skf_cells_kernel(const uint8_t *__restrict__ scratch, const SnapChunk *__restrict__ chunks, uint64_t upos, uint64_t uend,
                                                       uint8_t *__restrict__ cells, uint64_t base_cell, int *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    uint8_t *s_out = s_mem;
    uint32_t *s_tab = (uint32_t *)(s_mem + SNAP_CHUNK);
    uint32_t *s_red = s_tab + 1024;
    const uint32_t tid = threadIdx.x;
    const SnapChunk c = chunks[blockIdx.x];
    const uint32_t ulen = c.ulen;
    if (ulen > SNAP_CHUNK) return;                                      // step 1 has flagged it
    for (uint32_t i = tid; i < 1024; i += 256) s_tab[i] = g_crc_tab[i];
    {
        const uint4 *g = (const uint4 *)(scratch + (uint64_t)blockIdx.x * SNAP_CHUNK);
        for (uint32_t i = tid; i < (ulen + 15) / 16; i += 256) ((uint4 *)s_out)[i] = g[i];
    }
    __syncthreads();
    const uint32_t crc = mask_crc(block_crc32c(s_out, s_tab, ulen, tid, s_red));
    if (crc != c.crc) { if (tid == 0) atomicMax(status, 2); return; }

    // the part of this chunk inside the data section [upos, uend): chunk-local bytes [lo, hi)
    const uint64_t uoff = c.uoff;
    const uint32_t lo = (uint32_t)((uoff > upos ? uoff : upos) - uoff), hi = (uint32_t)((uoff + ulen < uend ? uoff + ulen : uend) - uoff);
    if (lo >= hi) return;
    const uint32_t vpar = (uint32_t)((upos - uoff + 1) & 1u);         // local offsets of this parity hold the cell values, the others 0x18
    int odd = 0;
    {
        const uint32_t d0 = (lo + 3) >> 2, d1 = hi >> 2;
        const uint32_t mask = vpar ? 0x00FF00FFu : 0xFF00FF00u, want = vpar ? 0x00180018u : 0x18001800u;
        for (uint32_t d = d0 + tid; d < d1; d += 256) odd |= (((const uint32_t *)s_out)[d] & mask) != want;
        if (tid < 8) {                                                // the ragged ends, byte by byte
            const uint32_t e0 = tid < 4 ? lo + tid : (max(d1, d0) << 2) + (tid - 4);
            const uint32_t lim = tid < 4 ? min(d0 << 2, hi) : hi;
            if (e0 >= lo && e0 < lim && (e0 & 1u) != vpar) odd |= s_out[e0] != 0x18;
        }
    }
    if (__syncthreads_or(odd)) { if (tid == 0) atomicMax(status, 3); return; }
    const uint32_t b0 = lo + (((lo & 1u) != vpar) ? 1u : 0u);         // first value byte
    if (b0 >= hi) return;
    const uint64_t ci_lo = (uoff + b0 - upos) >> 1;
    const uint64_t count = ((uint64_t)(hi - b0) + 1) >> 1;
    const uint64_t rel_lo = ci_lo - base_cell, rel_hi = rel_lo + count;
    for (uint64_t w = (rel_lo & ~7ull) + 8ull * tid; w < rel_hi; w += 2048) {
        if (w >= rel_lo && w + 8 <= rel_hi) {
            const uint8_t *p = s_out + b0 + 2 * (uint32_t)(w - rel_lo);
            uint64_t v = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) v |= (uint64_t)p[2 * j] << (8 * j);
            *(uint64_t *)(cells + w) = v;
        } else {
            for (int j = 0; j < 8; j++) {
                const uint64_t r = w + j;
                if (r >= rel_lo && r < rel_hi) cells[r] = s_out[b0 + 2 * (uint32_t)(r - rel_lo)];
            }
        }
    }
}

int launch_skf_decode_cells(int device, const uint8_t *src, const SnapChunk *chunks, uint32_t n_chunks, uint64_t upos, uint64_t uend, uint8_t *scratch,
                            uint8_t *cells, uint64_t base_cell, int *status, hipStream_t st)
{
    if (!n_chunks) return SKX_OK;
    SKX_TRY(upload_tables(device));
    const uint32_t lds = SNAP_CHUNK + 4096 + 16;
    (void)hipFuncSetAttribute((const void *)skf_cells_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(skf_decode_kernel, dim3((n_chunks + 3) / 4), dim3(256), 0, st, src, chunks, n_chunks, scratch, status);
    hipLaunchKernelGGL(skf_cells_kernel, dim3(n_chunks), dim3(256), lds, st, scratch, chunks, upos, uend, cells, base_cell, status);
    return SKX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- save
// One finished frame chunk per 64 KB of the data section: slot = [type][len24][masked crc][payload], sizes[chunk] = 8 + payload.
constexpr uint32_t RAW_LIMIT = SNAP_CHUNK - SNAP_CHUNK / 8;          // above this the chunk is stored uncompressed (snap's rule)

__global__ __launch_bounds__(64) void skf_encode_cells_kernel(const uint8_t *__restrict__ cells, uint64_t base_cell, uint64_t upos, uint64_t uoff0,
                                                              uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    uint8_t *s_in = s_mem;
    uint32_t *s_tab = (uint32_t *)(s_mem + SNAP_CHUNK);
    const uint32_t lane = threadIdx.x;
    const uint64_t rel0 = uoff0 + (uint64_t)blockIdx.x * SNAP_CHUNK - upos;        // offset of this chunk inside the data section
    uint8_t *slot = slots + (uint64_t)blockIdx.x * SKF_SLOT;
    for (uint32_t i = lane; i < 1024; i += 64) s_tab[i] = g_crc_tab[i];
    // the chunk's CBOR bytes: 16 per step = 8 cells, 0x18 before each
    const bool value_first = rel0 & 1u;
    for (uint32_t u = lane; u < SNAP_CHUNK / 16; u += 64) {
        const uint8_t *p = cells + (((rel0 + 16ull * u) >> 1) - base_cell);
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= (uint64_t)p[j] << (8 * j);
        uint64_t x = v & 0xFFFFFFFFull, y = v >> 32;
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
        y = (y | (y << 16)) & 0x0000FFFF0000FFFFull; y = (y | (y << 8)) & 0x00FF00FF00FF00FFull;
        if (value_first) { x |= 0x1800180018001800ull; y |= 0x1800180018001800ull; }
        else { x = (x << 8) | 0x0018001800180018ull; y = (y << 8) | 0x0018001800180018ull; }
        uint64_t *d = (uint64_t *)(s_in + 16 * u);
        d[0] = x; d[1] = y;
    }
    __syncthreads();
    // snappy elements, 64 granules of 16 bytes per round
    uint8_t *out = slot + 8;
    if (lane == 0) { out[0] = 0x80; out[1] = 0x80; out[2] = 0x04; }                 // varint(65536)
    uint32_t pos = 3;
    bool raw = false;
    for (uint32_t r = 0; r < SNAP_CHUNK / 1024; r++) {
        const uint32_t g = r * 64 + lane;
        const uint4 q = *(const uint4 *)(s_in + 16 * g);
        const uint32_t prev = g ? ((const uint32_t *)s_in)[4 * g - 1] : 0u;
        // bit i of mm: byte i of the granule differs from the byte two back
        auto nz = [](uint32_t d) -> uint32_t {
            uint32_t t = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
            return ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
        };
        uint32_t mm = nz(q.x ^ ((prev >> 16) | (q.x << 16))) | (nz(q.y ^ ((q.x >> 16) | (q.y << 16))) << 4) |
                      (nz(q.z ^ ((q.y >> 16) | (q.z << 16))) << 8) | (nz(q.w ^ ((q.z >> 16) | (q.w << 16))) << 12);
        if (g == 0) mm |= 3u;                                                       // nothing to copy from yet
        const bool same = mm == 0u;
        const uint64_t m = __ballot(same);
        // maximal runs of repeating granules become copies of up to 64 bytes (the longest a snappy copy can be); runs restart at
        // the round's first granule.  A granule that does not repeat: copy up to its first differing byte, the differing
        // span as a literal, copy after it.
        const uint64_t below = (1ull << lane) - 1ull;
        const uint64_t zb = ~m & below, za = ~m & ~(below | (1ull << lane));
        const uint32_t start = zb ? 64u - (uint32_t)__clzll(zb) : 0u, end = za ? (uint32_t)__ffsll((long long)za) - 1u : 64u;
        const uint32_t ncopy = (same && ((lane - start) & 3u) == 0u) ? min(4u, end - lane) : 0u;      // granules covered by this lane's copy
        const uint32_t first = same ? 0u : (uint32_t)__ffs((int)mm) - 1u, last = same ? 0u : 31u - (uint32_t)__clz((int)mm);
        const uint32_t lit = last - first + 1u, tail = 15u - last;
        const uint32_t sz = same ? (ncopy ? 3u : 0u) : ((first ? 3u : 0u) + 1u + lit + (tail ? 3u : 0u));
        uint32_t inc = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (pos + total > RAW_LIMIT) { raw = true; break; }
        uint32_t o = pos + inc - sz;
        if (same) {
            if (ncopy) { out[o] = (uint8_t)(((16u * ncopy - 1u) << 2) | 2u); out[o + 1] = 2; out[o + 2] = 0; }      // copy, 2-byte offset
        } else {
            if (first) { out[o] = (uint8_t)(((first - 1u) << 2) | 2u); out[o + 1] = 2; out[o + 2] = 0; o += 3; }
            out[o++] = (uint8_t)((lit - 1u) << 2);                                                           // literal
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            for (uint32_t j = first; j <= last; j++) out[o++] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
            if (tail) { out[o] = (uint8_t)(((tail - 1u) << 2) | 2u); out[o + 1] = 2; out[o + 2] = 0; }
        }
        pos += total;
    }
    uint32_t payload = pos;
    if (raw) {
        for (uint32_t i = lane; i < SNAP_CHUNK / 8; i += 64) ((uint64_t *)out)[i] = ((const uint64_t *)s_in)[i];
        payload = SNAP_CHUNK;
    }
    const uint32_t crc = mask_crc(wave_crc32c(s_in, s_tab, SNAP_CHUNK, lane));
    if (lane == 0) {
        const uint32_t flen = payload + 4;
        slot[0] = raw ? 1 : 0; slot[1] = (uint8_t)flen; slot[2] = (uint8_t)(flen >> 8); slot[3] = (uint8_t)(flen >> 16);
        slot[4] = (uint8_t)crc; slot[5] = (uint8_t)(crc >> 8); slot[6] = (uint8_t)(crc >> 16); slot[7] = (uint8_t)(crc >> 24);
        sizes[blockIdx.x] = 8 + payload;
    }
}

// pack the finished chunks back to back: dense[off[c] .. off[c] + sizes[c]) = slot c
__global__ __launch_bounds__(256) void skf_gather_kernel(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ sizes, const uint64_t *__restrict__ off,
                                                        uint8_t *__restrict__ dense)
{
    const uint8_t *s = slots + (uint64_t)blockIdx.x * SKF_SLOT;
    uint8_t *d = dense + off[blockIdx.x];
    const uint32_t n = sizes[blockIdx.x];
    const uint32_t head = min(n, (uint32_t)((8u - (uint32_t)((uintptr_t)d & 7u)) & 7u));      // bytes up to 8-byte alignment of the destination
    for (uint32_t i = threadIdx.x; i < head; i += 256) d[i] = s[i];
    const uint32_t nw = (n - head) >> 3;
    for (uint32_t i = threadIdx.x; i < nw; i += 256) {
        const uint8_t *p = s + head + 8 * i;
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= (uint64_t)p[j] << (8 * j);
        *(uint64_t *)(d + head + 8 * i) = v;
    }
    for (uint32_t i = head + 8 * nw + threadIdx.x; i < n; i += 256) d[i] = s[i];
}

int launch_skf_encode_cells(int device, const uint8_t *cells, uint64_t base_cell, uint64_t upos, uint64_t uoff0, uint32_t n_chunks, uint8_t *slots,
                            uint32_t *sizes, hipStream_t st)
{
    if (!n_chunks) return SKX_OK;
    SKX_TRY(upload_tables(device));
    const uint32_t lds = SNAP_CHUNK + 4096;
    (void)hipFuncSetAttribute((const void *)skf_encode_cells_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(skf_encode_cells_kernel, dim3(n_chunks), dim3(64), lds, st, cells, base_cell, upos, uoff0, slots, sizes);
    return SKX_OK;
}
void launch_skf_gather(const uint8_t *slots, const uint32_t *sizes, const uint64_t *off, uint32_t n_chunks, uint8_t *dense, hipStream_t st)
{
    if (!n_chunks) return;
    hipLaunchKernelGGL(skf_gather_kernel, dim3(n_chunks), dim3(256), 0, st, slots, sizes, off, dense);
}

}  // namespace skx
