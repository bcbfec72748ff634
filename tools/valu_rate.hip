// tools/valu_rate.hip: issue rate of a few VALU instructions on gfx950 (wave-instructions per SIMD and cycle, against v_xor_b32), to price
// the inner loops of the popcount kernels.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define KERNEL(name, text)                                                                                          \
    __global__ __launch_bounds__(256) void name(uint32_t *out, int iters)                                           \
    {                                                                                                               \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t x = blockIdx.x * 2654435761u + threadIdx.x, y = x * 40503u;                                       \
        for (int i = 0; i < iters; i++) {                                                                           \
            REP8(asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));) \
        }                                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
    }
// eight independent instructions per asm block, eight blocks per iteration: 64 instructions
KERNEL(k_xor, "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0x96\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0x96\n v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n v_bitop3_b32 %6, %8, %9, %6 bitop3:0x96\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7")
KERNEL(k_dot8, "v_dot8_u32_u4 %0, %8, %9, %0\n v_dot8_u32_u4 %1, %8, %9, %1\n v_dot8_u32_u4 %2, %8, %9, %2\n v_dot8_u32_u4 %3, %8, %9, %3\n v_dot8_u32_u4 %4, %8, %9, %4\n v_dot8_u32_u4 %5, %8, %9, %5\n v_dot8_u32_u4 %6, %8, %9, %6\n v_dot8_u32_u4 %7, %8, %9, %7")
KERNEL(k_add3, "v_add3_u32 %0, %8, %9, %0\n v_add3_u32 %1, %8, %9, %1\n v_add3_u32 %2, %8, %9, %2\n v_add3_u32 %3, %8, %9, %3\n v_add3_u32 %4, %8, %9, %4\n v_add3_u32 %5, %8, %9, %5\n v_add3_u32 %6, %8, %9, %6\n v_add3_u32 %7, %8, %9, %7")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7")
KERNEL(k_and_or, "v_and_or_b32 %0, %8, %9, %0\n v_and_or_b32 %1, %8, %9, %1\n v_and_or_b32 %2, %8, %9, %2\n v_and_or_b32 %3, %8, %9, %3\n v_and_or_b32 %4, %8, %9, %4\n v_and_or_b32 %5, %8, %9, %5\n v_and_or_b32 %6, %8, %9, %6\n v_and_or_b32 %7, %8, %9, %7")
KERNEL(k_sad, "v_sad_u8 %0, %8, %9, %0\n v_sad_u8 %1, %8, %9, %1\n v_sad_u8 %2, %8, %9, %2\n v_sad_u8 %3, %8, %9, %3\n v_sad_u8 %4, %8, %9, %4\n v_sad_u8 %5, %8, %9, %5\n v_sad_u8 %6, %8, %9, %6\n v_sad_u8 %7, %8, %9, %7")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %8, %0\n v_pk_add_u16 %1, %8, %1\n v_pk_add_u16 %2, %8, %2\n v_pk_add_u16 %3, %8, %3\n v_pk_add_u16 %4, %8, %4\n v_pk_add_u16 %5, %8, %5\n v_pk_add_u16 %6, %8, %6\n v_pk_add_u16 %7, %8, %7")

#define KERNEL64(name, text)                                                                                        \
    __global__ __launch_bounds__(256) void name(uint32_t *out, int iters)                                           \
    {                                                                                                               \
        uint64_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t x = (blockIdx.x * 2654435761u + threadIdx.x) & 31u, y = x * 40503u;                               \
        for (int i = 0; i < iters; i++) {                                                                           \
            REP8(asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));) \
        }                                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                   \
    }
KERNEL64(k_shl64, "v_lshlrev_b64 %0, %8, %0\n v_lshlrev_b64 %1, %8, %1\n v_lshlrev_b64 %2, %8, %2\n v_lshlrev_b64 %3, %8, %3\n v_lshlrev_b64 %4, %8, %4\n v_lshlrev_b64 %5, %8, %5\n v_lshlrev_b64 %6, %8, %6\n v_lshlrev_b64 %7, %8, %7")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7")
KERNEL64(k_cmp64, "v_cmp_gt_u64 vcc, %0, %1\n v_cmp_gt_u64 vcc, %1, %2\n v_cmp_gt_u64 vcc, %2, %3\n v_cmp_gt_u64 vcc, %3, %4\n v_cmp_gt_u64 vcc, %4, %5\n v_cmp_gt_u64 vcc, %5, %6\n v_cmp_gt_u64 vcc, %6, %7\n v_cmp_gt_u64 vcc, %7, %0")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7")
KERNEL(k_mulu24, "v_mul_u32_u24 %0, %8, %0\n v_mul_u32_u24 %1, %8, %1\n v_mul_u32_u24 %2, %8, %2\n v_mul_u32_u24 %3, %8, %3\n v_mul_u32_u24 %4, %8, %4\n v_mul_u32_u24 %5, %8, %5\n v_mul_u32_u24 %6, %8, %6\n v_mul_u32_u24 %7, %8, %7")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %8, %0, %9\n v_alignbit_b32 %1, %8, %1, %9\n v_alignbit_b32 %2, %8, %2, %9\n v_alignbit_b32 %3, %8, %3, %9\n v_alignbit_b32 %4, %8, %4, %9\n v_alignbit_b32 %5, %8, %5, %9\n v_alignbit_b32 %6, %8, %6, %9\n v_alignbit_b32 %7, %8, %7, %9")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc")

#define KERNELM(name, text)                                                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t *out, int iters)                                           \
    {                                                                                                               \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t x = blockIdx.x * 2654435761u + threadIdx.x, y = x * 40503u;                                       \
        const unsigned long long m = __ballot((threadIdx.x * 7u + blockIdx.x) & 1u);                                \
        for (int i = 0; i < iters; i++) {                                                                           \
            REP8(asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(m) : "vcc");) \
        }                                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
    }
KERNELM(k_cnd_s, "v_cndmask_b32_e64 %0, %8, %0, %10\n v_cndmask_b32_e64 %1, %8, %1, %10\n v_cndmask_b32_e64 %2, %8, %2, %10\n v_cndmask_b32_e64 %3, %8, %3, %10\n v_cndmask_b32_e64 %4, %8, %4, %10\n v_cndmask_b32_e64 %5, %8, %5, %10\n v_cndmask_b32_e64 %6, %8, %6, %10\n v_cndmask_b32_e64 %7, %8, %7, %10")
KERNELM(k_cmp_cnd, "v_cmp_lt_u32 vcc, %8, %0\n v_add_u32 %1, %9, %1\n v_xor_b32 %2, %9, %2\n v_cndmask_b32 %3, %8, %3, vcc\n v_cmp_lt_u32 vcc, %8, %4\n v_add_u32 %5, %9, %5\n v_xor_b32 %6, %9, %6\n v_cndmask_b32 %7, %8, %7, vcc")
KERNELM(k_cmp, "v_cmp_lt_u32 vcc, %8, %0\n v_cmp_lt_u32 vcc, %8, %1\n v_cmp_lt_u32 vcc, %8, %2\n v_cmp_lt_u32 vcc, %8, %3\n v_cmp_lt_u32 vcc, %8, %4\n v_cmp_lt_u32 vcc, %8, %5\n v_cmp_lt_u32 vcc, %8, %6\n v_cmp_lt_u32 vcc, %8, %7")
KERNELM(k_bfi, "v_bfi_b32 %0, %8, %9, %0\n v_bfi_b32 %1, %8, %9, %1\n v_bfi_b32 %2, %8, %9, %2\n v_bfi_b32 %3, %8, %9, %3\n v_bfi_b32 %4, %8, %9, %4\n v_bfi_b32 %5, %8, %9, %5\n v_bfi_b32 %6, %8, %9, %6\n v_bfi_b32 %7, %8, %9, %7")
KERNELM(k_addc, "v_addc_co_u32 %0, vcc, %8, %0, vcc\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc")

template <typename K> static void run(const char *name, K k, uint32_t *out, int waves_per_simd)
{
    const int iters = 20000, blocks = 256 * waves_per_simd;            // 256 threads = one wave per SIMD of a CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double wi = (double)blocks * 4 * iters * 64;               // wave-instructions
    printf("%-10s waves/SIMD %d: %8.2f ms, %.3e wave-instructions/s = %.2f per SIMD and 4 cycles at 2.4 GHz\n", name, waves_per_simd, ms, wi / (ms * 1e-3),
           wi / (ms * 1e-3) / (1024 * 2.4e9 / 4));
}
int main()
{
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 4}) {
        run("xor", k_xor, out, w); run("bcnt", k_bcnt, out, w); run("bitop3", k_bitop3, out, w); run("dot4_u8", k_dot4, out, w); run("dot8_u4", k_dot8, out, w);
        run("add3", k_add3, out, w); run("mbcnt", k_mbcnt, out, w); run("and_or", k_and_or, out, w); run("sad_u8", k_sad, out, w); run("pk_add_u16", k_pkadd, out, w);
        run("lshlrev_b64", k_shl64, out, w); run("mad_u64_u32", k_mad64, out, w); run("cmp_gt_u64", k_cmp64, out, w); run("mul_lo_u32", k_mullo, out, w);
        run("mul_u32_u24", k_mulu24, out, w); run("alignbit", k_alignbit, out, w); run("cndmask vcc", k_cndmask, out, w);
        run("cndmask sgpr", k_cnd_s, out, w); run("cmp+2+cndmask", k_cmp_cnd, out, w); run("cmp", k_cmp, out, w); run("bfi", k_bfi, out, w); run("addc", k_addc, out, w);
    }
    return 0;
}
