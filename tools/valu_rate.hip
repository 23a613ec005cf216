// valu_rate.hip -- issue cost of the integer / packed VALU instructions the ORB kernels are built from, on gfx950.
// Every SIMD holds 8 waves; each wave runs a long chain of INDEPENDENT instances of ONE instruction (8 accumulators),
// so the figure is the sustained per-SIMD throughput: cycles per wave64 instruction = t * clk * 1024 / wave-instructions.
// The clock is unknown from here: read the table relative to v_fma_f32 (MI355X_MICROARCH.md: 2 cycles).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate.bin tools/valu_rate.hip && tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFINE_KERNEL(NAME, ASM)                                                                                 \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, int iters, unsigned seed)                     \
    {                                                                                                            \
        unsigned a[8], b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;                                \
        for (int i = 0; i < 8; i++) a[i] = b + i * 977u;                                                         \
        for (int it = 0; it < iters; it++) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");   \
            }                                                                                                    \
        }                                                                                                        \
        unsigned r = 0;                                                                                          \
        for (int i = 0; i < 8; i++) r ^= a[i];                                                                   \
        if (r == 0x12345u) out[threadIdx.x] = r;                                                                 \
    }

DEFINE_KERNEL(fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEFINE_KERNEL(add_u32, "v_add_u32 %0, %1, %0")
DEFINE_KERNEL(and_b32, "v_and_b32 %0, %1, %0")
DEFINE_KERNEL(min_u32, "v_min_u32 %0, %1, %0")
DEFINE_KERNEL(max3_u32, "v_max3_u32 %0, %1, %2, %0")
DEFINE_KERNEL(min3_i32, "v_min3_i32 %0, %1, %2, %0")
DEFINE_KERNEL(med3_u32, "v_med3_u32 %0, %1, %2, %0")
DEFINE_KERNEL(max_sdwa, "v_max_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
DEFINE_KERNEL(add_sdwa, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")
DEFINE_KERNEL(pk_max_u16, "v_pk_max_u16 %0, %1, %0")
DEFINE_KERNEL(pk_min_i16, "v_pk_min_i16 %0, %1, %0")
DEFINE_KERNEL(pk_add_u16, "v_pk_add_u16 %0, %1, %0")
DEFINE_KERNEL(pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
DEFINE_KERNEL(pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")
DEFINE_KERNEL(pk_fma_f32_half, "v_pk_add_f16 %0, %1, %0")
DEFINE_KERNEL(perm_b32, "v_perm_b32 %0, %1, %0, %2")
DEFINE_KERNEL(alignbyte, "v_alignbyte_b32 %0, %1, %0, 3")
DEFINE_KERNEL(dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEFINE_KERNEL(dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEFINE_KERNEL(lshl_or, "v_lshl_or_b32 %0, %1, 8, %0")
DEFINE_KERNEL(and_or, "v_and_or_b32 %0, %1, %2, %0")
DEFINE_KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
DEFINE_KERNEL(bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
DEFINE_KERNEL(mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEFINE_KERNEL(sad_u8, "v_sad_u8 %0, %1, %2, %0")
DEFINE_KERNEL(msad_u8, "v_msad_u8 %0, %1, %2, %0")
DEFINE_KERNEL(lerp_u8, "v_lerp_u8 %0, %1, %2, %0")
DEFINE_KERNEL(cmp_cnd, "v_cmp_gt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc")
DEFINE_KERNEL(mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEFINE_KERNEL(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEFINE_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(or3, "v_or3_b32 %0, %1, %2, %0")
DEFINE_KERNEL(add3, "v_add3_u32 %0, %1, %2, %0")
DEFINE_KERNEL(mul_lo, "v_mul_lo_u32 %0, %1, %0")
DEFINE_KERNEL(cvt_f32_ubyte, "v_cvt_f32_ubyte0 %0, %1")
DEFINE_KERNEL(qsad, "v_add_u32 %0, %1, %0\n\tv_add_u32 %0, %2, %0")

// scalar: a chain of independent s_add per wave (8 accumulators held in SGPRs by the compiler)
__global__ __launch_bounds__(256) void k_salu(unsigned* out, int iters, unsigned seed)
{
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %2, %2, 7\n\ts_add_u32 %3, %3, 9\n\t"
                         "s_add_u32 %4, %4, 3\n\ts_add_u32 %5, %5, 5\n\ts_add_u32 %6, %6, 7\n\ts_add_u32 %7, %7, 9"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
    }
    const unsigned r = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// VALU and SALU streams side by side in every wave (do they share issue?)
__global__ __launch_bounds__(256) void k_valu_salu(unsigned* out, int iters, unsigned seed)
{
    unsigned a[8], b = threadIdx.x * 2654435761u + seed;
    for (int i = 0; i < 8; i++) a[i] = b + i * 977u;
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i += 2)
                asm volatile("v_min_u32 %0, %4, %0\n\ts_add_u32 %2, %2, 3\n\tv_min_u32 %1, %4, %1\n\ts_add_u32 %3, %3, 5"
                             : "+v"(a[i]), "+v"(a[i + 1]), "+s"(s0), "+s"(s1) : "v"(b) : "scc");
        }
    }
    unsigned r = s0 ^ s1 ^ s2 ^ s3;
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// LDS read throughput by width (conflict-free: consecutive lanes, consecutive elements)
template <int BYTES>
__global__ __launch_bounds__(256) void k_lds(unsigned* out, int iters, unsigned seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned*>(buf)[i] = i * seed;
    __syncthreads();
    unsigned acc = 0;
    const unsigned base = (threadIdx.x * BYTES) & 4095u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const unsigned off = base + ((it + u) & 7) * 1024u;
            if constexpr (BYTES == 1) acc += buf[off];
            else if constexpr (BYTES == 4) acc += *reinterpret_cast<const unsigned*>(buf + off);
            else if constexpr (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(buf + off); acc += v.x ^ v.y; }
            else { const uint4 v = *reinterpret_cast<const uint4*>(buf + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <typename K>
static double run(K kernel, unsigned* d, int iters, int perIter)
{
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters / 8, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * iters * perIter;
    // cycles at 2.4 GHz per wave-instruction per SIMD
    return ms * 1e-3 * 2.4e9 * 1024.0 / waveInstr;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 4096);
    const int iters = 4000;
    printf("cycles per wave64 instruction per SIMD, assuming 2.4 GHz (8 waves per SIMD, independent chains)\n");
#define RUN(NAME, N) printf("%-18s %.2f\n", #NAME, run(k_##NAME, d, iters, N));
    RUN(fma_f32, 64) RUN(add_u32, 64) RUN(and_b32, 64) RUN(min_u32, 64) RUN(max3_u32, 64) RUN(min3_i32, 64) RUN(med3_u32, 64)
    RUN(max_sdwa, 64) RUN(add_sdwa, 64) RUN(pk_max_u16, 64) RUN(pk_min_i16, 64) RUN(pk_add_u16, 64) RUN(pk_sub_i16, 64)
    RUN(pk_mad_u16, 64) RUN(pk_fma_f32_half, 64) RUN(perm_b32, 64) RUN(alignbyte, 64) RUN(dot4_u8, 64) RUN(dot2_u16, 64)
    RUN(lshl_or, 64) RUN(and_or, 64) RUN(bfe_u32, 64) RUN(bfi_b32, 64) RUN(mad_u24, 64) RUN(sad_u8, 64) RUN(msad_u8, 64)
    RUN(lerp_u8, 64) RUN(cmp_cnd, 128) RUN(mbcnt, 64) RUN(bcnt, 64) RUN(mov_dpp, 64) RUN(or3, 64) RUN(add3, 64)
    RUN(mul_lo, 64) RUN(cvt_f32_ubyte, 64) RUN(qsad, 128)
    printf("%-18s %.2f  (per s_add_u32)\n", "salu", run(k_salu, d, iters, 64));
    printf("%-18s %.2f  (per pair: one v_min_u32 + one s_add_u32)\n", "valu+salu", run(k_valu_salu, d, iters, 64));
    printf("%-18s %.2f  (per ds_read_u8)\n", "lds_u8", run(k_lds<1>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b32)\n", "lds_b32", run(k_lds<4>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b64)\n", "lds_b64", run(k_lds<8>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b128)\n", "lds_b128", run(k_lds<16>, d, iters / 4, 16));
    return 0;
}
