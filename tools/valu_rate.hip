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
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21");   \
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
// round 3: which instructions belong to the fast group (add / and ran at ~2.7 cycles in round 2)?
DEFINE_KERNEL(or_b32, "v_or_b32 %0, %1, %0")
DEFINE_KERNEL(xor_b32, "v_xor_b32 %0, %1, %0")
DEFINE_KERNEL(sub_u32, "v_sub_u32 %0, %1, %0")
DEFINE_KERNEL(lshlrev, "v_lshlrev_b32 %0, 3, %0")
DEFINE_KERNEL(lshrrev, "v_lshrrev_b32 %0, 3, %0")
DEFINE_KERNEL(ashrrev, "v_ashrrev_i32 %0, 3, %0")
DEFINE_KERNEL(mov_b32, "v_mov_b32 %0, %1")
DEFINE_KERNEL(not_b32, "v_not_b32 %0, %0")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_only, "v_cmp_gt_u32 vcc, %1, %0")
DEFINE_KERNEL(cmp_e64, "v_cmp_gt_u32_e64 s[20:21], %1, %0")
DEFINE_KERNEL(max_u16, "v_max_u16 %0, %1, %0")
DEFINE_KERNEL(min_u16, "v_min_u16 %0, %1, %0")
DEFINE_KERNEL(add_u16, "v_add_u16 %0, %1, %0")
DEFINE_KERNEL(sub_u16, "v_sub_u16 %0, %1, %0")
DEFINE_KERNEL(max_i32, "v_max_i32 %0, %1, %0")
DEFINE_KERNEL(min_i16, "v_min_i16 %0, %1, %0")
DEFINE_KERNEL(lshl_add, "v_lshl_add_u32 %0, %1, 2, %0")
DEFINE_KERNEL(add_lshl, "v_add_lshl_u32 %0, %1, %0, 2")
DEFINE_KERNEL(xad, "v_xad_u32 %0, %1, %2, %0")
DEFINE_KERNEL(mad_i24, "v_mad_i32_i24 %0, %1, %2, %0")
DEFINE_KERNEL(mul_u24, "v_mul_u32_u24 %0, %1, %0")
DEFINE_KERNEL(add_f32, "v_add_f32 %0, %1, %0")
DEFINE_KERNEL(mul_f32, "v_mul_f32 %0, %1, %0")
DEFINE_KERNEL(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
DEFINE_KERNEL(pk_lshl_u16, "v_pk_lshlrev_b16 %0, 1, %0")
DEFINE_KERNEL(pk_mul_u16, "v_pk_mul_lo_u16 %0, %1, %0")
DEFINE_KERNEL(sat_pk_u8, "v_sat_pk_u8_i16 %0, %0")
DEFINE_KERNEL(cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
DEFINE_KERNEL(readlane, "v_readlane_b32 s20, %0, 3")
DEFINE_KERNEL(bpermute_free, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(add_sdwa_b, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEFINE_KERNEL(and_sdwa, "v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEFINE_KERNEL(sub_sdwa16, "v_sub_u16_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0")

// round 5: the opcodes of fast_blur_kernel's emitted ISA that tools/isa_mix.py still priced by assumption
DEFINE_KERNEL(max_i16, "v_max_i16 %0, %1, %0")
DEFINE_KERNEL(cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEFINE_KERNEL(lshlrev_b16, "v_lshlrev_b16 %0, 3, %0")
DEFINE_KERNEL(lshrrev_b16, "v_lshrrev_b16 %0, 3, %0")
DEFINE_KERNEL(mul_lo_u16, "v_mul_lo_u16 %0, %1, %0")
DEFINE_KERNEL(readfirstlane, "v_readfirstlane_b32 s20, %0")
DEFINE_KERNEL(cmp_lt_i16, "v_cmp_lt_i16 vcc, %1, %0")
DEFINE_KERNEL(cmp_i16_e64, "v_cmp_lt_i16_e64 s[20:21], %1, %0")
DEFINE_KERNEL(mbcnt_hi, "v_mbcnt_hi_u32_b32 %0, %1, %0")

// 64-bit forms: accumulators are register pairs
#define DEFINE_KERNEL64(NAME, ASM)                                                                                \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, int iters, unsigned seed)                     \
    {                                                                                                            \
        unsigned long long a[8];                                                                                 \
        const unsigned b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;                                \
        const unsigned long long b64 = ((unsigned long long)b << 32) | c;                                        \
        for (int i = 0; i < 8; i++) a[i] = b64 + i * 977u;                                                       \
        for (int it = 0; it < iters; it++) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(b64) : "vcc", "s20", "s21");   \
            }                                                                                                    \
        }                                                                                                        \
        unsigned long long r = 0;                                                                                \
        for (int i = 0; i < 8; i++) r ^= a[i];                                                                   \
        if (r == 0x12345u) out[threadIdx.x] = (unsigned)r;                                                       \
    }
DEFINE_KERNEL64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %3")
DEFINE_KERNEL64(mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
DEFINE_KERNEL64(mad_i64_i32, "v_mad_i64_i32 %0, s[20:21], %1, %2, %0")
DEFINE_KERNEL64(mov_b64, "v_mov_b64 %0, %3")
DEFINE_KERNEL64(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")

// round 3b: the same fast-group instructions with THREE DISTINCT registers (dst, src0, src1 all different, sources rotating) --
// is the fast rate a property of the opcode or of the operand pattern of the chains above (src1 == dst, src0 constant)?
#define DEFINE_KERNEL3(NAME, ASM)                                                                                \
    __global__ __launch_bounds__(256) void k3_##NAME(unsigned* out, int iters, unsigned seed)                    \
    {                                                                                                            \
        unsigned a[8], b[8];                                                                                     \
        for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + seed + i * 977u; b[i] = a[i] ^ 0x5bd1e995u; } \
        for (int it = 0; it < iters; it++) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(a[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 3) & 7]));  \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 3) & 7]));  \
            }                                                                                                    \
        }                                                                                                        \
        unsigned r = 0;                                                                                          \
        for (int i = 0; i < 8; i++) r ^= a[i] ^ b[i];                                                            \
        if (r == 0x12345u) out[threadIdx.x] = r;                                                                 \
    }
DEFINE_KERNEL3(min_u16, "v_min_u16 %0, %1, %2")
DEFINE_KERNEL3(xor_b32, "v_xor_b32 %0, %1, %2")
DEFINE_KERNEL3(add_u32, "v_add_u32 %0, %1, %2")
DEFINE_KERNEL3(sub_u16, "v_sub_u16 %0, %1, %2")
DEFINE_KERNEL3(pk_min_u16, "v_pk_min_u16 %0, %1, %2")
DEFINE_KERNEL3(min_u32, "v_min_u32 %0, %1, %2")

// round 3c: dependent chains -- NCH independent chains per wave (the arc network of the FAST kernel has ~4), fast vs packed opcode
template <int NCH, int KIND>
__global__ __launch_bounds__(256) void k_chain(unsigned* out, int iters, unsigned seed)
{
    unsigned a[8], b = threadIdx.x * 2654435761u + seed;
    for (int i = 0; i < 8; i++) a[i] = b + i * 977u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 64 / NCH; u++)
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if constexpr (KIND == 0) asm volatile("v_min_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                else asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// MFMA beside VALU: every wave issues ONE v_mfma_i32_16x16x64_i8 (or a 4x4x4 f32 one) per VPER vector instructions.  If the matrix
// pipe takes its own issue slots, the time per iteration stays that of the VALU chain alone until the MFMA rate saturates.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VPER, int KIND>
__global__ __launch_bounds__(256) void k_mfma_mix(unsigned* out, int iters, unsigned seed)
{
    unsigned a[8], b = threadIdx.x * 2654435761u + seed;
    for (int i = 0; i < 8; i++) a[i] = b + i * 977u;
    v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, A = {(int)b, (int)b + 1, (int)b + 2, (int)b + 3}, B = {(int)seed, 2, 3, 4};
    v4f facc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (KIND == 1) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc0) : "v"(A), "v"(B));
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc1) : "v"(A), "v"(B));
            } else if constexpr (KIND == 2) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(facc) : "v"(__builtin_bit_cast(float, b)), "v"(1.0f));
            }
#pragma unroll
            for (int i = 0; i < VPER; i++) asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(a[i & 7]) : "v"(b));
        }
    }
    unsigned r = (unsigned)(acc0.x ^ acc0.y ^ acc1.x ^ acc1.w) ^ __builtin_bit_cast(unsigned, facc.x + facc.y);
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

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
    printf("-- round 3: fast group? --\n");
    RUN(or_b32, 64) RUN(xor_b32, 64) RUN(sub_u32, 64) RUN(lshlrev, 64) RUN(lshrrev, 64) RUN(ashrrev, 64) RUN(mov_b32, 64) RUN(not_b32, 64)
    RUN(cndmask, 64) RUN(cmp_only, 64) RUN(cmp_e64, 64) RUN(max_u16, 64) RUN(min_u16, 64) RUN(add_u16, 64) RUN(sub_u16, 64) RUN(max_i32, 64)
    RUN(min_i16, 64) RUN(lshl_add, 64) RUN(add_lshl, 64) RUN(xad, 64) RUN(mad_i24, 64) RUN(mul_u24, 64) RUN(add_f32, 64) RUN(mul_f32, 64)
    RUN(cvt_u32_f32, 64) RUN(pk_lshl_u16, 64) RUN(pk_mul_u16, 64) RUN(sat_pk_u8, 64) RUN(cvt_pk_u8, 64) RUN(readlane, 64)
    RUN(bpermute_free, 64) RUN(add_sdwa_b, 64) RUN(and_sdwa, 64) RUN(sub_sdwa16, 64)
    printf("-- round 5: the rest of fast_blur_kernel's opcodes --\n");
    RUN(max_i16, 64) RUN(cndmask_e64, 64) RUN(lshlrev_b16, 64) RUN(lshrrev_b16, 64) RUN(mul_lo_u16, 64) RUN(readfirstlane, 64)
    RUN(cmp_lt_i16, 64) RUN(cmp_i16_e64, 64) RUN(mbcnt_hi, 64)
    RUN(lshl_add_u64, 64) RUN(mad_u64_u32, 64) RUN(mad_i64_i32, 64) RUN(mov_b64, 64) RUN(lshlrev_b64, 64)
    printf("-- round 3b: three distinct registers per instruction --\n");
#define RUN3(NAME) printf("%-18s %.2f\n", #NAME "_3reg", run(k3_##NAME, d, iters, 64));
    RUN3(min_u16) RUN3(xor_b32) RUN3(add_u32) RUN3(sub_u16) RUN3(pk_min_u16) RUN3(min_u32)
    printf("-- round 3c: NCH dependent chains per wave, 8 waves per SIMD (cycles per instruction per SIMD) --\n");
#define RUNCH(N, K, LABEL) printf("%-28s %.2f\n", LABEL, run(k_chain<N, K>, d, iters, 64));
    RUNCH(1, 0, "min_u16 1 chain") RUNCH(2, 0, "min_u16 2 chains") RUNCH(4, 0, "min_u16 4 chains") RUNCH(8, 0, "min_u16 8 chains")
    RUNCH(1, 1, "pk_min_u16 1 chain") RUNCH(2, 1, "pk_min_u16 2 chains") RUNCH(4, 1, "pk_min_u16 4 chains") RUNCH(8, 1, "pk_min_u16 8 chains")
    printf("-- MFMA beside VALU: microseconds-equivalent cycles per ITERATION GROUP (one group = the MFMAs + VPER v_pk_max_u16) --\n");
#define RUNMIX(V, K, LABEL) printf("%-34s %.2f cycles per group\n", LABEL, run(k_mfma_mix<V, K>, d, iters, 8));
    RUNMIX(16, 0, "16 valu, no mfma")
    RUNMIX(16, 1, "16 valu + 2 mfma_i32_16x16x64_i8")
    RUNMIX(32, 0, "32 valu, no mfma")
    RUNMIX(32, 1, "32 valu + 2 mfma_i32_16x16x64_i8")
    RUNMIX(8, 1, "8 valu + 2 mfma_i32_16x16x64_i8")
    RUNMIX(0, 1, "2 mfma_i32_16x16x64_i8 alone")
    RUNMIX(16, 2, "16 valu + 1 mfma_f32_16x16x4_f32")
    RUNMIX(0, 2, "1 mfma_f32_16x16x4_f32 alone")
    printf("%-18s %.2f  (per s_add_u32)\n", "salu", run(k_salu, d, iters, 64));
    printf("%-18s %.2f  (per pair: one v_min_u32 + one s_add_u32)\n", "valu+salu", run(k_valu_salu, d, iters, 64));
    printf("%-18s %.2f  (per ds_read_u8)\n", "lds_u8", run(k_lds<1>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b32)\n", "lds_b32", run(k_lds<4>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b64)\n", "lds_b64", run(k_lds<8>, d, iters / 4, 16));
    printf("%-18s %.2f  (per ds_read_b128)\n", "lds_b128", run(k_lds<16>, d, iters / 4, 16));
    return 0;
}
