// kernels_fast.hip -- fused per-tile kernel: FAST-9/16 + corner score + 3x3 NMS + candidate
// compaction, AND the 5x5 Gaussian of the same tile, for every pyramid level of every frame.
//
// Replaces GpuFast::detect (src/cuda/Fast_gpu.cu:354-395): tileCalcKeypoints_kernel (:269-287,
// isKeyPoint2 :222-267, calcMask :67-182, cornerScore :193-216) and nonmaxSuppression (:289-319);
// folds the two-threshold retry of ComputeKeyPointsOctTree (src/ORBextractor.cc:449-482) into ONE
// pass; and replaces the cv::cuda Gaussian filter of ComputePyramid (:145,612,621; SPEC DECISION
// S1: Q8 taps {22,62,88,62,22}, REFLECT_101, one rounding).
//
// One pass for both FAST thresholds:
//   * the score (largest threshold at which the pixel is still a FAST-9 corner) is computed once
//     with minThFAST as the floor; a pixel is a corner of the iniThFAST pass iff score >= iniThFAST
//     (the segment test is monotone in the threshold);
//   * an NMS survivor of the low pass with score >= iniThFAST is exactly an NMS survivor of the
//     high pass (its neighbours that exist only in the low map score < iniThFAST <= its score);
//   so one candidate list + the score carried in each word serves both lists of the reference.
//   The quadtree kernel applies the retry rule and the caps from the per-level counters.
// No score map is ever written to HBM (the reference memsets and rewrites W0*H0*4 bytes per call,
// :355): scores live in LDS for the tile plus a 1-pixel halo.
//
// Tile: 64 x 32 pixels per 256-thread block; the level tile + 4-px halo (3 ring + 1 NMS; the blur
// needs 2) is staged ONCE in LDS with dword loads and feeds both computations.
//
// The kernel is bound by vector-instruction issue, with the LDS byte gathers of stage B as the second limit
// (DESIGN.md section 4.2).  gfx950 retires the plain 16-bit VOP2 integer instructions (v_min_u16 / v_max_u16 / v_sub_u16
// / v_max_i16) and add / and / or / xor at ~2.3 cycles per wave64 instruction, every packed v_pk_* form, v_perm, v_dot*,
// v_mbcnt and all 32-bit min / max at ~4.2 (tools/valu_rate.hip, profiles/r03_valu_rate.txt): ONE pixel per lane on
// the fast instructions costs what two pixels per lane cost on the packed ones, without the packing around them.
//   stage A  compass pre-test on every position: lane = column, a wave walks 17 rows down that column with a seven-row
//            register window (three byte loads + nine fast instructions per pixel); survivors (~20 % of the
//            positions) go to an LDS queue through a hand-scheduled compaction step;
//   stage B  segment test + corner score of one queued pixel per lane: the sixteen 9-arcs of both polarities by
//            prefix / suffix minima over the ring halves (42 v_min_u16 + 15 v_max_u16 per polarity);
//   stage C  NMS + compaction over the dense corner queue.
// Candidate order in HBM is not deterministic (one atomicAdd per block reserves the slots) -- every
// consumer is order-independent: it uses the raster key (y, x) carried in the word (S2b).
// Algorithmic bytes per pixel: 1 read (level) + 1 written (blurred level) + 4 per candidate.
#include <cstdlib>

#include "launch.h"
#include "device_math.h"
#include "fast_common.h"

namespace orbfe {

constexpr int kImgW = kFastTW + 8, kImgH = kFastTH + 8;   // 72 x 40 staged pixels (tile size: fast_common.h)
constexpr int kScH = kFastTH + 2;                         // 34 score rows; score columns 0..65
constexpr int kScW = kFastTW + 2;
// the score map uses the image's LDS pitch, so "score of the pixel staged at byte e" is one constant away
constexpr int kScPitch = kImgW;
constexpr int kScoreOfs = 3 * kImgW + 3;                  // score (sy, sx) <-> staged pixel (sy + 3, sx + 3)
constexpr int kMaxTileCand = (kFastTW / 2) * (kFastTH / 2);  // strict 8-neighbour maxima: <= 1 per 2x2
constexpr int kTmpH = kFastTH + 4;                        // 36 rows of horizontal blur sums
constexpr int kQCap = 2304;                               // queue A slots (>= 66 * 34), even
static_assert(kQCap >= kScH * kScW && kQCap % 2 == 0, "stage A geometry");

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));


// lane mask of (unsigned)x <= bound / of a > b, straight from the compare (a ballot of a combined predicate costs hipcc a
// v_cndmask + v_cmp pair)
__device__ __forceinline__ unsigned long long mask_le_u32(uint32_t x, uint32_t bound)
{
    unsigned long long m;
    asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "s"(bound));
    return m;
}

__device__ __forceinline__ unsigned long long mask_ge_u32(uint32_t x, uint32_t bound)
{
    unsigned long long m;
    asm("v_cmp_ge_u32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "s"(bound));
    return m;
}
__device__ __forceinline__ unsigned long long mask_gt_u32(uint32_t a, uint32_t b)
{
    unsigned long long m;
    asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
// per lane: mask bit set ? a : b
__device__ __forceinline__ uint32_t select_by_mask(unsigned long long mask, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
// LDS operations of the lanes of `mask` only (EXEC narrowed and restored inside the statement; addresses are LDS bytes)
__device__ __forceinline__ void masked_lds_add(unsigned long long mask, uint32_t addr, uint32_t v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_add_u32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "scc", "memory");
}
__device__ __forceinline__ void masked_lds_write_b32(unsigned long long mask, uint32_t addr, uint32_t v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "scc", "memory");
}
// lane masks of a < b (signed; the bound is a scalar, or a vector with `vbound`)
__device__ __forceinline__ unsigned long long mask_lt_i32(uint32_t a, int b, bool vbound = false)
{
    unsigned long long m;
    if (vbound) asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    else asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(b));
    return m;
}
__device__ __forceinline__ void masked_lds_write_b8(unsigned long long mask, uint32_t addr, uint32_t v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b8 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "scc", "memory");
}
__device__ __forceinline__ void masked_lds_write_b16(unsigned long long mask, uint32_t addr, uint32_t v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "scc", "memory");
}
// LDS add without a return value, issued by the calling lane(s) as written
__device__ __forceinline__ void lds_add(uint32_t* addr, uint32_t v)
{
    asm volatile("ds_add_u32 %0, %1" : : "v"((uint32_t)(uintptr_t)addr), "v"(v) : "memory");
}

// LDS fetch-add issued by the calling lane(s) as written (hipcc's atomic optimizer wraps a single-lane atomicAdd in a
// wave reduction)
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t* addr, uint32_t v)
{
    uint32_t old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"((uint32_t)(uintptr_t)addr), "v"(v) : "memory");
    return old;
}

// ---- round 3: the 16-bit VOP2 instructions below belong to the fast group of this chip (tools/valu_rate.hip,
// profiles/r03_valu_rate.txt: v_min_u16 / v_max_u16 / v_sub_u16 / v_min_i16, like v_add_u32 / v_and / v_or / v_xor / v_sub,
// retire in ~2.3 cycles per wave64 instruction; the packed v_pk_* forms, v_perm, v_lshl_or, three-input and 32-bit
// min / max take ~4.2).  One pixel per lane on the fast instructions costs what two pixels per lane cost on the packed
// ones -- without the byte-pair packing, the pair bookkeeping and the per-half predicates.  Inline assembly because
// hipcc widens 16-bit min / max to the slow 32-bit forms.  Operands are 0..255 (or small signed differences) in 32-bit
// registers; only the low 16 bits of a result are ever consumed.
__device__ __forceinline__ uint32_t min16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t max16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sub16(uint32_t a, uint32_t b)  // a - b, 16-bit two's complement
{
    uint32_t r;
    asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t maxi16(uint32_t a, uint32_t b)  // signed
{
    uint32_t r;
    asm("v_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max over the sixteen 9-arcs of the arc minimum of ONE pixel's ring p[0..15] (values 0..255).  Arc k = ring positions
// k..k+8.  With S[k] = min(p[k..7]), S'[k] = min(p[k..15]), P[k] = min(p[8..k]), P'[k] = min(p[0..k]):  arc k =
// min(S[k], P[k+8]) for k < 8 and min(S'[k], P'[k-8]) for k >= 8 -- 26 + 16 two-input minima, then 15 maxima
// (42 v_min_u16 + 15 v_max_u16).
__device__ __forceinline__ uint32_t arc_max_of_min(const uint32_t (&p)[16])
{
    uint32_t S[16], Pf[16];
    S[7] = p[7];
    S[15] = p[15];
#pragma unroll
    for (int k = 6; k >= 0; k--) {
        S[k] = min16(p[k], S[k + 1]);
        S[k + 8] = min16(p[k + 8], S[k + 9]);
    }
    Pf[0] = p[0];
    Pf[8] = p[8];
#pragma unroll
    for (int k = 1; k < 7; k++) {
        Pf[k] = min16(p[k], Pf[k - 1]);
        Pf[k + 8] = min16(p[k + 8], Pf[k + 7]);
    }
    Pf[7] = S[0];
    Pf[15] = S[8];
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a[k] = min16(S[k], Pf[k + 8]);
        a[k + 8] = min16(S[k + 8], Pf[k]);
    }
#pragma unroll
    for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
        for (int k = 0; k < st; k++) a[k] = max16(a[k], a[k + st]);
    return a[0];
}

// the dark polarity: min over the sixteen 9-arcs of the arc maximum (the same scheme with min / max exchanged)
__device__ __forceinline__ uint32_t arc_min_of_max(const uint32_t (&p)[16])
{
    uint32_t S[16], Pf[16];
    S[7] = p[7];
    S[15] = p[15];
#pragma unroll
    for (int k = 6; k >= 0; k--) {
        S[k] = max16(p[k], S[k + 1]);
        S[k + 8] = max16(p[k + 8], S[k + 9]);
    }
    Pf[0] = p[0];
    Pf[8] = p[8];
#pragma unroll
    for (int k = 1; k < 7; k++) {
        Pf[k] = max16(p[k], Pf[k - 1]);
        Pf[k + 8] = max16(p[k + 8], Pf[k + 7]);
    }
    Pf[7] = S[0];
    Pf[15] = S[8];
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a[k] = max16(S[k], Pf[k + 8]);
        a[k + 8] = max16(S[k + 8], Pf[k]);
    }
#pragma unroll
    for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
        for (int k = 0; k < st; k++) a[k] = min16(a[k], a[k + st]);
    return a[0];
}

// lane mask of th < (signed 16-bit) m, into an SGPR pair (the e64 compare; VCC stays free)
__device__ __forceinline__ unsigned long long mask_th_i16(int th, uint32_t m)
{
    unsigned long long r;
    asm("v_cmp_lt_i16_e64 %0, %1, %2" : "=s"(r) : "s"(th), "v"(m));
    return r;
}

// One compaction step of stage A (one pixel per lane), hand-scheduled (hipcc turns the same source into 9 vector
// instructions per step: it rebuilds the ballot through v_cndmask + v_cmp): the lanes of `mask` whose signed 16-bit margin
// m exceeds th append `entry` (the staged byte offset of their pixel) to the queue at byte address qNext + rank * step;
// returns how many did.  EXEC is narrowed and restored inside the statement.
__device__ __forceinline__ int queue_slot1(uint32_t m, int th, unsigned long long mask, uint32_t entry, int stepV, int qNext)
{
    unsigned long long save;
    uint32_t tmp;
    int cnt;
    asm volatile("v_cmp_lt_i16_e32 vcc, %8, %3\n\t"
                 "s_and_b64 vcc, vcc, %4\n\t"
                 "s_and_saveexec_b64 %0, vcc\n\t"
                 "v_mbcnt_lo_u32_b32 %1, vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %1, vcc_hi, %1\n\t"
                 "v_mad_i32_i24 %1, %1, %6, %7\n\t"
                 "ds_write_b16 %1, %5\n\t"
                 "s_mov_b64 exec, %0\n\t"
                 "s_bcnt1_i32_b64 %2, vcc"
                 : "=&s"(save), "=&v"(tmp), "=s"(cnt)
                 : "v"(m), "s"(mask), "v"(entry), "v"(stepV), "s"(qNext), "s"(th)
                 : "vcc", "scc", "memory");
    return cnt;
}

// MODE is a timing-only ablation switch (bit 0 = Gaussian, bit 1 = FAST, bit 2 = stop after stage A, bit 3 = stop
// after stage B); the shipped library instantiates MODE 3 only -- the other variants exist in the -DORBFE_ABLATION
// build (`make ablation`) used by tools/fast_stage_*.sh.
template <int MODE>
__global__ __launch_bounds__(256) void fast_blur_kernel(const PipelineDesc* __restrict__ P,
                                                        const uint32_t* __restrict__ tileInfo,
                                                        const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                        int gray0Pitch, int gray0Aligned4,
                                                        uint8_t* __restrict__ ws, uint32_t* __restrict__ cand,
                                                        uint32_t* __restrict__ counters,
                                                        uint16_t* __restrict__ tileRows)
{
    __shared__ uint32_t sRow[kFastTH];  // pre-NMS corners per tile row: low-pass count | high-pass count << 16
    __shared__ __attribute__((aligned(16))) uint8_t sImg[kImgH + 1][kImgW];  // + 1 spare row (the column walk of stage A reads at most row 39; kept so the arrays behind keep their offsets)
    __shared__ __attribute__((aligned(16))) uint32_t sTmp[kTmpH / 2][kFastTW];  // row pairs of horizontal sums
    __shared__ __attribute__((aligned(16))) uint8_t sScore[kScH + 1][kScPitch];
    __shared__ uint32_t sCand[kMaxTileCand];
    __shared__ uint32_t sCnt[4];  // tile: survivors, high survivors, pre-NMS low, pre-NMS high
    __shared__ uint32_t sBase;
    // queue A: staged-pixel byte offsets (r * 72 + c) of the positions that pass the compass test; wave 0 fills it
    // from slot 0 upwards, wave 1 from the last slot downwards (no reservation, no atomics)
    __shared__ __attribute__((aligned(4))) uint16_t sQA[kQCap];
    // the corner queue lives in the Gaussian's row-pair buffer: sTmp is dead once the vertical pass has run, and a
    // barrier (after stage A) separates its last read from the first corner-queue write
    static_assert(sizeof(uint16_t) * kScH * kScW <= sizeof(uint32_t) * (kTmpH / 2) * kFastTW, "corner queue must fit the blur buffer");
    uint16_t* const sQB = reinterpret_cast<uint16_t*>(&sTmp[0][0]);
    __shared__ uint32_t sQ[3];  // entries of queue A by wave 0 / wave 1, corner-queue entries

    const int f = blockIdx.x;
    const int tile = blockIdx.y;
    const int nL = P->nLevels;
    // tile -> (level, tile column, tile row) from one scalar load (host table, orbfe_create)
    const uint32_t ti = tileInfo[tile];
    const int l = (int)(ti >> 24);
    const LevelDesc& L = P->lv[l];
    const int w = L.w, h = L.h;
    const int x0 = (int)(ti & 0xfffu) * kFastTW;
    const int y0 = (int)((ti >> 12) & 0xfffu) * kFastTH;
    const int minTh = P->minTh, iniTh = P->iniTh;

    const uint8_t* src;
    int spitch;
    bool aligned;
    if (l == 0) {
        src = gray0 + (size_t)f * gray0FrameStride;
        spitch = gray0Pitch;
        aligned = gray0Aligned4 != 0;
    } else {
        src = ws + L.imgOff + (size_t)f * L.imgFrameStride;
        spitch = L.pitch;
        aligned = true;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (tid < 4) sCnt[tid] = 0;
    if (tid < 3) sQ[tid] = 0;
    if (tid < kFastTH) sRow[tid] = 0;

    // ---- stage the 72 x 40 tile (origin x0-4, y0-4).  Thread -> fixed dword column c4 (18 per row) and
    //      rows r0, r0+14, r0+28: the three loads are issued back to back.  Rows outside the level follow
    //      BORDER_REFLECT_101 through the row address; dwords that start inside the row are loaded whole (the
    //      bytes past w lie inside the pitch), columns outside the level are patched from LDS afterwards
    //      (only the 2 nearest are ever read, by the blur; FAST never looks outside the level) ----
    if (aligned) {
        if (tid < 14 * (kImgW / 4)) {
            const int r0 = tid / (kImgW / 4);
            const int c4 = tid - r0 * (kImgW / 4);
            const int gx = x0 - 4 + 4 * c4;
            const bool colIn = gx >= 0 && gx < w;
            uint32_t wv[3];
            // tiles whose 40 staged rows lie inside the level (block-uniform; all but the first and last tile rows) need
            // no reflection: 32-bit offsets from the uniform level base, one add per row
            if (y0 >= 4 && y0 + kFastTH + 4 <= h) {
                uint32_t off = __umul24((unsigned)(y0 - 4 + r0), (unsigned)spitch) + (unsigned)gx;
                const uint32_t step = 14u * (unsigned)spitch;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    wv[k] = 0;
                    if (colIn && r0 + 14 * k < kImgH) wv[k] = *reinterpret_cast<const uint32_t*>(src + off);
                    off += step;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int r = r0 + 14 * k;
                    wv[k] = 0;
                    if (colIn && r < kImgH)
                        wv[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)reflect_near(y0 - 4 + r, h) * spitch + gx);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int r = r0 + 14 * k;
                if (colIn && r < kImgH) *reinterpret_cast<uint32_t*>(&sImg[r][4 * c4]) = wv[k];
            }
        }
        const int cR = w - x0 + 4;  // LDS column of image column w
        const bool patchL = x0 == 0, patchR = cR + 1 < kImgW;
        if (patchL || patchR) {  // block-uniform: border tiles only
            __syncthreads();
            if (tid < kImgH) {
                if (patchL) {
                    sImg[tid][2] = sImg[tid][6];  // col -2 <- col 2
                    sImg[tid][3] = sImg[tid][5];  // col -1 <- col 1
                }
            } else if (tid >= 64 && tid < 64 + kImgH) {
                if (patchR) {
                    sImg[tid - 64][cR] = sImg[tid - 64][cR - 2];      // col w   <- col w-2
                    sImg[tid - 64][cR + 1] = sImg[tid - 64][cR - 3];  // col w+1 <- col w-3
                }
            }
        }
    } else {
        // level 0 handed over with an odd base address or pitch: byte loads
        for (int e = tid; e < kImgH * (kImgW / 4); e += 256) {
            const int r = e / (kImgW / 4);
            const int c4 = e - r * (kImgW / 4);
            const uint8_t* row = src + (size_t)reflect_near(y0 - 4 + r, h) * spitch;
            uint32_t wv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) wv |= (uint32_t)row[reflect_near(x0 - 4 + 4 * c4 + i, w)] << (8 * i);
            *reinterpret_cast<uint32_t*>(&sImg[r][4 * c4]) = wv;
        }
    }
    __syncthreads();

    if constexpr ((MODE & 1) != 0) {
    // ================= Gaussian 5x5 of the tile (S1) =================
    // horizontal: item = (row pair j, 4 output columns xl..xl+3); output column x of tmp row rr (image
    // row y0-2+rr) taps LDS cols x+2..x+6.  Byte windows via v_alignbyte, 4 taps per v_dot4_u32_u8:
    // o = dot4(bytes x+2..x+5, {22,62,88,62}) + dot4(bytes x+3..x+6, {0,0,0,22})  (<= 65280).
    // sTmp holds row PAIRS: dword [j][x] = tmp row 2j (low half) | tmp row 2j+1 (high half), so the vertical
    // pass is three v_dot2_u32_u16 per output.
    for (int e = tid; e < (kTmpH / 2) * (kFastTW / 4); e += 256) {
        const int j = e / (kFastTW / 4);
        const int xl = (e - j * (kFastTW / 4)) * 4;
        uint32_t o[2][4];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint32_t* rowp = reinterpret_cast<const uint32_t*>(&sImg[2 * j + half + 2][xl]);
            const uint32_t w0 = rowp[0], w1 = rowp[1], w2 = rowp[2];
            uint32_t win[5];
            win[0] = __builtin_amdgcn_alignbyte(w1, w0, 2);  // bytes 2..5
            win[1] = __builtin_amdgcn_alignbyte(w1, w0, 3);  // bytes 3..6
            win[2] = w1;                                     // bytes 4..7
            win[3] = __builtin_amdgcn_alignbyte(w2, w1, 1);  // bytes 5..8
            win[4] = __builtin_amdgcn_alignbyte(w2, w1, 2);  // bytes 6..9
#pragma unroll
            for (int i = 0; i < 4; i++)
                o[half][i] = __builtin_amdgcn_udot4(win[i + 1], 0x16000000u,
                                                    __builtin_amdgcn_udot4(win[i], 0x3E583E16u, 0u, false), false);
        }
        uint4 pk;
        pk.x = o[0][0] | (o[1][0] << 16);
        pk.y = o[0][1] | (o[1][1] << 16);
        pk.z = o[0][2] | (o[1][2] << 16);
        pk.w = o[0][3] | (o[1][3] << 16);
        *reinterpret_cast<uint4*>(&sTmp[j][xl]) = pk;
    }
    __syncthreads();
    }  // MODE & 1

    // ================= FAST =================
    // tested region 6 <= x <= w-6, 6 <= y <= h-6 (Fast_gpu.cu:275,365-368: strict compares
    // against border 5 and dim-5); scores are needed for the tile + 1-px halo (NMS): score position (sy, sx),
    // sy 0..33, sx 0..65, is the pixel staged at (sy + 3, sx + 3).
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id, provably uniform
    if (wv >= 2) {
        // waves 2 and 3 clear the score map and run the Gaussian's vertical pass while waves 0 and 1 run stage A
        // (spreading all three over the four waves -- 9/8/9/8-row strips, four queue segments -- was measured: 1 % slower)
        if constexpr ((MODE & 2) != 0)
            for (int e = tid - 128; e < (kScH + 1) * (kScPitch / 4); e += 128) reinterpret_cast<uint32_t*>(&sScore[0][0])[e] = 0;
        if constexpr ((MODE & 1) != 0) {
            // vertical pass of the Gaussian for the whole tile: thread -> 4 columns x output rows 2*yp, 2*yp+1 (tmp rows
            // 2yp..2yp+5 = pairs yp..yp+2), two row pairs per thread; one rounding (+32768 >> 16) folded into the
            // accumulator init; the result is byte 2 of each accumulator
            uint8_t* dst = ws + L.blurOff + (size_t)f * L.blurFrameStride;
            const int dpitch = L.pitch;
            const int xl = (tid & 15) * 4;
#pragma unroll 1
            for (int yp = (tid - 128) >> 4; yp < kFastTH / 2; yp += 8) {
                const int gy = y0 + 2 * yp, gx = x0 + xl;
                if (gy < h && gx < w) {
                    const uint4 p0 = *reinterpret_cast<const uint4*>(&sTmp[yp][xl]);
                    const uint4 p1 = *reinterpret_cast<const uint4*>(&sTmp[yp + 1][xl]);
                    const uint4 p2 = *reinterpret_cast<const uint4*>(&sTmp[yp + 2][xl]);
                    const uint32_t c0[4] = {p0.x, p0.y, p0.z, p0.w}, c1[4] = {p1.x, p1.y, p1.z, p1.w}, c2[4] = {p2.x, p2.y, p2.z, p2.w};
                    uint32_t ev[4], od[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const u16x2 a = __builtin_bit_cast(u16x2, c0[i]), b = __builtin_bit_cast(u16x2, c1[i]), c = __builtin_bit_cast(u16x2, c2[i]);
                        // even row 2yp: tmp rows 2yp..2yp+4 -> taps (22,62 | 88,62 | 22,-)
                        uint32_t acc = __builtin_amdgcn_udot2(a, __builtin_bit_cast(u16x2, 0x003E0016u), 32768u, false);
                        acc = __builtin_amdgcn_udot2(b, __builtin_bit_cast(u16x2, 0x003E0058u), acc, false);
                        ev[i] = __builtin_amdgcn_udot2(c, __builtin_bit_cast(u16x2, 0x00000016u), acc, false);
                        // odd row 2yp+1: tmp rows 2yp+1..2yp+5 -> taps (-,22 | 62,88 | 62,22)
                        acc = __builtin_amdgcn_udot2(a, __builtin_bit_cast(u16x2, 0x00160000u), 32768u, false);
                        acc = __builtin_amdgcn_udot2(b, __builtin_bit_cast(u16x2, 0x0058003Eu), acc, false);
                        od[i] = __builtin_amdgcn_udot2(c, __builtin_bit_cast(u16x2, 0x0016003Eu), acc, false);
                    }
                    // v_perm_b32(S0, S1, sel): selector 0-3 = bytes of S1, 4-7 = bytes of S0, 0x0c = 0x00
                    const uint32_t outE = __builtin_amdgcn_perm(ev[1], ev[0], 0x0c0c0602u) |
                                          (__builtin_amdgcn_perm(ev[3], ev[2], 0x0c0c0602u) << 16);
                    const uint32_t outO = __builtin_amdgcn_perm(od[1], od[0], 0x0c0c0602u) |
                                          (__builtin_amdgcn_perm(od[3], od[2], 0x0c0c0602u) << 16);
                    uint8_t* drow = dst + (size_t)gy * dpitch;
                    const bool odd = gy + 1 < h;
                    if (gx + 3 < w) {
                        *reinterpret_cast<uint32_t*>(drow + gx) = outE;  // dpitch % 64 == 0, gx % 4 == 0
                        if (odd) *reinterpret_cast<uint32_t*>(drow + dpitch + gx) = outO;
                    } else {
                        for (int i = 0; gx + i < w; i++) {
                            drow[gx + i] = (uint8_t)(outE >> (8 * i));
                            if (odd) drow[dpitch + gx + i] = (uint8_t)(outO >> (8 * i));
                        }
                    }
                }
            }
        }
    } else if constexpr ((MODE & 2) != 0) {
        // ---- stage A: compass test.  Ring bits 0 (+3,0), 4 (0,+3), 8 (-3,0), 12 (0,-3): nine consecutive ring
        //      positions always contain two ADJACENT compass points, the compass cycle 0-4-8-12 is bipartite
        //      ({0,8} vs {4,12}) and every cross pair is adjacent, so a corner needs
        //      min(max(N,S), max(E,W)) > v + th  or  max(min(N,S), min(E,W)) < v - th.
        //      ONE pixel per lane on the fast 16-bit instructions: lane = staged column 3 + lane, wave 0 walks the score
        //      rows 0..16 and wave 1 the rows 17..33 down that column with the seven column values (N, centre, S of
        //      the last rows) kept in registers -- three byte loads (W, E, the row three below) and nine arithmetic
        //      instructions per pixel; the two remaining score columns 64 and 65 are one extra step (lane = row) ----
        constexpr int kRowsA = kScH / 2;
        static_assert(2 * kRowsA == kScH && kScW == 66, "stage A geometry");
        const int R0 = wv ? 3 + kRowsA : 3;  // first staged row of this wave's strip
        // validity as lane / scalar masks: staged column c is tested iff cLo <= c <= cHi (score columns 0..65 inside the
        // FAST region kEdge < x < w - kEdge), staged row r iff rLo <= r <= rHi
        const int cLo = max(3, kEdge + 1 - (x0 - 4)), cHi = min(kScW + 2, w - kEdge - 1 - (x0 - 4));
        const int rLo = max(3, kEdge + 1 - (y0 - 4)), rHi = min(kScH + 2, h - kEdge - 1 - (y0 - 4));
        const uint32_t cSpan = (uint32_t)max(cHi - cLo, -1), rSpan = (uint32_t)max(rHi - rLo, -1);  // -1: nothing valid
        const bool anyValid = cSpan != 0xffffffffu && rSpan != 0xffffffffu;
        const unsigned long long colOK = anyValid ? mask_le_u32((uint32_t)(lane + 3 - cLo), cSpan) : 0ull;
        // compaction: wave 0 writes slots 0, 1, ... ; wave 1 writes slots kQCap-1, kQCap-2, ... -- the byte address of
        // the wave's next slot lives in a scalar, +-2 per entry (no reservation, no atomics)
        const int qStep = wv ? -2 : 2;
        const int qStepV = (tid & 64) ? -2 : 2;  // the same in a vector register (one SGPR operand per VOP3 on gfx9)
        const int qBase = (int)(reinterpret_cast<uintptr_t>(&sQA[0]) & 0xffffu);  // LDS byte address of the queue
        int qNext = qBase + (wv ? 2 * (kQCap - 1) : 0);
        uint32_t nq = 0;
        const uint8_t* base = &sImg[R0 - 3][lane];  // staged column c - 3 of the strip's first window row: W +0, column c +3, E +6
        uint32_t entry = (uint32_t)(R0 * kImgW + 3) + (uint32_t)lane;  // staged byte offset of the lane's pixel
        uint32_t col[7];  // sliding window: column values of staged rows r-3 .. r+3 (slot = window row mod 7)
#pragma unroll
        for (int k = 0; k < 6; k++) col[k] = base[k * kImgW + 3];
        uint32_t wN = base[3 * kImgW], eN = base[3 * kImgW + 6], sN = base[6 * kImgW + 3];  // requested one row ahead
#pragma unroll
        for (int i = 0; i < kRowsA; i++) {
            const uint32_t wq = wN, e = eN;
            col[(i + 6) % 7] = sN;
            if (i + 1 < kRowsA) {
                wN = base[(i + 4) * kImgW];
                eN = base[(i + 4) * kImgW + 6];
                sN = base[(i + 7) * kImgW + 3];
            }
            const uint32_t n = col[i % 7], c = col[(i + 3) % 7], so = col[(i + 6) % 7];
            const uint32_t hiV = min16(max16(n, so), max16(e, wq));
            const uint32_t loV = max16(min16(n, so), min16(e, wq));
            const uint32_t marg = maxi16(sub16(hiV, c), sub16(c, loV));  // max(bright, dark) margin: passes iff > minTh
            const unsigned long long rowMask = (uint32_t)(R0 + i - rLo) <= rSpan ? colOK : 0ull;  // scalar select
            const int cnt = queue_slot1(marg, minTh, rowMask, entry, qStepV, qNext);
            entry += (uint32_t)kImgW;
            nq += (uint32_t)cnt;
            qNext += cnt * qStep;
        }
        {
            // score columns 64 / 65 (staged columns 67 / 68): wave 0 / wave 1, lane = score row
            const int cx = kScW + 1 + wv;
            if (anyValid && (uint32_t)(cx - cLo) <= cSpan) {  // wave-uniform
                const int rr = min(lane, kScH - 1) + 3;
                const uint8_t* pc = &sImg[rr][cx];
                const uint32_t n = pc[-3 * kImgW], so = pc[3 * kImgW], c = pc[0], e = pc[3], wq = pc[-3];
                const uint32_t hiV = min16(max16(n, so), max16(e, wq));
                const uint32_t loV = max16(min16(n, so), min16(e, wq));
                const uint32_t marg = maxi16(sub16(hiV, c), sub16(c, loV));
                const unsigned long long m = mask_le_u32((uint32_t)lane, kScH - 1) & mask_le_u32((uint32_t)(rr - rLo), rSpan);
                const int cnt = queue_slot1(marg, minTh, m, (uint32_t)(rr * kImgW + cx), qStepV, qNext);
                nq += (uint32_t)cnt;
                qNext += cnt * qStep;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the queue stores above are not counted by the compiler
        if (lane == 0) sQ[wv] = nq;
    }

    if constexpr ((MODE & 2) != 0) {
    __syncthreads();
    if constexpr ((MODE & 512) != 0) {  // timing experiment: what do two more block barriers cost?
        __syncthreads();
        asm volatile("s_nop 0" ::: "memory");
        __syncthreads();
    }
    if constexpr ((MODE & 64) != 0) {
        // timing experiment (ablation build only): 128 independent packed instructions per lane on every wave.  If the kernel
        // is bound by vector issue its time grows by what they cost on a saturated SIMD; if it is bound by the waves' own
        // dependency / barrier chains it grows by a few per cent
        uint32_t d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = (uint32_t)tid * 2654435761u + (uint32_t)i;
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr ((MODE & 128) != 0) asm volatile("v_max_u16 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));  // fast-group opcode
                else asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            }
        if ((d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5] ^ d[6] ^ d[7]) == 0x12345u) sCnt[0] = 1;
    }
    // ---- stage B: segment test + corner score, one queued pixel per lane; corners -> score map + corner queue.
    //      hi = (max over the sixteen 9-arcs of the arc minimum) - v and lo = v - (min over the arcs of the arc maximum) are
    //      the largest margins by which a bright / dark 9-arc clears the centre: the pixel is a corner at threshold th iff
    //      max(hi, lo) > th (the segment test of Fast_gpu.cu:222-267) and max(hi, lo) - 1 is the score the reference finds
    //      by binary search (:193-216).  Both polarities for every pixel: scoring only the polarity whose compass test
    //      passed (inverted ring through one network) was built and measured -- 26 % of the queue passes BOTH compass
    //      tests on the bench stream (diagonal edges), and the second sweep + barrier they need cost more than the second
    //      network (fast_blur 1.27 vs 1.20 ms, profiles/r03_ab_experiments.json) ----
    if constexpr ((MODE & 4) == 0) {
        const int n0 = __builtin_amdgcn_readfirstlane((int)sQ[0]), n1 = __builtin_amdgcn_readfirstlane((int)sQ[1]);
        const int nA = n0 + n1;
        const uint8_t* img = &sImg[0][0];
        const uint32_t scoreLds = (uint32_t)(uintptr_t)&sScore[0][0] - (uint32_t)kScoreOfs, qbLds = (uint32_t)(uintptr_t)&sQB[0];
        constexpr uint32_t kIdle = (uint32_t)(20 * kImgW + 36);  // an interior position: lanes without a pixel read valid LDS
        // ring position k <-> (dy, dx): 0:(3,0) 1:(3,1) 2:(2,2) 3:(1,3) 4:(0,3) 5:(-1,3) 6:(-2,2) 7:(-3,1) 8:(-3,0)
        // 9:(-3,-1) 10:(-2,-2) 11:(-1,-3) 12:(0,-3) 13:(1,-3) 14:(2,-2) 15:(3,-1)
        constexpr int ro[16] = {3 * kImgW,      3 * kImgW + 1,  2 * kImgW + 2,  kImgW + 3,  3,  -kImgW + 3, -2 * kImgW + 2, -3 * kImgW + 1,
                                -3 * kImgW,     -3 * kImgW - 1, -2 * kImgW - 2, -kImgW - 3, -3, kImgW - 3,  2 * kImgW - 2,  3 * kImgW - 1};
#pragma unroll 1
        for (int i0 = wv * 64; i0 < nA; i0 += 256) {   // wave-uniform trip count
            const int i = i0 + lane;
            // wave 0's entry i = slot i; wave 1's entry j = slot kQCap - 1 - j
            uint32_t e = kIdle;
            if (i < nA) e = sQA[i < n0 ? i : kQCap - 1 + n0 - i];
            const uint8_t* pc = img + e;  // centre of the pixel in the staged tile
            uint32_t p[16];
            const uint32_t v = pc[0];
            if constexpr ((MODE & 32) != 0) {  // timing experiment: no ring loads
#pragma unroll
                for (int k = 0; k < 16; k++) p[k] = v + (uint32_t)((k * 37) & 63);
            } else {
                // (seven unaligned 8-byte reads per pixel instead of these sixteen byte reads were measured: fast_blur 2.5 ms
                // against 1.2 -- misaligned wide LDS reads are split by the hardware; profiles/r03_ab_experiments.json)
#pragma unroll
                for (int k = 0; k < 16; k++) p[k] = pc[ro[k]];
            }
            uint32_t margin;
            if constexpr ((MODE & 16) != 0) {  // timing experiment: no arc networks
                uint32_t acc = p[0];
#pragma unroll
                for (int k = 1; k < 16; k++) acc ^= p[k];
                margin = sub16(acc & 0xffu, v);
            } else {
                margin = maxi16(sub16(arc_max_of_min(p), v), sub16(v, arc_min_of_max(p)));
            }
            // corners of the wave: score - 1 into the score map, the pixel into the corner queue (one reservation per wave)
            const unsigned long long mc = mask_lt_i32((uint32_t)i, nA) & mask_th_i16(minTh, margin);
            masked_lds_write_b8(mc, scoreLds + e, sub16(margin, 1u));
            const uint32_t nc = (uint32_t)__popcll(mc);
            if (nc) {  // wave-uniform
                uint32_t qb = 0;
                if (lane == 0) qb = lds_add_rtn(&sQ[2], nc);
                qb = __builtin_amdgcn_readfirstlane(qb);
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc, 0u));
                masked_lds_write_b16(mc, qbLds + 2u * (qb + rk), e);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the masked LDS stores above are not counted by the compiler
    }
    __syncthreads();
    const int nB = (MODE & 8) ? 0 : (int)sQ[2];

    // ---- NMS (strictly greater than all 8 neighbours, Fast_gpu.cu:300-310) + tile compaction,
    //      again over the dense corner queue; halo corners only serve as neighbours.  The predicates live as lane masks
    //      in scalar registers from the compare on (a ballot of a combined predicate, and a one-lane atomicAdd behind
    //      hipcc's atomic optimizer, cost a dozen vector instructions each); the three plain counts are summed per
    //      wave on the scalar unit and reach LDS once, after the loop ----
    {
        uint32_t nPreW = 0, nPreHiW = 0, nKeepHiW = 0;
        const uint32_t rowLds = (uint32_t)(uintptr_t)&sRow[0];
        const uint32_t candLds = (uint32_t)(uintptr_t)&sCand[0];
        constexpr uint32_t kIdleEntry = (uint32_t)(20 * kImgW + 36);  // an interior position: inactive lanes read valid LDS
#pragma unroll 1
        for (int i0 = wv * 64; i0 < nB; i0 += 256) {  // wave-uniform
            const int i = i0 + lane;
            uint32_t e = kIdleEntry;
            if (i < nB) e = sQB[i];
            const unsigned long long mAct = mask_le_u32((uint32_t)i, (uint32_t)(nB - 1));
            const uint32_t r = (e * 3641u) >> 18;  // e / 72 for e < 2952
            const uint32_t ox = e - r * kImgW - 4u, oy = r - 4u;   // unsigned: positions left of / above the tile wrap
            // interior (counted once): 0 <= ox < kFastTW, 0 <= oy < kFastTH
            const unsigned long long mPre = mAct & mask_le_u32(ox, kFastTW - 1) & mask_le_u32(oy, kFastTH - 1);
            if (mPre == 0) continue;  // wave-uniform
            const uint8_t* sp = &sScore[0][0] + (e - kScoreOfs);
            const uint32_t sc = sp[0];
            const uint32_t n0 = max16(max16((uint32_t)sp[-kScPitch - 1], (uint32_t)sp[-kScPitch]), (uint32_t)sp[-kScPitch + 1]);
            const uint32_t n1 = max16(max16((uint32_t)sp[-1], (uint32_t)sp[1]), (uint32_t)sp[kScPitch - 1]);
            const uint32_t n2 = max16((uint32_t)sp[kScPitch], (uint32_t)sp[kScPitch + 1]);
            const uint32_t nmax = max16(max16(n0, n1), n2);  // strictly greater than all eight == greater than their maximum
            const unsigned long long mHi = mask_ge_u32(sc, (uint32_t)iniTh);
            const unsigned long long mKeep = mPre & mask_gt_u32(sc, nmax);
            const unsigned long long mPreHi = mPre & mHi, mKeepHi = mKeep & mHi;
            nPreW += (uint32_t)__popcll(mPre);
            nPreHiW += (uint32_t)__popcll(mPreHi);
            nKeepHiW += (uint32_t)__popcll(mKeepHi);
            // pre-NMS corners per tile row (only read by the exact-cap path of the quadtree kernel): low | high << 16
            masked_lds_add(mPre, rowLds + 4u * oy, select_by_mask(mHi, 0x10001u, 1u));
            if (mKeep) {  // wave-uniform
                uint32_t wbase = 0;
                if (lane == 0) wbase = lds_add_rtn(&sCnt[0], (uint32_t)__popcll(mKeep));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mKeep >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mKeep, 0u));
                masked_lds_write_b32(mKeep, candLds + 4u * (wbase + rank), pack_cand(x0 + (int)ox, y0 + (int)oy, (int)sc));
            }
        }
        if (lane == 0) {
            if (nPreW) lds_add(&sCnt[2], nPreW);
            if (nPreHiW) lds_add(&sCnt[3], nPreHiW);
            if (nKeepHiW) lds_add(&sCnt[1], nKeepHiW);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS operations above are not counted by the compiler
    }
    __syncthreads();

    }  // MODE & 2

    // per-tile-row pre-NMS counts, low pass | high pass << 8 (each <= 64, the tile's width): 64 B per tile, one coalesced store
    // instruction (zero when FAST is ablated).  Only the exact-cap path of the quadtree kernel reads them.  One total per tile
    // (4 B) with the row counts re-derived there was built and measured: this kernel -2.2 %, but the quadtree kernel 0.33 -> 0.81 ms
    // per 512 frames at the reference node's nFast = 1.6 x nFeatures, where the cap fires on every level
    // (profiles/r05_ab_experiments.json)
    if (tid < kFastTH) {
        const uint32_t v = sRow[tid];
        tileRows[((size_t)f * P->totalTiles + tile) * kFastTH + tid] = (uint16_t)((v & 0xffu) | ((v >> 16) << 8));
    }

    uint32_t* cnt = counters + ((size_t)f * nL + l) * kCntWords;
    const uint32_t nTile = sCnt[0];
    if (tid == 0) {
        uint32_t base = 0;
        if constexpr ((MODE & 256) != 0) base = (uint32_t)tile * 8u;  // timing experiment: no returning atomic (results wrong)
        else if (nTile) base = atomicAdd(&cnt[kCntCand], nTile);
        if (sCnt[1]) atomicAdd(&cnt[kCntHigh], sCnt[1]);
        if (sCnt[2]) atomicAdd(&cnt[kCntPreLow], sCnt[2]);
        if (sCnt[3]) atomicAdd(&cnt[kCntPreHigh], sCnt[3]);
        sBase = base;
    }
    __syncthreads();
    if (nTile) {
        const uint32_t base = sBase;
        uint32_t* out = cand + L.candOff + (size_t)f * L.candCap;
        for (uint32_t i = tid; i < nTile; i += 256) {
            if (base + i < (uint32_t)L.candCap) out[base + i] = sCand[i];
            else atomicOr(&cnt[kCntStatus], (uint32_t)kFlagCandOverflow);
        }
    }
}

void fast_tiles_for(int w, int h, int* tx, int* ty)
{
    *tx = (w + kFastTW - 1) / kFastTW;
    *ty = (h + kFastTH - 1) / kFastTH;
}

uint32_t fast_tile_info(int level, int tileX, int tileY) { return ((uint32_t)level << 24) | ((uint32_t)tileY << 12) | (uint32_t)tileX; }

void launch_fast_blur(hipStream_t s, int frames, int totalTiles, const PipelineDesc* dP, const uint32_t* dTileInfo,
                      const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, int gray0Aligned4, uint8_t* ws,
                      uint32_t* cand, uint32_t* counters, uint16_t* tileRows)
{
    dim3 block(256);
    dim3 grid(frames, totalTiles);
#define ORBFE_LAUNCH_FB(M)                                                                                          \
    hipLaunchKernelGGL(fast_blur_kernel<M>, grid, block, 0, s, dP, dTileInfo, gray0, gray0FrameStride, gray0Pitch, \
                       gray0Aligned4, ws, cand, counters, tileRows)
#ifdef ORBFE_ABLATION
    // timing-only build (liborbfe_ablation.so, `make ablation`): ORBFE_FAST_MODE selects a truncated kernel whose
    // RESULTS ARE WRONG unless it is 3.  The shipped library does not contain these variants and reads no variable.
    static const int mode = [] {
        const char* e = getenv("ORBFE_FAST_MODE");
        return e ? atoi(e) & 1023 : 3;
    }();
    switch (mode) {
    case 0: ORBFE_LAUNCH_FB(0); break;
    case 1: ORBFE_LAUNCH_FB(1); break;
    case 2: ORBFE_LAUNCH_FB(2); break;
    case 6: ORBFE_LAUNCH_FB(6); break;    // FAST stage A only
    case 10: ORBFE_LAUNCH_FB(10); break;  // FAST stages A + B only
    case 26: ORBFE_LAUNCH_FB(26); break;  // A + B, stage B without its arc network (what do the ring loads cost?)
    case 67: ORBFE_LAUNCH_FB(67); break;    // product kernel + 128 padding instructions per wave (is it issue-bound?)
    case 195: ORBFE_LAUNCH_FB(195); break;  // the same padding with a fast-group opcode
    case 259: ORBFE_LAUNCH_FB(259); break;  // product kernel without the returning atomicAdd of the candidate reservation
    case 515: ORBFE_LAUNCH_FB(515); break;  // product kernel + two extra block barriers
    case 42: ORBFE_LAUNCH_FB(42); break;  // A + B, stage B without its ring loads (what does the arithmetic cost?)
    default: ORBFE_LAUNCH_FB(3); break;
    }
#else
    ORBFE_LAUNCH_FB(3);
#endif
#undef ORBFE_LAUNCH_FB
}

}  // namespace orbfe
