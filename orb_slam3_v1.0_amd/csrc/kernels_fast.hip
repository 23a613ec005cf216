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
// The per-pixel FAST work is split into stages of rising cost and falling population (compass test ->
// segment test + corner score in one -> NMS) with LDS queues between them, so each stage runs on a dense
// set of lanes: with ~6 % of pixels being corners a monolithic per-pixel function makes nearly every
// 64-lane wave pay for the most expensive path.
// Candidate order in HBM is not deterministic (one atomicAdd per block reserves the slots) -- every
// consumer is order-independent: it uses the raster key (y, x) carried in the word (S2b).
// Algorithmic bytes per pixel: 1 read (level) + 1 written (blurred level) + 4 per candidate.
#include <cstdlib>

#include "launch.h"
#include "device_math.h"
#include "fast_common.h"

namespace orbfe {

constexpr int kImgW = kFastTW + 8, kImgH = kFastTH + 8;   // 72 x 40 staged pixels (tile size: fast_common.h)
constexpr int kScW = kFastTW + 2, kScH = kFastTH + 2;     // 66 x 34 scores
constexpr int kScPitch = 68;
constexpr int kMaxTileCand = (kFastTW / 2) * (kFastTH / 2);  // strict 8-neighbour maxima: <= 1 per 2x2
constexpr int kTmpH = kFastTH + 4;                        // 36 rows of horizontal blur sums

// max over the 16 circular 9-arcs of the minimum of a[] over the arc.  Three-input min / max (v_min3_i32, v_max3_i32):
// t[k] = min of 3 consecutive, arc minimum = min3(t[k], t[k+3], t[k+6]) -- 32 + 8 instructions instead of 64 + 16.
__device__ __forceinline__ int arc_max_min(const int (&a)[16])
{
    int t[16], m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = min(min(a[k], a[(k + 1) & 15]), a[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = min(min(t[k], t[(k + 3) & 15]), t[(k + 6) & 15]);
    int g[6];
#pragma unroll
    for (int k = 0; k < 5; k++) g[k] = max(max(m9[3 * k], m9[3 * k + 1]), m9[3 * k + 2]);
    g[5] = m9[15];
    return max(max(max(g[0], g[1]), g[2]), max(max(g[3], g[4]), g[5]));
}

// min over the 16 circular 9-arcs of the maximum of a[] over the arc (the dark polarity's mirror image)
__device__ __forceinline__ int arc_min_max(const int (&a)[16])
{
    int t[16], m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = max(max(a[k], a[(k + 1) & 15]), a[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = max(max(t[k], t[(k + 3) & 15]), t[(k + 6) & 15]);
    int g[6];
#pragma unroll
    for (int k = 0; k < 5; k++) g[k] = min(min(m9[3 * k], m9[3 * k + 1]), m9[3 * k + 2]);
    g[5] = m9[15];
    return min(min(min(g[0], g[1]), g[2]), min(min(g[3], g[4]), g[5]));
}

// Stages B + C in one: on the RAW ring values, hi = (max over 9-arcs of the arc minimum) - v and
// lo = v - (min over 9-arcs of the arc maximum) are the largest margins by which a bright / dark 9-arc clears the
// centre.  The pixel is a corner at threshold th iff max(hi, lo) > th (the segment test of Fast_gpu.cu:222-267), and
// max(hi, lo) - 1 is the score the reference finds by binary search (:193-216).  80 three-input min / max instead of
// two 16-bit ring masks (64 compare / shift-in + two 9-run tests) -- the same cost -- and no separate score pass.
__device__ __forceinline__ int corner_margin(const uint8_t (*img)[kImgW], int r, int c)
{
    const int v = img[r][c];
    const int p[16] = {img[r + 3][c],     img[r + 3][c + 1], img[r + 2][c + 2], img[r + 1][c + 3], img[r][c + 3],     img[r - 1][c + 3],
                       img[r - 2][c + 2], img[r - 3][c + 1], img[r - 3][c],     img[r - 3][c - 1], img[r - 2][c - 2], img[r - 1][c - 3],
                       img[r][c - 3],     img[r + 1][c - 3], img[r + 2][c - 2], img[r + 3][c - 1]};
    return max(arc_max_min(p) - v, v - arc_min_max(p));
}

// stage A: compass points = ring bits 0 (+3,0), 4 (0,+3), 8 (-3,0), 12 (0,-3).  Nine consecutive
// ring positions always contain two ADJACENT compass points (consecutive multiples of 4), so a
// corner needs an adjacent compass pair that is bright (both > th) or dark (both < -th).
// `p` points at the centre pixel inside the staged tile (row pitch kImgW).
__device__ __forceinline__ bool compass_pass_ptr(const uint8_t* p, int th)
{
    // the compass cycle 0-4-8-12 is bipartite ({0,8} vs {4,12}) and every cross pair is adjacent, so
    // "some adjacent pair is bright" == (0 or 8 bright) and (4 or 12 bright); same for dark
    const int v = p[0];
    const int a0 = p[3 * kImgW], a4 = p[3], a8 = p[-3 * kImgW], a12 = p[-3];
    const int hiPair = min(max(a0, a8), max(a4, a12));  // thresholds move to the centre value: no per-point subtraction
    const int loPair = max(min(a0, a8), min(a4, a12));
    return hiPair > v + th || loPair < v - th;
}

// append `flag`ged lanes' value to an LDS queue (one LDS atomic per wave)
__device__ __forceinline__ void queue_push(bool flag, uint16_t value, uint16_t* q, uint32_t* qCount, int lane)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;  // wave-uniform
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(qCount, (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);  // lane 0 is active: every thread of the block calls this
    if (flag) q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// MODE is a timing-only ablation switch (ORBFE_FAST_MODE env var): bit 0 = Gaussian, bit 1 = FAST;
// the product always runs MODE 3.
template <int MODE>
__global__ __launch_bounds__(256) void fast_blur_kernel(const PipelineDesc* __restrict__ P,
                                                        const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                        int gray0Pitch, int gray0Aligned4,
                                                        uint8_t* __restrict__ ws, uint32_t* __restrict__ cand,
                                                        uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ tileRows)
{
    __shared__ uint32_t sRow[kFastTH];  // pre-NMS corners per tile row: low-pass count | high-pass count << 16
    __shared__ __attribute__((aligned(16))) uint8_t sImg[kImgH][kImgW];
    __shared__ __attribute__((aligned(16))) uint32_t sTmp[kTmpH / 2][kFastTW];  // row pairs of horizontal sums
    __shared__ __attribute__((aligned(16))) uint8_t sScore[kScH][kScPitch];
    __shared__ uint32_t sCand[kMaxTileCand];
    __shared__ uint32_t sCnt[4];  // tile: survivors, high survivors, pre-NMS low, pre-NMS high
    __shared__ uint32_t sBase;
    __shared__ uint16_t sQA[kScH * kScW];  // stage queues; entry = sy << 7 | sx (score-map position)
    // queue B lives in the Gaussian's row-pair buffer: sTmp is dead once the vertical pass has run, and a barrier
    // (after stage A) separates its last read from the first queue-B write -- 4.4 KB less LDS: 8 blocks per CU instead of 7
    static_assert(sizeof(uint16_t) * kScH * kScW <= sizeof(uint32_t) * (kTmpH / 2) * kFastTW, "queue B must fit the blur buffer");
    uint16_t* const sQB = reinterpret_cast<uint16_t*>(&sTmp[0][0]);
    __shared__ uint32_t sQ[2];

    const int f = blockIdx.x;
    const int tile = blockIdx.y;
    const int nL = P->nLevels;
    // tile -> level without a chain of dependent scalar loads: the first 8 bases arrive in one load
    int l = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) l = tile >= P->tileBaseTab[i] ? i : l;
    while (l + 1 < nL && tile >= P->lv[l + 1].tileBase) l++;  // levels >= 8 (rare)
    const LevelDesc& L = P->lv[l];
    const int w = L.w, h = L.h;
    const int t = tile - L.tileBase;
    const int x0 = (t % L.tilesX) * kFastTW;
    const int y0 = (t / L.tilesX) * kFastTH;
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
    if (tid < 2) sQ[tid] = 0;
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
    // vertical: thread -> 4 columns x output rows 2*yp, 2*yp+1 (tmp rows 2yp..2yp+5 = pairs yp..yp+2), one
    // rounding (+32768 >> 16) folded into the accumulator init; the result is byte 2 of each accumulator
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        uint8_t* dst = ws + L.blurOff + (size_t)f * L.blurFrameStride;
        const int dpitch = L.pitch;
        const int xl = (tid & 15) * 4;
        const int yp = tid >> 4;  // 0..15
        const int gy = y0 + 2 * yp, gx = x0 + xl;
        if (gy < h && gx < w) {
            const uint4 p0 = *reinterpret_cast<const uint4*>(&sTmp[yp][xl]);
            const uint4 p1 = *reinterpret_cast<const uint4*>(&sTmp[yp + 1][xl]);
            const uint4 p2 = *reinterpret_cast<const uint4*>(&sTmp[yp + 2][xl]);
            const uint32_t c0[4] = {p0.x, p0.y, p0.z, p0.w}, c1[4] = {p1.x, p1.y, p1.z, p1.w}, c2[4] = {p2.x, p2.y, p2.z, p2.w};
            uint32_t ev[4], od[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const us2 a = __builtin_bit_cast(us2, c0[i]), b = __builtin_bit_cast(us2, c1[i]), c = __builtin_bit_cast(us2, c2[i]);
                // even row 2yp: tmp rows 2yp..2yp+4 -> taps (22,62 | 88,62 | 22,-)
                uint32_t acc = __builtin_amdgcn_udot2(a, __builtin_bit_cast(us2, 0x003E0016u), 32768u, false);
                acc = __builtin_amdgcn_udot2(b, __builtin_bit_cast(us2, 0x003E0058u), acc, false);
                ev[i] = __builtin_amdgcn_udot2(c, __builtin_bit_cast(us2, 0x00000016u), acc, false);
                // odd row 2yp+1: tmp rows 2yp+1..2yp+5 -> taps (-,22 | 62,88 | 62,22)
                acc = __builtin_amdgcn_udot2(a, __builtin_bit_cast(us2, 0x00160000u), 32768u, false);
                acc = __builtin_amdgcn_udot2(b, __builtin_bit_cast(us2, 0x0058003Eu), acc, false);
                od[i] = __builtin_amdgcn_udot2(c, __builtin_bit_cast(us2, 0x0016003Eu), acc, false);
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

    }  // MODE & 1

    if constexpr ((MODE & 2) != 0) {
    // ================= FAST =================
    // tested region 6 <= x <= w-6, 6 <= y <= h-6 (Fast_gpu.cu:275,365-368: strict compares
    // against border 5 and dim-5); scores are needed for the tile + 1-px halo (NMS)
    // stage A: compass test on every position, survivors -> queue A; scores default to 0.
    // Column strips: lane = score column, the 4 waves split the 34 score rows (9,9,8,8); the 68 positions of
    // score columns 64,65 (right halo) fill the ninth slot of waves 2 and 3, so every wave does nine
    // evaluations.  The pass flags stay wave-wide lane masks (the compare result itself): one LDS atomic per
    // wave reserves its queue slots and v_mbcnt ranks the lanes -- no per-lane bit masks, no block scan.
    for (int e = tid; e < kScH * (kScPitch / 4); e += 256) reinterpret_cast<uint32_t*>(&sScore[0][0])[e] = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id, provably uniform
    {
        const int syBeg = wv * 9 - (wv > 2 ? wv - 2 : 0);  // 0, 9, 18, 26
        const int px = x0 - 1 + lane;
        const bool xok = px > kEdge && px < w - kEdge;
        bool fl[9];
        int ent[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int sy = syBeg + k, sx = lane;
            bool ok = xok;
            if (k == 8 && wv >= 2) {  // wave-uniform: halo positions 0..63 (wave 2) and 64..67 (wave 3)
                const int hp = lane + 64 * (wv - 2);
                sy = min(hp >> 1, kScH - 1);
                sx = kFastTW + (hp & 1);
                const int qx = x0 - 1 + sx;
                ok = hp < 2 * kScH && qx > kEdge && qx < w - kEdge;
            }
            const int py = y0 - 1 + sy;
            ok = ok && py > kEdge && py < h - kEdge;
            fl[k] = compass_pass_ptr(&sImg[sy + 3][sx + 3], minTh) && ok;
            ent[k] = (sy << 7) | sx;
        }
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) tot += (uint32_t)__popcll(__ballot(fl[k]));
        uint32_t base = 0;
        if (tot) {  // wave-uniform
            if (lane == 0) base = atomicAdd(&sQ[0], tot);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const unsigned long long m = __ballot(fl[k]);
                if (fl[k])
                    sQA[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base))] = (uint16_t)ent[k];
                base += (uint32_t)__popcll(m);
            }
        }
    }
    __syncthreads();
    // stage B: corner margin on queue A (dense); corners -> score map + queue B
    if constexpr ((MODE & 4) == 0) {
        const int nA = (int)sQ[0];
        const int itB = (nA + 255) / 256;
#pragma unroll 1
        for (int it = 0; it < itB; it++) {
            const int i = tid + it * 256;
            bool corner = false;
            uint16_t e = 0;
            if (i < nA) {
                e = sQA[i];
                const int sy = e >> 7;
                const int sx = e & 127;
                const int margin = corner_margin(sImg, sy + 3, sx + 3);
                corner = margin > minTh;
                if (corner) sScore[sy][sx] = (uint8_t)(margin - 1);
            }
            queue_push(corner, e, sQB, &sQ[1], lane);
        }
    }
    __syncthreads();
    const int nB = (MODE & 8) ? 0 : (int)sQ[1];

    // ---- NMS (strictly greater than all 8 neighbours, Fast_gpu.cu:300-310) + tile compaction,
    //      again over the dense corner queue; halo corners only serve as neighbours ----
    {
        const int itN = (nB + 255) / 256;
#pragma unroll 1
        for (int it = 0; it < itN; it++) {
            const int i = tid + it * 256;
            bool pre = false, keep = false, hi = false;
            int ox = 0, oy = 0, s = 0;
            if (i < nB) {
                const uint32_t q = sQB[i];
                const int sy = (int)((q >> 7) & 63u);
                const int sx = (int)(q & 127u);
                ox = sx - 1;
                oy = sy - 1;
                pre = ox >= 0 && ox < kFastTW && oy >= 0 && oy < kFastTH;  // interior: counted once
                if (pre) {
                    s = sScore[sy][sx];
                    hi = s >= iniTh;
                    atomicAdd(&sRow[oy], hi ? 0x10001u : 1u);  // only read by the exact-cap path of the quadtree kernel
                    // strictly greater than all eight == greater than their maximum (three v_max3 + one v_max)
                    const int n0 = max(max((int)sScore[sy - 1][sx - 1], (int)sScore[sy - 1][sx]), (int)sScore[sy - 1][sx + 1]);
                    const int n1 = max(max((int)sScore[sy][sx - 1], (int)sScore[sy][sx + 1]), (int)sScore[sy + 1][sx - 1]);
                    const int n2 = max((int)sScore[sy + 1][sx], (int)sScore[sy + 1][sx + 1]);
                    keep = s > max(max(n0, n1), n2);
                }
            }
            const unsigned long long mPre = __ballot(pre);
            if (mPre == 0) continue;  // wave-uniform
            const unsigned long long mPreHi = __ballot(pre && hi);
            const unsigned long long mKeep = __ballot(keep);
            const unsigned long long mKeepHi = __ballot(keep && hi);
            uint32_t wbase = 0;
            if (lane == 0) {
                atomicAdd(&sCnt[2], (uint32_t)__popcll(mPre));
                if (mPreHi) atomicAdd(&sCnt[3], (uint32_t)__popcll(mPreHi));
                if (mKeepHi) atomicAdd(&sCnt[1], (uint32_t)__popcll(mKeepHi));
                if (mKeep) wbase = atomicAdd(&sCnt[0], (uint32_t)__popcll(mKeep));
            }
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (keep) {
                const uint32_t rank = (uint32_t)__popcll(mKeep & ((1ull << lane) - 1ull));
                sCand[wbase + rank] = pack_cand(x0 + ox, y0 + oy, s);
            }
        }
    }
    __syncthreads();

    }  // MODE & 2

    // per-tile-row pre-NMS counts (128 B per tile, plain coalesced store; zero when FAST is ablated)
    if (tid < kFastTH) tileRows[((size_t)f * P->totalTiles + tile) * kFastTH + tid] = sRow[tid];

    uint32_t* cnt = counters + ((size_t)f * nL + l) * kCntWords;
    const uint32_t nTile = sCnt[0];
    if (tid == 0) {
        uint32_t base = 0;
        if (nTile) base = atomicAdd(&cnt[kCntCand], nTile);
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

void launch_fast_blur(hipStream_t s, int frames, int totalTiles, const PipelineDesc* dP, const uint8_t* gray0,
                      size_t gray0FrameStride, int gray0Pitch, int gray0Aligned4, uint8_t* ws, uint32_t* cand,
                      uint32_t* counters, uint32_t* tileRows)
{
    dim3 block(256);
    dim3 grid(frames, totalTiles);
    static const int mode = [] {
        const char* e = getenv("ORBFE_FAST_MODE");  // timing experiments only; results are wrong unless 3
        return e ? atoi(e) & 15 : 3;
    }();
#define ORBFE_LAUNCH_FB(M)                                                                               \
    hipLaunchKernelGGL(fast_blur_kernel<M>, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch, \
                       gray0Aligned4, ws, cand, counters, tileRows)
    switch (mode) {
    case 0: ORBFE_LAUNCH_FB(0); break;
    case 1: ORBFE_LAUNCH_FB(1); break;
    case 2: ORBFE_LAUNCH_FB(2); break;
    case 6: ORBFE_LAUNCH_FB(6); break;    // FAST stage A only
    case 10: ORBFE_LAUNCH_FB(10); break;  // FAST stages A + B only
    default: ORBFE_LAUNCH_FB(3); break;
    }
#undef ORBFE_LAUNCH_FB
}

}  // namespace orbfe
