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
// The per-pixel FAST work is split into three stages of rising cost and falling population
// (compass test -> full segment test -> corner score) with LDS queues between them, so each stage
// runs on a dense set of lanes: with ~6 % of pixels being corners a monolithic per-pixel function
// makes nearly every 64-lane wave pay for the most expensive path.
// Candidate order in HBM is not deterministic (one atomicAdd per block reserves the slots) -- every
// consumer is order-independent: it uses the raster key (y, x) carried in the word (S2b).
// Algorithmic bytes per pixel: 1 read (level) + 1 written (blurred level) + 4 per candidate.
#include "launch.h"

namespace orbfe {

constexpr int kFastTW = 64, kFastTH = 32;
constexpr int kImgW = kFastTW + 8, kImgH = kFastTH + 8;   // 72 x 40 staged pixels
constexpr int kScW = kFastTW + 2, kScH = kFastTH + 2;     // 66 x 34 scores
constexpr int kScPitch = 68;
constexpr int kMaxTileCand = (kFastTW / 2) * (kFastTH / 2);  // strict 8-neighbour maxima: <= 1 per 2x2
constexpr int kTmpH = kFastTH + 4;                        // 36 rows of horizontal blur sums

// 16-bit circular mask contains >= 9 contiguous ones (== c_table lookup, Fast_gpu.cu:187-191)
__device__ __forceinline__ bool arc9(uint32_t m)
{
    uint32_t m2 = m | (m << 16);
    uint32_t r = m2 & (m2 >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m2 >> 8;
    return (r & 0xffffu) != 0;
}

// max over the 16 circular 9-arcs of the minimum of a[] over the arc
__device__ __forceinline__ int arc_max_min(const int (&a)[16])
{
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m2[k] = min(a[k], a[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m8[k] = min(m4[k], m4[(k + 4) & 15]);
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; k++) best = max(best, min(m8[k], a[(k + 8) & 15]));
    return best;
}

// stage A: compass points = ring bits 0 (+3,0), 4 (0,+3), 8 (-3,0), 12 (0,-3).  A 9-arc always
// holds two adjacent compass points, so fewer than two bright (or dark) ones => not a corner.
__device__ __forceinline__ bool compass_pass(const uint8_t (*img)[kImgW], int r, int c, int th)
{
    const int v = img[r][c];
    const int d0 = img[r + 3][c] - v, d4 = img[r][c + 3] - v, d8 = img[r - 3][c] - v, d12 = img[r][c - 3] - v;
    const int nb = (d0 > th) + (d4 > th) + (d8 > th) + (d12 > th);
    const int nd = (d0 < -th) + (d4 < -th) + (d8 < -th) + (d12 < -th);
    return nb >= 2 || nd >= 2;
}

__device__ __forceinline__ void ring_diffs(const uint8_t (*img)[kImgW], int r, int c, int (&d)[16])
{
    const int v = img[r][c];
    d[0] = img[r + 3][c] - v;      d[1] = img[r + 3][c + 1] - v;  d[2] = img[r + 2][c + 2] - v;
    d[3] = img[r + 1][c + 3] - v;  d[4] = img[r][c + 3] - v;      d[5] = img[r - 1][c + 3] - v;
    d[6] = img[r - 2][c + 2] - v;  d[7] = img[r - 3][c + 1] - v;  d[8] = img[r - 3][c] - v;
    d[9] = img[r - 3][c - 1] - v;  d[10] = img[r - 2][c - 2] - v; d[11] = img[r - 1][c - 3] - v;
    d[12] = img[r][c - 3] - v;     d[13] = img[r + 1][c - 3] - v; d[14] = img[r + 2][c - 2] - v;
    d[15] = img[r + 3][c - 1] - v;
}

// stage B: full 16-pixel segment test at threshold th
__device__ __forceinline__ bool segment_test(const int (&d)[16], int th)
{
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mb |= (uint32_t)(d[k] > th) << k;
        md |= (uint32_t)(d[k] < -th) << k;
    }
    return arc9(mb) || arc9(md);
}

// stage C: largest t such that some 9-arc has all |diff| > t  ==  max-min over arcs, minus 1
// (equals the binary search of cornerScore, Fast_gpu.cu:193-216)
__device__ __forceinline__ int corner_score(const int (&d)[16])
{
    int nd_[16];
#pragma unroll
    for (int k = 0; k < 16; k++) nd_[k] = -d[k];
    return max(arc_max_min(d), arc_max_min(nd_)) - 1;
}

// append `flag`ged lanes' value to an LDS queue (one LDS atomic per wave)
__device__ __forceinline__ void queue_push(bool flag, uint16_t value, uint16_t* q, uint32_t* qCount, int lane)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;  // wave-uniform
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(qCount, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (flag) q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// REFLECT_101 for small overshoots (|overshoot| < n), clamped for the don't-care region far outside
__device__ __forceinline__ int reflect_near(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__global__ __launch_bounds__(256) void fast_blur_kernel(const PipelineDesc* __restrict__ P,
                                                        const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                        int gray0Pitch, int gray0Aligned4,
                                                        uint8_t* __restrict__ ws, uint32_t* __restrict__ cand,
                                                        uint32_t* __restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint8_t sImg[kImgH][kImgW];
    __shared__ __attribute__((aligned(16))) uint16_t sTmp[kTmpH][kFastTW];
    __shared__ uint8_t sScore[kScH][kScPitch];
    __shared__ uint32_t sCand[kMaxTileCand];
    __shared__ uint32_t sCnt[4];  // tile: survivors, high survivors, pre-NMS low, pre-NMS high
    __shared__ uint32_t sBase;
    __shared__ uint16_t sQA[kScH * kScW];  // stage queues (position indices)
    __shared__ uint16_t sQB[kScH * kScW];
    __shared__ uint32_t sQ[2];

    const int f = blockIdx.x;
    const int tile = blockIdx.y;
    const int nL = P->nLevels;
    int l = 0;
    while (l + 1 < nL && tile >= P->lv[l + 1].tileBase) l++;
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

    // ---- stage the 72 x 40 tile (origin x0-4, y0-4); pixels outside the level follow
    //      BORDER_REFLECT_101 (needed by the blur; FAST never looks at them) ----
    for (int e = tid; e < kImgH * (kImgW / 4); e += 256) {
        const int r = e / (kImgW / 4);
        const int c4 = e - r * (kImgW / 4);
        const int gy = reflect_near(y0 - 4 + r, h);
        const int gx = x0 - 4 + 4 * c4;
        const uint8_t* row = src + (size_t)gy * spitch;
        uint32_t wv;
        if (aligned && gx >= 0 && gx + 3 < w) {
            wv = *reinterpret_cast<const uint32_t*>(row + gx);
        } else {
            wv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) wv |= (uint32_t)row[reflect_near(gx + i, w)] << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(&sImg[r][4 * c4]) = wv;
    }
    __syncthreads();

    // ================= Gaussian 5x5 of the tile (S1) =================
    // horizontal: output column xl (0..63) of row rr (image row y0-2+rr) taps LDS cols xl+2..xl+6
    for (int e = tid; e < kTmpH * (kFastTW / 4); e += 256) {
        const int rr = e / (kFastTW / 4);
        const int xl = (e - rr * (kFastTW / 4)) * 4;
        const uint32_t* rowp = reinterpret_cast<const uint32_t*>(&sImg[rr + 2][xl]);
        const uint32_t w0 = rowp[0], w1 = rowp[1], w2 = rowp[2];
        uint32_t b[12];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            b[i] = (w0 >> (8 * i)) & 0xff;
            b[4 + i] = (w1 >> (8 * i)) & 0xff;
            b[8 + i] = (w2 >> (8 * i)) & 0xff;
        }
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            o[i] = 22u * b[i + 2] + 62u * b[i + 3] + 88u * b[i + 4] + 62u * b[i + 5] + 22u * b[i + 6];  // <= 65280
        uint2 pk;
        pk.x = o[0] | (o[1] << 16);
        pk.y = o[2] | (o[3] << 16);
        *reinterpret_cast<uint2*>(&sTmp[rr][xl]) = pk;
    }
    __syncthreads();
    // vertical: thread -> 4 columns x 2 rows (rows yp and yp+16), one rounding (+32768 >> 16)
    {
        uint8_t* dst = ws + L.blurOff + (size_t)f * L.blurFrameStride;
        const int dpitch = L.pitch;
        const int xl = (tid & 15) * 4;
        const int yp = tid >> 4;  // 0..15
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int yl = yp + 16 * half;
            const int gy = y0 + yl, gx = x0 + xl;
            if (gy < h && gx < w) {
                uint32_t acc[4] = {0, 0, 0, 0};
                const uint32_t kw[5] = {22u, 62u, 88u, 62u, 22u};
#pragma unroll
                for (int tt = 0; tt < 5; tt++) {
                    const uint2 v = *reinterpret_cast<const uint2*>(&sTmp[yl + tt][xl]);
                    acc[0] += kw[tt] * (v.x & 0xffffu);
                    acc[1] += kw[tt] * (v.x >> 16);
                    acc[2] += kw[tt] * (v.y & 0xffffu);
                    acc[3] += kw[tt] * (v.y >> 16);
                }
                const uint32_t outw = ((acc[0] + 32768u) >> 16) | (((acc[1] + 32768u) >> 16) << 8) |
                                      (((acc[2] + 32768u) >> 16) << 16) | (((acc[3] + 32768u) >> 16) << 24);
                uint8_t* drow = dst + (size_t)gy * dpitch;
                if (gx + 3 < w) {
                    *reinterpret_cast<uint32_t*>(drow + gx) = outw;  // dpitch % 64 == 0, gx % 4 == 0
                } else {
                    for (int i = 0; gx + i < w; i++) drow[gx + i] = (uint8_t)(outw >> (8 * i));
                }
            }
        }
    }

    // ================= FAST =================
    // tested region 6 <= x <= w-6, 6 <= y <= h-6 (Fast_gpu.cu:275,365-368: strict compares
    // against border 5 and dim-5); scores are needed for the tile + 1-px halo (NMS)
    // stage A: compass test on every position, survivors -> queue A; scores default to 0
    constexpr int kPos = kScH * kScW;
    constexpr int kIterA = (kPos + 255) / 256;
#pragma unroll 1
    for (int it = 0; it < kIterA; it++) {  // uniform trip count: queue_push uses wave ballots
        const int e = tid + it * 256;
        bool pass = false;
        if (e < kPos) {
            const int sy = e / kScW;
            const int sx = e - sy * kScW;
            const int px = x0 - 1 + sx, py = y0 - 1 + sy;
            sScore[sy][sx] = 0;
            if (px > kEdge && px < w - kEdge && py > kEdge && py < h - kEdge)
                pass = compass_pass(sImg, sy + 3, sx + 3, minTh);
        }
        queue_push(pass, (uint16_t)e, sQA, &sQ[0], lane);
    }
    __syncthreads();
    // stage B: full segment test on queue A (dense), corners -> queue B
    {
        const int nA = (int)sQ[0];
        const int itB = (nA + 255) / 256;
#pragma unroll 1
        for (int it = 0; it < itB; it++) {
            const int i = tid + it * 256;
            bool corner = false;
            uint16_t e = 0;
            if (i < nA) {
                e = sQA[i];
                const int sy = e / kScW;
                const int sx = e - sy * kScW;
                int d[16];
                ring_diffs(sImg, sy + 3, sx + 3, d);
                corner = segment_test(d, minTh);
            }
            queue_push(corner, e, sQB, &sQ[1], lane);
        }
    }
    __syncthreads();
    // stage C: corner score on queue B (dense)
    {
        const int nB = (int)sQ[1];
        for (int i = tid; i < nB; i += 256) {
            const int e = sQB[i];
            const int sy = e / kScW;
            const int sx = e - sy * kScW;
            int d[16];
            ring_diffs(sImg, sy + 3, sx + 3, d);
            sScore[sy][sx] = (uint8_t)corner_score(d);
        }
    }
    __syncthreads();

    // ---- NMS (strictly greater than all 8 neighbours, Fast_gpu.cu:300-310) + tile compaction ----
    for (int e = tid; e < kFastTW * kFastTH; e += 256) {
        const int oy = e / kFastTW;
        const int ox = e - oy * kFastTW;
        const int s = sScore[oy + 1][ox + 1];
        bool keep = false;
        if (s > 0) {
            keep = s > sScore[oy][ox] && s > sScore[oy][ox + 1] && s > sScore[oy][ox + 2] &&
                   s > sScore[oy + 1][ox] && s > sScore[oy + 1][ox + 2] && s > sScore[oy + 2][ox] &&
                   s > sScore[oy + 2][ox + 1] && s > sScore[oy + 2][ox + 2];
        }
        const bool hi = s >= iniTh;
        const unsigned long long mPre = __ballot(s > 0);
        if (mPre == 0) continue;  // wave-uniform: nothing in these 64 pixels
        const unsigned long long mPreHi = __ballot(s > 0 && hi);
        const unsigned long long mKeep = __ballot(keep);
        const unsigned long long mKeepHi = __ballot(keep && hi);
        uint32_t wbase = 0;
        if (lane == 0) {
            atomicAdd(&sCnt[2], (uint32_t)__popcll(mPre));
            if (mPreHi) atomicAdd(&sCnt[3], (uint32_t)__popcll(mPreHi));
            if (mKeepHi) atomicAdd(&sCnt[1], (uint32_t)__popcll(mKeepHi));
            if (mKeep) wbase = atomicAdd(&sCnt[0], (uint32_t)__popcll(mKeep));
        }
        wbase = __shfl(wbase, 0);
        if (keep) {
            const uint32_t rank = (uint32_t)__popcll(mKeep & ((1ull << lane) - 1ull));
            sCand[wbase + rank] = pack_cand(x0 + ox, y0 + oy, s);
        }
    }
    __syncthreads();

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
                      uint32_t* counters)
{
    dim3 block(256);
    dim3 grid(frames, totalTiles);
    hipLaunchKernelGGL(fast_blur_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch,
                       gray0Aligned4, ws, cand, counters);
}

}  // namespace orbfe
