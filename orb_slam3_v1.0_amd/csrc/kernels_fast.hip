// kernels_fast.hip -- FAST-9/16 segment test + corner score + 3x3 non-max suppression, gfx950.
//
// Replaces GpuFast::detect (src/cuda/Fast_gpu.cu:354-395): tileCalcKeypoints_kernel (:269-287,
// isKeyPoint2 :222-267, calcMask :67-182, cornerScore :193-216) and nonmaxSuppression (:289-319),
// and folds the two-threshold retry of ComputeKeyPointsOctTree (src/ORBextractor.cc:449-482) into
// ONE pass over every pyramid level:
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
// Tile: 64 x 32 pixels per 256-thread block, image tile (+4 halo: 3 ring + 1 NMS) staged in LDS
// with coalesced dword loads.  Candidate order in HBM is not deterministic (one atomicAdd per
// block reserves the slots) -- every consumer is order-independent: it uses the raster key (y, x)
// carried in the word (SPEC DECISION S2b).
// Algorithmic bytes: 1 byte read per pixel (+4 bytes per surviving candidate).
#include "launch.h"

namespace orbfe {

constexpr int kFastTW = 64, kFastTH = 32;
constexpr int kImgW = kFastTW + 8, kImgH = kFastTH + 8;   // 72 x 40 staged pixels
constexpr int kScW = kFastTW + 2, kScH = kFastTH + 2;     // 66 x 34 scores
constexpr int kScPitch = 68;
constexpr int kMaxTileCand = (kFastTW / 2) * (kFastTH / 2);  // strict 8-neighbour maxima: <= 1 per 2x2

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

// FAST score of the pixel at LDS position (r, c) of the staged tile; 0 if not a corner at `th`.
__device__ __forceinline__ int fast_score_lds(const uint8_t (*img)[kImgW], int r, int c, int th)
{
    const int v = img[r][c];
    // compass points: ring bits 0 (+3,0), 4 (0,+3), 8 (-3,0), 12 (0,-3).  A 9-arc always holds
    // two adjacent compass points, so fewer than two bright (or dark) ones => not a corner.
    const int d0 = img[r + 3][c] - v, d4 = img[r][c + 3] - v, d8 = img[r - 3][c] - v, d12 = img[r][c - 3] - v;
    const int nb = (d0 > th) + (d4 > th) + (d8 > th) + (d12 > th);
    const int nd = (d0 < -th) + (d4 < -th) + (d8 < -th) + (d12 < -th);
    if (nb < 2 && nd < 2) return 0;
    int d[16];
    d[0] = d0; d[4] = d4; d[8] = d8; d[12] = d12;
    d[1] = img[r + 3][c + 1] - v;  d[2] = img[r + 2][c + 2] - v;  d[3] = img[r + 1][c + 3] - v;
    d[5] = img[r - 1][c + 3] - v;  d[6] = img[r - 2][c + 2] - v;  d[7] = img[r - 3][c + 1] - v;
    d[9] = img[r - 3][c - 1] - v;  d[10] = img[r - 2][c - 2] - v; d[11] = img[r - 1][c - 3] - v;
    d[13] = img[r + 1][c - 3] - v; d[14] = img[r + 2][c - 2] - v; d[15] = img[r + 3][c - 1] - v;
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mb |= (uint32_t)(d[k] > th) << k;
        md |= (uint32_t)(d[k] < -th) << k;
    }
    if (!(arc9(mb) || arc9(md))) return 0;
    // largest t such that some 9-arc has all |diff| > t  ==  max-min over arcs, minus 1
    int nd_[16];
#pragma unroll
    for (int k = 0; k < 16; k++) nd_[k] = -d[k];
    const int sb = arc_max_min(d);
    const int sd = arc_max_min(nd_);
    return max(sb, sd) - 1;
}

__global__ __launch_bounds__(256) void fast_kernel(const PipelineDesc* __restrict__ P,
                                                   const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                   int gray0Pitch, int gray0Aligned4,
                                                   uint8_t* __restrict__ ws, uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint8_t sImg[kImgH][kImgW];
    __shared__ uint8_t sScore[kScH][kScPitch];
    __shared__ uint32_t sCand[kMaxTileCand];
    __shared__ uint32_t sCnt[4];  // tile: survivors, high survivors, pre-NMS low, pre-NMS high
    __shared__ uint32_t sBase;

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
    if (tid < 4) sCnt[tid] = 0;

    // ---- stage the 72 x 40 tile (origin x0-4, y0-4); out-of-image bytes read as 0 ----
    for (int e = tid; e < kImgH * (kImgW / 4); e += 256) {
        const int r = e / (kImgW / 4);
        const int c4 = e - r * (kImgW / 4);
        const int gy = y0 - 4 + r;
        const int gx = x0 - 4 + 4 * c4;
        uint32_t wv = 0;
        if (gy >= 0 && gy < h) {
            const uint8_t* row = src + (size_t)gy * spitch;
            if (aligned && gx >= 0 && gx + 3 < w) {
                wv = *reinterpret_cast<const uint32_t*>(row + gx);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int xx = gx + i;
                    if (xx >= 0 && xx < w) wv |= (uint32_t)row[xx] << (8 * i);
                }
            }
        }
        *reinterpret_cast<uint32_t*>(&sImg[r][4 * c4]) = wv;
    }
    __syncthreads();

    // ---- scores for the tile + 1-px halo; tested region 6 <= x <= w-6, 6 <= y <= h-6
    //      (Fast_gpu.cu:275,365-368: strict compares against border 5 and dim-5) ----
    for (int e = tid; e < kScH * kScW; e += 256) {
        const int sy = e / kScW;
        const int sx = e - sy * kScW;
        const int px = x0 - 1 + sx, py = y0 - 1 + sy;
        int sc = 0;
        if (px > kEdge && px < w - kEdge && py > kEdge && py < h - kEdge)
            sc = fast_score_lds(sImg, sy + 3, sx + 3, minTh);
        sScore[sy][sx] = (uint8_t)sc;
    }
    __syncthreads();

    // ---- NMS (strictly greater than all 8 neighbours, Fast_gpu.cu:300-310) + tile compaction ----
    const int lane = tid & 63;
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
        const unsigned long long mPreHi = __ballot(s > 0 && hi);
        const unsigned long long mKeep = __ballot(keep);
        const unsigned long long mKeepHi = __ballot(keep && hi);
        uint32_t wbase = 0;
        if (lane == 0) {
            if (mPre) atomicAdd(&sCnt[2], (uint32_t)__popcll(mPre));
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

void launch_fast(hipStream_t s, int frames, int totalTiles, const PipelineDesc* dP, const uint8_t* gray0,
                 size_t gray0FrameStride, int gray0Pitch, int gray0Aligned4, uint8_t* ws, uint32_t* cand,
                 uint32_t* counters)
{
    dim3 block(256);
    dim3 grid(frames, totalTiles);
    hipLaunchKernelGGL(fast_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch,
                       gray0Aligned4, ws, cand, counters);
}

}  // namespace orbfe
