// match_proj.h -- shared between kernels_match_proj.hip (SearchByProjection pipeline: grid, top-K, claims) and
// kernels_match_kf.hip (thread-per-map-point key-frame searches built on the same per-level cell tables).
#pragma once
#include <string>

#include "match_common.h"

namespace orbfe {
namespace proj {

constexpr int kClaimFree = 0x7fffffff;
constexpr int kResolveThreads = 1024;
constexpr int kTopK = 24;         // stored candidates per map point
constexpr uint32_t kKey32None = 0xffffffffu;
constexpr int kRankBits = 20;
constexpr uint32_t kRankMask = (1u << kRankBits) - 1u;
constexpr int kSortLds = 2048;    // frames up to this many keypoints are sorted in LDS
constexpr int kMaxCells = 1 << 22;          // grid cells (cols * rows)

struct GridDesc {
    int cols, rows;
    float minX, minY, invW, invH;
};

struct ProjArgs {
    int B, M, kpStride;           // frames, map points per frame, keypoint stride per frame
    GridDesc g;
    float th, thFar, nnRatio;
    int farPoints, bFactor;
    int mode;                     // 0: SearchByProjection(F, vpMapPoints, ...); 1: relocalisation overload (kModeReloc)
    int dCut;                     // candidates with distance >= dCut cannot change any verdict (see proj_dcut)
    const orbfe_keypoint* kp;     // [B][kpStride]
    const uint8_t* desc;          // [B][kpStride][32]
    const int* nKp;               // [B]
    const orbfe_map_point* mps;   // [B][M]
    const uint8_t* mpDesc;        // [B][M][32]
    const int* initObs;           // [B][kpStride] or null
    const float* scaleFactors;    // [nLevels]
    int nLevels;
    // scratch
    int sortCap;                  // pow2 >= kpStride (global sort path)
    unsigned long long* sortKeys; // [B][sortCap] (only frames with more than kSortLds keypoints)
    int* order;                   // [B][kpStride] rank -> keypoint index
    uint8_t* octByRank;           // [B][kpStride] rank -> octave (clamped to 0..31)
    int* rankOf;                  // [B][kpStride] keypoint index -> rank (global sort path only)
    int4* rec;                    // [B][kpStride] per storage slot: {rank, octave | cell y << 8, x bits, y bits}
    unsigned long long* descS;    // [B][kpStride][4] descriptors in storage order
    int* colStart;                // [B][tabLevels][cols + 1]: first storage slot of (level, grid column cx)
    int tabLevels;                // min(nLevels, 32)
    int* cnt;                     // [B][M] number of candidates (dist < 256) per map point
    uint32_t* topk;               // [B][M][kTopK] sorted smallest keys (contiguous per map point)
    int* claimG;                  // [B][kpStride] fallback claim table (frames that do not fit the LDS image)
    int* perm;                    // [B][M] map points ordered by (level, tile): work assignment of the top-K pass
    int* dbg;                     // [B][4] diagnostics: sweeps, cooperative rescans, -, -
    int* matchOut;                // [B][kpStride]
    int* nMatches;                // [B]
};

constexpr int kModeReloc = 1;  // radius th * scale, levels [l-1, l+1], no ratio test (src/ORBmatcher.cc:1250-1283)

// per map point: search window of GetFeaturesInArea (src/Frame.cc:413-435) + validity (:40-47)
struct MpWindow {
    bool valid;
    float x, y, r;
    int minCX, maxCX, minCY, maxCY, minLevel, maxLevel;
};

// cell range of GetFeaturesInArea (src/Frame.cc:413-435 == src/KeyFrame.cc:798-812) for a square of half-size r
__device__ __forceinline__ void window_cells(const GridDesc& g, MpWindow& w);

__device__ __forceinline__ MpWindow mp_window(const ProjArgs& A, const orbfe_map_point& mp)
{
    MpWindow w;
    w.valid = mp.in_view && !(A.farPoints && mp.track_depth > A.thFar) && !mp.bad;
    const int lvl = w.valid ? min(max(mp.level, 0), A.nLevels - 1) : 0;  // the host API rejects out-of-range levels
    float r;
    if (A.mode == kModeReloc) r = A.th * A.scaleFactors[lvl];  // :1253
    else {
        r = mp.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :125-131
        if (A.bFactor) r = r * A.th;
        r = r * A.scaleFactors[lvl];
    }
    w.r = r;
    w.x = mp.proj_x;
    w.y = mp.proj_y;
    window_cells(A.g, w);
    w.minLevel = lvl - 1;
    w.maxLevel = A.mode == kModeReloc ? min(lvl + 1, A.nLevels - 1) : lvl;  // :1255 (octaves never exceed nLevels - 1)
    return w;
}

__device__ __forceinline__ void window_cells(const GridDesc& g, MpWindow& w)
{
    const float r = w.r;
    float t;
    t = w.x - g.minX; t = t - r; t = t * g.invW;
    w.minCX = max(0, (int)floorf(t));
    t = w.x - g.minX; t = t + r; t = t * g.invW;
    w.maxCX = min(g.cols - 1, (int)ceilf(t));
    t = w.y - g.minY; t = t - r; t = t * g.invH;
    w.minCY = max(0, (int)floorf(t));
    t = w.y - g.minY; t = t + r; t = t * g.invH;
    w.maxCY = min(g.rows - 1, (int)ceilf(t));
    if (w.minCX >= g.cols || w.maxCX < 0 || w.minCY >= g.rows || w.maxCY < 0) w.valid = false;
}

// median of three (v_med3_u32)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

__device__ __forceinline__ uint32_t make_key32(int dist, int rank) { return ((uint32_t)dist << kRankBits) | (uint32_t)rank; }

// host side (kernels_match_proj.hip)
int proj_dcut(float nnRatio);
// carve the device scratch after whatever `sc` already holds; sets A's scratch pointers
int proj_setup(MatchScratch& m, ProjArgs& A, Carver sc, size_t hostNeed, std::string& err, size_t* endOff = nullptr);
// grid only (sortMps == false: cell per keypoint, visit-order sort, level-major storage, column-start tables) or grid +
// (level, tile) order of the map points
void proj_prepare_launch(hipStream_t s, const ProjArgs& A, bool sortMps);
// the whole SearchByProjection pipeline (prepare, top-K, claims) for A.B frames
int proj_launch(hipStream_t s, ProjArgs& A, std::string& err);

}  // namespace proj
}  // namespace orbfe
