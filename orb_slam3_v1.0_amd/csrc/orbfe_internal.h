// orbfe_internal.h -- shared between the host orchestration and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbfe.h"

// Synchronous copies / fills that stay OFF the NULL stream.  A null-stream operation (hipMemcpy, hipMemset) waits for every
// other stream of the device; if another thread is capturing one of them into a graph (a per-frame chain on another handle,
// or the application's own graphs), HIP fails the copy AND invalidates that capture -- seen with the tracking thread capturing
// orbfe_track_frame while the mapping thread uploaded a key frame (tests/test_two_threads_gpu.py).  These run on the
// caller's own (non-blocking) stream and wait for it alone.
inline hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t memset_sync(void* dst, int value, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemsetAsync(dst, value, bytes, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}

namespace orbfe {

constexpr int kMaxLevels = ORBFE_MAX_LEVELS;
constexpr int kEdge = 5;          // EDGE_THRESHOLD, src/ORBextractor.cc:80
constexpr int kHalfPatch = 15;    // HALF_PATCH_SIZE, :79
constexpr int kPatch = 31;        // PATCH_SIZE, :78
constexpr int kPitchAlign = 64;   // bytes; rows of the device pyramid start 64-B aligned

// candidate word: score << 24 | y << 12 | x  (level dims <= 4095)
constexpr int kCoordBits = 12;
constexpr uint32_t kCoordMask = (1u << kCoordBits) - 1;
__host__ __device__ inline uint32_t pack_cand(int x, int y, int score)
{
    return ((uint32_t)score << 24) | ((uint32_t)y << kCoordBits) | (uint32_t)x;
}
__host__ __device__ inline int cand_x(uint32_t c) { return (int)(c & kCoordMask); }
__host__ __device__ inline int cand_y(uint32_t c) { return (int)((c >> kCoordBits) & kCoordMask); }
__host__ __device__ inline int cand_score(uint32_t c) { return (int)(c >> 24); }
__host__ __device__ inline uint32_t cand_key(uint32_t c) { return c & 0x00FFFFFFu; }  // raster key (y, x)

// per-(frame, level) counter block (u32 words)
enum Counter {
    kCntCand = 0,     // NMS survivors of the low-threshold pass appended to cand[]
    kCntHigh = 1,     // ... of which score >= iniThFAST
    kCntPreLow = 2,   // pre-NMS corners at minThFAST
    kCntPreHigh = 3,  // pre-NMS corners at iniThFAST
    kCntKp = 4,       // keypoints kept by the quadtree for this level
    kCntStatus = 5,   // device-side guard flags (0 == fine)
    kCntWords = 8
};

enum DeviceFlag : uint32_t {
    kFlagNodeOverflow = 1u,   // quadtree node table too small
    kFlagRoundLimit = 2u,     // quadtree did not terminate within the round guard
    kFlagCandOverflow = 4u,   // candidate array full (cannot happen: cap == ceil(w/2)*ceil(h/2))
};

struct LevelDesc {
    int w, h, pitch;          // pitch of the device-side level image (bytes)
    int nFeatures;            // mnFeaturesPerLevel[l]
    int nIni;                 // round(w / h)
    float hX;                 // w / nIni
    int nodeCap;              // max(nFeatures + 3, 4 * nIni)
    int candCap;              // ceil(w/2) * ceil(h/2)
    int tilesX, tilesY;       // FAST tiling
    int tileBase;             // first FAST tile index of this level (prefix over levels)
    int kpBase;               // first slot of this level in the per-frame level-keypoint array
    float invScale;           // mvInvScaleFactor[l]
    int scaledPatch;          // (int)(31 * invScale)
    uint32_t pad_;
    size_t imgOff;            // byte offset of frame 0's unblurred level image in the workspace
    size_t imgFrameStride;    // bytes between consecutive frames of this level
    size_t blurOff, blurFrameStride;
    size_t candOff;           // u32 index of frame 0's candidate array (stride candCap)
    size_t xtabOff, ytabOff;  // u32 index of the resize tables of this level (l >= 1)
};

struct PipelineDesc {
    int nLevels;
    int nFast;                // nFastFeatures
    int iniTh, minTh;
    int kpCapFrame;           // sum of nodeCap == orbfe_max_keypoints
    int totalTiles;           // FAST tiles per frame over all levels
    int tileBaseTab[8];       // lv[0..7].tileBase in one s_load_dwordx8 (INT_MAX beyond nLevels)
    LevelDesc lv[kMaxLevels];
};

}  // namespace orbfe
