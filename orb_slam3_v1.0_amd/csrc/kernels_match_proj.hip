// kernels_match_proj.hip -- ORBmatcher::SearchByProjection(Frame, MapPoints, ...) on gfx950.
//
// Replaces src/ORBmatcher.cc:31-123 (+ RadiusByViewingCos :125-131) and the Frame helpers it leans
// on: AssignFeaturesToGrid / PosInGrid (src/Frame.cc:157-176,470-480) and GetFeaturesInArea
// (:404-468).  Distances are __popcll over 4 x 64-bit XOR words (== DescriptorDistance
// :1375-1391).  No MFMA: the path is bitwise/integer.
//
// Exactness of the order-dependent parts:
//  * The sequential "best / second best" scan (:92-104) returns the two smallest candidates under
//    the total order (distance, visit position); the visit position of GetFeaturesInArea is (cell x,
//    cell y, keypoint index).  That key is packed into one 64-bit word per candidate.
//  * The function is greedy: a keypoint already holding a map point with observations is skipped
//    (:77-79), including points written earlier in the SAME loop.  Let claim[idx] = smallest index
//    of an accepted map point (with observations) whose best match is idx.  Evaluating every map
//    point i with "idx is free iff claim[idx] >= i", rebuilding claim[] from the results and
//    repeating until nothing changes reaches a fixed point, and by induction over i every fixed
//    point equals the sequential result.  Map points are resolved in chunks of 1024 in index order,
//    each chunk iterated to its own fixed point while all earlier chunks are already final.
//  * Only the K smallest keys of a map point are kept (sorted).  A sweep takes the first two that
//    are still free; every candidate that was not stored is larger than all stored ones, so this is
//    exact whenever two free entries are found or the list was not truncated.  When the list runs dry
//    the verdict is often already implied by the last stored distance; otherwise the map point is
//    rescanned exactly (one wave, LDS-resident frame, only the index range of its two levels).
//
// Pipeline per call (B frames), all asynchronous on one stream, no host round trip:
//   prep (grid cell per keypoint) -> top-K candidate keys per map point (thread per map point,
//   keypoints + descriptors staged in LDS) -> ONE persistent block per frame resolves the claims
//   (claim table in LDS) and writes the final matches.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "match_common.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

constexpr int kClaimFree = 0x7fffffff;
constexpr int kResolveThreads = 1024;
constexpr int kTopK = 16;         // stored candidates per map point
constexpr int kCandChunk = 512;   // keypoints staged in LDS per pass (48 B each -> 24 KB, 6 blocks per CU)

struct GridDesc {
    int cols, rows;
    float minX, minY, invW, invH;
};

struct ProjArgs {
    int B, M, kpStride;           // frames, map points per frame, keypoint stride per frame
    GridDesc g;
    float th, thFar, nnRatio;
    int farPoints, bFactor;
    const orbfe_keypoint* kp;     // [B][kpStride]
    const uint8_t* desc;          // [B][kpStride][32]
    const int* nKp;               // [B]
    const orbfe_map_point* mps;   // [B][M]
    const uint8_t* mpDesc;        // [B][M][32]
    const int* initObs;           // [B][kpStride] or null
    const float* scaleFactors;    // [nLevels]
    int nLevels;
    // scratch
    int* cellXY;                  // [B][kpStride] : cx | cy << 16, or -1
    int* cnt;                     // [B][M] number of candidates (dist < 256) per map point
    unsigned long long* topk;     // [B][kTopK][M] sorted smallest keys (entry-major: coalesced per sweep)
    int* claimG;                  // [B][kpStride] fallback claim table (n > kLdsClaims)
    int* perm;                    // [B][M] map points ordered by pyramid level (work assignment of the top-K pass)
    int* dbg;                     // [B][4] diagnostics: sweeps, cooperative rescans, chunks, -
    int* lvlStart;                // [B][34] first keypoint index per level (level-major input), [33] = sorted flag
    int* matchOut;                // [B][kpStride]
    int* nMatches;                // [B]
};

// per map point: search window of GetFeaturesInArea (src/Frame.cc:413-435) + validity (:40-47)
struct MpWindow {
    bool valid;
    float x, y, r;
    int minCX, maxCX, minCY, maxCY, minLevel, maxLevel;
};

__device__ __forceinline__ MpWindow mp_window(const ProjArgs& A, const orbfe_map_point& mp)
{
    MpWindow w;
    w.valid = mp.in_view && !(A.farPoints && mp.track_depth > A.thFar) && !mp.bad;
    const int lvl = w.valid ? min(max(mp.level, 0), A.nLevels - 1) : 0;  // the host API rejects out-of-range levels
    float r = mp.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :125-131
    if (A.bFactor) r = r * A.th;
    r = r * A.scaleFactors[lvl];
    w.r = r;
    w.x = mp.proj_x;
    w.y = mp.proj_y;
    float t;
    t = w.x - A.g.minX; t = t - r; t = t * A.g.invW;
    w.minCX = max(0, (int)floorf(t));
    t = w.x - A.g.minX; t = t + r; t = t * A.g.invW;
    w.maxCX = min(A.g.cols - 1, (int)ceilf(t));
    t = w.y - A.g.minY; t = t - r; t = t * A.g.invH;
    w.minCY = max(0, (int)floorf(t));
    t = w.y - A.g.minY; t = t + r; t = t * A.g.invH;
    w.maxCY = min(A.g.rows - 1, (int)ceilf(t));
    if (w.minCX >= A.g.cols || w.maxCX < 0 || w.minCY >= A.g.rows || w.maxCY < 0) w.valid = false;
    w.minLevel = lvl - 1;
    w.maxLevel = lvl;
    return w;
}

// candidate test of GetFeaturesInArea for a keypoint in grid cell `cell` (cx | cy << 16, -1 = none)
__device__ __forceinline__ bool in_window(const MpWindow& w, int cell, float kx, float ky, int oct)
{
    if (cell < 0) return false;
    const int cx = cell & 0xffff, cy = cell >> 16;
    if (cx < w.minCX || cx > w.maxCX || cy < w.minCY || cy > w.maxCY) return false;
    const bool checkLevels = (w.minLevel > 0) || (w.maxLevel >= 0);  // src/Frame.cc:437
    if (checkLevels && (oct < w.minLevel || (w.maxLevel >= 0 && oct > w.maxLevel))) return false;
    const float dx = kx - w.x, dy = ky - w.y;
    return fabsf(dx) < w.r && fabsf(dy) < w.r;  // src/Frame.cc:461
}

// (distance, cell x, cell y, index): total order == visit order of the reference
__device__ __forceinline__ unsigned long long make_key(int dist, int cell, int idx)
{
    return ((unsigned long long)dist << 52) | ((unsigned long long)(cell & 0xffff) << 36) |
           ((unsigned long long)(cell >> 16) << 20) | (unsigned long long)idx;
}

// Frame::PosInGrid (src/Frame.cc:470-480): round(), only the LINEAR index is validated, so a
// keypoint with posX == cols lands in column 0 of the next row.
__global__ void proj_prep_kernel(ProjArgs A)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = A.nKp[f];
    if (i >= A.kpStride) return;
    int cell = -1;
    if (i < n) {
        const orbfe_keypoint& k = A.kp[(size_t)f * A.kpStride + i];
        float px = k.x - A.g.minX;
        px = px * A.g.invW;
        float py = k.y - A.g.minY;
        py = py * A.g.invH;
        const int posX = (int)roundf(px), posY = (int)roundf(py);
        const int lin = posY * A.g.cols + posX;
        if (lin >= 0 && lin < A.g.cols * A.g.rows) cell = (lin % A.g.cols) | ((lin / A.g.cols) << 16);
    }
    A.cellXY[(size_t)f * A.kpStride + i] = cell;
    A.matchOut[(size_t)f * A.kpStride + i] = -1;
}

// Counting sort of a frame's map points by (pyramid level, 8 x 6 spatial tile); the last bucket
// holds the invalid ones.  Only the WORK ASSIGNMENT of the top-K pass uses this order: the 64 lanes
// of a wave then search the same levels in overlapping windows, so a keypoint is relevant either
// for most lanes or for none and the whole wave skips it.  Results stay indexed by the original
// map point order.
constexpr int kTilesX = 8, kTilesY = 6;
constexpr int kSortBuckets = 32 * kTilesX * kTilesY + 1;

__device__ __forceinline__ int sort_bucket(const ProjArgs& A, const orbfe_map_point& mp)
{
    const bool valid = mp.in_view && !(A.farPoints && mp.track_depth > A.thFar) && !mp.bad;
    if (!valid) return kSortBuckets - 1;
    const int lvl = min(max(mp.level, 0), 31);
    const int cx = (int)((mp.proj_x - A.g.minX) * A.g.invW), cy = (int)((mp.proj_y - A.g.minY) * A.g.invH);
    const int tx = min(max(cx * kTilesX / A.g.cols, 0), kTilesX - 1);
    const int ty = min(max(cy * kTilesY / A.g.rows, 0), kTilesY - 1);
    return (lvl * kTilesY + ty) * kTilesX + tx;
}

__global__ __launch_bounds__(1024) void proj_sort_kernel(ProjArgs A)
{
    __shared__ int sHist[kSortBuckets];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    for (int b = tid; b < kSortBuckets; b += 1024) sHist[b] = 0;
    __syncthreads();
    const orbfe_map_point* mps = A.mps + (size_t)f * A.M;
    for (int i = tid; i < A.M; i += 1024) atomicAdd(&sHist[sort_bucket(A, mps[i])], 1);
    __syncthreads();
    if (tid == 0) {  // exclusive prefix in place (1537 entries, once per frame)
        int acc = 0;
        for (int b = 0; b < kSortBuckets; b++) {
            const int c = sHist[b];
            sHist[b] = acc;
            acc += c;
        }
    }
    __syncthreads();
    for (int i = tid; i < A.M; i += 1024) {
        const int pos = atomicAdd(&sHist[sort_bucket(A, mps[i])], 1);
        A.perm[(size_t)f * A.M + pos] = i;
    }

    // Keypoint index range per pyramid level.  The extractor emits keypoints level-major
    // (src/ORBextractor.cc:499-500); when that holds, lvlStart[l] = first index with octave >= l and a
    // block of the top-K pass only walks the levels its map points can match.  For any other order
    // lvlStart[33] = 0 ("not sorted") and the pass walks everything.
    __shared__ int sStart[34];
    __syncthreads();
    const int n = A.nKp[f];
    const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
    if (tid < 33) sStart[tid] = n;
    if (tid == 33) sStart[33] = 1;
    __syncthreads();
    for (int j = tid; j < n; j += 1024) {
        const int o = min(max(kp[j].octave, 0), 32);
        atomicMin(&sStart[o], j);
        if (j + 1 < n && kp[j + 1].octave < kp[j].octave) sStart[33] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        for (int l2 = 31; l2 >= 0; l2--) sStart[l2] = min(sStart[l2], sStart[l2 + 1]);
        for (int l2 = 0; l2 < 34; l2++) A.lvlStart[f * 34 + l2] = sStart[l2];
    }
}

// Thread per map point; a block owns 256 map points of one frame and stages the frame's keypoints
// (cell, x, y, octave, descriptor) in LDS in chunks, so the inner loop never waits on global memory.
// Every lane looks at the same keypoint at the same time -> LDS broadcast reads.
__global__ __launch_bounds__(256) void proj_topk_kernel(ProjArgs A)
{
    __shared__ int4 sKp[kCandChunk];  // {cell, octave, x bits, y bits}: one ds_read_b128 per keypoint
    __shared__ unsigned long long sDesc[kCandChunk][4];
    const int f = blockIdx.y;
    const int slotIdx = blockIdx.x * 256 + threadIdx.x;
    const bool live = slotIdx < A.M;
    const int i = live ? A.perm[(size_t)f * A.M + slotIdx] : 0;  // level-coherent waves
    const int n = A.nKp[f];
    const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
    const int* cellXY = A.cellXY + (size_t)f * A.kpStride;
    const unsigned long long* desc = reinterpret_cast<const unsigned long long*>(A.desc + (size_t)f * A.kpStride * 32);
    MpWindow w;
    w.valid = false;
    w.minLevel = w.maxLevel = 0;
    if (live) w = mp_window(A, A.mps[(size_t)f * A.M + i]);
    unsigned long long d4[4] = {0, 0, 0, 0};
    if (live && w.valid) {
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * A.M + i) * 32);
        d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
    }
    unsigned long long keys[kTopK];
#pragma unroll
    for (int t = 0; t < kTopK; t++) keys[t] = kKeyNone;
    int total = 0;
    // keypoint index range this block has to look at: the levels [min(lvl)-1, max(lvl)] of its map points
    __shared__ int sLvlLo, sLvlHi;
    if (threadIdx.x == 0) { sLvlLo = 64; sLvlHi = -1; }
    __syncthreads();
    if (w.valid) {
        atomicMin(&sLvlLo, w.minLevel);
        atomicMax(&sLvlHi, w.maxLevel);
    }
    __syncthreads();
    int jlo = 0, jhi = n;
    {
        const int* ls = A.lvlStart + f * 34;
        if (sLvlHi < 0) jhi = 0;  // no valid map point in this block
        else if (ls[33]) {        // level-major keypoints
            jlo = ls[min(max(sLvlLo, 0), 32)];
            jhi = ls[min(sLvlHi + 1, 32)];
        }
    }
    for (int base = jlo; base < jhi; base += kCandChunk) {
        const int m = min(kCandChunk, jhi - base);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += 256) {
            const orbfe_keypoint k = kp[base + j];
            sKp[j] = make_int4(cellXY[base + j], k.octave, __float_as_int(k.x), __float_as_int(k.y));
        }
        for (int j = threadIdx.x; j < m * 4; j += 256) sDesc[0][j] = desc[(size_t)base * 4 + j];
        __syncthreads();
        if (w.valid) {
            const bool checkLevels = (w.minLevel > 0) || (w.maxLevel >= 0);  // src/Frame.cc:437
            for (int j = 0; j < m; j++) {
                const int4 kq = sKp[j];  // broadcast read
                const int cell = kq.x, oct = kq.y;
                // level filter first: waves are level-coherent, so most keypoints fail for all 64 lanes
                if (checkLevels && (oct < w.minLevel || (w.maxLevel >= 0 && oct > w.maxLevel))) continue;
                if (!in_window(w, cell, __int_as_float(kq.z), __int_as_float(kq.w), oct)) continue;
                const int dist = __popcll(sDesc[j][0] ^ d4[0]) + __popcll(sDesc[j][1] ^ d4[1]) +
                                 __popcll(sDesc[j][2] ^ d4[2]) + __popcll(sDesc[j][3] ^ d4[3]);
                if (dist >= 256) continue;  // can enter neither slot (the reference's bests start at 256)
                total++;
                unsigned long long key = make_key(dist, cell, base + j);
                if (key < keys[kTopK - 1]) {  // sorted insert
#pragma unroll
                    for (int t = 0; t < kTopK; t++) {
                        const unsigned long long lo = key < keys[t] ? key : keys[t];
                        key = key < keys[t] ? keys[t] : key;
                        keys[t] = lo;
                    }
                }
            }
        }
    }
    if (live) {
        A.cnt[(size_t)f * A.M + i] = total;
#pragma unroll
        for (int t = 0; t < kTopK; t++) A.topk[((size_t)f * kTopK + t) * A.M + i] = keys[t];
    }
}

// One persistent block per frame, one THREAD per map point, chunks of 1024 map points in index order.
// The frame's keypoints (cell, octave, x, y) and descriptors are staged in LDS once (frames up to
// kResN keypoints), so sweeps and the exact rescans of starved map points never touch global memory.
constexpr int kResN = 2048;

struct ResolveLds {
    int claim[kResN];
    int4 kp[kResN];                       // {cell, octave, x bits, y bits}
    unsigned long long desc[kResN][4];
};

// exact rescan for a map point whose stored top-K ran dry: one WAVE scans all keypoints of the
// frame (lane-strided) and reduces the two smallest free keys; all lanes get the result.
// The claim test goes first: a starved map point sits in a region where nearly everything is taken.
template <bool LDS>
__device__ __forceinline__ void full_scan_top2_wave(const ProjArgs& A, int f, int i, const int* claim,
                                                    const ResolveLds* S, int lane, unsigned long long& k1,
                                                    unsigned long long& k2)
{
    const MpWindow w = mp_window(A, A.mps[(size_t)f * A.M + i]);
    k1 = kKeyNone;
    k2 = kKeyNone;
    if (w.valid) {  // wave-uniform
        const int n = A.nKp[f];
        const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
        const int* cellXY = A.cellXY + (size_t)f * A.kpStride;
        const uint8_t* desc = A.desc + (size_t)f * A.kpStride * 32;
        unsigned long long d4[4];
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * A.M + i) * 32);
        d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
        // level-major keypoints: only the index range of levels [minLevel, maxLevel] can pass the filter
        int jlo = 0, jhi = n;
        {
            const int* ls = A.lvlStart + f * 34;
            if (ls[33]) {
                jlo = ls[min(max(w.minLevel, 0), 32)];
                jhi = ls[min(max(w.maxLevel, -1) + 1, 32)];
            }
        }
        for (int idx = jlo + lane; idx < jhi; idx += 64) {
            if (claim[idx] < i) continue;
            int cell, oct, dist;
            float kx, ky;
            if (LDS) {
                const int4 q = S->kp[idx];
                cell = q.x; oct = q.y; kx = __int_as_float(q.z); ky = __int_as_float(q.w);
            } else {
                cell = cellXY[idx];
                const orbfe_keypoint k = kp[idx];
                oct = k.octave; kx = k.x; ky = k.y;
            }
            if (!in_window(w, cell, kx, ky, oct)) continue;
            if (LDS)
                dist = __popcll(S->desc[idx][0] ^ d4[0]) + __popcll(S->desc[idx][1] ^ d4[1]) +
                       __popcll(S->desc[idx][2] ^ d4[2]) + __popcll(S->desc[idx][3] ^ d4[3]);
            else
                dist = hamming256(reinterpret_cast<const uint2*>(desc + (size_t)idx * 32), d4);
            if (dist >= 256) continue;
            const unsigned long long key = make_key(dist, cell, idx);
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
        }
    }
    wave_top2(k1, k2);
}

template <bool LDS>
__global__ __launch_bounds__(kResolveThreads) void proj_resolve_kernel(ProjArgs A)
{
    __shared__ ResolveLds S;
    __shared__ int sChanged;
    __shared__ int sCount;
    __shared__ int sFbCount;                             // starved map points of the current sweep
    __shared__ int sFbMp[kResolveThreads];
    __shared__ unsigned long long sFbK1[kResolveThreads], sFbK2[kResolveThreads];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = A.nKp[f];
    const int M = A.M;
    int* claim = LDS ? S.claim : A.claimG + (size_t)f * A.kpStride;
    const orbfe_map_point* mps = A.mps + (size_t)f * M;
    const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
    const int* initObs = A.initObs ? A.initObs + (size_t)f * A.kpStride : nullptr;

    // claim[idx] = -1 if the slot holds a map point with observations on entry (:77-79); later the
    // smallest accepted map point (with observations) whose best match is idx
    for (int i = tid; i < n; i += kResolveThreads) claim[i] = (initObs && initObs[i] > 0) ? -1 : kClaimFree;
    if (LDS) {
        const int* cellXY = A.cellXY + (size_t)f * A.kpStride;
        const unsigned long long* desc = reinterpret_cast<const unsigned long long*>(A.desc + (size_t)f * A.kpStride * 32);
        for (int j = tid; j < n; j += kResolveThreads) {
            const orbfe_keypoint k = kp[j];
            S.kp[j] = make_int4(cellXY[j], k.octave, __float_as_int(k.x), __float_as_int(k.y));
        }
        for (int j = tid; j < n * 4; j += kResolveThreads) S.desc[0][j] = desc[j];
    }
    if (tid == 0) sCount = 0;
    int nAccepted = 0;

    for (int chunk = 0; chunk < M; chunk += kResolveThreads) {
        const int i = chunk + tid;
        const bool live = i < M;
        int c = 0, obs = 0;
        unsigned long long keys[kTopK];
#pragma unroll
        for (int t = 0; t < kTopK; t++) keys[t] = kKeyNone;
        if (live) {
            c = A.cnt[(size_t)f * M + i];
            obs = mps[i].observations;
            if (c > 0) {
#pragma unroll
                for (int t = 0; t < kTopK; t++) keys[t] = A.topk[((size_t)f * kTopK + t) * M + i];
            }
        }
        int res = -1;
        for (int iter = 0; iter <= kResolveThreads + 1; iter++) {
            __syncthreads();
            // drop the tentative claims of this chunk (entries >= chunk), keep earlier chunks' final ones
            for (int k = tid; k < n; k += kResolveThreads)
                if (claim[k] >= chunk) claim[k] = kClaimFree;
            if (tid == 0) { sChanged = 0; sFbCount = 0; }
            __syncthreads();
            if (res >= 0 && obs > 0) atomicMin(&claim[res], i);
            __syncthreads();
            int result = -1;
            unsigned long long k1 = kKeyNone, k2 = kKeyNone;
            int slot = -1;
            if (c > 0) {
#pragma unroll
                for (int t = 0; t < kTopK; t++) {  // ascending: the first two free entries are the answer
                    const unsigned long long key = keys[t];
                    if (key != kKeyNone && claim[(int)(key & 0xFFFFF)] >= i) {
                        if (k1 == kKeyNone) k1 = key;
                        else if (k2 == kKeyNone) k2 = key;
                    }
                }
                if (k2 == kKeyNone && c > kTopK) {
                    // The stored list ran dry.  Every candidate that was not stored has a key above
                    // keys[K-1], i.e. a distance >= dK.  Two cases are decided without looking at them:
                    //  - nothing free and dK > TH_HIGH: the best free candidate fails :108 -> no match;
                    //  - one free entry with best <= nnRatio * dK: the ratio test :110 cannot reject
                    //    (the float product is monotone in the unknown second distance >= dK).
                    const int dK = (int)(keys[kTopK - 1] >> 52);
                    bool decided = false;
                    if (k1 == kKeyNone) decided = dK > ORBFE_TH_HIGH;
                    else {
                        const int bd = (int)(k1 >> 52);
                        decided = bd > ORBFE_TH_HIGH || (A.nnRatio > 0.f && !((float)bd > A.nnRatio * (float)dK));
                        if (decided) k2 = keys[kTopK - 1];  // stand-in with distance dK: same verdict as the true second
                    }
                    if (!decided) {  // exact rescan, done cooperatively below
                        slot = atomicAdd(&sFbCount, 1);
                        sFbMp[slot] = i;
                        atomicAdd(&A.dbg[f * 4 + 1], 1);
                    }
                }
            }
            __syncthreads();
            {
                const int nFb = sFbCount;
                for (int q = tid >> 6; q < nFb; q += kResolveThreads / 64) {  // one wave per starved map point
                    unsigned long long a1, a2;
                    full_scan_top2_wave<LDS>(A, f, sFbMp[q], claim, &S, tid & 63, a1, a2);
                    if ((tid & 63) == 0) { sFbK1[q] = a1; sFbK2[q] = a2; }
                }
            }
            __syncthreads();
            if (slot >= 0) { k1 = sFbK1[slot]; k2 = sFbK2[slot]; }
            if (k1 != kKeyNone) {
                const int bestDist = (int)(k1 >> 52), bestIdx = (int)(k1 & 0xFFFFF);
                const int bestLevel = LDS ? S.kp[bestIdx].y : kp[bestIdx].octave;
                int bestDist2 = 256, bestLevel2 = -1;
                if (k2 != kKeyNone) {
                    bestDist2 = (int)(k2 >> 52);
                    const int i2 = (int)(k2 & 0xFFFFF);
                    bestLevel2 = LDS ? S.kp[i2].y : kp[i2].octave;
                }
                if (bestDist <= ORBFE_TH_HIGH) {  // :108-117
                    const bool reject = bestLevel == bestLevel2 && (float)bestDist > A.nnRatio * (float)bestDist2;
                    if (!reject) result = bestIdx;
                }
            }
            if (result != res) {
                res = result;
                sChanged = 1;
            }
            __syncthreads();
            if (tid == 0) atomicAdd(&A.dbg[f * 4 + 0], 1);
            if (!sChanged) break;
        }
        // F->mvpMapPoints[bestIdx] = pMP in map-point order: the last writer wins; nmatches counts accepts
        if (res >= 0) {
            atomicMax(&A.matchOut[(size_t)f * A.kpStride + res], i);
            nAccepted++;
        }
    }
    if (nAccepted) atomicAdd(&sCount, nAccepted);
    __syncthreads();
    if (tid == 0) A.nMatches[f] = sCount;
}

// carve the device scratch after whatever `sc` already holds; sets A's scratch pointers
int proj_setup(MatchScratch& m, ProjArgs& A, Carver sc, size_t hostNeed, std::string& err)
{
    const int B = A.B, M = A.M;
    const size_t oCell = sc.take((size_t)B * A.kpStride * sizeof(int));
    const size_t oCnt = sc.take((size_t)B * std::max(M, 1) * sizeof(int));
    const size_t oTopk = sc.take((size_t)B * kTopK * std::max(M, 1) * sizeof(unsigned long long));
    const size_t oClaim = sc.take((size_t)B * A.kpStride * sizeof(int));
    const size_t oPerm = sc.take((size_t)B * std::max(M, 1) * sizeof(int));
    const size_t oDbg = sc.take((size_t)B * 4 * sizeof(int));
    const size_t oLvl = sc.take((size_t)B * 34 * sizeof(int));
    int rc = ensure(m, sc.off, hostNeed + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    A.cellXY = reinterpret_cast<int*>(dp + oCell);
    A.cnt = reinterpret_cast<int*>(dp + oCnt);
    A.topk = reinterpret_cast<unsigned long long*>(dp + oTopk);
    A.claimG = reinterpret_cast<int*>(dp + oClaim);
    A.perm = reinterpret_cast<int*>(dp + oPerm);
    A.dbg = reinterpret_cast<int*>(dp + oDbg);
    A.lvlStart = reinterpret_cast<int*>(dp + oLvl);
    return ORBFE_OK;
}

int proj_launch(hipStream_t s, ProjArgs& A, std::string& err)
{
    const dim3 blk(256);
    hipLaunchKernelGGL(proj_prep_kernel, dim3((A.kpStride + 255) / 256, A.B), blk, 0, s, A);
    if (A.M == 0) {
        MCHK(hipMemsetAsync(A.nMatches, 0, (size_t)A.B * sizeof(int), s));
        return ORBFE_OK;
    }
    MCHK(hipMemsetAsync(A.dbg, 0, (size_t)A.B * 4 * sizeof(int), s));
    hipLaunchKernelGGL(proj_sort_kernel, dim3(A.B), dim3(1024), 0, s, A);
    hipLaunchKernelGGL(proj_topk_kernel, dim3((A.M + 255) / 256, A.B), blk, 0, s, A);
    if (A.kpStride <= kResN)  // nKp[f] <= kpStride: the whole frame fits the LDS image
        hipLaunchKernelGGL(proj_resolve_kernel<true>, dim3(A.B), dim3(kResolveThreads), 0, s, A);
    else
        hipLaunchKernelGGL(proj_resolve_kernel<false>, dim3(A.B), dim3(kResolveThreads), 0, s, A);
    MCHK(hipGetLastError());
    return ORBFE_OK;
}

}  // namespace

void match_scratch_free(MatchScratch& m)
{
    if (m.d) (void)hipFree(m.d);
    if (m.hpin) (void)hipHostFree(m.hpin);
    m = MatchScratch();
}

int match_projection_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F, int M, const orbfe_map_point* mps,
                         const uint8_t* mpDesc, const int* initObs, float th, int farPoints, float thFar,
                         float nnRatio, int* matchOut, int* nMatches, std::string& err)
{
    const int n = F->n;
    for (int i = 0; i < n; i++) matchOut[i] = -1;
    *nMatches = 0;
    if (n == 0 || M == 0) return ORBFE_OK;
    if (n >= (1 << 20) || F->grid_cols > 65535 || F->grid_rows > 32767 || F->n_levels < 1) return ORBFE_ERR_UNSUPPORTED;
    for (int i = 0; i < M; i++)
        if (mps[i].in_view && (mps[i].level < 0 || mps[i].level >= F->n_levels)) return ORBFE_ERR_INVALID_ARG;

    // one pinned staging block -> one H2D copy
    Carver in;
    const size_t oKp = in.take((size_t)n * sizeof(orbfe_keypoint));
    const size_t oDesc = in.take((size_t)n * 32);
    const size_t oMp = in.take((size_t)M * sizeof(orbfe_map_point));
    const size_t oMpDesc = in.take((size_t)M * 32);
    const size_t oObs = in.take((size_t)n * sizeof(int));
    const size_t oSf = in.take((size_t)F->n_levels * sizeof(float));
    const size_t oN = in.take(sizeof(int));
    const size_t oMatch = in.take((size_t)n * sizeof(int));
    const size_t oNM = in.take(sizeof(int));
    const size_t inBytes = in.off;
    const size_t hostNeed = inBytes + (size_t)n * sizeof(int) + 64;

    ProjArgs A{};
    A.B = 1; A.M = M; A.kpStride = n;
    A.g = GridDesc{F->grid_cols, F->grid_rows, F->min_x, F->min_y, F->grid_inv_w, F->grid_inv_h};
    A.th = th; A.thFar = thFar; A.nnRatio = nnRatio; A.farPoints = farPoints; A.bFactor = th != 1.0;
    A.nLevels = F->n_levels;
    int rc = proj_setup(m, A, in, hostNeed, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oKp, F->kp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(hp + oDesc, F->desc, (size_t)n * 32);
    memcpy(hp + oMp, mps, (size_t)M * sizeof(orbfe_map_point));
    memcpy(hp + oMpDesc, mpDesc, (size_t)M * 32);
    if (initObs) memcpy(hp + oObs, initObs, (size_t)n * sizeof(int));
    memcpy(hp + oSf, F->scale_factors, (size_t)F->n_levels * sizeof(float));
    memcpy(hp + oN, &n, sizeof(int));
    MCHK(hipMemcpyAsync(dp, hp, oMatch, hipMemcpyHostToDevice, s));
    A.kp = reinterpret_cast<const orbfe_keypoint*>(dp + oKp);
    A.desc = dp + oDesc;
    A.nKp = reinterpret_cast<const int*>(dp + oN);
    A.mps = reinterpret_cast<const orbfe_map_point*>(dp + oMp);
    A.mpDesc = dp + oMpDesc;
    A.initObs = initObs ? reinterpret_cast<const int*>(dp + oObs) : nullptr;
    A.scaleFactors = reinterpret_cast<const float*>(dp + oSf);
    A.matchOut = reinterpret_cast<int*>(dp + oMatch);
    A.nMatches = reinterpret_cast<int*>(dp + oNM);
    int* hNM = reinterpret_cast<int*>(hp + inBytes);
    int* hMatch = hNM + 2;
    rc = proj_launch(s, A, err);
    if (rc != ORBFE_OK) return rc;
    MCHK(hipMemcpyAsync(hMatch, A.matchOut, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipMemcpyAsync(hNM, A.nMatches, sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    if (getenv("ORBFE_DEBUG_MATCH")) {  // diagnostics only
        int dbg[4] = {0, 0, 0, 0};
        (void)hipMemcpy(dbg, A.dbg, sizeof dbg, hipMemcpyDeviceToHost);
        fprintf(stderr, "[orbfe] match_projection: n=%d M=%d sweeps=%d cooperative_rescans=%d\n", n, M, dbg[0], dbg[1]);
    }
    memcpy(matchOut, hMatch, (size_t)n * sizeof(int));
    *nMatches = *hNM;
    return ORBFE_OK;
}

int match_projection_batch_device(MatchScratch& m, hipStream_t s, int B, const orbfe_keypoint* dKp, const uint8_t* dDesc,
                                  const int* dN, int kpStride, int gridCols, int gridRows, float minX, float minY,
                                  float invW, float invH, const float* dScaleFactors, int nLevels, int M,
                                  const orbfe_map_point* dMps, const uint8_t* dMpDesc, const int* dInitObs, float th,
                                  int farPoints, float thFar, float nnRatio, int* dMatchOut, int* dNMatches,
                                  std::string& err)
{
    if (kpStride >= (1 << 20) || gridCols > 65535 || gridRows > 32767) return ORBFE_ERR_UNSUPPORTED;
    ProjArgs A{};
    A.B = B; A.M = M; A.kpStride = kpStride;
    A.g = GridDesc{gridCols, gridRows, minX, minY, invW, invH};
    A.th = th; A.thFar = thFar; A.nnRatio = nnRatio; A.farPoints = farPoints; A.bFactor = th != 1.0;
    A.kp = dKp; A.desc = dDesc; A.nKp = dN; A.mps = dMps; A.mpDesc = dMpDesc; A.initObs = dInitObs;
    A.scaleFactors = dScaleFactors; A.nLevels = nLevels;
    A.matchOut = dMatchOut;
    A.nMatches = dNMatches;
    int rc = proj_setup(m, A, Carver(), 64, err);
    if (rc != ORBFE_OK) return rc;
    return proj_launch(s, A, err);
}

}  // namespace orbfe
