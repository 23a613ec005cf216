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
//    cell y, keypoint index).  The keypoints of a frame are SORTED by that visit position once per call
//    (their position in the sorted order is the "rank"), so a candidate key is one 32-bit word
//    distance << 20 | rank.  The keypoints are STORED level-major, in visit order inside a level, with one
//    column-start table per level: the grid columns minCX..maxCX of level l are one contiguous slot range, and a
//    map point predicted at level L only walks levels L-1 and L (src/Frame.cc:437-452) -- two ranges and four
//    table loads per map point; the cell-row test runs on the record of every visited keypoint.
//  * The function is greedy: a keypoint already holding a map point with observations is skipped
//    (:77-79), including points written earlier in the SAME loop.  Let claim[rank] = smallest index
//    of an accepted map point (with observations) whose best match is that keypoint.  Evaluating every
//    map point i with "free iff claim >= i", rebuilding claim[] from the results and
//    repeating until nothing changes reaches a fixed point, and by induction over i every fixed
//    point equals the sequential result.  Map points are resolved in chunks of 1024 in index order,
//    each chunk iterated to its own fixed point while all earlier chunks are already final.
//  * Only the K smallest keys of a map point are kept (sorted).  A sweep takes the first two that
//    are still free; every candidate that was not stored is larger than all stored ones, so this is
//    exact whenever two free entries are found or the list was not truncated.  When the list runs dry
//    the verdict is often already implied by the last stored distance; otherwise the map point is
//    rescanned exactly (one wave over the rank range of its window columns, LDS-resident frame).
//
// Pipeline per call (B frames), all asynchronous on one stream, no host round trip:
//   grid (cell per keypoint, sort by visit position, column-start tables) -> map points ordered by
//   (level, tile) -> top-K candidate keys per map point (thread per map point) -> ONE persistent block
//   per frame resolves the claims (claim table in LDS) and writes the final matches.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device_math.h"
#include "match_common.h"

#include "match_proj.h"

#pragma clang fp contract(off)

namespace orbfe {

using namespace proj;

namespace {


// ---------------------------------------------------------------------------------------------
// grid: Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:157-176,470-480) + visit-order sort.
// PosInGrid rounds and validates only the LINEAR index, so a keypoint with posX == cols lands in
// column 0 of the next row.  One block per frame: 64-bit keys (visit cell v = cx * rows + cy) << 20 | index
// are bitonic-sorted (LDS for frames up to kSortLds keypoints, else in global scratch); keypoints outside
// the grid get v = cols*rows and sort to the end, where no window ever looks.
// ---------------------------------------------------------------------------------------------
template <bool LDS, class KeyFn>
__device__ __forceinline__ void block_sort_keys(unsigned long long* sKeys, unsigned long long* gKeys, int n, int P, int tid, KeyFn keyOf)
{
    for (int j = tid; j < P; j += 1024) {
        const unsigned long long key = j < n ? keyOf(j) : ~0ull;
        if constexpr (LDS) sKeys[j] = key;
        else gKeys[j] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1));
                const int hi = lo | j2;
                const bool up = (lo & k2) == 0;
                unsigned long long a, b;
                if constexpr (LDS) { a = sKeys[lo]; b = sKeys[hi]; }
                else { a = gKeys[lo]; b = gKeys[hi]; }
                if ((a > b) == up) {
                    if constexpr (LDS) { sKeys[lo] = b; sKeys[hi] = a; }
                    else { gKeys[lo] = b; gKeys[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// The same network for P <= 1024 with ONE key per thread held in a register: the exchanges at distance < 64 stay inside
// a wave (lane shuffles, no barrier); only distances >= 64 go through LDS, double-buffered in the two halves of sKeys
// (one barrier per pass).  For P = 1024 that is 10 barriers instead of 55 -- the sort is latency, not work.
// Leaves the sorted keys in sKeys[0..P) like block_sort_keys<true>.
template <class KeyFn>
__device__ __forceinline__ void block_sort_keys_reg(unsigned long long* sKeys, int n, int P, int tid, KeyFn keyOf)
{
    unsigned long long key = tid < n ? keyOf(tid) : ~0ull;
    int buf = 0;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            unsigned long long other;
            if (j2 >= 64) {
                unsigned long long* b = sKeys + buf * 1024;
                b[tid] = key;
                __syncthreads();
                other = b[tid ^ j2];
                buf ^= 1;
            } else {
                other = shfl_xor_u64(key, j2);
            }
            const bool up = (tid & k2) == 0;
            const bool lower = (tid & j2) == 0;
            const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    __syncthreads();  // the last LDS pass may still be read by others
    sKeys[tid] = key;
    __syncthreads();
}

constexpr int kCellBits = 22;  // kMaxCells

template <bool LDS>
__device__ __forceinline__ void proj_grid_body(const ProjArgs& A)
{
    __shared__ unsigned long long sKeys[LDS ? kSortLds : 1];
    __shared__ int sRankOf[LDS ? kSortLds : 1];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = min(A.nKp[f], A.kpStride);
    const int nCells = A.g.cols * A.g.rows;
    const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
    unsigned long long* gKeys = A.sortKeys + (size_t)f * A.sortCap;
    int* gRankOf = A.rankOf + (size_t)f * A.kpStride;
    int P = 1;
    while (P < n) P <<= 1;
    for (int j = tid; j < A.kpStride; j += 1024) A.matchOut[(size_t)f * A.kpStride + j] = -1;
    auto cellOf = [&](int j) {
        const orbfe_keypoint& k = kp[j];
        float px = k.x - A.g.minX;
        px = px * A.g.invW;
        float py = k.y - A.g.minY;
        py = py * A.g.invH;
        const int posX = (int)roundf(px), posY = (int)roundf(py);
        const int lin = posY * A.g.cols + posX;
        int v = nCells;
        if (lin >= 0 && lin < nCells) v = (lin % A.g.cols) * A.g.rows + lin / A.g.cols;
        return v;
    };
    // sort 1: visit order (cell x, cell y, index) -> rank
    auto key1 = [&](int j) { return ((unsigned long long)cellOf(j) << kRankBits) | (unsigned long long)j; };
    const bool regSort = LDS && P <= 1024;  // block-uniform
    if (regSort) block_sort_keys_reg(sKeys, n, P, tid, key1);
    else block_sort_keys<LDS>(sKeys, gKeys, n, P, tid, key1);
    for (int r = tid; r < n; r += 1024) {
        unsigned long long key;
        if constexpr (LDS) key = sKeys[r];
        else key = gKeys[r];
        const int idx = (int)(key & kRankMask);
        A.order[(size_t)f * A.kpStride + r] = idx;
        A.octByRank[(size_t)f * A.kpStride + r] = (uint8_t)min(max(kp[idx].octave, 0), 31);
        if constexpr (LDS) sRankOf[idx] = r;
        else gRankOf[idx] = r;
    }
    __syncthreads();
    // sort 2: storage order (level, cell x, cell y, index)
    auto key2 = [&](int j) {
        const unsigned long long lvl = (unsigned long long)min(max(kp[j].octave, 0), 31);
        return (lvl << (kRankBits + kCellBits + 1)) | ((unsigned long long)cellOf(j) << kRankBits) | (unsigned long long)j;
    };
    if (regSort) block_sort_keys_reg(sKeys, n, P, tid, key2);
    else block_sort_keys<LDS>(sKeys, gKeys, n, P, tid, key2);
    const unsigned long long* desc = reinterpret_cast<const unsigned long long*>(A.desc + (size_t)f * A.kpStride * 32);
    for (int p = tid; p < n; p += 1024) {
        unsigned long long key;
        if constexpr (LDS) key = sKeys[p];
        else key = gKeys[p];
        const int idx = (int)(key & kRankMask);
        const int v = (int)((key >> kRankBits) & ((1u << (kCellBits + 1)) - 1u));
        const int lvl = (int)(key >> (kRankBits + kCellBits + 1));
        const orbfe_keypoint k = kp[idx];
        int rank;
        if constexpr (LDS) rank = sRankOf[idx];
        else rank = gRankOf[idx];
        const int cy = v < nCells ? v % A.g.rows : 0x7fff;
        A.rec[(size_t)f * A.kpStride + p] = make_int4(rank, lvl | (cy << 8), __float_as_int(k.x), __float_as_int(k.y));
        unsigned long long* dd = A.descS + ((size_t)f * A.kpStride + p) * 4;
        dd[0] = desc[(size_t)idx * 4 + 0];
        dd[1] = desc[(size_t)idx * 4 + 1];
        dd[2] = desc[(size_t)idx * 4 + 2];
        dd[3] = desc[(size_t)idx * 4 + 3];
    }
    // colStart[l][cx] = first storage slot whose (level, cell) is >= (l, cx * rows) (lower bound), cx = 0..cols:
    // the keypoints of grid columns [a, b] of level l are the slots [colStart[l][a], colStart[l][b + 1])
    const int tabStride = A.g.cols + 1;
    int* cs = A.colStart + (size_t)f * A.tabLevels * tabStride;
    const int nTab = A.tabLevels * tabStride;
    for (int e = tid; e < nTab; e += 1024) {
        const int l = e / tabStride;
        const int v = (e - l * tabStride) * A.g.rows;
        const unsigned long long want = ((unsigned long long)l << (kCellBits + 1)) | (unsigned long long)v;
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            unsigned long long key;
            if constexpr (LDS) key = sKeys[mid];
            else key = gKeys[mid];
            if ((key >> kRankBits) < want) lo = mid + 1;
            else hi = mid;
        }
        cs[e] = lo;
    }
}

// Counting sort of a frame's map points by (pyramid level, 8 x 6 spatial tile); the last bucket
// holds the invalid ones.  Only the WORK ASSIGNMENT of the top-K pass uses this order: the 64 lanes
// of a wave then walk windows of similar size in the same part of the frame (similar trip counts,
// shared cache lines).  Results stay indexed by the original map point order.
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

__device__ __forceinline__ void proj_sort_body(const ProjArgs& A)
{
    __shared__ int sHist[kSortBuckets];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    for (int b = tid; b < kSortBuckets; b += 1024) sHist[b] = 0;
    __syncthreads();
    const orbfe_map_point* mps = A.mps + (size_t)f * A.M;
    for (int i = tid; i < A.M; i += 1024) atomicAdd(&sHist[sort_bucket(A, mps[i])], 1);
    __syncthreads();
    {  // exclusive prefix in place: two buckets per thread, wave scan by shuffles, 16 wave totals through LDS
        static_assert(kSortBuckets <= 2048, "two buckets per thread");
        __shared__ int sWaveTot[16];
        const int b0 = 2 * tid, b1 = 2 * tid + 1;
        const int c0 = b0 < kSortBuckets ? sHist[b0] : 0, c1 = b1 < kSortBuckets ? sHist[b1] : 0;
        const int lane = tid & 63, wv = tid >> 6;
        int incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sWaveTot[wv] = incl;
        __syncthreads();
        int base = incl - (c0 + c1);
        for (int q = 0; q < wv; q++) base += sWaveTot[q];
        if (b0 < kSortBuckets) sHist[b0] = base;
        if (b1 < kSortBuckets) sHist[b1] = base + c0;
    }
    __syncthreads();
    for (int i = tid; i < A.M; i += 1024) {
        const int pos = atomicAdd(&sHist[sort_bucket(A, mps[i])], 1);
        A.perm[(size_t)f * A.M + pos] = i;
    }
}

// One block per frame prepares both sides of the search: the keypoint grid (visit-order sort, level-major storage,
// column-start tables) and, for SearchByProjection, the (level, tile) order of the map points.
template <bool LDS, bool SORT_MPS>
__global__ __launch_bounds__(1024) void proj_prepare_kernel(ProjArgs A)
{
    // the two sides are independent: with SORT_MPS the launch has two blocks per frame (blockIdx.y), one per side -- a single
    // frame's call then pays the longer of the two instead of their sum
    if constexpr (SORT_MPS) {
        if (blockIdx.y == 1) {
            proj_sort_body(A);
            return;
        }
    }
    proj_grid_body<LDS>(A);
}

// Thread per map point: walk the storage range of every (level, grid column) of the window (cells
// minCY..maxCY of one column are contiguous), keep the kTopK smallest keys sorted (min/max network on
// 32-bit keys).  A block owns 256 map points of one frame, ordered by (level, tile): the storage segment
// of the levels they can match is staged in LDS (records + descriptors, 48 B per keypoint), because the
// per-lane gathers of this loop are ~8x cheaper from LDS than through the vector L1.
constexpr int kTopkLds = 768;  // staged keypoints per block (36 KB -> 4 blocks per CU); larger segments read global memory

struct TopkLds {
    int4 rec[kTopkLds];
    unsigned long long desc[kTopkLds][4];
};

template <bool LDS, bool WIDE>
__device__ __forceinline__ int topk_scan(const ProjArgs& A, const MpWindow& w, const int* cs, int tabStride, const int4* rec,
                                         const unsigned long long* descS, const TopkLds* S, int segBase,
                                         unsigned long long d0, unsigned long long d1, unsigned long long d2,
                                         unsigned long long d3, uint32_t (&keys)[kTopK])
{
    int total = 0;
    // One contiguous slot range per level: all rows of the window's grid columns (columns are contiguous in the
    // level-major visit order); the cell-row test runs per keypoint on the record.  Compared with one range per
    // (level, column) this visits about 1.6x more records, but needs 4 table loads per map point instead of ~60
    // dependent ones, and the trip counts of the 64 lanes of a wave are far more uniform.
    const int lFirst = max(w.minLevel, 0), l1 = w.maxLevel;
    // WIDE (relocalisation, three levels): the two-level body runs once more for the third level
    for (int l0 = lFirst; l0 <= (WIDE ? l1 : lFirst); l0 += 2) {
    int plo[2], phi[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int l = min(l0 + k, l1);
        const int* csl = cs + (size_t)l * tabStride;
        plo[k] = csl[w.minCX];
        phi[k] = (l0 + k <= l1) ? csl[w.maxCX + 1] : plo[k];
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {  // levels that pass src/Frame.cc:437-452 (at most two)
        int4 qn = make_int4(0, 0, 0, 0);  // record of the next slot, requested one iteration ahead
        if (plo[k] < phi[k]) {
            if constexpr (LDS) qn = S->rec[plo[k] - segBase];
            else qn = rec[plo[k]];
        }
        for (int p = plo[k]; p < phi[k]; p++) {
            const int4 q = qn;
            if (p + 1 < phi[k]) {
                if constexpr (LDS) qn = S->rec[p + 1 - segBase];
                else qn = rec[p + 1];
            }
            const int cy = q.y >> 8;
            if (cy < w.minCY || cy > w.maxCY) continue;
            const float dx = __int_as_float(q.z) - w.x, dy = __int_as_float(q.w) - w.y;
            if (!(fabsf(dx) < w.r && fabsf(dy) < w.r)) continue;  // src/Frame.cc:461
            int dist;
            if constexpr (LDS) {
                const unsigned long long* kd = S->desc[p - segBase];
                dist = __popcll(kd[0] ^ d0) + __popcll(kd[1] ^ d1) + __popcll(kd[2] ^ d2) + __popcll(kd[3] ^ d3);
            } else {
                const unsigned long long* kd = descS + (size_t)p * 4;
                dist = __popcll(kd[0] ^ d0) + __popcll(kd[1] ^ d1) + __popcll(kd[2] ^ d2) + __popcll(kd[3] ^ d3);
            }
            if (dist >= A.dCut) continue;  // cannot change the verdict (proj_dcut)
            total++;
            uint32_t key = make_key32(dist, q.x);
            if (key < keys[kTopK - 1]) {
                // sorted insert without a dependency chain: after the insert, slot t holds the median of
                // (old keys[t-1], key, old keys[t]) -- one v_med3_u32 per slot, all 24 independent (top-down, in place)
#pragma unroll
                for (int t = kTopK - 1; t > 0; t--) keys[t] = umed3(keys[t - 1], key, keys[t]);
                keys[0] = min(keys[0], key);
            }
        }
    }
    }
    return total;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void proj_topk_kernel(ProjArgs A)
{
    __shared__ TopkLds S;
    __shared__ int sLvlLo, sLvlHi;
    const int f = blockIdx.y;
    const int slotIdx = blockIdx.x * 256 + threadIdx.x;
    const bool live = slotIdx < A.M;
    const int i = live ? A.perm[(size_t)f * A.M + slotIdx] : 0;
    MpWindow w;
    w.valid = false;
    w.minLevel = w.maxLevel = 0;
    if (live) w = mp_window(A, A.mps[(size_t)f * A.M + i]);
    if (threadIdx.x == 0) { sLvlLo = 64; sLvlHi = -1; }
    __syncthreads();
    if (w.valid) {
        atomicMin(&sLvlLo, max(w.minLevel, 0));
        atomicMax(&sLvlHi, w.maxLevel);
    }
    __syncthreads();
    const int tabStride = A.g.cols + 1;
    const int* cs = A.colStart + (size_t)f * A.tabLevels * tabStride;
    const int4* rec = A.rec + (size_t)f * A.kpStride;
    const unsigned long long* descS = A.descS + (size_t)f * A.kpStride * 4;
    // storage segment of the levels this block can match: [first slot of level lo, first slot of level hi+1)
    int segBase = 0, segEnd = 0;
    if (sLvlHi >= 0) {
        segBase = cs[(size_t)sLvlLo * tabStride];
        segEnd = sLvlHi + 1 < A.tabLevels ? cs[(size_t)(sLvlHi + 1) * tabStride] : min(A.nKp[f], A.kpStride);
    }
    const bool useLds = segEnd - segBase <= kTopkLds;  // block-uniform
    if (useLds) {
        // records and descriptors of the segment are contiguous: 48 B per keypoint as 3 x 16-byte pieces, four
        // loads in flight per thread before the LDS stores (a load / store pair per iteration would serialise
        // the global latency)
        const int nSeg = segEnd - segBase;
        const uint4* gRec = reinterpret_cast<const uint4*>(rec + segBase);
        const uint4* gDesc = reinterpret_cast<const uint4*>(descS + (size_t)segBase * 4);
        uint4* lRec = reinterpret_cast<uint4*>(S.rec);
        uint4* lDesc = reinterpret_cast<uint4*>(&S.desc[0][0]);
        for (int j0 = 0; j0 < nSeg * 2; j0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg * 2) v[k] = gDesc[j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg * 2) lDesc[j] = v[k];
            }
        }
        for (int j0 = 0; j0 < nSeg; j0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg) v[k] = gRec[j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg) lRec[j] = v[k];
            }
        }
    }
    __syncthreads();
    if (!live) return;
    uint32_t keys[kTopK];
#pragma unroll
    for (int t = 0; t < kTopK; t++) keys[t] = kKey32None;
    int total = 0;
    if (w.valid) {
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * A.M + i) * 32);
        const unsigned long long d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
        if (useLds) total = topk_scan<true, WIDE>(A, w, cs, tabStride, rec, descS, &S, segBase, d0, d1, d2, d3, keys);
        else total = topk_scan<false, WIDE>(A, w, cs, tabStride, rec, descS, &S, segBase, d0, d1, d2, d3, keys);
    }
    A.cnt[(size_t)f * A.M + i] = total;
    // map-point-major: the keys of one map point are contiguous (96 B; i is a permuted index: entry-major
    // 4-byte stores would dirty 16 different sectors per map point)
    uint4* dst = reinterpret_cast<uint4*>(A.topk + ((size_t)f * A.M + i) * kTopK);
#pragma unroll
    for (int t = 0; t < kTopK / 4; t++) dst[t] = make_uint4(keys[4 * t], keys[4 * t + 1], keys[4 * t + 2], keys[4 * t + 3]);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-map-point form of the same pass, for SMALL launches (one or a few frames; proj_launch selects it).  The 64
// lanes of a wave take 64 consecutive storage slots of ONE map point's range at a time (the ranges of its two or three
// levels laid end to end): window test, distance and cut-off run once per record; the few keys below the cut-off are
// appended to a wave-private LDS list (ballot + mbcnt), and at the end every key finds its position by counting the
// smaller ones (v_readlane broadcast) and is stored straight to that slot -- no sorted insert.  More than 64 survivors
// (huge radii) are reduced to the best 64 by the same ranking before the next chunk.  Window parameters, table lookups
// and descriptors are computed / loaded lane-parallel (lane j <-> map point j of the wave) and broadcast per map point;
// nothing inside the sequential loop waits on global memory.  `mpw` map points per wave: few for a single frame, so
// its map points spread over the whole chip.  The lists and counts are identical to proj_topk_kernel's.
// Measured at 256 frames x 2000 map points it is 2.6x SLOWER than the thread-per-map-point kernel (half-empty chunks,
// per-map-point scalar bookkeeping); for one frame it shortens the call by 10-17 %.
// ---------------------------------------------------------------------------------------------
constexpr int kWaveListCap = 128;

// ranks the n (<= 128) distinct keys of `list` and writes those with fewer than `limit` smaller keys to out[rank]
__device__ __forceinline__ void rank_emit_lds(const uint32_t* list, int n, int limit, uint32_t* out, int lane)
{
    const uint32_t e0 = lane < n ? list[lane] : kKey32None;
    const uint32_t e1 = lane + 64 < n ? list[lane + 64] : kKey32None;
    int r0 = 0, r1 = 0;
    for (int b = 0; b < n; b++) {
        const uint32_t kb = list[b];  // uniform address: LDS broadcast
        r0 += kb < e0;
        r1 += kb < e1;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < n && r0 < limit) out[r0] = e0;
    if (lane + 64 < n && r1 < limit) out[r1] = e1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS only: a fence would also wait for the global stores in flight
    __builtin_amdgcn_wave_barrier();
}

template <bool WIDE, bool LDS>
__device__ __forceinline__ void topk_wave_body(const ProjArgs& A, const TopkLds& S, uint32_t (*sList)[2][kWaveListCap], int f, int lane,
                                               int wv, int nMine, int i, const MpWindow& w, const int* cs, int tabStride,
                                               const int4* rec, const unsigned long long* descS, int segBase)
{
    constexpr int NLV = WIDE ? 3 : 2;
    // lane-parallel table lookups of the own map point: one contiguous slot range per level (see topk_scan)
    int plo[NLV], phi[NLV];
    {
        const int l0 = max(w.minLevel, 0), l1 = w.maxLevel;
#pragma unroll
        for (int k = 0; k < NLV; k++) {
            const int l = min(l0 + k, l1);
            const int* csl = cs + (size_t)l * tabStride;
            plo[k] = w.valid ? csl[w.minCX] : 0;
            phi[k] = (w.valid && l0 + k <= l1) ? csl[w.maxCX + 1] : plo[k];
        }
    }
    const int validI = w.valid ? 1 : 0;
    // own map point's descriptor, broadcast later with v_readlane (a load inside the per-map-point loop would put a
    // memory round trip on every iteration of a sequential loop)
    uint32_t dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (w.valid) {
        const uint4* dp = reinterpret_cast<const uint4*>(A.mpDesc + ((size_t)f * A.M + i) * 32);
        const uint4 a = dp[0], b = dp[1];
        dq[0] = a.x; dq[1] = a.y; dq[2] = a.z; dq[3] = a.w; dq[4] = b.x; dq[5] = b.y; dq[6] = b.z; dq[7] = b.w;
    }
    // everything loaded from global memory is consumed here, before the sequential loop: inside it, a wait for one
    // of these loads (vmcnt) would also wait for the previous map point's stores on every iteration
#pragma unroll
    for (int k = 0; k < NLV; k++) asm volatile("" : "+v"(plo[k]), "+v"(phi[k]));
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("" : "+v"(dq[k]));
    {
        int iv = i;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(iv));
    }
    for (int j = 0; j < nMine; j++) {  // wave-uniform trip count
        const int mp = __builtin_amdgcn_readlane(i, j);
        uint32_t* dst = A.topk + ((size_t)f * A.M + mp) * kTopK;
        int total = 0, listN = 0, cur = 0;
        if (__builtin_amdgcn_readlane(validI, j)) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.x), j));
            const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.y), j));
            const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.r), j));
            const int minCY = __builtin_amdgcn_readlane(w.minCY, j), maxCY = __builtin_amdgcn_readlane(w.maxCY, j);
            unsigned long long d[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                d[k] = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dq[2 * k], j) |
                       ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dq[2 * k + 1], j) << 32);
            // the slot ranges of the (two or three) levels laid end to end: lane <-> position in the concatenation
            int a[NLV], len[NLV], lenAll = 0;
#pragma unroll
            for (int k = 0; k < NLV; k++) {
                a[k] = __builtin_amdgcn_readlane(plo[k], j);
                len[k] = __builtin_amdgcn_readlane(phi[k], j) - a[k];
                lenAll += len[k];
            }
            for (int base = 0; base < lenAll; base += 64) {
                if (listN > 64) {  // keep the best 64 (the best kTopK are among them)
                    rank_emit_lds(sList[wv][cur], listN, 64, sList[wv][cur ^ 1], lane);
                    cur ^= 1;
                    listN = 64;
                }
                const int idx = base + lane;
                bool q = false;
                uint32_t key = kKey32None;
                if (idx < lenAll) {
                    int p = a[0] + idx;
                    if (idx >= len[0]) p = a[1] + (idx - len[0]);
                    if constexpr (NLV == 3)
                        if (idx >= len[0] + len[1]) p = a[2] + (idx - len[0] - len[1]);
                    int4 rq;
                    if constexpr (LDS) rq = S.rec[p - segBase];
                    else rq = rec[p];
                    const int cy = rq.y >> 8;
                    const float dx = __int_as_float(rq.z) - x, dy = __int_as_float(rq.w) - y;
                    if (cy >= minCY && cy <= maxCY && fabsf(dx) < r && fabsf(dy) < r) {  // src/Frame.cc:461
                        int dist;
                        if constexpr (LDS) {
                            const unsigned long long* kd = S.desc[p - segBase];
                            dist = __popcll(kd[0] ^ d[0]) + __popcll(kd[1] ^ d[1]) + __popcll(kd[2] ^ d[2]) + __popcll(kd[3] ^ d[3]);
                        } else {
                            const unsigned long long* kd = descS + (size_t)p * 4;
                            dist = __popcll(kd[0] ^ d[0]) + __popcll(kd[1] ^ d[1]) + __popcll(kd[2] ^ d[2]) + __popcll(kd[3] ^ d[3]);
                        }
                        if (dist < A.dCut) {  // proj_dcut
                            q = true;
                            key = make_key32(dist, rq.x);
                        }
                    }
                }
                const unsigned long long mask = __ballot(q);
                if (mask) {  // wave-uniform
                    if (q)
                        sList[wv][cur][listN + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = key;
                    const int c = __popcll(mask);
                    listN += c;
                    total += c;
                }
            }
            if (total) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS only: a fence would also wait for the global stores in flight
                __builtin_amdgcn_wave_barrier();
                if (listN > 64) {
                    rank_emit_lds(sList[wv][cur], listN, 64, sList[wv][cur ^ 1], lane);
                    cur ^= 1;
                    listN = 64;
                }
                const uint32_t e = lane < listN ? sList[wv][cur][lane] : kKey32None;
                int rk = 0;
                for (int b = 0; b < listN; b++) rk += (uint32_t)__builtin_amdgcn_readlane((int)e, b) < e;
                // map-point-major list: sorted smallest keys, padded with kKey32None
                if (lane < listN && rk < kTopK) dst[rk] = e;
                if (lane >= listN && lane < kTopK) dst[lane] = kKey32None;
            }
        }
        if (lane == 0) A.cnt[(size_t)f * A.M + mp] = total;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void proj_topk_wave_kernel(ProjArgs A, int mpw)
{
    __shared__ TopkLds S;
    __shared__ int sLvlLo, sLvlHi;
    __shared__ uint32_t sList[4][2][kWaveListCap];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waveBase = (blockIdx.x * 4 + wv) * mpw;  // first map point (in perm order) of this wave
    const int nMine = min(mpw, A.M - waveBase);         // may be <= 0
    const bool live = lane < nMine;
    const int i = live ? A.perm[(size_t)f * A.M + waveBase + lane] : 0;
    MpWindow w;
    w.valid = false;
    w.minLevel = w.maxLevel = 0;
    w.x = w.y = w.r = 0.0f;
    w.minCX = w.maxCX = w.minCY = w.maxCY = 0;
    if (live) w = mp_window(A, A.mps[(size_t)f * A.M + i]);
    if (threadIdx.x == 0) { sLvlLo = 64; sLvlHi = -1; }
    __syncthreads();
    if (w.valid) {
        atomicMin(&sLvlLo, max(w.minLevel, 0));
        atomicMax(&sLvlHi, w.maxLevel);
    }
    __syncthreads();
    const int tabStride = A.g.cols + 1;
    const int* cs = A.colStart + (size_t)f * A.tabLevels * tabStride;
    const int4* rec = A.rec + (size_t)f * A.kpStride;
    const unsigned long long* descS = A.descS + (size_t)f * A.kpStride * 4;
    int segBase = 0, segEnd = 0;
    if (sLvlHi >= 0) {
        segBase = cs[(size_t)sLvlLo * tabStride];
        segEnd = sLvlHi + 1 < A.tabLevels ? cs[(size_t)(sLvlHi + 1) * tabStride] : min(A.nKp[f], A.kpStride);
    }
    const bool useLds = segEnd - segBase <= kTopkLds;  // block-uniform
    if (useLds) {  // same staging as proj_topk_kernel
        const int nSeg = segEnd - segBase;
        const uint4* gRec = reinterpret_cast<const uint4*>(rec + segBase);
        const uint4* gDesc = reinterpret_cast<const uint4*>(descS + (size_t)segBase * 4);
        uint4* lRec = reinterpret_cast<uint4*>(S.rec);
        uint4* lDesc = reinterpret_cast<uint4*>(&S.desc[0][0]);
        for (int j0 = 0; j0 < nSeg * 2; j0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg * 2) v[k] = gDesc[j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg * 2) lDesc[j] = v[k];
            }
        }
        for (int j0 = 0; j0 < nSeg; j0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg) v[k] = gRec[j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nSeg) lRec[j] = v[k];
            }
        }
    }
    __syncthreads();
    if (useLds) topk_wave_body<WIDE, true>(A, S, sList, f, lane, wv, nMine, i, w, cs, tabStride, rec, descS, segBase);
    else topk_wave_body<WIDE, false>(A, S, sList, f, lane, wv, nMine, i, w, cs, tabStride, rec, descS, segBase);
}

// One persistent block per frame, one THREAD per map point, chunks of 1024 map points in index order.
// The frame's keypoints (cell, octave, x, y) and descriptors, in rank order, are staged in LDS once (frames
// up to kResN keypoints), so sweeps and the exact rescans of starved map points never touch global memory.
constexpr int kResN = 2048;

constexpr int kResNDesc = 1280;  // frames up to this many keypoints also keep their descriptors in LDS (77 KB per block)

template <int N, bool DESC>
struct ResolveLdsT {
    int claim[3][N];                      // by rank; three rotating tables (final / this sweep / next sweep)
    uint8_t oct[N];                       // by rank
    int4 rec[N];                          // by storage slot: {rank, octave | cell y << 8, x bits, y bits}
    // descriptors by storage slot: the exact rescans of starved map points (hundreds per frame when many map points
    // compete for look-alike keypoints) then never wait on global memory
    unsigned long long desc[DESC ? N * 4 : 4];
};

__device__ __forceinline__ void wave_top2_u32(uint32_t& k1, uint32_t& k2)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o1 = __shfl_xor(k1, d), o2 = __shfl_xor(k2, d);
        const uint32_t lo = min(k1, o1), hi = max(k1, o1);
        k2 = min(hi, min(k2, o2));
        k1 = lo;
    }
}

// exact rescan for a map point whose stored top-K ran dry: one WAVE scans the contiguous rank range of the
// window's grid columns (lane-strided) and reduces the two smallest free keys; all lanes get the result.
// The claim test goes first: a starved map point sits in a region where nearly everything is taken.
template <bool LDS, bool DESC, typename SLds>
__device__ __forceinline__ void full_scan_top2_wave(const ProjArgs& A, int f, int i, const MpWindow& w,
                                                    const unsigned long long (&d4)[4], const int* cs, const int* claim,
                                                    const SLds* S, int lane, uint32_t& k1, uint32_t& k2)
{
    k1 = kKey32None;
    k2 = kKey32None;
    if (w.valid) {  // wave-uniform
        const int tabStride = A.g.cols + 1;
        const int4* rec = A.rec + (size_t)f * A.kpStride;
        const unsigned long long* descS = A.descS + (size_t)f * A.kpStride * 4;
        for (int l = max(w.minLevel, 0); l <= w.maxLevel; l++) {
            const int* csl = cs + (size_t)l * tabStride;
            const int plo = csl[w.minCX], phi = csl[w.maxCX + 1];  // all rows of the window's columns
            for (int p = plo + lane; p < phi; p += 64) {
                int4 q;
                if constexpr (LDS) q = S->rec[p];
                else q = rec[p];
                if (claim[q.x] < i) continue;
                const int cy = q.y >> 8;
                if (cy < w.minCY || cy > w.maxCY) continue;
                const float dx = __int_as_float(q.z) - w.x, dy = __int_as_float(q.w) - w.y;
                if (!(fabsf(dx) < w.r && fabsf(dy) < w.r)) continue;
                int dist;
                // descriptors of the few survivors: from the block's LDS image when the frame is small enough for it, else L2
                if constexpr (DESC) dist = hamming256(reinterpret_cast<const uint2*>(S->desc + (size_t)p * 4), d4);
                else dist = hamming256(reinterpret_cast<const uint2*>(descS + (size_t)p * 4), d4);
                if (dist >= A.dCut) continue;
                const uint32_t key = make_key32(dist, q.x);
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
        }
    }
    wave_top2_u32(k1, k2);
}

// starved map points whose window + descriptor are parked in LDS by their own threads (one memory latency for all of
// them) instead of being fetched by the rescanning wave, one after the other
constexpr int kFbLds = 192;
constexpr int kResTabLds = 1024;  // column-start tables up to this many entries are kept in LDS too

// THREADS: 1024 for small launches (fewest ordered chunks: shortest call); 256 when the chip is full anyway -- a 1024-thread
// block with its register and LDS footprint keeps a whole CU to itself while it mostly waits on barriers.
template <bool LDS, int THREADS, int N = kResN, bool DESC = false>
__global__ __launch_bounds__(THREADS) void proj_resolve_kernel(ProjArgs A)
{
    __shared__ ResolveLdsT<N, DESC> S;
    __shared__ int sChanged[2];
    __shared__ int sGaveUp[2];                           // a claim HOLDER moved away in this sweep (by sweep parity): ranks came free
    __shared__ int sCount;
    __shared__ int sFbCount[2];                          // starved map points of the current sweep (by sweep parity, like sChanged)
    __shared__ int sFbMp[THREADS];
    __shared__ uint32_t sFbK1[THREADS], sFbK2[THREADS];
    __shared__ MpWindow sFbWin[kFbLds];
    __shared__ unsigned long long sFbDesc[kFbLds][4];
    __shared__ int sTab[LDS ? kResTabLds : 1];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = min(A.nKp[f], A.kpStride);
    const int M = A.M;
    // Claim tables (by rank): value = -1 if the slot holds a map point with observations on entry (:77-79), else the
    // smallest accepted map point (with observations) whose best match is that keypoint, else kClaimFree.
    // Three tables rotate so that a sweep needs TWO block barriers instead of five: T[fin] holds the final claims of
    // the earlier chunks; T[cur] = T[fin] + the tentative claims of this sweep (written by atomicMin in phase 1, read
    // in phase 2); T[nxt] is refilled with a copy of T[fin] during phase 2 for the next sweep.  A chunk that has
    // converged promotes T[cur] to final.
    int* T[3];
#pragma unroll
    for (int q = 0; q < 3; q++) T[q] = LDS ? S.claim[q] : A.claimG + ((size_t)f * 3 + q) * A.kpStride;
    int fin = 0, cur = 1, nxt = 2;
    const orbfe_map_point* mps = A.mps + (size_t)f * M;
    const int* order = A.order + (size_t)f * A.kpStride;
    const int4* rec = A.rec + (size_t)f * A.kpStride;
    const int* initObs = A.initObs ? A.initObs + (size_t)f * A.kpStride : nullptr;

    for (int r = tid; r < n; r += THREADS) {
        const int v = (initObs && initObs[order[r]] > 0) ? -1 : kClaimFree;
        T[fin][r] = v;
        T[cur][r] = v;
    }
    const uint8_t* octByRank = A.octByRank + (size_t)f * A.kpStride;
    if constexpr (LDS) {
        for (int r = tid; r < n; r += THREADS) {
            S.rec[r] = rec[r];
            S.oct[r] = octByRank[r];
        }
    }
    if constexpr (DESC) {
        const unsigned long long* dsrc = A.descS + (size_t)f * A.kpStride * 4;
        for (int q = tid; q < n * 4; q += THREADS) S.desc[q] = dsrc[q];
    }
    const int tabStride = A.g.cols + 1;
    const int* csG = A.colStart + (size_t)f * A.tabLevels * tabStride;
    const bool tabInLds = LDS && A.tabLevels * tabStride <= kResTabLds;  // block-uniform
    if (tabInLds)
        for (int k = tid; k < A.tabLevels * tabStride; k += THREADS) sTab[k] = csG[k];
    const int* cs = tabInLds ? sTab : csG;
    if (tid == 0) sCount = 0;
    int nAccepted = 0;
    __syncthreads();  // claim tables and LDS image complete before the first sweep

    for (int chunk = 0; chunk < M; chunk += THREADS) {
        const int i = chunk + tid;
        const bool live = i < M;
        int c = 0, obs = 0;
        uint32_t keys[kTopK];
#pragma unroll
        for (int t = 0; t < kTopK; t++) keys[t] = kKey32None;
        if (live) {
            c = A.cnt[(size_t)f * M + i];
            obs = mps[i].observations;
            if (c > 0) {
                const uint4* src = reinterpret_cast<const uint4*>(A.topk + ((size_t)f * M + i) * kTopK);
#pragma unroll
                for (int t = 0; t < kTopK / 4; t++) {
                    const uint4 q = src[t];
                    keys[4 * t] = q.x; keys[4 * t + 1] = q.y; keys[4 * t + 2] = q.z; keys[4 * t + 3] = q.w;
                }
            }
        }
        int res = -1;  // rank of the accepted keypoint
        // the pair the verdict was made from survives the sweep: a later sweep looks again only if it has to
        uint32_t k1 = kKey32None, k2 = kKey32None;
        bool k2StandIn = false;  // k2 is the K-th key standing in for an unseen second best: taken itself, nothing to watch
        // (T[cur] == T[fin] here: initialised above / refilled at the end of the previous chunk)
        for (int iter = 0; iter <= THREADS + 1; iter++) {
            // ---- phase 1: this sweep's tentative claims on top of the final ones ----
            if (tid == 0) { sChanged[iter & 1] = 0; sFbCount[iter & 1] = 0; sGaveUp[iter & 1] = 0; }  // last read two barriers ago
            if (res >= 0 && obs > 0) atomicMin(&T[cur][res], i);
            __syncthreads();
            // ---- phase 2: the map points whose verdict may have moved look for their two best free candidates ----
            // T[cur] is rebuilt every sweep from the final claims + every lane's current verdict.  Unless a claim HOLDER
            // moved away in the previous sweep (sGaveUp), every entry of it is <= the previous sweep's: what was taken
            // stays taken, so a lane whose best and second best are still free would find the same pair again -- it keeps
            // its verdict without looking (2 lookups instead of kTopK; a wave whose lanes all keep theirs skips the rest).
            const int* claim = T[cur];
            bool look = iter == 0 || sGaveUp[(iter & 1) ^ 1] != 0;
            if (!look && c > 0) {
                const int c1 = k1 != kKey32None ? claim[(int)(k1 & kRankMask)] : kClaimFree;
                const int c2 = (k2 != kKey32None && !k2StandIn) ? claim[(int)(k2 & kRankMask)] : kClaimFree;
                look = c1 < i || c2 < i;
            }
            int slot = -1;
            if (c > 0 && look) {
                k1 = kKey32None;
                k2 = kKey32None;
                k2StandIn = false;
                // ascending list: the first two free entries are the two smallest free keys -- a running (min, second) over
                // m = free ? key : none costs v_min_u32 + v_med3_u32 per entry (a <= b: second(a, b, m) = med3(a, m, b))
                // instead of a compare / select chain; all kTopK claim lookups are issued before the first is used
                int cl[kTopK];
#pragma unroll
                for (int t = 0; t < kTopK; t++) cl[t] = claim[keys[t] != kKey32None ? (int)(keys[t] & kRankMask) : 0];
#pragma unroll
                for (int t = 0; t < kTopK; t++) {
                    const uint32_t m = cl[t] >= i ? keys[t] : kKey32None;  // (an empty entry is kKey32None already)
                    k2 = umed3(k1, m, k2);
                    k1 = min(k1, m);
                }
                if (k2 == kKey32None && c > kTopK) {
                    // The stored list ran dry.  Every candidate that was not stored has a key above
                    // keys[K-1], i.e. a distance >= dK.  Two cases are decided without looking at them:
                    //  - nothing free and dK > TH_HIGH: the best free candidate fails :108 -> no match;
                    //  - one free entry with best <= nnRatio * dK: the ratio test :110 cannot reject
                    //    (the float product is monotone in the unknown second distance >= dK).
                    const int dK = (int)(keys[kTopK - 1] >> kRankBits);
                    bool decided = false;
                    if (k1 == kKey32None) decided = dK > ORBFE_TH_HIGH;
                    else {
                        const int bd = (int)(k1 >> kRankBits);
                        decided = bd > ORBFE_TH_HIGH || (A.nnRatio > 0.f && !((float)bd > A.nnRatio * (float)dK));
                        if (decided) {  // stand-in with distance dK: same verdict as the true second
                            k2 = keys[kTopK - 1];
                            k2StandIn = true;
                        }
                    }
                    if (!decided) {  // exact rescan, done cooperatively below
                        slot = atomicAdd(&sFbCount[iter & 1], 1);
                        sFbMp[slot] = i;
                        if (slot < kFbLds) {
                            sFbWin[slot] = mp_window(A, mps[i]);
                            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * M + i) * 32);
#pragma unroll
                            for (int t = 0; t < 4; t++) sFbDesc[slot][t] = dp[t];
                        }
#ifdef ORBFE_DIAG
                        atomicAdd(&A.dbg[f * 4 + 1], 1);  // diagnostics build only: exact rescans of this frame
#endif
                    }
                }
            }
            // the table of the next sweep starts as a copy of the final claims (nobody reads or writes T[nxt] in this phase)
            for (int k = tid; k < n; k += THREADS) T[nxt][k] = T[fin][k];
            auto verdict = [&]() {
                int result = -1;
                if (k1 != kKey32None) {
                    const int bestDist = (int)(k1 >> kRankBits), bestRank = (int)(k1 & kRankMask);
                    int bestLevel;
                    if constexpr (LDS) bestLevel = S.oct[bestRank];
                    else bestLevel = octByRank[bestRank];
                    int bestDist2 = 256, bestLevel2 = -1;
                    if (k2 != kKey32None) {
                        bestDist2 = (int)(k2 >> kRankBits);
                        const int r2 = (int)(k2 & kRankMask);
                        if constexpr (LDS) bestLevel2 = S.oct[r2];
                        else bestLevel2 = octByRank[r2];
                    }
                    if (bestDist <= ORBFE_TH_HIGH) {  // :108-117
                        const bool reject = bestLevel == bestLevel2 && (float)bestDist > A.nnRatio * (float)bestDist2;
                        if (!reject) result = bestRank;
                    }
                }
                return result;
            };
            // a lane that held the claim on its old rank and now moves away frees that rank: everybody looks again next sweep
            auto move_to = [&](int result) {
                if (result != res) {
                    if (res >= 0 && obs > 0 && claim[res] == i) sGaveUp[iter & 1] = 1;
                    res = result;
                    sChanged[iter & 1] = 1;
                }
            };
            if (slot < 0 && look && c > 0) move_to(verdict());
            __syncthreads();
            if (sFbCount[iter & 1] > 0) {  // block-uniform, rare: some stored list ran dry undecided -- exact rescan, one wave per map point
                const int nFb = sFbCount[iter & 1];
                // (four map points per wave on 16-lane rows was measured: 0.355 ms against 0.324 -- a sweep rarely holds more
                // starved map points than the block has waves, and a row needs four times the iterations per window)
                for (int q = __builtin_amdgcn_readfirstlane(tid >> 6); q < nFb; q += THREADS / 64) {
                    uint32_t a1, a2;
                    const int mp = sFbMp[q];
                    MpWindow w;
                    unsigned long long d4[4];
                    if (q < kFbLds) {
                        w = sFbWin[q];
#pragma unroll
                        for (int t = 0; t < 4; t++) d4[t] = sFbDesc[q][t];
                    } else {
                        w = mp_window(A, mps[mp]);
                        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * M + mp) * 32);
#pragma unroll
                        for (int t = 0; t < 4; t++) d4[t] = dp[t];
                    }
                    full_scan_top2_wave<LDS, DESC>(A, f, mp, w, d4, cs, claim, &S, tid & 63, a1, a2);
                    if ((tid & 63) == 0) { sFbK1[q] = a1; sFbK2[q] = a2; }
                }
                __syncthreads();
                if (slot >= 0) {
                    k1 = sFbK1[slot];
                    k2 = sFbK2[slot];
                    move_to(verdict());
                }
                __syncthreads();
            }
#ifdef ORBFE_DIAG
            if (tid == 0) atomicAdd(&A.dbg[f * 4 + 0], 1);  // diagnostics build only: sweeps of this frame
#endif
            if (!sChanged[iter & 1]) break;  // fixed point: T[cur] = final claims + this chunk's
            // next sweep writes its claims into the fresh copy
            const int tq = cur;
            cur = nxt;
            nxt = tq;
        }
        // the converged table becomes the final one; the next chunk starts from a copy of it
        {
            const int tq = fin;
            fin = cur;
            cur = tq;
        }
        if (chunk + THREADS < M) {
            for (int k = tid; k < n; k += THREADS) T[cur][k] = T[fin][k];
            __syncthreads();
        }
        // F->mvpMapPoints[bestIdx] = pMP in map-point order: the last writer wins; nmatches counts accepts
        if (res >= 0) {
            atomicMax(&A.matchOut[(size_t)f * A.kpStride + order[res]], i);
            nAccepted++;
        }
    }
    if (nAccepted) atomicAdd(&sCount, nAccepted);
    __syncthreads();
    if (tid == 0) A.nMatches[f] = sCount;
}

}  // namespace

namespace proj {

void proj_prepare_launch(hipStream_t s, const ProjArgs& A, bool sortMps)
{
    const bool lds = A.kpStride <= kSortLds;
    if (sortMps) {
        if (lds) hipLaunchKernelGGL((proj_prepare_kernel<true, true>), dim3(A.B, 2), dim3(1024), 0, s, A);
        else hipLaunchKernelGGL((proj_prepare_kernel<false, true>), dim3(A.B, 2), dim3(1024), 0, s, A);
    } else {
        if (lds) hipLaunchKernelGGL((proj_prepare_kernel<true, false>), dim3(A.B), dim3(1024), 0, s, A);
        else hipLaunchKernelGGL((proj_prepare_kernel<false, false>), dim3(A.B), dim3(1024), 0, s, A);
    }
}

// Distance cut-off.  The verdict of src/ORBmatcher.cc:108-117 only depends on candidates below a bound:
// a best candidate needs dist <= TH_HIGH, and a second-best with nnRatio * d2 >= TH_HIGH can never reject a
// best <= TH_HIGH (the float product is monotone in d2) -- it acts exactly like "no second candidate".
// So every candidate with dist >= max(TH_HIGH + 1, min{d : !(TH_HIGH > nnRatio * d)}) is skipped: with
// nnRatio 0.85 that is 118, which removes ~90 % of the unrelated keypoints (their distances cluster at 128)
// from the sorted inserts and keeps the stored lists complete.
int proj_dcut(float nnRatio)
{
    int D = 256;
    for (int d = 0; d < 256; d++)
        if (!((float)ORBFE_TH_HIGH > nnRatio * (float)d)) {
            D = d;
            break;
        }
    return std::max(D, ORBFE_TH_HIGH + 1);
}

// carve the device scratch after whatever `sc` already holds; sets A's scratch pointers
int proj_setup(MatchScratch& m, ProjArgs& A, Carver sc, size_t hostNeed, std::string& err, size_t* endOff)
{
    const int B = A.B, M = A.M;
    A.sortCap = 1;
    while (A.sortCap < A.kpStride) A.sortCap <<= 1;
    const bool ldsSort = A.kpStride <= kSortLds;
    const size_t oKeys = sc.take(ldsSort ? 8 : (size_t)B * A.sortCap * sizeof(unsigned long long));
    A.tabLevels = std::min(std::max(A.nLevels, 1), 32);
    A.dCut = proj_dcut(A.nnRatio);
    const size_t oOrder = sc.take((size_t)B * A.kpStride * sizeof(int));
    const size_t oOct = sc.take((size_t)B * A.kpStride);
    const size_t oRankOf = sc.take(ldsSort ? 8 : (size_t)B * A.kpStride * sizeof(int));
    const size_t oRec = sc.take((size_t)B * A.kpStride * sizeof(int4));
    const size_t oDescS = sc.take((size_t)B * A.kpStride * 32);
    const size_t oCol = sc.take((size_t)B * A.tabLevels * ((size_t)A.g.cols + 1) * sizeof(int));
    const size_t oCnt = sc.take((size_t)B * std::max(M, 1) * sizeof(int));
    const size_t oTopk = sc.take((size_t)B * kTopK * std::max(M, 1) * sizeof(uint32_t));
    const size_t oClaim = sc.take((size_t)B * 3 * A.kpStride * sizeof(int));  // three rotating claim tables per frame
    const size_t oPerm = sc.take((size_t)B * std::max(M, 1) * sizeof(int));
    const size_t oDbg = sc.take((size_t)B * 4 * sizeof(int));
    if (endOff) *endOff = sc.off;
    int rc = ensure(m, sc.off, hostNeed + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    A.sortKeys = reinterpret_cast<unsigned long long*>(dp + oKeys);
    A.order = reinterpret_cast<int*>(dp + oOrder);
    A.octByRank = dp + oOct;
    A.rankOf = reinterpret_cast<int*>(dp + oRankOf);
    A.rec = reinterpret_cast<int4*>(dp + oRec);
    A.descS = reinterpret_cast<unsigned long long*>(dp + oDescS);
    A.colStart = reinterpret_cast<int*>(dp + oCol);
    A.cnt = reinterpret_cast<int*>(dp + oCnt);
    A.topk = reinterpret_cast<uint32_t*>(dp + oTopk);
    A.claimG = reinterpret_cast<int*>(dp + oClaim);
    A.perm = reinterpret_cast<int*>(dp + oPerm);
    A.dbg = reinterpret_cast<int*>(dp + oDbg);
    return ORBFE_OK;
}

int proj_launch(hipStream_t s, ProjArgs& A, std::string& err)
{
    if (A.M == 0) {
        proj_prepare_launch(s, A, false);
        MCHK(hipMemsetAsync(A.nMatches, 0, (size_t)A.B * sizeof(int), s));
        return ORBFE_OK;
    }
#ifdef ORBFE_DIAG
    MCHK(hipMemsetAsync(A.dbg, 0, (size_t)A.B * 4 * sizeof(int), s));
#endif
    proj_prepare_launch(s, A, true);
    // Large launches: thread per map point (164 VALU + 38 SALU instructions per map point, 0.27 ms for 256 x 2000).
    // Small launches (a single frame has 8 blocks of 256 map points): the wave-per-map-point form spreads the frame's
    // map points over the chip -- it costs more instructions (181 VALU + 138 SALU per map point, 0.70 ms at the large
    // size) but a live per-frame call finishes 10-17 % sooner.
    const int blocks256 = (A.M + 255) / 256;
    if ((long long)blocks256 * A.B >= 128) {
        if (A.mode == kModeReloc)
            hipLaunchKernelGGL(proj_topk_kernel<true>, dim3(blocks256, A.B), dim3(256), 0, s, A);
        else
            hipLaunchKernelGGL(proj_topk_kernel<false>, dim3(blocks256, A.B), dim3(256), 0, s, A);
    } else {
        // map points per wave: as few as it takes to put >= 512 blocks on the chip, down to ONE -- a wave's map points are a
        // serial chain of latency-bound steps, and a single frame's call waits for the longest chain (one frame, 2000 map
        // points: 36.5 us with eight per wave, 13.2 us with one; orbfe_track_frame 0.306 -> 0.272 ms)
        int mpw = 64;
        while (mpw > 1 && (long long)((A.M + 4 * mpw - 1) / (4 * mpw)) * A.B < 512) mpw >>= 1;
        const dim3 grid((A.M + 4 * mpw - 1) / (4 * mpw), A.B);
        if (A.mode == kModeReloc)
            hipLaunchKernelGGL(proj_topk_wave_kernel<true>, grid, dim3(256), 0, s, A, mpw);
        else
            hipLaunchKernelGGL(proj_topk_wave_kernel<false>, grid, dim3(256), 0, s, A, mpw);
    }
    const bool lds = A.kpStride <= kResN;  // nKp[f] <= kpStride: the whole frame fits the LDS image
    // Block shape of the resolve pass: a frame is a dependent chain of sweeps, so the call is shortest with the fewest
    // chunks (1024 threads).  Frames of up to kResNDesc keypoints also stage their descriptors in LDS: the exact rescans
    // of starved map points (one wave each, ~250 per frame on the bench stream) then run out of LDS on all 16 waves.
#ifdef ORBFE_DIAG
    static const int envT = getenv("ORBFE_RESOLVE_THREADS") ? atoi(getenv("ORBFE_RESOLVE_THREADS")) : 0;  // tuning experiments
    const int rt = envT ? envT : kResolveThreads;
#else
    const int rt = kResolveThreads;  // the shipped library reads no environment variable here (liborbfe_diag.so does)
#endif
    if (lds && A.kpStride <= kResNDesc && rt == kResolveThreads) {
        hipLaunchKernelGGL((proj_resolve_kernel<true, kResolveThreads, kResNDesc, true>), dim3(A.B), dim3(kResolveThreads), 0, s, A);
    } else if (rt == 256) {
        if (lds) hipLaunchKernelGGL((proj_resolve_kernel<true, 256>), dim3(A.B), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((proj_resolve_kernel<false, 256>), dim3(A.B), dim3(256), 0, s, A);
    } else {
        if (lds) hipLaunchKernelGGL((proj_resolve_kernel<true, kResolveThreads>), dim3(A.B), dim3(kResolveThreads), 0, s, A);
        else hipLaunchKernelGGL((proj_resolve_kernel<false, kResolveThreads>), dim3(A.B), dim3(kResolveThreads), 0, s, A);
    }
    MCHK(hipGetLastError());
    return ORBFE_OK;
}

}  // namespace proj

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
    if (n >= (1 << 20) || F->grid_cols > 65535 || F->grid_rows > 32767 || F->n_levels < 1 ||
        (long long)F->grid_cols * F->grid_rows > kMaxCells)
        return ORBFE_ERR_UNSUPPORTED;
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
#ifdef ORBFE_DIAG
    if (getenv("ORBFE_DEBUG_MATCH")) {  // liborbfe_diag.so only (tools/diag_match.py, tools/resolve_stats.py)
        int dbg[4] = {0, 0, 0, 0};
        (void)copy_sync(dbg, A.dbg, sizeof dbg, hipMemcpyDeviceToHost, s);
        fprintf(stderr, "[orbfe] match_projection: n=%d M=%d sweeps=%d cooperative_rescans=%d\n", n, M, dbg[0], dbg[1]);
    }
#endif
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
    if (kpStride >= (1 << 20) || gridCols > 65535 || gridRows > 32767 || (long long)gridCols * gridRows > kMaxCells)
        return ORBFE_ERR_UNSUPPORTED;
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
