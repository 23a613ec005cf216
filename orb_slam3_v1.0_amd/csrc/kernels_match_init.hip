// kernels_match_init.hip -- ORBmatcher::SearchForInitialization on gfx950.
//
// Replaces src/ORBmatcher.cc:329-439 (caller src/Tracking.cc:605-607): for every LEVEL-0 keypoint of
// frame 1, in index order, the best / second-best Hamming match among the level-0 keypoints of
// frame 2 inside a square window (Frame::GetFeaturesInArea on frame 2's grid), where a candidate
// is skipped if it already holds a match at a distance <= ours (vMatchedDistance, :368-369) and an
// accepted match STEALS the keypoint from its previous owner (:387-391); then the rotation
// histogram filter (:411-435), whose bin counts include matches that were stolen later.
//
// The running per-target minimum makes every step depend on all earlier ones, and the function
// runs once per frame pair during map initialisation on <= N_0 + 3 keypoints, so the exact design
// is the simple one: ONE block walks frame 1's level-0 keypoints sequentially; for each of them
// all threads scan frame 2 (staged in LDS: cell, octave, position, descriptor), reduce the two
// smallest keys under the total order (distance, visit position) and thread 0 applies the update.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "match_common.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

constexpr int kInitThreads = 1024;
constexpr int kInitN = 2048;  // frame-2 keypoints that fit the LDS image
constexpr int kIntMax = 0x7fffffff;
constexpr int kFastN0 = 1024;     // level-0 keypoints of frame 1 the fast kernel handles
constexpr int kFastCand = 8192;   // candidate keys of the rows longer than four that fit the LDS image

struct InitRowHdr;
struct InitArgs {
    int n1, n2;       // n2: frame 2's keypoint count -- or, with n2Dev set, its capacity (the count is still on the device)
    const int* n2Dev; // fused per-frame chain (init_track_launch): the extraction that runs in front wrote the count here
    int candStride;   // row stride of `cand` (n2 on the host path, the capacity on the fused one)
    const orbfe_keypoint* kp1;
    const uint8_t* desc1;
    const orbfe_keypoint* kp2;
    const uint8_t* desc2;
    int cols, rows;
    float minX, minY, invW, invH;
    float r;  // windowSize
    float nnRatio;
    int checkOrientation;
    int* matches12;   // [n1]
    int* nMatches;    // [1]
    int* matched21;   // [n2] scratch: vnMatches21
    int* matchedDist; // [n2] scratch: vMatchedDistance
    int* cell2;       // [n2] scratch
    int* list0;       // [n1] scratch: level-0 keypoints of frame 1 in index order
    int* binOf;       // [n1] scratch: rotation bin pushed for i1, or -1
    unsigned long long* cand;  // [n0][n2] scratch of the fast path: candidate keys per level-0 keypoint of frame 1
    struct InitRowHdr* hdr;    // [n0] row lengths + the four smallest keys
};

struct InitLds {
    int4 kp[kInitN];  // {cell, octave, x bits, y bits}
    unsigned long long desc[kInitN][4];
    int dist[kInitN];  // vMatchedDistance
};

template <bool LDS>
__global__ __launch_bounds__(kInitThreads) void init_match_kernel(InitArgs A)
{
    __shared__ InitLds S;
    __shared__ unsigned long long sK1[kInitThreads / 64], sK2[kInitThreads / 64];
    __shared__ int sHist[ORBFE_HISTO_LENGTH];
    __shared__ int sWave[kInitThreads / 64];
    __shared__ int sN0, sNm, sInd[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n1 = A.n1, n2 = A.n2Dev ? min(*A.n2Dev, A.n2) : A.n2;

    // ---- frame 2: grid cell per keypoint (Frame::PosInGrid, src/Frame.cc:470-480) + state ----
    for (int j = tid; j < n2; j += kInitThreads) {
        const orbfe_keypoint k = A.kp2[j];
        float px = k.x - A.minX; px = px * A.invW;
        float py = k.y - A.minY; py = py * A.invH;
        const int posX = (int)roundf(px), posY = (int)roundf(py);
        const int lin = posY * A.cols + posX;
        const int cell = (lin >= 0 && lin < A.cols * A.rows) ? ((lin % A.cols) | ((lin / A.cols) << 16)) : -1;
        A.cell2[j] = cell;
        A.matched21[j] = -1;
        if (LDS) {
            S.kp[j] = make_int4(cell, k.octave, __float_as_int(k.x), __float_as_int(k.y));
            S.dist[j] = kIntMax;
        } else {
            A.matchedDist[j] = kIntMax;
        }
    }
    if (LDS) {
        const unsigned long long* d2 = reinterpret_cast<const unsigned long long*>(A.desc2);
        for (int j = tid; j < n2 * 4; j += kInitThreads) S.desc[0][j] = d2[j];
    }
    if (tid < ORBFE_HISTO_LENGTH) sHist[tid] = 0;
    if (tid == 0) { sN0 = 0; sNm = 0; }
    __syncthreads();

    // ---- frame 1: ordered list of level-0 keypoints (level1 > 0 -> continue, :346-347) ----
    for (int base = 0; base < n1; base += kInitThreads) {
        const int i = base + tid;
        const int flag = (i < n1 && A.kp1[i].octave <= 0) ? 1 : 0;
        if (i < n1) { A.matches12[i] = -1; A.binOf[i] = -1; }
        int incl = flag;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sWave[wv] = incl;
        __syncthreads();
        int pos = sN0 + incl - flag;
        for (int q = 0; q < wv; q++) pos += sWave[q];
        if (flag) A.list0[pos] = i;
        __syncthreads();
        if (tid == kInitThreads - 1) sN0 = pos + flag;
        __syncthreads();
    }
    const int n0 = sN0;
    int* mdist = LDS ? S.dist : A.matchedDist;
    const float factor = 1.0f / ORBFE_HISTO_LENGTH;

    for (int t = 0; t < n0; t++) {
        const int i1 = A.list0[t];
        const orbfe_keypoint k1p = A.kp1[i1];
        // GetFeaturesInArea(x, y, windowSize, level1, level1) on frame 2, src/Frame.cc:413-435
        const float x = k1p.x, y = k1p.y, r = A.r;
        float tt;
        tt = x - A.minX; tt = tt - r; tt = tt * A.invW;
        const int minCX = max(0, (int)floorf(tt));
        tt = x - A.minX; tt = tt + r; tt = tt * A.invW;
        const int maxCX = min(A.cols - 1, (int)ceilf(tt));
        tt = y - A.minY; tt = tt - r; tt = tt * A.invH;
        const int minCY = max(0, (int)floorf(tt));
        tt = y - A.minY; tt = tt + r; tt = tt * A.invH;
        const int maxCY = min(A.rows - 1, (int)ceilf(tt));
        const bool any = !(minCX >= A.cols || maxCX < 0 || minCY >= A.rows || maxCY < 0);
        const int level1 = k1p.octave;
        unsigned long long k1 = kKeyNone, k2 = kKeyNone;
        if (any) {
            unsigned long long d4[4];
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.desc1 + (size_t)i1 * 32);
            d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
            for (int j = tid; j < n2; j += kInitThreads) {
                int cell, oct;
                float kx, ky;
                if (LDS) {
                    const int4 q = S.kp[j];
                    cell = q.x; oct = q.y; kx = __int_as_float(q.z); ky = __int_as_float(q.w);
                } else {
                    cell = A.cell2[j];
                    const orbfe_keypoint k = A.kp2[j];
                    oct = k.octave; kx = k.x; ky = k.y;
                }
                if (cell < 0) continue;
                const int cx = cell & 0xffff, cy = cell >> 16;
                if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
                // bCheckLevels = (minLevel > 0) || (maxLevel >= 0) with minLevel = maxLevel = level1
                const bool checkLevels = (level1 > 0) || (level1 >= 0);
                if (checkLevels && (oct < level1 || (level1 >= 0 && oct > level1))) continue;
                const float dx = kx - x, dy = ky - y;
                if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                int dist;
                if (LDS)
                    dist = __popcll(S.desc[j][0] ^ d4[0]) + __popcll(S.desc[j][1] ^ d4[1]) + __popcll(S.desc[j][2] ^ d4[2]) +
                           __popcll(S.desc[j][3] ^ d4[3]);
                else
                    dist = hamming256(reinterpret_cast<const uint2*>(A.desc2 + (size_t)j * 32), d4);
                if (mdist[j] <= dist) continue;  // :368-369
                const unsigned long long key = ((unsigned long long)dist << 52) | ((unsigned long long)cx << 36) |
                                               ((unsigned long long)cy << 20) | (unsigned long long)j;
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
        }
        wave_top2(k1, k2);
        if (lane == 0) { sK1[wv] = k1; sK2[wv] = k2; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long b1 = kKeyNone, b2 = kKeyNone;
            for (int q = 0; q < kInitThreads / 64; q++) top2_merge(b1, b2, sK1[q], sK2[q]);
            if (b1 != kKeyNone) {
                const int bestDist = (int)(b1 >> 52), bestIdx2 = (int)(b1 & 0xFFFFF);
                const int bestDist2 = b2 == kKeyNone ? kIntMax : (int)(b2 >> 52);
                if (bestDist <= ORBFE_TH_LOW && (float)bestDist < (float)bestDist2 * A.nnRatio) {  // :383-385
                    const int prev = A.matched21[bestIdx2];
                    if (prev >= 0) { A.matches12[prev] = -1; sNm--; }
                    A.matches12[i1] = bestIdx2;
                    A.matched21[bestIdx2] = i1;
                    mdist[bestIdx2] = bestDist;
                    sNm++;
                    if (A.checkOrientation) {
                        float rot = k1p.angle - A.kp2[bestIdx2].angle;
                        if (rot < 0.0) rot = rot + 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == ORBFE_HISTO_LENGTH) bin = 0;
                        sHist[bin]++;
                        A.binOf[i1] = bin;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- rotation histogram filter (:411-435; ComputeThreeMaxima :1328-1370) ----
    if (A.checkOrientation) {
        if (tid == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; i++) {
                const int s = sHist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
        }
        __syncthreads();
        for (int i = tid; i < n1; i += kInitThreads) {
            const int b = A.binOf[i];
            if (b >= 0 && b != sInd[0] && b != sInd[1] && b != sInd[2] && A.matches12[i] >= 0) {
                A.matches12[i] = -1;
                atomicSub(&sNm, 1);
            }
        }
        __syncthreads();
    }
    if (tid == 0) *A.nMatches = sNm;
}


// ---------------------------------------------------------------------------------------------
// Fast variant (frame 2 <= kInitN keypoints, frame 1 <= kFastN0 level-0 keypoints).  The distances do not depend
// on the running state, only the skip test "vMatchedDistance[i2] <= dist" (:368-369) does.  So
//   init_cand_kernel   (one WAVE per level-0 keypoint of frame 1, spread over the chip): window + level tests and
//       the Hamming distance for every frame-2 keypoint -> candidate keys (distance, cell x, cell y, index) in a
//       row per keypoint, plus the row's four smallest keys, sorted;
//   init_order_kernel  (one block; the order-dependent part runs in ONE wave, keypoints in index order): a
//       candidate is skipped iff its target already holds a match at a distance <= its own, so the best and
//       second-best of a step are the first two of the row's sorted keys that pass.  Lanes 0-3 test the four
//       pre-selected keys: if two pass (or the row has no more than four) the step is decided by a ballot; else
//       the whole row (in LDS when all rows fit) is wave-reduced.  Lane 0 applies the accept / steal update of
//       :383-406.  No block barrier inside the ordered loop.
// Same keys, same order, same float comparisons as init_match_kernel: the results are identical.
// ---------------------------------------------------------------------------------------------
struct InitRowHdr {
    int cnt, pad;
    unsigned long long top[4];  // the row's smallest keys, ascending, kKeyNone-padded
};

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void init_cand_kernel(InitArgs A, int n0)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (int)(threadIdx.x >> 6);  // row == position in frame 1's level-0 list
    if (t >= n0) return;
    const int n2 = A.n2Dev ? min(*A.n2Dev, A.n2) : A.n2;
    const int i1 = A.list0[t];
    const orbfe_keypoint k1p = A.kp1[i1];
    // GetFeaturesInArea(x, y, windowSize, level1, level1) on frame 2, src/Frame.cc:413-435
    const float x = k1p.x, y = k1p.y, r = A.r;
    float tt;
    tt = x - A.minX; tt = tt - r; tt = tt * A.invW;
    const int minCX = max(0, (int)floorf(tt));
    tt = x - A.minX; tt = tt + r; tt = tt * A.invW;
    const int maxCX = min(A.cols - 1, (int)ceilf(tt));
    tt = y - A.minY; tt = tt - r; tt = tt * A.invH;
    const int minCY = max(0, (int)floorf(tt));
    tt = y - A.minY; tt = tt + r; tt = tt * A.invH;
    const int maxCY = min(A.rows - 1, (int)ceilf(tt));
    const bool any = !(minCX >= A.cols || maxCX < 0 || minCY >= A.rows || maxCY < 0);
    const int level1 = k1p.octave;
    int count = 0;
    unsigned long long top[4] = {kKeyNone, kKeyNone, kKeyNone, kKeyNone};
    if (any) {
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.desc1 + (size_t)i1 * 32);
        const unsigned long long d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
        unsigned long long* row = A.cand + (size_t)t * A.candStride;
        for (int j0 = 0; j0 < n2; j0 += 64) {
            const int j = j0 + lane;
            bool ok = j < n2;
            unsigned long long key = kKeyNone;
            if (ok) {
                const orbfe_keypoint k = A.kp2[j];
                // Frame::PosInGrid (src/Frame.cc:470-480): only the LINEAR index is validated
                float px = k.x - A.minX; px = px * A.invW;
                float py = k.y - A.minY; py = py * A.invH;
                const int posX = (int)roundf(px), posY = (int)roundf(py);
                const int lin = posY * A.cols + posX;
                ok = lin >= 0 && lin < A.cols * A.rows;
                const int cx = ok ? lin % A.cols : 0, cy = ok ? lin / A.cols : 0;
                if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) ok = false;
                // bCheckLevels = (minLevel > 0) || (maxLevel >= 0) with minLevel = maxLevel = level1
                const bool checkLevels = (level1 > 0) || (level1 >= 0);
                if (checkLevels && (k.octave < level1 || (level1 >= 0 && k.octave > level1))) ok = false;
                const float dx = k.x - x, dy = k.y - y;
                if (!(fabsf(dx) < r && fabsf(dy) < r)) ok = false;
                if (ok) {
                    const unsigned long long* kd = reinterpret_cast<const unsigned long long*>(A.desc2 + (size_t)j * 32);
                    const int dist = __popcll(kd[0] ^ d0) + __popcll(kd[1] ^ d1) + __popcll(kd[2] ^ d2) + __popcll(kd[3] ^ d3);
                    key = ((unsigned long long)dist << 52) | ((unsigned long long)cx << 36) | ((unsigned long long)cy << 20) |
                          (unsigned long long)j;
                }
            }
            unsigned long long mask = __ballot(ok);
            if (ok) row[count + __popcll(mask & ((1ull << lane) - 1ull))] = key;
            count += __popcll(mask);
            while (mask) {  // wave-uniform: fold the chunk's keys into the sorted four
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                unsigned long long kk = readlane_u64(key, l);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned long long lo = kk < top[q] ? kk : top[q];
                    kk = kk < top[q] ? top[q] : kk;
                    top[q] = lo;
                }
            }
        }
    }
    if (lane == 0) {
        InitRowHdr h;
        h.cnt = count;
        h.pad = 0;
        h.top[0] = top[0]; h.top[1] = top[1]; h.top[2] = top[2]; h.top[3] = top[3];
        A.hdr[t] = h;
    }
}

struct FastState {
    unsigned short dist[kInitN];    // vMatchedDistance (0xFFFF == INT_MAX)
    short matched21[kInitN];        // vnMatches21 (position in the level-0 list)
    float angle2[kInitN];
    int i1[kFastN0];                // level-0 keypoints of frame 1 in index order
    float angle1[kFastN0];
    int off[kFastN0];               // row offsets in sCand; after the ordered loop: the rotation bin of the row's accepted match
    int cnt[kFastN0];
    short acc[kFastN0];             // the frame-2 keypoint row t accepted (:383-406), -1 = none -- whoever owns it in the end
    unsigned long long top[kFastN0][4];
};

__global__ __launch_bounds__(kInitThreads) void init_order_kernel(InitArgs A, int n0)
{
    __shared__ unsigned long long sCand[kFastCand];
    __shared__ FastState T;
    __shared__ int sHist[ORBFE_HISTO_LENGTH];
    __shared__ int sWave[kInitThreads / 64];
    __shared__ int sNm, sTotal, sInd[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n1 = A.n1, n2 = A.n2Dev ? min(*A.n2Dev, A.n2) : A.n2;
    for (int j = tid; j < n2; j += kInitThreads) {
        T.dist[j] = 0xFFFF;
        T.matched21[j] = -1;
        T.angle2[j] = A.kp2[j].angle;
    }
    for (int i = tid; i < n1; i += kInitThreads) A.matches12[i] = -1;
    for (int t = tid; t < n0; t += kInitThreads) {
        const int i1 = A.list0[t];
        const InitRowHdr h = A.hdr[t];
        T.i1[t] = i1;
        T.angle1[t] = A.kp1[i1].angle;
        T.cnt[t] = h.cnt;
        T.acc[t] = -1;
        T.top[t][0] = h.top[0]; T.top[t][1] = h.top[1]; T.top[t][2] = h.top[2]; T.top[t][3] = h.top[3];
    }
    if (tid < ORBFE_HISTO_LENGTH) sHist[tid] = 0;
    if (tid == 0) sNm = 0;
    __syncthreads();
    // ---- row offsets (only rows longer than four are ever read in full); copy them into LDS when they fit ----
    {
        int run = 0;
        for (int base = 0; base < n0; base += kInitThreads) {
            const int t = base + tid;
            int c = t < n0 ? T.cnt[t] : 0;
            if (c <= 4) c = 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) sWave[wv] = incl;
            __syncthreads();
            int pos = run + incl - c;
            int tot = 0;
            for (int q = 0; q < kInitThreads / 64; q++) {
                if (q < wv) pos += sWave[q];
                tot += sWave[q];
            }
            if (t < n0) T.off[t] = pos;
            run += tot;
            __syncthreads();
        }
        if (tid == 0) sTotal = run;
        __syncthreads();
    }
    const bool inLds = sTotal <= kFastCand;
    if (inLds) {
        for (int t = wv; t < n0; t += kInitThreads / 64) {
            const int c = T.cnt[t], o = T.off[t];
            if (c <= 4) continue;
            const unsigned long long* row = A.cand + (size_t)t * A.candStride;
            for (int k = lane; k < c; k += 64) sCand[o + k] = row[k];
        }
    }
    __syncthreads();

    // ---- the order-dependent part, one wave.  The running state is vMatchedDistance and vnMatches21 alone: vnMatches12, the
    //      match count and the rotation histogram follow from "which row accepted which keypoint" (T.acc) and "who owns it in
    //      the end" (T.matched21) and are derived by all threads afterwards -- a row accepts at most once, its match survives iff
    //      nobody took the keypoint later (:387-391), and the histogram counts every accept, stolen later or not (:395-405).
    //      A step is then one dependent LDS read (the targets' distances) between the prefetched row header and three LDS
    //      writes by lane 0; the LDS executes a wave's operations in order, so the next step's reads see them ----
    if (wv == 0 && n0 > 0) {
        unsigned long long keyN = lane < 4 ? T.top[0][lane] : kKeyNone;
        int cN = T.cnt[0];
        for (int t = 0; t < n0; t++) {
            const unsigned long long key = keyN;
            const int c = cN;
            if (t + 1 < n0) {  // the next row's header does not depend on the state: requested a step ahead
                keyN = lane < 4 ? T.top[t + 1][lane] : kKeyNone;
                cN = T.cnt[t + 1];
            }
            if (c == 0) continue;  // wave-uniform
            unsigned long long k1 = kKeyNone, k2 = kKeyNone;
            {
                bool pass = false;
                if (key != kKeyNone) {
                    const int j = (int)(key & 0xFFFFF), dist = (int)(key >> 52);
                    pass = !((int)T.dist[j] <= dist);  // :368-369 (0xFFFF stands for INT_MAX: never <= a distance)
                }
                const unsigned long long m = __ballot(pass);
                if (__popcll(m) >= 2 || c <= 4) {
                    if (m) {
                        k1 = readlane_u64(key, __builtin_ctzll(m));
                        const unsigned long long m2 = m & (m - 1);
                        if (m2) k2 = readlane_u64(key, __builtin_ctzll(m2));
                    }
                } else {  // fewer than two of the pre-selected keys are usable: reduce the whole row
                    const unsigned long long* L = inLds ? sCand + T.off[t] : A.cand + (size_t)t * A.candStride;
                    for (int k = lane; k < c; k += 64) {
                        const unsigned long long kk = L[k];
                        const int j = (int)(kk & 0xFFFFF), dist = (int)(kk >> 52);
                        if ((int)T.dist[j] <= dist) continue;
                        if (kk < k1) { k2 = k1; k1 = kk; }
                        else if (kk < k2) k2 = kk;
                    }
                    wave_top2(k1, k2);
                }
            }
            if (lane == 0 && k1 != kKeyNone) {
                const int bestDist = (int)(k1 >> 52), bestIdx2 = (int)(k1 & 0xFFFFF);
                const int bestDist2 = k2 == kKeyNone ? kIntMax : (int)(k2 >> 52);
                if (bestDist <= ORBFE_TH_LOW && (float)bestDist < (float)bestDist2 * A.nnRatio) {  // :383-385
                    T.matched21[bestIdx2] = (short)t;  // position in the level-0 list (< kFastN0); the previous owner loses it
                    T.dist[bestIdx2] = (unsigned short)bestDist;
                    T.acc[t] = (short)bestIdx2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // lane 0's LDS updates stay in front of the next step's reads
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- rotation histogram over every accept (:395-405), ComputeThreeMaxima (:1328-1370) ----
    if (A.checkOrientation) {
        const float factor = 1.0f / ORBFE_HISTO_LENGTH;
        for (int t = tid; t < n0; t += kInitThreads) {
            const int j = T.acc[t];
            if (j < 0) continue;
            float rot = T.angle1[t] - T.angle2[j];
            if (rot < 0.0) rot = rot + 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            atomicAdd(&sHist[bin], 1);
            T.off[t] = bin;  // (the row offsets are no longer needed)
        }
        __syncthreads();
        if (tid == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; i++) {
                const int s = sHist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
        }
        __syncthreads();
    }
    // ---- vnMatches12 and the count: keypoint j of frame 2 belongs to the row that owns it in the end, unless the rotation
    //      filter drops that row's bin (:411-435) ----
    int local = 0;
    for (int j = tid; j < n2; j += kInitThreads) {
        const int t = T.matched21[j];
        if (t < 0) continue;
        if (A.checkOrientation) {
            const int b = T.off[t];
            if (b != sInd[0] && b != sInd[1] && b != sInd[2]) continue;
        }
        A.matches12[T.i1[t]] = j;
        local++;
    }
    if (local) atomicAdd(&sNm, local);
    __syncthreads();
    if (tid == 0) *A.nMatches = sNm;
}

}  // namespace

int match_initialization_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F1, const orbfe_frame_view* F2,
                             int windowSize, float nnRatio, int checkOrientation, int* matches12Out, int* nMatches,
                             std::string& err)
{
    const int n1 = F1->n, n2 = F2->n;
    for (int i = 0; i < n1; i++) matches12Out[i] = -1;
    *nMatches = 0;
    if (n1 == 0 || n2 == 0) return ORBFE_OK;
    if (n2 >= (1 << 20) || F2->grid_cols > 65535 || F2->grid_rows > 32767) return ORBFE_ERR_UNSUPPORTED;
    Carver c;
    const size_t oKp1 = c.take((size_t)n1 * sizeof(orbfe_keypoint));
    const size_t oD1 = c.take((size_t)n1 * 32);
    const size_t oKp2 = c.take((size_t)n2 * sizeof(orbfe_keypoint));
    const size_t oD2 = c.take((size_t)n2 * 32);
    const size_t oL0 = c.take((size_t)n1 * sizeof(int));  // uploaded for the fast path, written by the sequential kernel otherwise
    const size_t inBytes = c.off;
    const size_t oM12 = c.take((size_t)n1 * sizeof(int));
    const size_t oNM = c.take(sizeof(int));
    const size_t outBytes = c.off - oM12;
    const size_t oM21 = c.take((size_t)n2 * sizeof(int));
    const size_t oMD = c.take((size_t)n2 * sizeof(int));
    const size_t oCell = c.take((size_t)n2 * sizeof(int));
    const size_t oBin = c.take((size_t)n1 * sizeof(int));
    int n0 = 0;  // level-0 keypoints of frame 1 (:346-347): selects the kernel and sizes its candidate scratch
    for (int i = 0; i < n1; i++) n0 += F1->kp[i].octave <= 0;
    bool fast = n2 <= kInitN && n0 <= kFastN0;
    const size_t oCand = c.take(fast ? (size_t)std::max(n0, 1) * n2 * sizeof(unsigned long long) : 8);
    const size_t oHdr = c.take(fast ? (size_t)std::max(n0, 1) * sizeof(InitRowHdr) : 8);
    int rc = ensure(m, c.off, inBytes + outBytes + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oKp1, F1->kp, (size_t)n1 * sizeof(orbfe_keypoint));
    memcpy(hp + oD1, F1->desc, (size_t)n1 * 32);
    memcpy(hp + oKp2, F2->kp, (size_t)n2 * sizeof(orbfe_keypoint));
    memcpy(hp + oD2, F2->desc, (size_t)n2 * 32);
    if (fast) {  // row t of the candidate pass = t-th level-0 keypoint of frame 1 (level1 > 0 -> continue, :346-347)
        int* l0 = reinterpret_cast<int*>(hp + oL0);
        for (int i = 0, t = 0; i < n1; i++)
            if (F1->kp[i].octave <= 0) l0[t++] = i;
    }
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    InitArgs A{};
    A.n1 = n1; A.n2 = n2;
    A.n2Dev = nullptr;
    A.candStride = n2;
    A.kp1 = reinterpret_cast<const orbfe_keypoint*>(dp + oKp1);
    A.desc1 = dp + oD1;
    A.kp2 = reinterpret_cast<const orbfe_keypoint*>(dp + oKp2);
    A.desc2 = dp + oD2;
    A.cols = F2->grid_cols; A.rows = F2->grid_rows;
    A.minX = F2->min_x; A.minY = F2->min_y; A.invW = F2->grid_inv_w; A.invH = F2->grid_inv_h;
    A.r = (float)windowSize;
    A.nnRatio = nnRatio;
    A.checkOrientation = checkOrientation;
    A.matches12 = reinterpret_cast<int*>(dp + oM12);
    A.nMatches = reinterpret_cast<int*>(dp + oNM);
    A.matched21 = reinterpret_cast<int*>(dp + oM21);
    A.matchedDist = reinterpret_cast<int*>(dp + oMD);
    A.cell2 = reinterpret_cast<int*>(dp + oCell);
    A.list0 = reinterpret_cast<int*>(dp + oL0);
    A.binOf = reinterpret_cast<int*>(dp + oBin);
    A.cand = reinterpret_cast<unsigned long long*>(dp + oCand);
    A.hdr = reinterpret_cast<InitRowHdr*>(dp + oHdr);
#ifdef ORBFE_DIAG
    if (getenv("ORBFE_INIT_SLOW")) fast = false;  // liborbfe_diag.so only: force the sequential block kernel
#endif
    if (fast) {
        if (n0 > 0) hipLaunchKernelGGL(init_cand_kernel, dim3((n0 + 3) / 4), dim3(256), 0, s, A, n0);
        hipLaunchKernelGGL(init_order_kernel, dim3(1), dim3(kInitThreads), 0, s, A, n0);
    }
    else if (n2 <= kInitN)
        hipLaunchKernelGGL(init_match_kernel<true>, dim3(1), dim3(kInitThreads), 0, s, A);
    else
        hipLaunchKernelGGL(init_match_kernel<false>, dim3(1), dim3(kInitThreads), 0, s, A);
    MCHK(hipGetLastError());
    uint8_t* hOut = hp + inBytes;
    MCHK(hipMemcpyAsync(hOut, dp + oM12, outBytes, hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(matches12Out, hOut, (size_t)n1 * sizeof(int));
    *nMatches = *reinterpret_cast<int*>(hOut + (oNM - oM12));
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------
// The fused form (orbfe_track_initialization): frame 1 is the RESIDENT initial frame (mInitialFrame, Tracking.cc:569-602),
// frame 2 the current frame whose keypoints, descriptors and count the extraction chain wrote a kernel earlier.  No copy,
// no synchronisation: capturable.  The same kernels as match_initialization_run, with frame 2's count read on the device.
// ---------------------------------------------------------------------------------------------
namespace {
struct InitTrackCarve {
    size_t oM21, oMD, oCell, oBin, oCand, oHdr, bytes;
    bool fast;
};
InitTrackCarve init_track_carve(int n1, int n0, int cap)
{
    InitTrackCarve C{};
    Carver c;
    C.fast = cap <= kInitN && n0 <= kFastN0;
    C.oM21 = c.take((size_t)cap * sizeof(int));
    C.oMD = c.take((size_t)cap * sizeof(int));
    C.oCell = c.take((size_t)cap * sizeof(int));
    C.oBin = c.take((size_t)std::max(n1, 1) * sizeof(int));
    C.oCand = c.take(C.fast ? (size_t)std::max(n0, 1) * cap * sizeof(unsigned long long) : 8);
    C.oHdr = c.take(C.fast ? (size_t)std::max(n0, 1) * sizeof(InitRowHdr) : 8);
    C.bytes = c.off;
    return C;
}
}  // namespace

size_t init_track_scratch_bytes(int n1, int n0, int cap) { return init_track_carve(n1, n0, cap).bytes; }

int init_track_launch(hipStream_t s, int n1, int n0, const orbfe_keypoint* dKp1, const uint8_t* dDesc1, const int* dList0,
                      const orbfe_keypoint* dKp2, const uint8_t* dDesc2, const int* dN2, int cap, int gridCols, int gridRows,
                      float minX, float minY, float invW, float invH, int windowSize, float nnRatio, int checkOrientation,
                      int* dMatches12, int* dNMatches, uint8_t* scratch, std::string& err)
{
    if (cap >= (1 << 20) || gridCols > 65535 || gridRows > 32767) return ORBFE_ERR_UNSUPPORTED;
    const InitTrackCarve C = init_track_carve(n1, n0, cap);
    InitArgs A{};
    A.n1 = n1; A.n2 = cap;
    A.n2Dev = dN2;
    A.candStride = cap;
    A.kp1 = dKp1; A.desc1 = dDesc1;
    A.kp2 = dKp2; A.desc2 = dDesc2;
    A.cols = gridCols; A.rows = gridRows;
    A.minX = minX; A.minY = minY; A.invW = invW; A.invH = invH;
    A.r = (float)windowSize;
    A.nnRatio = nnRatio;
    A.checkOrientation = checkOrientation;
    A.matches12 = dMatches12;
    A.nMatches = dNMatches;
    A.matched21 = reinterpret_cast<int*>(scratch + C.oM21);
    A.matchedDist = reinterpret_cast<int*>(scratch + C.oMD);
    A.cell2 = reinterpret_cast<int*>(scratch + C.oCell);
    A.list0 = const_cast<int*>(dList0);  // (read only on the fast path; the sequential kernel rebuilds the same list in place)
    A.binOf = reinterpret_cast<int*>(scratch + C.oBin);
    A.cand = reinterpret_cast<unsigned long long*>(scratch + C.oCand);
    A.hdr = reinterpret_cast<InitRowHdr*>(scratch + C.oHdr);
    if (n1 == 0) {  // nothing to match: the count only (matches12 has no entries)
        MCHK(hipMemsetAsync(dNMatches, 0, sizeof(int), s));
        return ORBFE_OK;
    }
    if (C.fast) {
        if (n0 > 0) hipLaunchKernelGGL(init_cand_kernel, dim3((n0 + 3) / 4), dim3(256), 0, s, A, n0);
        hipLaunchKernelGGL(init_order_kernel, dim3(1), dim3(kInitThreads), 0, s, A, n0);
    } else if (cap <= kInitN) {
        hipLaunchKernelGGL(init_match_kernel<true>, dim3(1), dim3(kInitThreads), 0, s, A);
    } else {
        hipLaunchKernelGGL(init_match_kernel<false>, dim3(1), dim3(kInitThreads), 0, s, A);
    }
    MCHK(hipGetLastError());
    return ORBFE_OK;
}

}  // namespace orbfe
