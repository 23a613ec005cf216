// kernels_match.hip -- 256-bit Hamming matchers on gfx950.
//
// Replaces the hot loops of ORBmatcher::SearchByProjection(Frame, MapPoints, ...)
// (src/ORBmatcher.cc:31-123) incl. Frame::GetFeaturesInArea / PosInGrid (src/Frame.cc:404-480),
// and of ORBmatcher::SearchByBoW (src/ORBmatcher.cc:133-327) incl. ComputeThreeMaxima (:1328-1370).
// Distances are __popcll over 4 x 64-bit XOR words (== DescriptorDistance :1375-1391); min scans
// are wavefront shuffles.  No MFMA: the path is bitwise/integer.
//
// Exactness of the order-dependent parts:
//  * The reference's sequential "best / second best" scan (:92-104) returns the two smallest
//    candidates under the total order (distance, visit position), where the visit position of
//    GetFeaturesInArea is (cell x, cell y, keypoint index) -- so a wave can reduce it in any order.
//  * SearchByProjection is greedy: a keypoint claimed by an earlier map point (with observations)
//    is skipped by later ones (:77-79).  Let claim[idx] = smallest map-point index whose accepted
//    best match is idx.  Evaluating every map point i against "idx is free iff claim[idx] >= i" and
//    rebuilding claim[] from the results, repeated until nothing changes, reaches a fixed point;
//    by induction over i every fixed point equals the sequential result.  Each iteration is fully
//    parallel (one wave per map point).
//  * SearchByBoW's claims never cross vocabulary nodes (a frame feature belongs to one node), so one
//    wave walks the key-frame features of a node sequentially and scans the node's frame features
//    in parallel.
#include <algorithm>
#include <cstring>
#include <vector>

#include "match.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

constexpr int kClaimFree = 0x7fffffff;
constexpr unsigned long long kKeyNone = ~0ull;

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
    int* claim;                   // [B][kpStride]
    int* resA;                    // [B][M]
    int* resB;                    // [B][M]
    int* changed;                 // [1]
    int* matchOut;                // [B][kpStride]
    int* nMatches;                // [B]
};

__device__ __forceinline__ int hamming256(const uint2* a, const unsigned long long* b4)
{
    const unsigned long long* a4 = reinterpret_cast<const unsigned long long*>(a);
    return __popcll(a4[0] ^ b4[0]) + __popcll(a4[1] ^ b4[1]) + __popcll(a4[2] ^ b4[2]) + __popcll(a4[3] ^ b4[3]);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((unsigned long long)hi << 32) | lo;
}

// merge two (smallest, second smallest) pairs
__device__ __forceinline__ void top2_merge(unsigned long long& k1, unsigned long long& k2, unsigned long long o1,
                                           unsigned long long o2)
{
    const unsigned long long lo = k1 < o1 ? k1 : o1;
    const unsigned long long hi = k1 < o1 ? o1 : k1;
    const unsigned long long s2 = k2 < o2 ? k2 : o2;
    k1 = lo;
    k2 = hi < s2 ? hi : s2;
}

__device__ __forceinline__ void wave_top2(unsigned long long& k1, unsigned long long& k2)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o1 = shfl_xor_u64(k1, d), o2 = shfl_xor_u64(k2, d);
        top2_merge(k1, k2, o1, o2);
    }
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
    if (i == 0) A.nMatches[f] = 0;
}

// claim[idx] = -1 if the slot holds a map point with observations on entry (:77-79), else the
// smallest accepted map point index (with observations) whose best match is idx.
__global__ __launch_bounds__(256) void proj_claims_kernel(ProjArgs A, const int* __restrict__ res)
{
    const int f = blockIdx.x;
    const int n = A.nKp[f];
    int* claim = A.claim + (size_t)f * A.kpStride;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        claim[i] = (A.initObs && A.initObs[(size_t)f * A.kpStride + i] > 0) ? -1 : kClaimFree;
    __syncthreads();
    if (res) {
        for (int i = threadIdx.x; i < A.M; i += blockDim.x) {
            const int r = res[(size_t)f * A.M + i];
            if (r >= 0 && A.mps[(size_t)f * A.M + i].observations > 0) atomicMin(&claim[r], i);
        }
    }
}

// one wave per map point
__global__ __launch_bounds__(256) void proj_match_kernel(ProjArgs A, const int* __restrict__ resPrev,
                                                         int* __restrict__ resOut)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= A.M) return;
    const orbfe_map_point mp = A.mps[(size_t)f * A.M + i];
    int result = -1;
    const bool valid = mp.in_view && !(A.farPoints && mp.track_depth > A.thFar) && !mp.bad;  // :40-47
    if (valid) {
        const int lvl = mp.level;
        float r = mp.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :125-131
        if (A.bFactor) r = r * A.th;
        r = r * A.scaleFactors[lvl];
        // GetFeaturesInArea cell range, src/Frame.cc:413-435
        const float x = mp.proj_x, y = mp.proj_y;
        float t;
        t = x - A.g.minX; t = t - r; t = t * A.g.invW;
        const int minCX = max(0, (int)floorf(t));
        t = x - A.g.minX; t = t + r; t = t * A.g.invW;
        const int maxCX = min(A.g.cols - 1, (int)ceilf(t));
        t = y - A.g.minY; t = t - r; t = t * A.g.invH;
        const int minCY = max(0, (int)floorf(t));
        t = y - A.g.minY; t = t + r; t = t * A.g.invH;
        const int maxCY = min(A.g.rows - 1, (int)ceilf(t));
        const bool any = !(minCX >= A.g.cols || maxCX < 0 || minCY >= A.g.rows || maxCY < 0);
        const int minLevel = lvl - 1, maxLevel = lvl;
        const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);  // :437

        unsigned long long d4[4];
        {
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.mpDesc + ((size_t)f * A.M + i) * 32);
            d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
        }
        unsigned long long k1 = kKeyNone, k2 = kKeyNone;
        if (any) {
            const int n = A.nKp[f];
            const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
            const int* cellXY = A.cellXY + (size_t)f * A.kpStride;
            const int* claim = A.claim + (size_t)f * A.kpStride;
            const uint8_t* desc = A.desc + (size_t)f * A.kpStride * 32;
            for (int idx = lane; idx < n; idx += 64) {
                const int cell = cellXY[idx];
                if (cell < 0) continue;
                const int cx = cell & 0xffff, cy = cell >> 16;
                if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
                const orbfe_keypoint k = kp[idx];
                if (checkLevels && (k.octave < minLevel || (maxLevel >= 0 && k.octave > maxLevel))) continue;
                const float dx = k.x - x, dy = k.y - y;
                if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;       // :461
                if (claim[idx] < i) continue;                           // :77-79 (greedy claim)
                const int dist = hamming256(reinterpret_cast<const uint2*>(desc + (size_t)idx * 32), d4);
                if (dist >= 256) continue;  // can enter neither slot (initial bests are 256)
                const unsigned long long key = ((unsigned long long)dist << 52) | ((unsigned long long)cx << 36) |
                                               ((unsigned long long)cy << 20) | (unsigned long long)idx;
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
        }
        wave_top2(k1, k2);
        if (k1 != kKeyNone) {
            const orbfe_keypoint* kp = A.kp + (size_t)f * A.kpStride;
            const int bestDist = (int)(k1 >> 52), bestIdx = (int)(k1 & 0xFFFFF);
            const int bestLevel = kp[bestIdx].octave;
            int bestDist2 = 256, bestLevel2 = -1;
            if (k2 != kKeyNone) {
                bestDist2 = (int)(k2 >> 52);
                bestLevel2 = kp[(int)(k2 & 0xFFFFF)].octave;
            }
            if (bestDist <= ORBFE_TH_HIGH) {  // :108-117
                const bool reject = bestLevel == bestLevel2 && (float)bestDist > A.nnRatio * (float)bestDist2;
                if (!reject) result = bestIdx;
            }
        }
    }
    if (lane == 0) {
        resOut[(size_t)f * A.M + i] = result;
        if (resPrev[(size_t)f * A.M + i] != result) atomicAdd(A.changed, 1);
    }
}

// F->mvpMapPoints[bestIdx] = pMP in map-point order: the last writer wins; nmatches counts accepts.
__global__ __launch_bounds__(256) void proj_finalize_kernel(ProjArgs A, const int* __restrict__ res)
{
    const int f = blockIdx.x;
    int local = 0;
    for (int i = threadIdx.x; i < A.M; i += blockDim.x) {
        const int r = res[(size_t)f * A.M + i];
        if (r >= 0) {
            atomicMax(&A.matchOut[(size_t)f * A.kpStride + r], i);
            local++;
        }
    }
    if (local) atomicAdd(&A.nMatches[f], local);
}

__global__ void fill_kernel(int* p, int v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// SearchByBoW
// ------------------------------------------------------------------------------------------------
struct BowArgs {
    int G;
    const int *kfOff, *kfIdx, *fOff, *fIdx;
    const uint8_t *kfDesc, *fDesc, *kfHasMP;
    const float *kfAngle, *fAngle;
    int nF;
    float nnRatio;
    int checkOrientation;
    int* matchOut;   // [nF], -1 initialised
    int* binOf;      // [nF]
    int* nMatches;   // [1]
};

__global__ __launch_bounds__(256) void bow_match_kernel(BowArgs A)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= A.G) return;
    const int k0 = A.kfOff[g], k1e = A.kfOff[g + 1];
    const int f0 = A.fOff[g], f1 = A.fOff[g + 1];
    const float factor = 1.0f / ORBFE_HISTO_LENGTH;
    for (int iKF = k0; iKF < k1e; iKF++) {  // sequential: later KF features skip matched frame features (:188)
        const int realIdxKF = A.kfIdx[iKF];
        if (!A.kfHasMP[realIdxKF]) continue;
        unsigned long long d4[4];
        {
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.kfDesc + (size_t)realIdxKF * 32);
            d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
        }
        unsigned long long k1 = kKeyNone, k2 = kKeyNone;
        for (int iF = f0 + lane; iF < f1; iF += 64) {
            const int realIdxF = A.fIdx[iF];
            if (A.matchOut[realIdxF] >= 0) continue;
            const int dist = hamming256(reinterpret_cast<const uint2*>(A.fDesc + (size_t)realIdxF * 32), d4);
            if (dist >= 256) continue;
            const unsigned long long key = ((unsigned long long)dist << 32) | (unsigned)(iF - f0);
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
        }
        wave_top2(k1, k2);
        if (k1 == kKeyNone) continue;
        const int bestDist1 = (int)(k1 >> 32);
        const int bestDist2 = k2 == kKeyNone ? 256 : (int)(k2 >> 32);
        if (bestDist1 <= ORBFE_TH_LOW && (float)bestDist1 < A.nnRatio * (float)bestDist2) {  // :237-239
            const int bestIdxF = A.fIdx[f0 + (int)(k1 & 0xffffffffu)];
            if (lane == 0) {
                A.matchOut[bestIdxF] = realIdxKF;
                if (A.checkOrientation) {
                    float rot = A.kfAngle[realIdxKF] - A.fAngle[bestIdxF];
                    if (rot < 0.0) rot = rot + 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == ORBFE_HISTO_LENGTH) bin = 0;
                    A.binOf[bestIdxF] = bin;
                }
            }
            __threadfence_block();  // later iterations of this wave read matchOut
        }
    }
}

// rotation-histogram filter (:304-322) + count; single block
__global__ __launch_bounds__(256) void bow_finalize_kernel(BowArgs A)
{
    __shared__ int hist[ORBFE_HISTO_LENGTH];
    __shared__ int sInd[3];
    __shared__ int sCount;
    const int tid = threadIdx.x;
    if (tid < ORBFE_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sCount = 0;
    __syncthreads();
    int local = 0;
    for (int j = tid; j < A.nF; j += blockDim.x)
        if (A.matchOut[j] >= 0) {
            local++;
            if (A.checkOrientation) atomicAdd(&hist[A.binOf[j]], 1);
        }
    __syncthreads();
    if (tid == 0) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        if (A.checkOrientation) {  // ComputeThreeMaxima :1328-1370
            int max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        }
        sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
    }
    __syncthreads();
    if (A.checkOrientation) {
        for (int j = tid; j < A.nF; j += blockDim.x)
            if (A.matchOut[j] >= 0) {
                const int b = A.binOf[j];
                if (b != sInd[0] && b != sInd[1] && b != sInd[2]) {
                    A.matchOut[j] = -1;
                    local--;
                }
            }
    }
    if (local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) *A.nMatches = sCount;
}

// grow-only arenas
int ensure(MatchScratch& m, size_t dBytes, size_t hBytes, std::string& err)
{
    if (dBytes > m.dBytes) {
        if (m.d) (void)hipFree(m.d);
        m.d = nullptr;
        m.dBytes = 0;
        const size_t want = dBytes + dBytes / 2;
        if (hipMalloc(&m.d, want) != hipSuccess) { err = "hipMalloc(match scratch) failed"; return ORBFE_ERR_OUT_OF_MEMORY; }
        m.dBytes = want;
    }
    if (hBytes > m.hBytes) {
        if (m.hpin) (void)hipHostFree(m.hpin);
        m.hpin = nullptr;
        m.hBytes = 0;
        const size_t want = hBytes + hBytes / 2;
        if (hipHostMalloc(&m.hpin, want) != hipSuccess) { err = "hipHostMalloc(match scratch) failed"; return ORBFE_ERR_OUT_OF_MEMORY; }
        m.hBytes = want;
    }
    return ORBFE_OK;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; }
};

#define MCHK(call)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return ORBFE_ERR_HIP; } \
    } while (0)

}  // namespace

void match_scratch_free(MatchScratch& m)
{
    if (m.d) (void)hipFree(m.d);
    if (m.hpin) (void)hipHostFree(m.hpin);
    m = MatchScratch();
}

// Runs the fixed-point iteration on device-resident inputs.  Scratch pointers in A must be set.
static int proj_iterate(hipStream_t s, ProjArgs& A, int* hChanged /*pinned*/, std::string& err)
{
    const dim3 blk(256);
    hipLaunchKernelGGL(proj_prep_kernel, dim3((A.kpStride + 255) / 256, A.B), blk, 0, s, A);
    const size_t nres = (size_t)A.B * A.M;
    if (nres == 0) return ORBFE_OK;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nres + 255) / 256)), blk, 0, s, A.resA, -2, nres);
    int* prev = A.resA;
    int* cur = A.resB;
    bool first = true;
    const int maxIter = A.M + 2;
    for (int it = 0; it < maxIter;) {
        // two iterations per convergence check (one host sync per pair)
        for (int rep = 0; rep < 2; rep++, it++) {
            MCHK(hipMemsetAsync(A.changed, 0, sizeof(int), s));
            hipLaunchKernelGGL(proj_claims_kernel, dim3(A.B), blk, 0, s, A, first ? (const int*)nullptr : (const int*)prev);
            hipLaunchKernelGGL(proj_match_kernel, dim3((A.M + 3) / 4, A.B), blk, 0, s, A, (const int*)prev, cur);
            std::swap(prev, cur);
            first = false;
        }
        MCHK(hipMemcpyAsync(hChanged, A.changed, sizeof(int), hipMemcpyDeviceToHost, s));
        MCHK(hipStreamSynchronize(s));
        if (*hChanged == 0) break;
    }
    hipLaunchKernelGGL(proj_finalize_kernel, dim3(A.B), blk, 0, s, A, (const int*)prev);
    MCHK(hipGetLastError());
    return ORBFE_OK;
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
    const size_t inBytes = in.off;
    Carver sc = in;
    const size_t oCell = sc.take((size_t)n * sizeof(int));
    const size_t oClaim = sc.take((size_t)n * sizeof(int));
    const size_t oResA = sc.take((size_t)M * sizeof(int));
    const size_t oResB = sc.take((size_t)M * sizeof(int));
    const size_t oChanged = sc.take(sizeof(int));
    const size_t oMatch = sc.take((size_t)n * sizeof(int));
    const size_t oNM = sc.take(sizeof(int));
    const size_t hOut = (size_t)n * sizeof(int) + 2 * sizeof(int);
    int rc = ensure(m, sc.off, inBytes + hOut + 256, err);
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
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));

    ProjArgs A{};
    A.B = 1; A.M = M; A.kpStride = n;
    A.g = GridDesc{F->grid_cols, F->grid_rows, F->min_x, F->min_y, F->grid_inv_w, F->grid_inv_h};
    A.th = th; A.thFar = thFar; A.nnRatio = nnRatio; A.farPoints = farPoints; A.bFactor = th != 1.0;
    A.kp = reinterpret_cast<const orbfe_keypoint*>(dp + oKp);
    A.desc = dp + oDesc;
    A.nKp = reinterpret_cast<const int*>(dp + oN);
    A.mps = reinterpret_cast<const orbfe_map_point*>(dp + oMp);
    A.mpDesc = dp + oMpDesc;
    A.initObs = initObs ? reinterpret_cast<const int*>(dp + oObs) : nullptr;
    A.scaleFactors = reinterpret_cast<const float*>(dp + oSf);
    A.nLevels = F->n_levels;
    A.cellXY = reinterpret_cast<int*>(dp + oCell);
    A.claim = reinterpret_cast<int*>(dp + oClaim);
    A.resA = reinterpret_cast<int*>(dp + oResA);
    A.resB = reinterpret_cast<int*>(dp + oResB);
    A.changed = reinterpret_cast<int*>(dp + oChanged);
    A.matchOut = reinterpret_cast<int*>(dp + oMatch);
    A.nMatches = reinterpret_cast<int*>(dp + oNM);
    int* hChanged = reinterpret_cast<int*>(hp + inBytes);
    int* hNM = hChanged + 1;
    int* hMatch = hChanged + 2;
    rc = proj_iterate(s, A, hChanged, err);
    if (rc != ORBFE_OK) return rc;
    MCHK(hipMemcpyAsync(hMatch, A.matchOut, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipMemcpyAsync(hNM, A.nMatches, sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
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
    Carver sc;
    const size_t oCell = sc.take((size_t)B * kpStride * sizeof(int));
    const size_t oClaim = sc.take((size_t)B * kpStride * sizeof(int));
    const size_t oResA = sc.take((size_t)B * M * sizeof(int));
    const size_t oResB = sc.take((size_t)B * M * sizeof(int));
    const size_t oChanged = sc.take(sizeof(int));
    int rc = ensure(m, sc.off, 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    ProjArgs A{};
    A.B = B; A.M = M; A.kpStride = kpStride;
    A.g = GridDesc{gridCols, gridRows, minX, minY, invW, invH};
    A.th = th; A.thFar = thFar; A.nnRatio = nnRatio; A.farPoints = farPoints; A.bFactor = th != 1.0;
    A.kp = dKp; A.desc = dDesc; A.nKp = dN; A.mps = dMps; A.mpDesc = dMpDesc; A.initObs = dInitObs;
    A.scaleFactors = dScaleFactors; A.nLevels = nLevels;
    A.cellXY = reinterpret_cast<int*>(dp + oCell);
    A.claim = reinterpret_cast<int*>(dp + oClaim);
    A.resA = reinterpret_cast<int*>(dp + oResA);
    A.resB = reinterpret_cast<int*>(dp + oResB);
    A.changed = reinterpret_cast<int*>(dp + oChanged);
    A.matchOut = dMatchOut;
    A.nMatches = dNMatches;
    return proj_iterate(s, A, static_cast<int*>(m.hpin), err);
}

int match_bow_run(MatchScratch& m, hipStream_t s, int G, const int* kfOff, const int* kfIdx, const int* fOff,
                  const int* fIdx, int nKF, const uint8_t* kfDesc, const float* kfAngle, const uint8_t* kfHasMP, int nF,
                  const uint8_t* fDesc, const float* fAngle, float nnRatio, int checkOrientation, int* matchOut,
                  int* nMatches, std::string& err)
{
    for (int i = 0; i < nF; i++) matchOut[i] = -1;
    *nMatches = 0;
    if (G == 0 || nF == 0 || nKF == 0) return ORBFE_OK;
    const int nKfIdx = kfOff[G], nFIdx = fOff[G];
    for (int g = 0; g < G; g++)
        if (kfOff[g + 1] < kfOff[g] || fOff[g + 1] < fOff[g]) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < nKfIdx; i++)
        if (kfIdx[i] < 0 || kfIdx[i] >= nKF) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < nFIdx; i++)
        if (fIdx[i] < 0 || fIdx[i] >= nF) return ORBFE_ERR_INVALID_ARG;

    Carver in;
    const size_t oKfOff = in.take((size_t)(G + 1) * sizeof(int));
    const size_t oFOff = in.take((size_t)(G + 1) * sizeof(int));
    const size_t oKfIdx = in.take((size_t)std::max(nKfIdx, 1) * sizeof(int));
    const size_t oFIdx = in.take((size_t)std::max(nFIdx, 1) * sizeof(int));
    const size_t oKfDesc = in.take((size_t)nKF * 32);
    const size_t oFDesc = in.take((size_t)nF * 32);
    const size_t oKfHas = in.take((size_t)nKF);
    const size_t oKfAng = in.take((size_t)nKF * sizeof(float));
    const size_t oFAng = in.take((size_t)nF * sizeof(float));
    const size_t inBytes = in.off;
    Carver sc = in;
    const size_t oMatch = sc.take((size_t)nF * sizeof(int));
    const size_t oBin = sc.take((size_t)nF * sizeof(int));
    const size_t oNM = sc.take(sizeof(int));
    int rc = ensure(m, sc.off, inBytes + (size_t)nF * sizeof(int) + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oKfOff, kfOff, (size_t)(G + 1) * sizeof(int));
    memcpy(hp + oFOff, fOff, (size_t)(G + 1) * sizeof(int));
    memcpy(hp + oKfIdx, kfIdx, (size_t)nKfIdx * sizeof(int));
    memcpy(hp + oFIdx, fIdx, (size_t)nFIdx * sizeof(int));
    memcpy(hp + oKfDesc, kfDesc, (size_t)nKF * 32);
    memcpy(hp + oFDesc, fDesc, (size_t)nF * 32);
    memcpy(hp + oKfHas, kfHasMP, (size_t)nKF);
    if (kfAngle) memcpy(hp + oKfAng, kfAngle, (size_t)nKF * sizeof(float));
    if (fAngle) memcpy(hp + oFAng, fAngle, (size_t)nF * sizeof(float));
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));

    BowArgs A{};
    A.G = G;
    A.kfOff = reinterpret_cast<const int*>(dp + oKfOff);
    A.fOff = reinterpret_cast<const int*>(dp + oFOff);
    A.kfIdx = reinterpret_cast<const int*>(dp + oKfIdx);
    A.fIdx = reinterpret_cast<const int*>(dp + oFIdx);
    A.kfDesc = dp + oKfDesc;
    A.fDesc = dp + oFDesc;
    A.kfHasMP = dp + oKfHas;
    A.kfAngle = reinterpret_cast<const float*>(dp + oKfAng);
    A.fAngle = reinterpret_cast<const float*>(dp + oFAng);
    A.nF = nF;
    A.nnRatio = nnRatio;
    A.checkOrientation = checkOrientation;
    A.matchOut = reinterpret_cast<int*>(dp + oMatch);
    A.binOf = reinterpret_cast<int*>(dp + oBin);
    A.nMatches = reinterpret_cast<int*>(dp + oNM);
    const dim3 blk(256);
    hipLaunchKernelGGL(fill_kernel, dim3((nF + 255) / 256), blk, 0, s, A.matchOut, -1, (size_t)nF);
    hipLaunchKernelGGL(bow_match_kernel, dim3((G + 3) / 4), blk, 0, s, A);
    hipLaunchKernelGGL(bow_finalize_kernel, dim3(1), blk, 0, s, A);
    MCHK(hipGetLastError());
    int* hMatch = reinterpret_cast<int*>(hp + inBytes);
    int* hNM = hMatch + nF;
    MCHK(hipMemcpyAsync(hMatch, A.matchOut, (size_t)nF * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipMemcpyAsync(hNM, A.nMatches, sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(matchOut, hMatch, (size_t)nF * sizeof(int));
    *nMatches = *hNM;
    return ORBFE_OK;
}

}  // namespace orbfe
