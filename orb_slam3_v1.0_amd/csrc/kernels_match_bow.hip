// kernels_match_bow.hip -- ORBmatcher::SearchByBoW on gfx950 (src/ORBmatcher.cc:133-327 incl.
// ComputeThreeMaxima :1328-1370).  A frame feature belongs to exactly one vocabulary node, so the
// greedy "already matched" skips (:188) never cross nodes: one wave walks the key-frame features of
// a node sequentially and scans the node's frame features in parallel (top-2 under the total order
// (distance, position in the node's list)); a single block then applies the rotation histogram.
#include <algorithm>
#include <cstring>

#include "keyframe.h"
#include "match_common.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

// ------------------------------------------------------------------------------------------------
// SearchByBoW
// ------------------------------------------------------------------------------------------------
struct BowArgs {
    int G;
    const int *kfOff, *kfIdx, *fOff, *fIdx;
    const uint8_t *kfDesc, *fDesc, *kfHasMP;
    const float *kfAngle, *fAngle;
    int nF;
    int nLeft;       // F->Nleft: frame features >= nLeft belong to the right camera (-1: one camera)
    float nnRatio;
    int checkOrientation;
    int* matchOut;   // [nF], -1 initialised
    int* binOf;      // [nF]
    int* nMatches;   // [1]
};

// One vocabulary node shared by key frame and frame (:166-300): the key-frame features kfIdx[k0 .. k1e) of the node in
// DBoW2 order, sequentially (later ones skip frame features matched by earlier ones, :188); the node's frame features
// fList[0 .. fCount) scanned by the wave.  IdxT: the frame-side list lives in HBM (CSR from the host) or in LDS (built by
// the wave from the per-feature node ids); AngKF / AngF: orientation of a key-frame / frame feature.
template <typename IdxT, typename AngKF, typename AngF>
__device__ __forceinline__ void bow_walk_node(const int* __restrict__ kfIdx, int k0, int k1e, const IdxT* fList, int fCount,
                                              const uint8_t* __restrict__ kfDesc, const uint8_t* __restrict__ kfHasMP,
                                              const uint8_t* __restrict__ fDesc, int nLeftArg, float nnRatio, int checkOrientation,
                                              int* matchOut, int* binOf, AngKF kfAngle, AngF fAngle, int lane)
{
    const float factor = 1.0f / ORBFE_HISTO_LENGTH;
    for (int iKF = k0; iKF < k1e; iKF++) {  // sequential: later KF features skip matched frame features (:188)
        const int realIdxKF = kfIdx[iKF];
        if (!kfHasMP[realIdxKF]) continue;
        unsigned long long d4[4];
        {
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(kfDesc + (size_t)realIdxKF * 32);
            d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
        }
        // two camera sides (F->Nleft != -1, :205-233): separate best / second best for the left and the right features
        const int nLeft = nLeftArg < 0 ? 0x7fffffff : nLeftArg;
        unsigned long long k1 = kKeyNone, k2 = kKeyNone, r1 = kKeyNone, r2 = kKeyNone;
        for (int iF = lane; iF < fCount; iF += 64) {
            const int realIdxF = (int)fList[iF];
            if (matchOut[realIdxF] >= 0) continue;
            const int dist = hamming256(reinterpret_cast<const uint2*>(fDesc + (size_t)realIdxF * 32), d4);
            if (dist >= 256) continue;
            const unsigned long long key = ((unsigned long long)dist << 32) | (unsigned)iF;
            if (realIdxF < nLeft) {
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            } else {
                if (key < r1) { r2 = r1; r1 = key; }
                else if (key < r2) r2 = key;
            }
        }
        wave_top2(k1, k2);
        if (nLeftArg >= 0) wave_top2(r1, r2);  // wave-uniform
        if (k1 == kKeyNone) continue;  // bestDist1 == 256 > TH_LOW: neither side is looked at (:237)
        const int bestDist1 = (int)(k1 >> 32);
        const int bestDist2 = k2 == kKeyNone ? 256 : (int)(k2 >> 32);
        if (bestDist1 <= ORBFE_TH_LOW) {  // :237
            bool wrote = false;
            if ((float)bestDist1 < nnRatio * (float)bestDist2) {  // :239
                const int bestIdxF = (int)fList[(int)(k1 & 0xffffffffu)];
                if (lane == 0) {
                    matchOut[bestIdxF] = realIdxKF;
                    if (checkOrientation) {
                        float rot = kfAngle(realIdxKF) - fAngle(bestIdxF);
                        if (rot < 0.0) rot = rot + 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == ORBFE_HISTO_LENGTH) bin = 0;
                        binOf[bestIdxF] = bin;
                    }
                }
                wrote = true;
            }
            // right camera (:263-286): accepted whenever its best distance passes TH_LOW (the ratio test is "|| true")
            if (r1 != kKeyNone && (int)(r1 >> 32) <= ORBFE_TH_LOW) {
                const int bestIdxFR = (int)fList[(int)(r1 & 0xffffffffu)];
                if (lane == 0) {
                    matchOut[bestIdxFR] = realIdxKF;
                    if (checkOrientation) {
                        float rot = kfAngle(realIdxKF) - fAngle(bestIdxFR);
                        if (rot < 0.0) rot = rot + 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == ORBFE_HISTO_LENGTH) bin = 0;
                        binOf[bestIdxFR] = bin;
                    }
                }
                wrote = true;
            }
            if (wrote) __threadfence_block();  // later iterations of this wave read matchOut
        }
    }
}

__global__ __launch_bounds__(256) void bow_match_kernel(BowArgs A)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= A.G) return;
    const int f0 = A.fOff[g];
    bow_walk_node(A.kfIdx, A.kfOff[g], A.kfOff[g + 1], A.fIdx + f0, A.fOff[g + 1] - f0, A.kfDesc, A.kfHasMP, A.fDesc, A.nLeft,
                  A.nnRatio, A.checkOrientation, A.matchOut, A.binOf, [&](int i) { return A.kfAngle[i]; },
                  [&](int i) { return A.fAngle[i]; }, lane);
}

// rotation-histogram filter (:304-322) + count; single block
__device__ __forceinline__ void bow_finalize_block(int* matchOut, const int* binOf, int nF, int checkOrientation, int* nMatches)
{
    __shared__ int hist[ORBFE_HISTO_LENGTH];
    __shared__ int sInd[3];
    __shared__ int sCount;
    const int tid = threadIdx.x;
    if (tid < ORBFE_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sCount = 0;
    __syncthreads();
    int local = 0;
    for (int j = tid; j < nF; j += blockDim.x)
        if (matchOut[j] >= 0) {
            local++;
            if (checkOrientation) atomicAdd(&hist[binOf[j]], 1);
        }
    __syncthreads();
    if (tid == 0) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        if (checkOrientation) {  // ComputeThreeMaxima :1328-1370
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
    if (checkOrientation) {
        for (int j = tid; j < nF; j += blockDim.x)
            if (matchOut[j] >= 0) {
                const int b = binOf[j];
                if (b != sInd[0] && b != sInd[1] && b != sInd[2]) {
                    matchOut[j] = -1;
                    local--;
                }
            }
    }
    if (local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) *nMatches = sCount;
}

__global__ __launch_bounds__(256) void bow_finalize_kernel(BowArgs A)
{
    bow_finalize_block(A.matchOut, A.binOf, A.nF, A.checkOrientation, A.nMatches);
}

// ------------------------------------------------------------------------------------------------
// SearchByBoW of the frame the extraction chain has just produced against a key frame resident in HBM
// (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:825-835): the frame's FeatureVector exists only as the node id of
// every feature (the vocabulary descent ran a kernel earlier, nothing has been on the host), so the wave of a key-frame
// node collects the node's frame features itself -- ascending feature index, the order DBoW2 stores them in
// (TemplatedVocabulary.h:1157-1170) -- into its LDS list and then walks the node exactly like the CSR kernel.
// A node the frame has no feature in is left after the collection (the lockstep walk of the two FeatureVectors,
// :150-165, only stops at shared nodes).
// ------------------------------------------------------------------------------------------------
// wave64 minimum with DPP row operations (one VALU instruction per step instead of an LDS permute): quad swaps, row
// half-mirror and mirror leave every lane of a 16-lane row with the row minimum; row_bcast 15 / 31 fold the rows into lane
// 63, which is read back as a scalar.  Lanes a row mask leaves out receive the identity.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// wave-wide top-2 of per-lane (smallest, second smallest) pairs of 32-bit keys, distinct except for the "none" value:
// the minimum, then the minimum with its owner's smallest key replaced by that lane's second
__device__ __forceinline__ void wave_top2_u32(unsigned& k1, unsigned& k2)
{
    const unsigned m1 = wave_min_u32(k1);
    const unsigned m2 = wave_min_u32(k1 == m1 ? k2 : k1);
    k1 = m1;
    k2 = m2;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// One wave per block, one vocabulary node of the key frame at a time.  The node ids of the frame's features are staged in
// LDS once per block.  For a node the wave gathers the frame side -- descriptors and orientations: the first 64 features in
// registers, the rest in LDS (two 16-byte halves, each its own array: conflict-free b128 reads); "matched" as a bit per
// feature in its owning lane --, loads the key-frame side 64 features at a time into registers (lane l holds feature l of
// the chunk) and walks it sequentially: one descriptor broadcast per step (readlane), distances, a DPP wave top-2.  No
// step of the walk waits on HBM, and for a node of <= 64 frame features none touches LDS.  (Computing every feature's two
// best candidates up front and replaying the order with scalar steps was measured: 220 us per call against 206 -- in a
// node where most features get matched a later feature has nearly always lost one of its two.)  Nodes with more than
// kBowNodeCap frame features (degenerate vocabularies: levelsup >= L puts a whole frame into one node) take the general walk.
constexpr int kBowNodeCap = 512;

__global__ __launch_bounds__(64) void bow_track_kernel(BowTrackArgs A)
{
    extern __shared__ uint4 sMem4[];
    const int lane = threadIdx.x;
    uint4* sLo = sMem4;
    uint4* sHi = sLo + kBowNodeCap;
    float* sAng = reinterpret_cast<float*>(sHi + kBowNodeCap);
    int* sNode = reinterpret_cast<int*>(sAng + kBowNodeCap);                    // [cap] node of every frame feature
    uint16_t* list = reinterpret_cast<uint16_t*>(sNode + A.cap);                // [cap] frame features of the current node
    const BowKfRef R = *A.ref;
    if ((int)blockIdx.x >= R.G) return;
    const int nF = min(*A.nF, A.cap);
    // stopped words (weight 0) never enter the FeatureVector (TemplatedVocabulary.h:1168-1172, 1196-1200): such a feature is in
    // no node and SearchByBoW cannot match it; -1 equals no key-frame node (keyframe_create keeps node >= 0 only)
    for (int i = lane; i < nF; i += 64) sNode[i] = A.weight[A.fLeaf[i]] > 0.0 ? A.fBow[2 * i + 1] : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr unsigned kNone = 0xffffffffu;
    for (int g = blockIdx.x; g < R.G; g += gridDim.x) {
        const int nid = R.nodeList[g];
        const int k0 = R.nodeOff[g], cntK = R.nodeOff[g + 1] - k0;
        int cnt = 0;
        for (int base = 0; base < nF; base += 64) {
            const int i = base + lane;
            const bool hit = i < nF && sNode[i] == nid;
            const unsigned long long m = __ballot(hit);
            if (hit) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
            cnt += __popcll(m);
        }
        if (cnt == 0) continue;  // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (cnt <= kBowNodeCap) {
            // frame side: the first 64 features of the node in registers (lane l holds list position l), the rest in LDS;
            // "matched" as one bit per 64-feature chunk in the owning lane (position p belongs to lane p & 63)
            uint4 lo0 = make_uint4(0, 0, 0, 0), hi0 = lo0;
            float ang0 = 0.f;
            int idx0 = 0;
            for (int p = lane; p < cnt; p += 64) {
                const int f = (int)list[p];
                const uint4* dp = reinterpret_cast<const uint4*>(A.fDesc + (size_t)f * 32);
                const uint4 lo = dp[0], hi = dp[1];
                const float ang = A.fKp[f].angle;
                if (p < 64) { lo0 = lo; hi0 = hi; ang0 = ang; idx0 = f; }
                else { sLo[p] = lo; sHi[p] = hi; sAng[p] = ang; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            unsigned taken = 0;  // bit c: list position c * 64 + lane is matched
            for (int kc = 0; kc < cntK; kc += 64) {
                const int nk = min(64, cntK - kc);
                const bool kOn = lane < nk;
                const int myK = kOn ? R.order[k0 + kc + lane] : 0;
                unsigned long long kd[4] = {0, 0, 0, 0};
                float kAng = 0.f;
                int kHas = 0;
                if (kOn) {
                    const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(R.desc + (size_t)myK * 32);
                    kd[0] = dp[0]; kd[1] = dp[1]; kd[2] = dp[2]; kd[3] = dp[3];
                    kAng = R.kp[myK].angle;
                    kHas = A.kfHasMP[myK];
                }
                for (int j = 0; j < nk; j++) {  // sequential: later KF features skip matched frame features (:188)
                    if (!__builtin_amdgcn_readlane(kHas, j)) continue;
                    const unsigned long long d0 = readlane_u64(kd[0], j), d1 = readlane_u64(kd[1], j), d2 = readlane_u64(kd[2], j),
                                             d3 = readlane_u64(kd[3], j);
                    unsigned k1 = kNone, k2 = kNone;
                    for (int p = lane, c = 0; p < cnt; p += 64, c++) {
                        if ((taken >> c) & 1u) continue;
                        const uint4 lo = c == 0 ? lo0 : sLo[p], hi = c == 0 ? hi0 : sHi[p];
                        const int dist = __popcll((((unsigned long long)lo.y << 32) | lo.x) ^ d0) + __popcll((((unsigned long long)lo.w << 32) | lo.z) ^ d1) +
                                         __popcll((((unsigned long long)hi.y << 32) | hi.x) ^ d2) + __popcll((((unsigned long long)hi.w << 32) | hi.z) ^ d3);
                        if (dist >= 256) continue;
                        const unsigned key = ((unsigned)dist << 16) | (unsigned)p;  // (distance, position in the node's list)
                        if (key < k1) { k2 = k1; k1 = key; }
                        else if (key < k2) k2 = key;
                    }
                    wave_top2_u32(k1, k2);
                    if (k1 == kNone) continue;  // bestDist1 == 256 > TH_LOW (:237)
                    const int bestDist1 = (int)(k1 >> 16);
                    const int bestDist2 = k2 == kNone ? 256 : (int)(k2 >> 16);
                    if (bestDist1 <= ORBFE_TH_LOW && (float)bestDist1 < A.nnRatio * (float)bestDist2) {  // :237-239
                        const int realIdxKF = __builtin_amdgcn_readlane(myK, j);
                        const float angKF = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kAng), j));
                        const int pos = (int)(k1 & 0xffffu);
                        if (lane == (pos & 63)) {  // the owner of the position
                            const int c = pos >> 6;
                            taken |= 1u << c;
                            const int bestIdxF = c == 0 ? idx0 : (int)list[pos];
                            A.matchOut[bestIdxF] = realIdxKF;
                            if (A.checkOrientation) {
                                float rot = angKF - (c == 0 ? ang0 : sAng[pos]);
                                if (rot < 0.0) rot = rot + 360.0f;
                                int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
                                if (bin == ORBFE_HISTO_LENGTH) bin = 0;
                                A.binOf[bestIdxF] = bin;
                            }
                        }
                    }
                }
            }
        } else {
            bow_walk_node(R.order, k0, k0 + cntK, list, cnt, R.desc, A.kfHasMP, A.fDesc, -1, A.nnRatio, A.checkOrientation,
                          A.matchOut, A.binOf, [&](int i) { return R.kp[i].angle; }, [&](int i) { return A.fKp[i].angle; }, lane);
        }
        __builtin_amdgcn_wave_barrier();  // the LDS arrays are rewritten for the block's next node
    }
}

__global__ __launch_bounds__(256) void bow_track_finalize_kernel(BowTrackArgs A)
{
    bow_finalize_block(A.matchOut, A.binOf, min(*A.nF, A.cap), A.checkOrientation, A.nMatches);
}

}  // namespace

int match_bow_run(MatchScratch& m, hipStream_t s, int G, const int* kfOff, const int* kfIdx, const int* fOff,
                  const int* fIdx, int nKF, const uint8_t* kfDesc, const float* kfAngle, const uint8_t* kfHasMP, int nF,
                  const uint8_t* fDesc, const float* fAngle, int nLeft, float nnRatio, int checkOrientation, int* matchOut,
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
    A.nLeft = nLeft;
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


int bow_track_launch(hipStream_t s, const BowTrackArgs& A, std::string& err)
{
    if (A.cap <= 0 || A.cap > 7168) {  // 16-bit list entries; node ids + one list of cap entries in LDS (<= 61 KB)
        err = "bow_track_launch: frame capacity outside (0, 7168]";
        return ORBFE_ERR_UNSUPPORTED;
    }
    const dim3 blk(256);
    // (A.matchOut arrives filled with -1: the vocabulary descent that runs before this wrote it)
    const size_t lds = (size_t)kBowNodeCap * (2 * sizeof(uint4) + sizeof(float)) + (size_t)A.cap * (sizeof(int) + sizeof(uint16_t));
    hipLaunchKernelGGL(bow_track_kernel, dim3(512), dim3(64), lds, s, A);
    hipLaunchKernelGGL(bow_track_finalize_kernel, dim3(1), blk, 0, s, A);
    MCHK(hipGetLastError());
    return ORBFE_OK;
}

}  // namespace orbfe
