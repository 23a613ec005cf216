// kernels_match_kf.hip -- the key-frame searches of ORBmatcher on gfx950: the search part of both Fuse overloads
// (src/ORBmatcher.cc:678-836, 864-975), SearchBySim3 (:977-1200) and the relocalisation overload of SearchByProjection
// (:1202-1326).  Thread per map point on the per-level cell tables that kernels_match_proj.hip builds for a frame
// (proj_prepare_launch); the relocalisation overload is order-dependent and runs on the projection pipeline itself
// (proj_launch in kModeReloc) between its own projection and rotation-histogram kernels.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device_math.h"
#include "match_common.h"
#include "match_proj.h"
#include "keyframe.h"

#pragma clang fp contract(off)

namespace orbfe {

using namespace proj;

namespace {

// ---------------------------------------------------------------------------------------------
// The search part of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:678-836), SURVEY 8f row f2:
// thread per map point -- projection (SPEC DECISION S8 arithmetic, as frustum_kernel), KeyFrame::IsInImage,
// PredictScale, KeyFrame::GetFeaturesInArea through the per-level cell-range tables, the chi-square gate
// (:794-817) and the nearest descriptor under (distance, visit position) == the reference's strict "<" scan.
// ---------------------------------------------------------------------------------------------
// GATHER: the map points are entries ids[i] of a resident map of mapCap entries (orbfe_map): id >= 0 the entry, ~id (negative)
// the entry with this call's skip flag set, outside the map no point -- pts / mpDesc are then the map's arrays.
template <bool GATHER>
__global__ __launch_bounds__(256) void fuse_search_kernel(ProjArgs A, orbfe_frustum F, float th, int M,
                                                          const int* __restrict__ ids, int mapCap,
                                                          const orbfe_world_point* __restrict__ pts,
                                                          const uint8_t* __restrict__ mpDesc,
                                                          const float* __restrict__ invLevelSigma2,
                                                          const float* __restrict__ uRight, int chi2Gate,
                                                          const unsigned long long* __restrict__ rightDesc, int nRight,
                                                          int* __restrict__ bestIdxOut, int* __restrict__ bestDistOut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (one wave per block for a single call: see fuse_search_run)
    if (i >= M) return;
    int src = i;
    bool skipped = false;
    if (GATHER) {
        const int id = ids[i];
        src = id < 0 ? ~id : id;
        skipped = id < 0 || src >= mapCap;
        if (src >= mapCap) src = 0;
    }
    const orbfe_world_point p = pts[src];
    int bestIdx = -1, bestDist = 256;
    do {
        if (skipped || (!GATHER && p.skip) || p.bad) break;  // :706-721 (a resident entry's own skip member is per frame: ignored)
        const float X = p.x, Y = p.y, Z = p.z;
        const float pcx = ((F.rcw[0] * X + F.rcw[1] * Y) + F.rcw[2] * Z) + F.tcw[0];
        const float pcy = ((F.rcw[3] * X + F.rcw[4] * Y) + F.rcw[5] * Z) + F.tcw[1];
        const float pcz = ((F.rcw[6] * X + F.rcw[7] * Y) + F.rcw[8] * Z) + F.tcw[2];
        if (pcz < 0.0f) break;  // :725
        const float invz = 1.0f / pcz;
        float u, v;
        camera_project(F, pcx, pcy, pcz, u, v);
        if (!(u >= F.min_x && u < F.max_x && v >= F.min_y && v < F.max_y)) break;  // KeyFrame::IsInImage
        const float ur = u - F.mbf * invz;
        const float maxD = 1.1f * p.max_distance, minD = 0.9f * p.min_distance;
        const float ox = X - F.twc[0], oy = Y - F.twc[1], oz = Z - F.twc[2];
        const float dist3D = sqrtf((ox * ox + oy * oy) + oz * oz);
        if (dist3D < minD || dist3D > maxD) break;  // :748
        const float ratio = p.max_distance / dist3D;  // PredictScale
        const float q = spec_logf(ratio) / F.log_scale_factor;
        int lvl;
        if (!(q > 0.0f)) lvl = 0;
        else if (q >= (float)F.n_levels) lvl = F.n_levels - 1;
        else {
            lvl = (int)ceilf(q);
            if (lvl >= F.n_levels) lvl = F.n_levels - 1;
        }
        MpWindow w;
        w.valid = true;
        w.x = u;
        w.y = v;
        w.r = th * A.scaleFactors[lvl];  // :766
        window_cells(A.g, w);
        if (!w.valid) break;
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(mpDesc + (size_t)src * 32);
        const unsigned long long d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
        const int tabStride = A.g.cols + 1;
        uint32_t best = kKey32None;
        for (int l = max(lvl - 1, 0); l <= min(lvl, A.tabLevels - 1); l++) {  // :787
            const int* csl = A.colStart + (size_t)l * tabStride;
            const float invS2 = chi2Gate ? invLevelSigma2[l] : 0.0f;
            {
                const int s = csl[w.minCX], e = csl[w.maxCX + 1];  // all rows of the window's columns
                for (int sl = s; sl < e; sl++) {
                    const int4 rq = A.rec[sl];
                    const int cy = rq.y >> 8;
                    if (cy < w.minCY || cy > w.maxCY) continue;
                    const float kx = __int_as_float(rq.z), ky = __int_as_float(rq.w);
                    if (!(fabsf(kx - u) < w.r && fabsf(ky - v) < w.r)) continue;  // src/KeyFrame.cc:826
                    const float ex = u - kx, ey = v - ky;
                    float kur = -1.0f;
                    if (uRight) kur = uRight[A.order[rq.x]];
                    if (!chi2Gate) {
                        // the Sim3 overload (:864-975) has no reprojection gate
                    } else if (kur >= 0) {  // :792-805
                        const float er = ur - kur;
                        const float e2 = (ex * ex + ey * ey) + er * er;
                        if ((double)(e2 * invS2) > 7.8) continue;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * invS2) > 5.99) continue;
                    }
                    const unsigned long long* kd = A.descS + (size_t)sl * 4;
                    if (rightDesc) {  // bRight (:820): the left feature passed the gates, its right twin's row is compared
                        const int idx = A.order[rq.x];
                        if (idx >= nRight) continue;
                        kd = rightDesc + (size_t)idx * 4;
                    }
                    const int dist = __popcll(kd[0] ^ d0) + __popcll(kd[1] ^ d1) + __popcll(kd[2] ^ d2) + __popcll(kd[3] ^ d3);
                    best = min(best, make_key32(dist, rq.x));
                }
            }
        }
        if (best != kKey32None) {
            bestIdx = A.order[best & kRankMask] + (rightDesc ? A.kpStride : 0);
            bestDist = (int)(best >> kRankBits);
        }
    } while (false);
    bestIdxOut[i] = bestIdx;
    bestDistOut[i] = bestDist;
}

// PredictScale (src/MapPoint.cc:580-612) on the pinned logarithm, SPEC DECISION S8
__device__ __forceinline__ int predict_scale(float maxDistance, float dist, float logScaleFactor, int nLevels)
{
    const float ratio = maxDistance / dist;
    const float q = spec_logf(ratio) / logScaleFactor;
    int lvl;
    if (!(q > 0.0f)) lvl = 0;
    else if (q >= (float)nLevels) lvl = nLevels - 1;
    else {
        lvl = (int)ceilf(q);
        if (lvl >= nLevels) lvl = nLevels - 1;
    }
    return lvl;
}

// ---------------------------------------------------------------------------------------------
// One direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1017-1092, :1094-1170): thread per map point of
// the source key frame -- source pose, similarity, the pinhole expression the function writes out itself
// (:1038-1043), KeyFrame::IsInImage, PredictScale, KeyFrame::GetFeaturesInArea on the target key frame's tables
// (A), nearest descriptor on levels [l-1, l] under (distance, visit position), bestDist <= TH_HIGH.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sim3_search_kernel(ProjArgs A, orbfe_sim3_view D, float th, int n,
                                                          const orbfe_world_point* __restrict__ pts,
                                                          const uint8_t* __restrict__ mpDesc, int* __restrict__ vnMatch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const orbfe_world_point p = pts[i];
    int bestIdx = -1;
    do {
        if (p.skip || p.bad) break;  // :1021-1025
        const float X = p.x, Y = p.y, Z = p.z;
        const float ax = ((D.rcw[0] * X + D.rcw[1] * Y) + D.rcw[2] * Z) + D.tcw[0];
        const float ay = ((D.rcw[3] * X + D.rcw[4] * Y) + D.rcw[5] * Z) + D.tcw[1];
        const float az = ((D.rcw[6] * X + D.rcw[7] * Y) + D.rcw[8] * Z) + D.tcw[2];
        const float bx = ((D.sr[0] * ax + D.sr[1] * ay) + D.sr[2] * az) + D.t[0];
        const float by = ((D.sr[3] * ax + D.sr[4] * ay) + D.sr[5] * az) + D.t[1];
        const float bz = ((D.sr[6] * ax + D.sr[7] * ay) + D.sr[8] * az) + D.t[2];
        if (bz < 0.0f) break;  // :1032
        const float invz = 1.0f / bz;
        const float x = bx * invz, y = by * invz;
        const float u = D.fx * x + D.cx, v = D.fy * y + D.cy;
        if (!(u >= D.min_x && u < D.max_x && v >= D.min_y && v < D.max_y)) break;  // KeyFrame::IsInImage
        const float maxD = 1.1f * p.max_distance, minD = 0.9f * p.min_distance;
        const float dist3D = sqrtf((bx * bx + by * by) + bz * bz);
        if (dist3D < minD || dist3D > maxD) break;  // :1052
        const int lvl = predict_scale(p.max_distance, dist3D, D.log_scale_factor, D.n_levels);
        MpWindow w;
        w.valid = true;
        w.x = u;
        w.y = v;
        w.r = th * A.scaleFactors[lvl];  // :1060
        window_cells(A.g, w);
        if (!w.valid) break;
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(mpDesc + (size_t)i * 32);
        const unsigned long long d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
        const int tabStride = A.g.cols + 1;
        uint32_t best = kKey32None;
        for (int l = max(lvl - 1, 0); l <= min(lvl, A.tabLevels - 1); l++) {  // :1078
            const int* csl = A.colStart + (size_t)l * tabStride;
            const int s = csl[w.minCX], e = csl[w.maxCX + 1];
            for (int sl = s; sl < e; sl++) {
                const int4 rq = A.rec[sl];
                const int cy = rq.y >> 8;
                if (cy < w.minCY || cy > w.maxCY) continue;
                const float kx = __int_as_float(rq.z), ky = __int_as_float(rq.w);
                if (!(fabsf(kx - u) < w.r && fabsf(ky - v) < w.r)) continue;  // src/KeyFrame.cc:826
                const unsigned long long* kd = A.descS + (size_t)sl * 4;
                const int dist = __popcll(kd[0] ^ d0) + __popcll(kd[1] ^ d1) + __popcll(kd[2] ^ d2) + __popcll(kd[3] ^ d3);
                best = min(best, make_key32(dist, rq.x));
            }
        }
        if (best != kKey32None && (int)(best >> kRankBits) <= ORBFE_TH_HIGH) bestIdx = A.order[best & kRankMask];  // :1088
    } while (false);
    vnMatch[i] = bestIdx;
}

// agreement check of SearchBySim3 (:1172-1189): match12[i1] = idx2 iff vnMatch2[vnMatch1[i1]] == i1
__global__ __launch_bounds__(256) void sim3_agree_kernel(int n1, const int* __restrict__ vn1, const int* __restrict__ vn2,
                                                         int* __restrict__ match12, int* __restrict__ nFound)
{
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    int hit = 0;
    if (i1 < n1) {
        const int idx2 = vn1[i1];
        int m = -1;
        if (idx2 >= 0 && vn2[idx2] == i1) {
            m = idx2;
            hit = 1;
        }
        match12[i1] = m;
    }
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(nFound, __popcll(b));
}

// ---------------------------------------------------------------------------------------------
// Relocalisation overload SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, checkOrientation)
// (src/ORBmatcher.cc:1202-1326).  reloc_project_kernel is the per-map-point head (:1225-1253) and writes the
// matcher's input records; the greedy "first free best" scan is the projection pipeline in kModeReloc (every
// accepted point blocks its keypoint: observations = 1); reloc_finalize_kernel is the rotation histogram.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reloc_project_kernel(orbfe_frustum F, int M, const orbfe_world_point* __restrict__ pts,
                                                            orbfe_map_point* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const orbfe_world_point p = pts[i];
    orbfe_map_point o;
    o.proj_x = -1.0f;
    o.proj_y = -1.0f;
    o.view_cos = 1.0f;
    o.track_depth = 0.0f;
    o.level = 0;
    o.in_view = 0;
    o.bad = 0;
    o.observations = 1;  // any accepted map point occupies its keypoint (:1268 tests the pointer only)
    do {
        if (p.skip || p.bad) break;  // :1221-1225
        const float X = p.x, Y = p.y, Z = p.z;
        const float pcx = ((F.rcw[0] * X + F.rcw[1] * Y) + F.rcw[2] * Z) + F.tcw[0];
        const float pcy = ((F.rcw[3] * X + F.rcw[4] * Y) + F.rcw[5] * Z) + F.tcw[1];
        const float pcz = ((F.rcw[6] * X + F.rcw[7] * Y) + F.rcw[8] * Z) + F.tcw[2];
        float u, v;
        camera_project(F, pcx, pcy, pcz, u, v);  // no depth test in this overload
        if (u < F.min_x || u > F.max_x) break;   // :1232-1235
        if (v < F.min_y || v > F.max_y) break;
        if (u != u || v != v) break;  // NaN passes the tests above; the reference's grid query is then empty
        const float ox = X - F.twc[0], oy = Y - F.twc[1], oz = Z - F.twc[2];
        const float dist3D = sqrtf((ox * ox + oy * oy) + oz * oz);
        const float maxD = 1.1f * p.max_distance, minD = 0.9f * p.min_distance;
        if (dist3D < minD || dist3D > maxD) break;  // :1245
        o.level = predict_scale(p.max_distance, dist3D, F.log_scale_factor, F.n_levels);
        o.proj_x = u;
        o.proj_y = v;
        o.in_view = 1;
    } while (false);
    out[i] = o;
}

// rotation histogram + three maxima (:1287-1323) over the claimed keypoints; single block
__global__ __launch_bounds__(256) void reloc_finalize_kernel(int n, const orbfe_keypoint* __restrict__ kp,
                                                             const float* __restrict__ kfAngle, int checkOrientation,
                                                             int* __restrict__ matchOut, int* __restrict__ nMatches)
{
    __shared__ int hist[ORBFE_HISTO_LENGTH];
    __shared__ int sInd[3];
    __shared__ int sCount;
    const int tid = threadIdx.x;
    if (tid < ORBFE_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sCount = 0;
    __syncthreads();
    const float factor = 1.0f / ORBFE_HISTO_LENGTH;
    int local = 0;
    for (int j = tid; j < n; j += blockDim.x) {
        const int i = matchOut[j];
        if (i < 0) continue;
        local++;
        if (checkOrientation) {
            float rot = kfAngle[i] - kp[j].angle;
            if (rot < 0.0f) rot = rot + 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            atomicAdd(&hist[bin], 1);
        }
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
        for (int j = tid; j < n; j += blockDim.x) {
            const int i = matchOut[j];
            if (i < 0) continue;
            float rot = kfAngle[i] - kp[j].angle;
            if (rot < 0.0f) rot = rot + 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) {
                matchOut[j] = -1;
                local--;
            }
        }
    }
    if (local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) *nMatches = sCount;
}

}  // namespace

int fuse_search_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* KF, const float* invLevelSigma2,
                    const float* uRight, const orbfe_frustum* F, float th, int M, const orbfe_world_point* pts,
                    const uint8_t* mpDesc, int chi2Gate, int nRight, int* bestIdxOut, int* bestDistOut, std::string& err)
{
    if (!chi2Gate) {
        invLevelSigma2 = nullptr;
        uRight = nullptr;
    }
    for (int i = 0; i < M; i++) {
        bestIdxOut[i] = -1;
        bestDistOut[i] = 256;
    }
    const int n = KF->n;
    if (n == 0 || M == 0) return ORBFE_OK;
    if (n >= (1 << 20) || KF->grid_cols > 65535 || KF->grid_rows > 32767 || KF->n_levels < 1 ||
        (long long)KF->grid_cols * KF->grid_rows > kMaxCells)
        return ORBFE_ERR_UNSUPPORTED;
    if (F->n_levels > KF->n_levels) return ORBFE_ERR_INVALID_ARG;  // predicted levels index the key frame's scale tables
    Carver in;
    const size_t oKp = in.take((size_t)n * sizeof(orbfe_keypoint));
    const size_t oDesc = in.take((size_t)n * 32);
    const size_t oPts = in.take((size_t)M * sizeof(orbfe_world_point));
    const size_t oMpDesc = in.take((size_t)M * 32);
    const size_t oSf = in.take((size_t)KF->n_levels * sizeof(float));
    const size_t oIs2 = in.take((size_t)KF->n_levels * sizeof(float));
    const size_t oUr = in.take((size_t)n * sizeof(float));
    const size_t oN = in.take(sizeof(int));
    const size_t oRight = in.take((size_t)(nRight > 0 ? nRight : 0) * 32);  // rows NLeft .. of mDescriptors (bRight)
    const size_t inBytes = in.off;
    const size_t oMatch = in.take((size_t)n * sizeof(int));  // the grid kernel clears a match array
    const size_t oBest = in.take((size_t)M * 2 * sizeof(int));
    ProjArgs A{};
    A.B = 1; A.M = 1; A.kpStride = n;
    A.g = GridDesc{KF->grid_cols, KF->grid_rows, KF->min_x, KF->min_y, KF->grid_inv_w, KF->grid_inv_h};
    A.nnRatio = 1.0f;
    A.nLevels = KF->n_levels;
    int rc = proj_setup(m, A, in, inBytes + (size_t)M * 2 * sizeof(int) + 64, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oKp, KF->kp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(hp + oDesc, KF->desc, (size_t)n * 32);
    memcpy(hp + oPts, pts, (size_t)M * sizeof(orbfe_world_point));
    memcpy(hp + oMpDesc, mpDesc, (size_t)M * 32);
    memcpy(hp + oSf, KF->scale_factors, (size_t)KF->n_levels * sizeof(float));
    if (invLevelSigma2) memcpy(hp + oIs2, invLevelSigma2, (size_t)KF->n_levels * sizeof(float));
    if (uRight) memcpy(hp + oUr, uRight, (size_t)n * sizeof(float));
    memcpy(hp + oN, &n, sizeof(int));
    if (nRight > 0) memcpy(hp + oRight, KF->desc + (size_t)n * 32, (size_t)nRight * 32);
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    A.kp = reinterpret_cast<const orbfe_keypoint*>(dp + oKp);
    A.desc = dp + oDesc;
    A.nKp = reinterpret_cast<const int*>(dp + oN);
    A.scaleFactors = reinterpret_cast<const float*>(dp + oSf);
    A.matchOut = reinterpret_cast<int*>(dp + oMatch);
    proj_prepare_launch(s, A, false);
    int* dBest = reinterpret_cast<int*>(dp + oBest);
    hipLaunchKernelGGL(fuse_search_kernel<false>, dim3((M + 63) / 64), dim3(64), 0, s, A, *F, th, M, nullptr, 0,
                       reinterpret_cast<const orbfe_world_point*>(dp + oPts), dp + oMpDesc,
                       reinterpret_cast<const float*>(dp + oIs2), uRight ? reinterpret_cast<const float*>(dp + oUr) : nullptr,
                       chi2Gate, nRight >= 0 ? reinterpret_cast<const unsigned long long*>(dp + oRight) : nullptr, nRight, dBest,
                       dBest + M);
    MCHK(hipGetLastError());
    int* hBest = reinterpret_cast<int*>(hp + inBytes);
    MCHK(hipMemcpyAsync(hBest, dBest, (size_t)M * 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(bestIdxOut, hBest, (size_t)M * sizeof(int));
    memcpy(bestDistOut, hBest + M, (size_t)M * sizeof(int));
    return ORBFE_OK;
}

int keyframe_set_grid(KeyFrameDev* K, hipStream_t s, int gridCols, int gridRows, float minX, float minY, float invW, float invH,
                      const float* invLevelSigma2, const float* uRight, std::string& err)
{
    if (!K || !invLevelSigma2 || gridCols < 1 || gridRows < 1) return ORBFE_ERR_INVALID_ARG;
    const int n = K->n;
    if (n >= (1 << 20) || gridCols > 65535 || gridRows > 32767 || (long long)gridCols * gridRows > kMaxCells) return ORBFE_ERR_UNSUPPORTED;
    K->hasGrid = false;
    Carver in;
    const size_t oN = in.take(sizeof(int));
    const size_t oIs2 = in.take((size_t)K->nLevels * sizeof(float));
    const size_t oUr = in.take((size_t)std::max(n, 1) * sizeof(float));
    const size_t inBytes = in.off;
    const size_t oMatch = in.take((size_t)std::max(n, 1) * sizeof(int));  // the grid kernel clears a match array
    ProjArgs A{};
    A.B = 1; A.M = 1; A.kpStride = std::max(n, 1);
    A.g = GridDesc{gridCols, gridRows, minX, minY, invW, invH};
    A.nnRatio = 1.0f;
    A.nLevels = K->nLevels;
    int rc = proj_setup(K->gridMem, A, in, inBytes, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(K->gridMem.hpin);
    uint8_t* dp = static_cast<uint8_t*>(K->gridMem.d);
    memcpy(hp + oN, &n, sizeof(int));
    memcpy(hp + oIs2, invLevelSigma2, (size_t)K->nLevels * sizeof(float));
    if (uRight && n) memcpy(hp + oUr, uRight, (size_t)n * sizeof(float));
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    A.kp = K->kp;
    A.desc = K->desc;
    A.nKp = reinterpret_cast<const int*>(dp + oN);
    A.scaleFactors = K->sf;
    A.matchOut = reinterpret_cast<int*>(dp + oMatch);
    if (n) proj_prepare_launch(s, A, false);
    MCHK(hipGetLastError());
    MCHK(hipStreamSynchronize(s));
    (void)hipHostFree(K->gridMem.hpin);  // the staging block has done its work; the device arena stays with the key frame
    K->gridMem.hpin = nullptr;
    K->gridMem.hBytes = 0;
    K->grid = A;
    K->invLevelSigma2 = reinterpret_cast<const float*>(dp + oIs2);
    K->uRight = uRight ? reinterpret_cast<const float*>(dp + oUr) : nullptr;
    K->hasGrid = true;
    return ORBFE_OK;
}

int fuse_search_keyframe_run(MatchScratch& m, hipStream_t s, const KeyFrameDev* K, int mapCap, const orbfe_world_point* mapPts,
                             const uint8_t* mapDesc, int M, const int* ids, const orbfe_frustum* F, float th, int* bestIdxOut,
                             int* bestDistOut, std::string& err)
{
    for (int i = 0; i < M; i++) {
        bestIdxOut[i] = -1;
        bestDistOut[i] = 256;
    }
    if (!K->hasGrid) {
        err = "orbfe_fuse_search_keyframe: the key frame has no grid (orbfe_keyframe_set_grid)";
        return ORBFE_ERR_INVALID_ARG;
    }
    if (K->n == 0 || M == 0) return ORBFE_OK;
    if (F->n_levels > K->nLevels) return ORBFE_ERR_INVALID_ARG;  // predicted levels index the key frame's scale tables
    Carver c;
    const size_t oIds = c.take((size_t)M * sizeof(int));
    const size_t oBest = c.take((size_t)M * 2 * sizeof(int));
    int rc = ensure(m, c.off, c.off, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oIds, ids, (size_t)M * sizeof(int));
    MCHK(hipMemcpyAsync(dp + oIds, hp + oIds, (size_t)M * sizeof(int), hipMemcpyHostToDevice, s));
    int* dBest = reinterpret_cast<int*>(dp + oBest);
    hipLaunchKernelGGL(fuse_search_kernel<true>, dim3((M + 63) / 64), dim3(64), 0, s, K->grid, *F, th, M,
                       reinterpret_cast<const int*>(dp + oIds), mapCap, mapPts, mapDesc, K->invLevelSigma2, K->uRight, 1,
                       static_cast<const unsigned long long*>(nullptr), -1, dBest, dBest + M);
    MCHK(hipGetLastError());
    int* hBest = reinterpret_cast<int*>(hp + oBest);
    MCHK(hipMemcpyAsync(hBest, dBest, (size_t)M * 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(bestIdxOut, hBest, (size_t)M * sizeof(int));
    memcpy(bestDistOut, hBest + M, (size_t)M * sizeof(int));
    return ORBFE_OK;
}

namespace {

int view_unsupported(const orbfe_frame_view* V)
{
    return V->n >= (1 << 20) || V->grid_cols > 65535 || V->grid_rows > 32767 || V->n_levels < 1 ||
           (long long)V->grid_cols * V->grid_rows > kMaxCells;
}

// host staging offsets of one key frame's inputs for the thread-per-point searches
struct ViewStage {
    size_t oKp, oDesc, oSf, oN, oMatch;
};

ViewStage view_carve(Carver& in, const orbfe_frame_view* V)
{
    ViewStage st;
    st.oKp = in.take((size_t)V->n * sizeof(orbfe_keypoint));
    st.oDesc = in.take((size_t)V->n * 32);
    st.oSf = in.take((size_t)V->n_levels * sizeof(float));
    st.oN = in.take(sizeof(int));
    return st;
}

void view_fill(uint8_t* hp, const ViewStage& st, const orbfe_frame_view* V)
{
    memcpy(hp + st.oKp, V->kp, (size_t)V->n * sizeof(orbfe_keypoint));
    memcpy(hp + st.oDesc, V->desc, (size_t)V->n * 32);
    memcpy(hp + st.oSf, V->scale_factors, (size_t)V->n_levels * sizeof(float));
    memcpy(hp + st.oN, &V->n, sizeof(int));
}

void view_bind(ProjArgs& A, uint8_t* dp, const ViewStage& st)
{
    A.kp = reinterpret_cast<const orbfe_keypoint*>(dp + st.oKp);
    A.desc = dp + st.oDesc;
    A.nKp = reinterpret_cast<const int*>(dp + st.oN);
    A.scaleFactors = reinterpret_cast<const float*>(dp + st.oSf);
    A.matchOut = reinterpret_cast<int*>(dp + st.oMatch);
}

ProjArgs view_args(const orbfe_frame_view* V)
{
    ProjArgs A{};
    A.B = 1; A.M = 1; A.kpStride = V->n;
    A.g = GridDesc{V->grid_cols, V->grid_rows, V->min_x, V->min_y, V->grid_inv_w, V->grid_inv_h};
    A.nnRatio = 1.0f;
    A.nLevels = V->n_levels;
    return A;
}

void view_prepare(hipStream_t s, const ProjArgs& A)
{
    proj_prepare_launch(s, A, false);
}

}  // namespace

int search_by_sim3_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* KF1, const orbfe_frame_view* KF2,
                       const orbfe_sim3_view* d12, const orbfe_sim3_view* d21, const orbfe_world_point* mp1,
                       const uint8_t* mpDesc1, const orbfe_world_point* mp2, const uint8_t* mpDesc2, float th,
                       int* match12Out, int* nFound, std::string& err)
{
    const int n1 = KF1->n, n2 = KF2->n;
    for (int i = 0; i < n1; i++) match12Out[i] = -1;
    *nFound = 0;
    if (n1 == 0 || n2 == 0) return ORBFE_OK;
    if (view_unsupported(KF1) || view_unsupported(KF2)) return ORBFE_ERR_UNSUPPORTED;
    // predicted levels index the target key frame's scale tables
    if (d12->n_levels > KF2->n_levels || d21->n_levels > KF1->n_levels || d12->n_levels < 1 || d21->n_levels < 1)
        return ORBFE_ERR_INVALID_ARG;
    Carver in;
    ViewStage s1 = view_carve(in, KF1), s2 = view_carve(in, KF2);
    const size_t oP1 = in.take((size_t)n1 * sizeof(orbfe_world_point));
    const size_t oD1 = in.take((size_t)n1 * 32);
    const size_t oP2 = in.take((size_t)n2 * sizeof(orbfe_world_point));
    const size_t oD2 = in.take((size_t)n2 * 32);
    const size_t inBytes = in.off;
    s1.oMatch = in.take((size_t)n1 * sizeof(int));  // the grid kernels clear a match array each
    s2.oMatch = in.take((size_t)n2 * sizeof(int));
    const size_t oVn1 = in.take((size_t)n1 * sizeof(int));
    const size_t oVn2 = in.take((size_t)n2 * sizeof(int));
    const size_t oOut = in.take((size_t)(n1 + 1) * sizeof(int));
    const size_t hostNeed = inBytes + (size_t)(n1 + 1) * sizeof(int) + 64;
    ProjArgs A1 = view_args(KF1), A2 = view_args(KF2);
    size_t end1 = 0;
    int rc = proj_setup(m, A1, in, hostNeed, err, &end1);
    if (rc != ORBFE_OK) return rc;
    Carver after1;
    after1.off = end1;
    const void* d0 = m.d;
    rc = proj_setup(m, A2, after1, hostNeed, err);
    if (rc != ORBFE_OK) return rc;
    if (m.d != d0) {  // the arena grew: bind the first table set again (no further growth: the size is covered now)
        rc = proj_setup(m, A1, in, hostNeed, err);
        if (rc != ORBFE_OK) return rc;
    }
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    view_fill(hp, s1, KF1);
    view_fill(hp, s2, KF2);
    memcpy(hp + oP1, mp1, (size_t)n1 * sizeof(orbfe_world_point));
    memcpy(hp + oD1, mpDesc1, (size_t)n1 * 32);
    memcpy(hp + oP2, mp2, (size_t)n2 * sizeof(orbfe_world_point));
    memcpy(hp + oD2, mpDesc2, (size_t)n2 * 32);
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    view_bind(A1, dp, s1);
    view_bind(A2, dp, s2);
    view_prepare(s, A1);
    view_prepare(s, A2);
    int* vn1 = reinterpret_cast<int*>(dp + oVn1);
    int* vn2 = reinterpret_cast<int*>(dp + oVn2);
    int* dOut = reinterpret_cast<int*>(dp + oOut);
    // key frame 1's map points into key frame 2 (tables A2), then the other way round
    hipLaunchKernelGGL(sim3_search_kernel, dim3((n1 + 63) / 64), dim3(64), 0, s, A2, *d12, th, n1,
                       reinterpret_cast<const orbfe_world_point*>(dp + oP1), dp + oD1, vn1);
    hipLaunchKernelGGL(sim3_search_kernel, dim3((n2 + 63) / 64), dim3(64), 0, s, A1, *d21, th, n2,
                       reinterpret_cast<const orbfe_world_point*>(dp + oP2), dp + oD2, vn2);
    MCHK(hipMemsetAsync(dOut + n1, 0, sizeof(int), s));
    hipLaunchKernelGGL(sim3_agree_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, n1, vn1, vn2, dOut, dOut + n1);
    MCHK(hipGetLastError());
    int* hOut = reinterpret_cast<int*>(hp + inBytes);
    MCHK(hipMemcpyAsync(hOut, dOut, (size_t)(n1 + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(match12Out, hOut, (size_t)n1 * sizeof(int));
    *nFound = hOut[n1];
    return ORBFE_OK;
}

int match_projection_kf_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F, const orbfe_frustum* Fr, int M,
                            const orbfe_world_point* pts, const uint8_t* mpDesc, const float* kfAngle,
                            const uint8_t* frameHasMP, float th, int checkOrientation, int* matchOut, int* nMatches,
                            std::string& err)
{
    const int n = F->n;
    for (int i = 0; i < n; i++) matchOut[i] = -1;
    *nMatches = 0;
    if (n == 0 || M == 0) return ORBFE_OK;
    if (view_unsupported(F)) return ORBFE_ERR_UNSUPPORTED;
    if (Fr->n_levels > F->n_levels || Fr->n_levels < 1) return ORBFE_ERR_INVALID_ARG;
    Carver in;
    ViewStage st = view_carve(in, F);
    const size_t oPts = in.take((size_t)M * sizeof(orbfe_world_point));
    const size_t oMpDesc = in.take((size_t)M * 32);
    const size_t oAng = in.take((size_t)M * sizeof(float));
    const size_t oObs = in.take((size_t)n * sizeof(int));
    const size_t inBytes = in.off;
    st.oMatch = in.take((size_t)(n + 1) * sizeof(int));  // matches, then the count
    const size_t oMp = in.take((size_t)M * sizeof(orbfe_map_point));
    const size_t hostNeed = inBytes + (size_t)(n + 1) * sizeof(int) + 64;
    ProjArgs A = view_args(F);
    A.M = M;
    A.mode = kModeReloc;
    A.th = th;
    A.nnRatio = 3.0e38f;  // "best <= nnRatio * second" always holds: this overload has no ratio test
    int rc = proj_setup(m, A, in, hostNeed, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    view_fill(hp, st, F);
    memcpy(hp + oPts, pts, (size_t)M * sizeof(orbfe_world_point));
    memcpy(hp + oMpDesc, mpDesc, (size_t)M * 32);
    if (kfAngle) memcpy(hp + oAng, kfAngle, (size_t)M * sizeof(float));
    int* hObs = reinterpret_cast<int*>(hp + oObs);
    for (int i = 0; i < n; i++) hObs[i] = (frameHasMP && frameHasMP[i]) ? 1 : -1;  // :1268: an occupied slot is never a candidate
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    view_bind(A, dp, st);
    A.mps = reinterpret_cast<const orbfe_map_point*>(dp + oMp);
    A.mpDesc = dp + oMpDesc;
    A.initObs = reinterpret_cast<const int*>(dp + oObs);
    A.nMatches = A.matchOut + n;
    hipLaunchKernelGGL(reloc_project_kernel, dim3((M + 255) / 256), dim3(256), 0, s, *Fr, M,
                       reinterpret_cast<const orbfe_world_point*>(dp + oPts), reinterpret_cast<orbfe_map_point*>(dp + oMp));
    rc = proj_launch(s, A, err);
    if (rc != ORBFE_OK) return rc;
    hipLaunchKernelGGL(reloc_finalize_kernel, dim3(1), dim3(256), 0, s, n, A.kp, reinterpret_cast<const float*>(dp + oAng),
                       checkOrientation, A.matchOut, A.nMatches);
    MCHK(hipGetLastError());
    int* hOut = reinterpret_cast<int*>(hp + inBytes);
    MCHK(hipMemcpyAsync(hOut, A.matchOut, (size_t)(n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(matchOut, hOut, (size_t)n * sizeof(int));
    *nMatches = hOut[n];
    return ORBFE_OK;
}

}  // namespace orbfe
