// kernels_match_tri.hip -- ORBmatcher::SearchForTriangulation on gfx950.
//
// Replaces src/ORBmatcher.cc:441-676 for one pinhole camera per key frame (callers src/LocalMapping.cc:488, up
// to 30 neighbour key frames per new key frame) and Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:104-131).
// This fork never sets vbMatched2, so every key-frame-1 feature picks its partner independently of the others:
// one thread per feature of key frame 1 walks the features of key frame 2 that share its vocabulary node
// (sequentially, so the "dist > bestDist -> continue" rule keeps the LAST candidate among equal distances, as in
// the reference), then one block applies the rotation histogram (ComputeThreeMaxima).  Distances are __popcll
// over 4 x 64-bit words; the geometric gates are SPEC DECISION S8 binary32 arithmetic.
#include <algorithm>
#include <cstring>

#include <vector>

#include "match_common.h"
#include "device_math.h"
#include "keyframe.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

struct TriArgs {
    int nPos1;                    // entries of idx1 (features of key frame 1 that sit in a shared node)
    const int* idx1;              // [nPos1]
    const int* grp1;              // [nPos1] group of every entry
    const int* off2;              // [G + 1]
    const int* idx2;
    const orbfe_keypoint* kp1;
    const orbfe_keypoint* kp2;
    const uint8_t* desc1;
    const uint8_t* desc2;
    const uint8_t* hasMP1;
    const uint8_t* hasMP2;
    const uint8_t* stereo1;       // or null
    const uint8_t* stereo2;
    const float* sf2;
    float F12[9];
    float epx, epy;
    int onlyStereo, coarse, checkOrientation;
    int model1, model2, kf1HasCamera2;
    float cam1[8], cam2[8], kbPrecision;
    float R12[9], t12[3];
    float sigma2_1[kMaxLevels];
    int n1;
    int* match12;                 // [n1]
    int* binOf;                   // [n1]
    int* nMatches;
};


// ---- KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220 -> TriangulateMatches :306-370) ----
// SPEC DECISION S10.  binary32, one operation per line, no contraction, except the null vector of the 4x4 system, which the
// reference takes from an Eigen JacobiSVD (not reproducible): here A^T A in binary64, eight cyclic Jacobi sweeps in the
// fixed pair order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3), eigenvector of the smallest eigenvalue (lowest index on ties).
// tan(theta) of unproject is sin / cos of the S5 sequences.  Same sequence as oracle/match_oracle.c kb8_epipolar.
struct CamP {
    float fx, fy, cx, cy, k1, k2, k3, k4;
    int camera_model;
};

__device__ __forceinline__ CamP cam_of(const float (&c)[8], int model)
{
    return CamP{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], model};
}

__device__ inline void cam_unproject(const CamP& C, float precision, float u, float v, float& rx, float& ry)
{
    const float pwx = (u - C.cx) / C.fx;
    const float pwy = (v - C.cy) / C.fy;
    rx = pwx;
    ry = pwy;
    if (C.camera_model == 0) return;  // Pinhole::unproject (src/CameraModels/Pinhole.cpp:57-60)
    // KannalaBrandt8::unproject (:115-142): Newton on theta (1 + k1 theta^2 + ...) = theta_d
    float scale = 1.0f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    const float kHalfPi = 0x1.921fb6p+0f;
    theta_d = fminf(fmaxf(-kHalfPi, theta_d), kHalfPi);
    if (theta_d > 1e-8f) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0t2 = C.k1 * theta2, k1t4 = C.k2 * theta4, k2t6 = C.k3 * theta6, k3t8 = C.k4 * theta8;
            const float num = theta * ((((1.0f + k0t2) + k1t4) + k2t6) + k3t8) - theta_d;
            const float den = (((1.0f + 3.0f * k0t2) + 5.0f * k1t4) + 7.0f * k2t6) + 9.0f * k3t8;
            const float fix = num / den;
            theta = theta - fix;
            if (fabsf(fix) < precision) break;
        }
        float c, sn;
        cos_sin_deg(theta * 0x1.ca5dc2p+5f, c, sn);  // theta in [0, pi/2] as degrees
        scale = (sn / c) / theta_d;
    }
    rx = pwx * scale;
    ry = pwy * scale;
}

// smallest-eigenvalue eigenvector of the symmetric 4x4 matrix M (destroyed)
__device__ inline void sym4_min_eigenvector(double (&M)[4][4], double (&vOut)[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 8; sweep++) {
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                const double apq = M[p][q];
                if (apq == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double sn = t * c;
                for (int k = 0; k < 4; k++) {  // columns p, q of M
                    const double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = c * mkp - sn * mkq;
                    M[k][q] = sn * mkp + c * mkq;
                }
                for (int k = 0; k < 4; k++) {  // rows p, q of M
                    const double mpk = M[p][k], mqk = M[q][k];
                    M[p][k] = c * mpk - sn * mqk;
                    M[q][k] = sn * mpk + c * mqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int i = 1; i < 4; i++)
        if (M[i][i] < M[m][m]) m = i;
    for (int k = 0; k < 4; k++) vOut[k] = V[k][m];
}

__device__ inline bool kb8_epipolar(const CamP& C1, const CamP& C2, float precision, float u1, float v1, float u2, float v2,
                                    const float (&R12)[9], const float (&t12)[3], float sigmaLevel, float unc)
{
    float r1x, r1y, r2x, r2y;
    cam_unproject(C1, precision, u1, v1, r1x, r1y);  // rays (x, y, 1)
    cam_unproject(C2, precision, u2, v2, r2x, r2y);
    // parallax (:313-319)
    const float r21x = (R12[0] * r2x + R12[1] * r2y) + R12[2];
    const float r21y = (R12[3] * r2x + R12[4] * r2y) + R12[5];
    const float r21z = (R12[6] * r2x + R12[7] * r2y) + R12[8];
    const float dot = (r1x * r21x + r1y * r21y) + r21z;
    const float n1 = sqrtf((r1x * r1x + r1y * r1y) + 1.0f);
    const float n2 = sqrtf((r21x * r21x + r21y * r21y) + r21z * r21z);
    const float cosParallax = dot / (n1 * n2);
    if ((double)cosParallax > 0.9998) return false;
    // Tcw2 = [R21 | -R21 t12], R21 = R12^T (:333-336)
    float R21[9], tc[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) tc[i] = -((R21[3 * i] * t12[0] + R21[3 * i + 1] * t12[1]) + R21[3 * i + 2] * t12[2]);
    // A (:401-405) with Tcw1 = [I | 0]
    float A[4][4];
    A[0][0] = -1.0f; A[0][1] = 0.0f; A[0][2] = r1x; A[0][3] = 0.0f;
    A[1][0] = 0.0f; A[1][1] = -1.0f; A[1][2] = r1y; A[1][3] = 0.0f;
    for (int j = 0; j < 3; j++) {
        A[2][j] = r2x * R21[6 + j] - R21[j];
        A[3][j] = r2y * R21[6 + j] - R21[3 + j];
    }
    A[2][3] = r2x * tc[2] - tc[0];
    A[3][3] = r2y * tc[2] - tc[1];
    double M[4][4], vv[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc = acc + (double)A[k][i] * (double)A[k][j];
            M[i][j] = acc;
        }
    sym4_min_eigenvector(M, vv);
    const float X = (float)(vv[0] / vv[3]), Y = (float)(vv[1] / vv[3]), Z = (float)(vv[2] / vv[3]);
    if (!(Z > 0.0f)) return false;  // :343-346 (NaN fails, as a comparison "<= 0" on NaN would pass in the reference: S10)
    const float z2 = ((R21[6] * X + R21[7] * Y) + R21[8] * Z) + tc[2];
    if (!(z2 > 0.0f)) return false;  // :348-351
    float pu, pv;
    camera_project(C1, X, Y, Z, pu, pv);  // :354-361
    const float e1x = pu - u1, e1y = pv - v1;
    if ((double)(e1x * e1x + e1y * e1y) > 5.991 * (double)sigmaLevel) return false;
    const float X2 = ((R21[0] * X + R21[1] * Y) + R21[2] * Z) + tc[0];
    const float Y2 = ((R21[3] * X + R21[4] * Y) + R21[5] * Z) + tc[1];
    camera_project(C2, X2, Y2, z2, pu, pv);  // :363-371
    const float e2x = pu - u2, e2y = pv - v2;
    if ((double)(e2x * e2x + e2y * e2y) > 5.991 * (double)unc) return false;
    return Z > 0.0001f;  // :218-219
}

__global__ __launch_bounds__(256) void tri_match_kernel(TriArgs A)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= A.nPos1) return;
    const int idx1 = A.idx1[pos];
    if (A.hasMP1[idx1]) return;  // :506-509
    const bool bStereo1 = A.stereo1 && A.stereo1[idx1];
    if (A.onlyStereo && !bStereo1) return;
    const orbfe_keypoint k1 = A.kp1[idx1];
    unsigned long long d4[4];
    const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(A.desc1 + (size_t)idx1 * 32);
    d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
    // epipolar line of k1 in image 2, Pinhole.cpp:112-114
    const float a = (k1.x * A.F12[0] + k1.y * A.F12[3]) + A.F12[6];
    const float b = (k1.x * A.F12[1] + k1.y * A.F12[4]) + A.F12[7];
    const float c = (k1.x * A.F12[2] + k1.y * A.F12[5]) + A.F12[8];
    const float den = a * a + b * b;
    const int g = A.grp1[pos];
    int bestDist = ORBFE_TH_LOW, bestIdx2 = -1;
    for (int i2 = A.off2[g]; i2 < A.off2[g + 1]; i2++) {
        const int idx2 = A.idx2[i2];
        if (A.hasMP2[idx2]) continue;  // :531
        const bool bStereo2 = A.stereo2 && A.stereo2[idx2];
        if (A.onlyStereo && !bStereo2) continue;
        const int dist = hamming256(reinterpret_cast<const uint2*>(A.desc2 + (size_t)idx2 * 32), d4);
        if (dist > ORBFE_TH_LOW || dist > bestDist) continue;  // :545
        const orbfe_keypoint k2 = A.kp2[idx2];
        if (!bStereo1 && !bStereo2 && !A.kf1HasCamera2) {  // :551-565
            const float distex = A.epx - k2.x, distey = A.epy - k2.y;
            const float err = distex * distex + distey * distey;
            if (err < 100 * A.sf2[k2.octave]) continue;
        }
        bool ok = false;
        if (A.model1 == ORBFE_CAMERA_KANNALA_BRANDT8) {  // block-uniform
            if (!A.coarse)  // (the reference evaluates it either way; its result is only used without bCoarse)
                ok = kb8_epipolar(cam_of(A.cam1, A.model1), cam_of(A.cam2, A.model2), A.kbPrecision, k1.x, k1.y, k2.x, k2.y, A.R12,
                                  A.t12, A.sigma2_1[k1.octave], 1.0f);
        } else {
            const float num = (a * k2.x + b * k2.y) + c;
            if (den != 0) {
                const float dsqr = num * num / den;
                ok = (double)dsqr < 3.84 * 1.0;
            }
        }
        if (A.coarse || ok) {
            bestIdx2 = idx2;
            bestDist = dist;
        }
    }
    if (bestIdx2 >= 0) {
        A.match12[idx1] = bestIdx2;
        if (A.checkOrientation) {
            float rot = k1.angle - A.kp2[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            A.binOf[idx1] = bin;
        }
    }
}

// rotation-histogram filter (:633-661) + count; single block
__global__ __launch_bounds__(256) void tri_finalize_kernel(TriArgs A)
{
    __shared__ int hist[ORBFE_HISTO_LENGTH];
    __shared__ int sInd[3];
    __shared__ int sCount;
    const int tid = threadIdx.x;
    if (tid < ORBFE_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sCount = 0;
    __syncthreads();
    int local = 0;
    for (int j = tid; j < A.n1; j += blockDim.x)
        if (A.match12[j] >= 0) {
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
        for (int j = tid; j < A.n1; j += blockDim.x)
            if (A.match12[j] >= 0) {
                const int bb = A.binOf[j];
                if (bb != sInd[0] && bb != sInd[1] && bb != sInd[2]) {
                    A.match12[j] = -1;
                    local--;
                }
            }
    }
    if (local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) *A.nMatches = sCount;
}


// ---------------------------------------------------------------------------------------------
// One key frame against K neighbours in ONE launch (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:455-488: up to
// 30 SearchForTriangulation calls per new key frame).  Key frames are RESIDENT in HBM (keyframe.h): keypoints,
// descriptors and the FeatureVector as "features sorted by (vocabulary node, index)" + the list of distinct nodes, so a
// call moves only what changes between calls -- the has-map-point flags and the per-pair geometry.  blockIdx.y = neighbour,
// one thread per feature of key frame 1: binary search of its node in the neighbour's node list (staged in LDS), then the
// sequential walk of that node's features exactly as tri_match_kernel does.
// The rotation histogram is NOT applied here: between two neighbours the host turns matches into map points
// (:500-700), and a feature of key frame 1 that received one is skipped for all later neighbours (:506-509) -- which also
// changes their histograms.  The kernel therefore returns the RAW match and its rotation bin per (neighbour, feature),
// computed from the has-map-point flags at the time of the call; orbfe_triangulation_select applies "skip the features
// that got a map point meanwhile", the histogram and ComputeThreeMaxima for one neighbour at a time on the host.  Exact:
// vbMatched2 is never set in this fork (:485,537), so every feature of key frame 1 chooses on its own and dropping one
// never changes another's choice.
// ---------------------------------------------------------------------------------------------
struct TriNeighbour {  // device-side argument block of one (key frame 1, neighbour) pair
    const orbfe_keypoint* kp2;
    const uint8_t* desc2;
    const uint8_t* stereo2;   // or null
    const float* sf2;
    const int* nodeList2;     // [G2] distinct vocabulary nodes of the neighbour, ascending
    const int* nodeOff2;      // [G2 + 1]
    const int* order2;        // features of the neighbour sorted by (node, index)
    const uint8_t* hasMP2;    // [n2] flags of THIS call
    int G2, n2;
    float F12[9];
    float epx, epy;
    int onlyStereo, coarse, checkOrientation;
    int model1, model2, kf1HasCamera2;
    float cam1[8], cam2[8], kbPrecision;
    float R12[9], t12[3];
    float sigma2_1[kMaxLevels];
};

constexpr int kTriNodeLds = 4096;  // distinct nodes of a neighbour staged in LDS (more: the search reads global memory)

__global__ __launch_bounds__(256) void tri_batch_kernel(const TriNeighbour* __restrict__ nbs, int n1,
                                                        const orbfe_keypoint* __restrict__ kp1, const uint8_t* __restrict__ desc1,
                                                        const int* __restrict__ node1, const uint8_t* __restrict__ stereo1,
                                                        const uint8_t* __restrict__ hasMP1, int* __restrict__ rawMatch,
                                                        uint8_t* __restrict__ rawBin)
{
    __shared__ int sNodes[kTriNodeLds];
    const int k = blockIdx.y;
    const TriNeighbour& A = nbs[k];
    const int G2 = A.G2;
    const bool nodesInLds = G2 <= kTriNodeLds;  // block-uniform
    if (nodesInLds)
        for (int g = threadIdx.x; g < G2; g += 256) sNodes[g] = A.nodeList2[g];
    __syncthreads();
    const int idx1 = blockIdx.x * 256 + threadIdx.x;
    if (idx1 >= n1) return;
    int bestDist = ORBFE_TH_LOW, bestIdx2 = -1, bin = 0;
    const int nid = node1[idx1];
    const bool bStereo1 = stereo1 && stereo1[idx1];
    if (nid >= 0 && !hasMP1[idx1] && !(A.onlyStereo && !bStereo1)) {  // :506-509
        int lo = 0, hi = G2;  // lower bound of nid in the neighbour's node list
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int v = nodesInLds ? sNodes[mid] : A.nodeList2[mid];
            if (v < nid) lo = mid + 1;
            else hi = mid;
        }
        const bool shared = lo < G2 && (nodesInLds ? sNodes[lo] : A.nodeList2[lo]) == nid;
        if (shared) {
            const orbfe_keypoint k1 = kp1[idx1];
            unsigned long long d4[4];
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(desc1 + (size_t)idx1 * 32);
            d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
            const float a = (k1.x * A.F12[0] + k1.y * A.F12[3]) + A.F12[6];  // Pinhole.cpp:112-114
            const float b = (k1.x * A.F12[1] + k1.y * A.F12[4]) + A.F12[7];
            const float c = (k1.x * A.F12[2] + k1.y * A.F12[5]) + A.F12[8];
            const float den = a * a + b * b;
            for (int i2 = A.nodeOff2[lo]; i2 < A.nodeOff2[lo + 1]; i2++) {
                const int idx2 = A.order2[i2];
                if (A.hasMP2[idx2]) continue;  // :531
                const bool bStereo2 = A.stereo2 && A.stereo2[idx2];
                if (A.onlyStereo && !bStereo2) continue;
                const int dist = hamming256(reinterpret_cast<const uint2*>(A.desc2 + (size_t)idx2 * 32), d4);
                if (dist > ORBFE_TH_LOW || dist > bestDist) continue;  // :545
                const orbfe_keypoint k2 = A.kp2[idx2];
                if (!bStereo1 && !bStereo2 && !A.kf1HasCamera2) {  // :551-565
                    const float distex = A.epx - k2.x, distey = A.epy - k2.y;
                    const float err = distex * distex + distey * distey;
                    if (err < 100 * A.sf2[k2.octave]) continue;
                }
                bool ok = false;
                if (A.model1 == ORBFE_CAMERA_KANNALA_BRANDT8) {  // block-uniform
                    if (!A.coarse)
                        ok = kb8_epipolar(cam_of(A.cam1, A.model1), cam_of(A.cam2, A.model2), A.kbPrecision, k1.x, k1.y, k2.x, k2.y,
                                          A.R12, A.t12, A.sigma2_1[k1.octave], 1.0f);
                } else {
                    const float num = (a * k2.x + b * k2.y) + c;
                    if (den != 0) {
                        const float dsqr = num * num / den;
                        ok = (double)dsqr < 3.84 * 1.0;
                    }
                }
                if (A.coarse || ok) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {  // :619-627 (the bin is computed always; select uses it only with checkOrientation)
                float rot = k1.angle - A.kp2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
                if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            }
        }
    }
    rawMatch[(size_t)k * n1 + idx1] = bestIdx2;
    rawBin[(size_t)k * n1 + idx1] = (uint8_t)bin;
}

}  // namespace

int match_triangulation_run(MatchScratch& m, hipStream_t s, int G, const int* off1, const int* idx1, const int* off2,
                            const int* idx2, int n1, const orbfe_keypoint* kp1, const uint8_t* desc1, const uint8_t* hasMP1,
                            const uint8_t* stereo1, int n2, const orbfe_keypoint* kp2, const uint8_t* desc2,
                            const uint8_t* hasMP2, const uint8_t* stereo2, const float* sf2, int nLevels2,
                            const orbfe_tri_params* P, int* matches12, int* nMatches, std::string& err)
{
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    *nMatches = 0;
    if (G == 0 || n1 == 0 || n2 == 0) return ORBFE_OK;
    const int nPos1 = off1[G], nPos2 = off2[G];
    for (int g = 0; g < G; g++)
        if (off1[g + 1] < off1[g] || off2[g + 1] < off2[g]) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < nPos1; i++)
        if (idx1[i] < 0 || idx1[i] >= n1) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < nPos2; i++)
        if (idx2[i] < 0 || idx2[i] >= n2) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < n2; i++)
        if (kp2[i].octave < 0 || kp2[i].octave >= nLevels2) return ORBFE_ERR_INVALID_ARG;  // indexes mvScaleFactors
    if (nPos1 == 0) return ORBFE_OK;

    Carver in;
    const size_t oIdx1 = in.take((size_t)nPos1 * sizeof(int));
    const size_t oGrp1 = in.take((size_t)nPos1 * sizeof(int));
    const size_t oOff2 = in.take((size_t)(G + 1) * sizeof(int));
    const size_t oIdx2 = in.take((size_t)std::max(nPos2, 1) * sizeof(int));
    const size_t oKp1 = in.take((size_t)n1 * sizeof(orbfe_keypoint));
    const size_t oKp2 = in.take((size_t)n2 * sizeof(orbfe_keypoint));
    const size_t oD1 = in.take((size_t)n1 * 32);
    const size_t oD2 = in.take((size_t)n2 * 32);
    const size_t oH1 = in.take((size_t)n1);
    const size_t oH2 = in.take((size_t)n2);
    const size_t oS1 = in.take((size_t)n1);
    const size_t oS2 = in.take((size_t)n2);
    const size_t oSf = in.take((size_t)nLevels2 * sizeof(float));
    const size_t inBytes = in.off;
    Carver sc = in;
    const size_t oMatch = sc.take((size_t)n1 * sizeof(int));
    const size_t oBin = sc.take((size_t)n1 * sizeof(int));
    const size_t oNM = sc.take(sizeof(int));
    int rc = ensure(m, sc.off, inBytes + (size_t)n1 * sizeof(int) + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oIdx1, idx1, (size_t)nPos1 * sizeof(int));
    int* grp = reinterpret_cast<int*>(hp + oGrp1);
    for (int g = 0; g < G; g++)
        for (int i = off1[g]; i < off1[g + 1]; i++) grp[i] = g;
    memcpy(hp + oOff2, off2, (size_t)(G + 1) * sizeof(int));
    memcpy(hp + oIdx2, idx2, (size_t)nPos2 * sizeof(int));
    memcpy(hp + oKp1, kp1, (size_t)n1 * sizeof(orbfe_keypoint));
    memcpy(hp + oKp2, kp2, (size_t)n2 * sizeof(orbfe_keypoint));
    memcpy(hp + oD1, desc1, (size_t)n1 * 32);
    memcpy(hp + oD2, desc2, (size_t)n2 * 32);
    memcpy(hp + oH1, hasMP1, (size_t)n1);
    memcpy(hp + oH2, hasMP2, (size_t)n2);
    if (stereo1) memcpy(hp + oS1, stereo1, (size_t)n1);
    if (stereo2) memcpy(hp + oS2, stereo2, (size_t)n2);
    memcpy(hp + oSf, sf2, (size_t)nLevels2 * sizeof(float));
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));

    TriArgs A{};
    A.nPos1 = nPos1;
    A.idx1 = reinterpret_cast<const int*>(dp + oIdx1);
    A.grp1 = reinterpret_cast<const int*>(dp + oGrp1);
    A.off2 = reinterpret_cast<const int*>(dp + oOff2);
    A.idx2 = reinterpret_cast<const int*>(dp + oIdx2);
    A.kp1 = reinterpret_cast<const orbfe_keypoint*>(dp + oKp1);
    A.kp2 = reinterpret_cast<const orbfe_keypoint*>(dp + oKp2);
    A.desc1 = dp + oD1;
    A.desc2 = dp + oD2;
    A.hasMP1 = dp + oH1;
    A.hasMP2 = dp + oH2;
    A.stereo1 = stereo1 ? dp + oS1 : nullptr;
    A.stereo2 = stereo2 ? dp + oS2 : nullptr;
    A.sf2 = reinterpret_cast<const float*>(dp + oSf);
    for (int i = 0; i < 9; i++) A.F12[i] = P->f12[i];
    A.epx = P->ep_x;
    A.epy = P->ep_y;
    A.onlyStereo = P->only_stereo;
    A.coarse = P->coarse;
    A.checkOrientation = P->check_orientation;
    A.model1 = P->camera_model1;
    A.model2 = P->camera_model2;
    A.kf1HasCamera2 = P->kf1_has_camera2;
    for (int i = 0; i < 8; i++) {
        A.cam1[i] = P->cam1[i];
        A.cam2[i] = P->cam2[i];
    }
    A.kbPrecision = P->kb_precision;
    for (int i = 0; i < 9; i++) A.R12[i] = P->r12[i];
    for (int i = 0; i < 3; i++) A.t12[i] = P->t12[i];
    for (int i = 0; i < kMaxLevels; i++) A.sigma2_1[i] = P->level_sigma2_1[i];
    if (P->camera_model1 == ORBFE_CAMERA_KANNALA_BRANDT8)
        for (int i = 0; i < n1; i++)
            if (kp1[i].octave < 0 || kp1[i].octave >= kMaxLevels) return ORBFE_ERR_INVALID_ARG;  // indexes mvLevelSigma2
    A.n1 = n1;
    A.match12 = reinterpret_cast<int*>(dp + oMatch);
    A.binOf = reinterpret_cast<int*>(dp + oBin);
    A.nMatches = reinterpret_cast<int*>(dp + oNM);
    const dim3 blk(256);
    hipLaunchKernelGGL(fill_kernel, dim3((n1 + 255) / 256), blk, 0, s, A.match12, -1, (size_t)n1);
    hipLaunchKernelGGL(tri_match_kernel, dim3((nPos1 + 63) / 64), dim3(64), 0, s, A);
    hipLaunchKernelGGL(tri_finalize_kernel, dim3(1), blk, 0, s, A);
    MCHK(hipGetLastError());
    int* hMatch = reinterpret_cast<int*>(hp + inBytes);
    int* hNM = hMatch + n1;
    MCHK(hipMemcpyAsync(hMatch, A.match12, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipMemcpyAsync(hNM, A.nMatches, sizeof(int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(matches12, hMatch, (size_t)n1 * sizeof(int));
    *nMatches = *hNM;
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------
// resident key frames (keyframe.h)
// ---------------------------------------------------------------------------------------------
int keyframe_create(int n, const orbfe_keypoint* kp, const uint8_t* desc, const int* nodeId, const uint8_t* stereo,
                    const float* sf, int nLevels, hipStream_t s, KeyFrameDev** out, std::string& err)
{
    *out = nullptr;
    if (n < 0 || nLevels < 1 || nLevels > kMaxLevels || (n > 0 && (!kp || !desc || !nodeId)) || !sf) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++)
        if (kp[i].octave < 0 || kp[i].octave >= nLevels) return ORBFE_ERR_INVALID_ARG;  // indexes mvScaleFactors / mvLevelSigma2
    // FeatureVector as sorted arrays: features with a node, by (node, feature index) -- DBoW2 appends the features of a
    // node in index order (TemplatedVocabulary.h:1157-1170), so this IS the order the reference walks them in
    std::vector<int> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; i++)
        if (nodeId[i] >= 0) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nodeId[a] < nodeId[b]; });
    std::vector<int> nodeList, nodeOff;
    for (size_t p = 0; p < order.size(); p++)
        if (p == 0 || nodeId[order[p]] != nodeId[order[p - 1]]) {
            nodeList.push_back(nodeId[order[p]]);
            nodeOff.push_back((int)p);
        }
    nodeOff.push_back((int)order.size());
    Carver c;
    const size_t oKp = c.take((size_t)std::max(n, 1) * sizeof(orbfe_keypoint));
    const size_t oDesc = c.take((size_t)std::max(n, 1) * 32);
    const size_t oNode = c.take((size_t)std::max(n, 1) * sizeof(int));
    const size_t oStereo = c.take((size_t)std::max(n, 1));
    const size_t oSf = c.take((size_t)kMaxLevels * sizeof(float));
    const size_t oOrder = c.take((order.size() + 1) * sizeof(int));
    const size_t oList = c.take((nodeList.size() + 1) * sizeof(int));
    const size_t oOff = c.take(nodeOff.size() * sizeof(int));
    std::vector<uint8_t> img(c.off, 0);
    if (n) {
        memcpy(&img[oKp], kp, (size_t)n * sizeof(orbfe_keypoint));
        memcpy(&img[oDesc], desc, (size_t)n * 32);
        memcpy(&img[oNode], nodeId, (size_t)n * sizeof(int));
        if (stereo) memcpy(&img[oStereo], stereo, (size_t)n);
    }
    memcpy(&img[oSf], sf, (size_t)nLevels * sizeof(float));
    if (!order.empty()) memcpy(&img[oOrder], order.data(), order.size() * sizeof(int));
    if (!nodeList.empty()) memcpy(&img[oList], nodeList.data(), nodeList.size() * sizeof(int));
    memcpy(&img[oOff], nodeOff.data(), nodeOff.size() * sizeof(int));
    KeyFrameDev* K = new (std::nothrow) KeyFrameDev();
    if (!K) return ORBFE_ERR_OUT_OF_MEMORY;
    if (hipMalloc(&K->block, c.off) != hipSuccess) {
        (void)hipGetLastError();
        delete K;
        err = "hipMalloc(key frame) failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    if (copy_sync(K->block, img.data(), c.off, hipMemcpyHostToDevice, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(K->block);
        delete K;
        err = "upload of the key frame failed";
        return ORBFE_ERR_HIP;
    }
    uint8_t* b = static_cast<uint8_t*>(K->block);
    K->n = n;
    K->nLevels = nLevels;
    K->G = (int)nodeList.size();
    K->hasStereo = stereo != nullptr;
    K->kp = reinterpret_cast<const orbfe_keypoint*>(b + oKp);
    K->desc = b + oDesc;
    K->node = reinterpret_cast<const int*>(b + oNode);
    K->stereo = stereo ? b + oStereo : nullptr;
    K->sf = reinterpret_cast<const float*>(b + oSf);
    K->order = reinterpret_cast<const int*>(b + oOrder);
    K->nodeList = reinterpret_cast<const int*>(b + oList);
    K->nodeOff = reinterpret_cast<const int*>(b + oOff);
    K->bytes = c.off;
    *out = K;
    return ORBFE_OK;
}

void keyframe_destroy(KeyFrameDev* K)
{
    if (!K) return;
    if (K->block) (void)hipFree(K->block);
    match_scratch_free(K->gridMem);  // the cell tables of keyframe_set_grid, if any
    delete K;
}

int match_triangulation_batch_run(MatchScratch& m, hipStream_t s, const KeyFrameDev* kf1, const uint8_t* hasMP1, int K,
                                  const KeyFrameDev* const* kf2, const uint8_t* const* hasMP2, const orbfe_tri_params* P,
                                  int* rawMatch, uint8_t* rawBin, std::string& err)
{
    const int n1 = kf1->n;
    if (K == 0 || n1 == 0) return ORBFE_OK;
    for (int k = 0; k < K; k++) {
        if (!kf2[k] || (!hasMP2[k] && kf2[k]->n > 0)) return ORBFE_ERR_INVALID_ARG;
        if (P[k].struct_size != (int)sizeof(orbfe_tri_params)) {
            err = "orbfe_tri_params.struct_size does not match this library (rebuild the caller against include/orbfe.h)";
            return ORBFE_ERR_INVALID_ARG;
        }
    }
    // one pinned block up: [argument blocks | has_mp1 | has_mp2 of every neighbour]; one block down: [raw match | raw bin]
    Carver in;
    const size_t oArgs = in.take((size_t)K * sizeof(TriNeighbour));
    const size_t oH1 = in.take((size_t)n1);
    std::vector<size_t> oH2((size_t)K);
    for (int k = 0; k < K; k++) oH2[(size_t)k] = in.take((size_t)std::max(kf2[k]->n, 1));
    const size_t inBytes = in.off;
    Carver sc = in;
    const size_t oMatch = sc.take((size_t)K * n1 * sizeof(int));
    const size_t oBin = sc.take((size_t)K * n1);
    const size_t outBytes = sc.off - oMatch;
    int rc = ensure(m, sc.off, inBytes + outBytes + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    TriNeighbour* args = reinterpret_cast<TriNeighbour*>(hp + oArgs);
    memcpy(hp + oH1, hasMP1, (size_t)n1);
    for (int k = 0; k < K; k++) {
        const KeyFrameDev* F = kf2[k];
        if (F->n) memcpy(hp + oH2[(size_t)k], hasMP2[k], (size_t)F->n);
        TriNeighbour& A = args[k];
        A.kp2 = F->kp; A.desc2 = F->desc; A.stereo2 = F->stereo; A.sf2 = F->sf;
        A.nodeList2 = F->nodeList; A.nodeOff2 = F->nodeOff; A.order2 = F->order;
        A.hasMP2 = dp + oH2[(size_t)k];
        A.G2 = F->G; A.n2 = F->n;
        const orbfe_tri_params& Q = P[k];
        for (int i = 0; i < 9; i++) A.F12[i] = Q.f12[i];
        A.epx = Q.ep_x; A.epy = Q.ep_y;
        A.onlyStereo = Q.only_stereo; A.coarse = Q.coarse; A.checkOrientation = Q.check_orientation;
        A.model1 = Q.camera_model1; A.model2 = Q.camera_model2; A.kf1HasCamera2 = Q.kf1_has_camera2;
        for (int i = 0; i < 8; i++) { A.cam1[i] = Q.cam1[i]; A.cam2[i] = Q.cam2[i]; }
        A.kbPrecision = Q.kb_precision;
        for (int i = 0; i < 9; i++) A.R12[i] = Q.r12[i];
        for (int i = 0; i < 3; i++) A.t12[i] = Q.t12[i];
        for (int i = 0; i < kMaxLevels; i++) A.sigma2_1[i] = Q.level_sigma2_1[i];
    }
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(tri_batch_kernel, dim3((n1 + 255) / 256, K), dim3(256), 0, s, reinterpret_cast<const TriNeighbour*>(dp + oArgs), n1,
                       kf1->kp, kf1->desc, kf1->node, kf1->stereo, dp + oH1, reinterpret_cast<int*>(dp + oMatch), dp + oBin);
    MCHK(hipGetLastError());
    uint8_t* hOut = hp + inBytes;
    MCHK(hipMemcpyAsync(hOut, dp + oMatch, outBytes, hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    memcpy(rawMatch, hOut, (size_t)K * n1 * sizeof(int));
    memcpy(rawBin, hOut + (oBin - oMatch), (size_t)K * n1);
    return ORBFE_OK;
}

}  // namespace orbfe
