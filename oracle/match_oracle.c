/*
 * match_oracle.c -- CPU ORACLE for the ORBmatcher hot loops.  TEST INFRASTRUCTURE ONLY.
 * Literal restatement of src/ORBmatcher.cc:31-131 (SearchByProjection, mono), :133-327
 * (SearchByBoW, mono), :1328-1370 (ComputeThreeMaxima), :1375-1391 (DescriptorDistance) and of
 * the Frame grid they lean on, src/Frame.cc:145-176 (AssignFeaturesToGrid), :404-468
 * (GetFeaturesInArea), :470-480 (PosInGrid).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_LOW 30        /* include/ORBmatcher.h:73 */
#define TH_HIGH 100      /* :74 */
#define HISTO_LENGTH 30  /* :75 */

/* DescriptorDistance, src/ORBmatcher.cc:1375-1391 (bit-hack popcount on 8 int32 words) */
int orc_hamming(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* Frame::PosInGrid, src/Frame.cc:470-480 (round(), linear-index-only validation) */
static int pos_in_grid(const orc_frame_view *F, const orc_keypoint *kp, int *posX, int *posY)
{
    *posX = (int)roundf((kp->x - F->minX) * F->gridInvW);
    *posY = (int)roundf((kp->y - F->minY) * F->gridInvH);
    int linearIdx = *posY * F->gridCols + *posX;
    int size = F->gridCols * F->gridRows;
    return (linearIdx >= 0) & (linearIdx < size);
}

void orc_assign_grid(const orc_frame_view *F, int *cellOut)
{
    for (int i = 0; i < F->n; i++) {
        int px, py;
        cellOut[i] = pos_in_grid(F, &F->kp[i], &px, &py) ? py * F->gridCols + px : -1;
    }
}

typedef struct {
    int *idx;
    int n, cap;
} cell_t;

static cell_t *build_grid(const orc_frame_view *F)
{
    int nCells = F->gridCols * F->gridRows;
    cell_t *g = (cell_t *)calloc((size_t)nCells, sizeof(cell_t));
    for (int i = 0; i < F->n; i++) { /* AssignFeaturesToGrid, src/Frame.cc:157-176 */
        int px, py;
        if (pos_in_grid(F, &F->kp[i], &px, &py)) {
            cell_t *c = &g[py * F->gridCols + px];
            if (c->n == c->cap) {
                c->cap = c->cap ? 2 * c->cap : 4;
                c->idx = (int *)realloc(c->idx, sizeof(int) * (size_t)c->cap);
            }
            c->idx[c->n++] = i;
        }
    }
    return g;
}

static void free_grid(const orc_frame_view *F, cell_t *g)
{
    int nCells = F->gridCols * F->gridRows;
    for (int i = 0; i < nCells; i++) free(g[i].idx);
    free(g);
}

/* Frame::GetFeaturesInArea, src/Frame.cc:404-468 */
static int features_in_area(const orc_frame_view *F, const cell_t *g, float x, float y, float r,
                            int minLevel, int maxLevel, int *out)
{
    int n = 0;
    float factorX = r, factorY = r;
    int nMinCellX = (int)floorf((x - F->minX - factorX) * F->gridInvW);
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= F->gridCols) return 0;
    int nMaxCellX = (int)ceilf((x - F->minX + factorX) * F->gridInvW);
    if (nMaxCellX > F->gridCols - 1) nMaxCellX = F->gridCols - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - F->minY - factorY) * F->gridInvH);
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= F->gridRows) return 0;
    int nMaxCellY = (int)ceilf((y - F->minY + factorY) * F->gridInvH);
    if (nMaxCellY > F->gridRows - 1) nMaxCellY = F->gridRows - 1;
    if (nMaxCellY < 0) return 0;

    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const cell_t *c = &g[iy * F->gridCols + ix];
            for (int j = 0; j < c->n; j++) {
                const orc_keypoint *kpUn = &F->kp[c->idx[j]];
                if (bCheckLevels) {
                    if (kpUn->octave < minLevel || (maxLevel >= 0 && kpUn->octave > maxLevel)) continue;
                }
                const float distx = kpUn->x - x;
                const float disty = kpUn->y - y;
                if (fabsf(distx) < factorX && fabsf(disty) < factorY) out[n++] = c->idx[j];
            }
        }
    return n;
}

/* RadiusByViewingCos, src/ORBmatcher.cc:125-131 */
static float radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }

int orc_search_by_projection(const orc_frame_view *F, int M, const orc_map_point *mps,
                             const uint8_t *mpDesc, const int *initObs, float th, int bFarPoints,
                             float thFarPoints, float nnRatio, int *matchOut)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;
    cell_t *g = build_grid(F);
    int *vIndices = (int *)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
    /* slotObs[i]: Observations() of the map point currently in F->mvpMapPoints[i], -1 if none */
    int *slotObs = (int *)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
    for (int i = 0; i < F->n; i++) {
        slotObs[i] = initObs ? initObs[i] : -1;
        matchOut[i] = -1;
    }
    for (int iMP = 0; iMP < M; iMP++) {
        const orc_map_point *pMP = &mps[iMP];
        if (!pMP->inView) continue;                                   /* :40-41 (no right view in mono) */
        if (bFarPoints && pMP->trackDepth > thFarPoints) continue;    /* :43-44 */
        if (pMP->bad) continue;                                       /* :46-47 */
        const int nPredictedLevel = pMP->level;
        float r = radius_by_viewing_cos(pMP->viewCos);
        if (bFactor) r *= th;
        r *= F->scaleFactors[nPredictedLevel];
        int nI = features_in_area(F, g, pMP->projX, pMP->projY, r, nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (nI == 0) continue;
        const uint8_t *MPdescriptor = mpDesc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nI; k++) {
            const int idx = vIndices[k];
            if (slotObs[idx] > 0) continue;                           /* :77-79 */
            /* :81-86 stereo check is inert in mono (mvuRight < 0) */
            const int dist = orc_hamming(MPdescriptor, F->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F->kp[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F->kp[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {                                    /* :108-117 */
            if (bestLevel == bestLevel2 && (float)bestDist > nnRatio * (float)bestDist2) continue;
            if (bestLevel != bestLevel2 || (float)bestDist <= nnRatio * (float)bestDist2) {
                matchOut[bestIdx] = iMP;
                slotObs[bestIdx] = pMP->observations;
                nmatches++;
            }
        }
    }
    free(vIndices);
    free(slotObs);
    free_grid(F, g);
    return nmatches;
}

/* ComputeThreeMaxima, src/ORBmatcher.cc:1328-1370 */
static void compute_three_maxima(const int *histoSize, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histoSize[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

/* nLeft = F->Nleft: -1 for one camera, else frame features >= nLeft belong to the right camera (:205-233, :263-286) */
int orc_search_by_bow_rig(int G, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx,
                          int nKF, const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfHasMP,
                          int nF, const uint8_t *fDesc, const float *fAngle, int nLeft, float nnRatio,
                          int checkOrientation, int *matchOut)
{
    (void)nKF;
    int nmatches = 0;
    for (int i = 0; i < nF; i++) matchOut[i] = -1;
    int *rotHist[HISTO_LENGTH];
    int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
        rotN[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;

    for (int g = 0; g < G; g++) { /* one shared vocabulary node, :161-287 */
        for (int iKF = kfOff[g]; iKF < kfOff[g + 1]; iKF++) {
            const int realIdxKF = kfIdx[iKF];
            if (!kfHasMP[realIdxKF]) continue;                 /* :172-176 */
            const uint8_t *dKF = kfDesc + (size_t)realIdxKF * 32;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
            for (int iF = fOff[g]; iF < fOff[g + 1]; iF++) {
                const int realIdxF = fIdx[iF];
                if (matchOut[realIdxF] >= 0) continue;         /* :188-189 / :206-207 */
                const int dist = orc_hamming(dKF, fDesc + (size_t)realIdxF * 32);
                if (nLeft == -1 || realIdxF < nLeft) {         /* :185-203 / :213-220 */
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                } else {                                       /* :222-229 */
                    if (dist < bestDist1R) {
                        bestDist2R = bestDist1R;
                        bestDist1R = dist;
                        bestIdxFR = realIdxF;
                    } else if (dist < bestDist2R) {
                        bestDist2R = dist;
                    }
                }
            }
            if (bestDist1 <= TH_LOW) {                          /* :237 */
                if ((float)bestDist1 < nnRatio * (float)bestDist2) {
                    matchOut[bestIdxF] = realIdxKF;
                    if (checkOrientation) {
                        float rot = kfAngle[realIdxKF] - fAngle[bestIdxF];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin][rotN[bin]++] = bestIdxF;
                    }
                    nmatches++;
                }
                if (bestDist1R <= TH_LOW) {                     /* :263-286: the ratio test there is "|| true" */
                    (void)bestDist2R;
                    matchOut[bestIdxFR] = realIdxKF;
                    if (checkOrientation) {
                        float rot = kfAngle[realIdxKF] - fAngle[bestIdxFR];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin][rotN[bin]++] = bestIdxFR;
                    }
                    nmatches++;
                }
            }
        }
    }
    if (checkOrientation) { /* :304-322 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                matchOut[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    return nmatches;
}

int orc_search_by_bow(int G, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx,
                      int nKF, const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfHasMP,
                      int nF, const uint8_t *fDesc, const float *fAngle, float nnRatio,
                      int checkOrientation, int *matchOut)
{
    return orc_search_by_bow_rig(G, kfOff, kfIdx, fOff, fIdx, nKF, kfDesc, kfAngle, kfHasMP, nF, fDesc, fAngle, -1, nnRatio,
                                 checkOrientation, matchOut);
}

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:329-439 (caller src/Tracking.cc:605-607).
 * F1 contributes keypoints + descriptors, F2 additionally its grid (GetFeaturesInArea on F2, :349).
 * matches12Out[i1] = index in F2 or -1 (vnMatches12); returns nmatches. */
int orc_search_for_initialization(const orc_frame_view *F1, const orc_frame_view *F2, int windowSize,
                                  float nnRatio, int checkOrientation, int *matches12Out)
{
    int nmatches = 0;
    const int n1 = F1->n, n2 = F2->n;
    for (int i = 0; i < n1; i++) matches12Out[i] = -1;
    int *rotHist[HISTO_LENGTH];
    int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
        rotN[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *vIndices2 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) {
        vMatchedDistance[i] = 0x7fffffff;
        vnMatches21[i] = -1;
    }
    cell_t *g = build_grid(F2);
    for (int i1 = 0; i1 < n1; i1++) {
        const orc_keypoint *kp1 = &F1->kp[i1];
        const int level1 = kp1->octave;
        if (level1 > 0) continue;                                                    /* :346-347 */
        const int nI = features_in_area(F2, g, kp1->x, kp1->y, (float)windowSize, level1, level1, vIndices2);
        if (nI == 0) continue;
        const uint8_t *d1 = F1->desc + (size_t)i1 * 32;
        int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
        for (int k = 0; k < nI; k++) {
            const int i2 = vIndices2[k];
            const int dist = orc_hamming(d1, F2->desc + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;                              /* :368-369 */
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {                                                    /* :383 */
            if ((float)bestDist < (float)bestDist2 * nnRatio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12Out[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12Out[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) {
                    float rot = F1->kp[i1].angle - F2->kp[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin][rotN[bin]++] = i1;
                }
            }
        }
    }
    if (checkOrientation) {                                                          /* :411-435 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                const int idx1 = rotHist[i][j];
                if (matches12Out[idx1] >= 0) {
                    matches12Out[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(vMatchedDistance);
    free(vnMatches21);
    free(vIndices2);
    free_grid(F2, g);
    return nmatches;
}

/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup),
 * Thirdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1227-1270, with FORB::distance
 * (Thirdparty/DBoW2/src/DBoW2/FORB.cpp:81-101).  The tree is given as CSR children lists in DBoW2's
 * child order; node 0 is the root; a node without children is a leaf.  S7: when a leaf is reached above
 * level L - levelsup the reference leaves nid uninitialised; the oracle returns the leaf's id. */
void orc_vocab_transform(int nNodes, const int *childOff, const int *childIdx, const uint8_t *nodeDesc,
                         const int *wordId, const double *weight, int L, const uint8_t *desc, int n,
                         int levelsup, int *wordOut, int *nodeOut, double *weightOut)
{
    (void)nNodes;
    for (int f = 0; f < n; f++) {
        const uint8_t *feature = desc + (size_t)f * 32;
        const int nid_level = L - levelsup;
        int nid = -1;
        if (nid_level <= 0) nid = 0;
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int c0 = childOff[final_id], c1 = childOff[final_id + 1];
            final_id = childIdx[c0];
            double best_d = (double)orc_hamming(feature, nodeDesc + (size_t)final_id * 32);
            for (int j = c0 + 1; j < c1; j++) {
                const int id = childIdx[j];
                const double d = (double)orc_hamming(feature, nodeDesc + (size_t)id * 32);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (childOff[final_id + 1] != childOff[final_id]);
        if (nid < 0) nid = final_id; /* S7 */
        wordOut[f] = wordId[final_id];
        nodeOut[f] = nid;
        if (weightOut) weightOut[f] = weight[final_id];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Frame::isInFrustum (src/Frame.cc:272-331) for a batch of map points, with
 * MapPoint::PredictScale (src/MapPoint.cc:572-587) and Pinhole::project (src/CameraModels/Pinhole.cpp:41-47).
 * SPEC DECISION S8: every float expression is evaluated left to right in binary32 with no
 * contraction (the reference's Eigen expressions compile to whatever -march=native allows), norms are
 * sqrtf((x*x + y*y) + z*z), and log() is orc_spec_logf below (the reference calls the platform libm).
 * ------------------------------------------------------------------------------------------ */
/* GeometricCamera::project of the two camera models (src/CameraModels/Pinhole.cpp:41-47,
 * src/CameraModels/KannalaBrandt8.cpp:66-83).  KannalaBrandt8 uses the S5 polynomial atan2 / cos / sin (the
 * reference calls libm): psi in radians is turned into degrees in [0, 360) for orc_cos_sin_deg. */
static void camera_project(const orc_frustum *F, float x, float y, float z, float *u, float *v)
{
    if (F->cameraModel == 0) {
        *u = F->fx * x / z + F->cx;
        *v = F->fy * y / z + F->cy;
        return;
    }
    const float x2_plus_y2 = x * x + y * y;
    const float theta = orc_spec_atan2f(sqrtf(x2_plus_y2), z);
    const float psi = orc_spec_atan2f(y, x);
    const float theta2 = theta * theta;
    const float theta3 = theta * theta2;
    const float theta5 = theta3 * theta2;
    const float theta7 = theta5 * theta2;
    const float theta9 = theta7 * theta2;
    const float r = (((theta + F->k1 * theta3) + F->k2 * theta5) + F->k3 * theta7) + F->k4 * theta9;
    float deg = psi * 0x1.ca5dc2p+5f; /* 180 / pi */
    if (deg < 0.0f) deg = deg + 360.0f;
    float c, s;
    orc_cos_sin_deg(deg, &c, &s);
    *u = F->fx * r * c + F->cx;
    *v = F->fy * r * s + F->cy;
}

float orc_spec_logf(float x)
{
    /* x = m * 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716:
     * 2s (1 + z/3 + z^2/5 + z^3/7 + z^4/9), z = s^2 (the next term is below 2^-28 relative) */
    if (!(x > 0.0f)) return -INFINITY;
    if (x > 3.0e38f) return INFINITY;
    int e;
    float m = frexpf(x, &e); /* exact */
    if (m < 0x1.6a09e6p-1f) {
        m = m * 2.0f;
        e -= 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 0x1.c71c72p-4f;
    p = p * z + 0x1.24924ap-3f;
    p = p * z + 0x1.99999ap-3f;
    p = p * z + 0x1.555556p-2f;
    p = p * z;
    const float t = s + s;
    const float r = t + t * p;
    const float ef = (float)e;
    return ef * 0x1.62ep-1f + (r + ef * 0x1.0bfbe8p-15f);
}

void orc_is_in_frustum(const orc_frustum *F, int n, const orc_world_point *pts, orc_map_point *out, float *projXR)
{
    for (int i = 0; i < n; i++) {
        const orc_world_point *p = &pts[i];
        orc_map_point *o = &out[i];
        o->projX = -1.0f; /* :275-276 */
        o->projY = -1.0f;
        o->viewCos = 0.0f;
        o->trackDepth = 0.0f;
        o->level = 0;
        o->inView = 0;
        o->bad = p->bad;
        o->observations = p->observations;
        if (projXR) projXR[i] = 0.0f;
        if (p->skip || p->bad) continue; /* src/Tracking.cc:1066-1069 */
        const float X = p->x, Y = p->y, Z = p->z;
        const float pcx = ((F->rcw[0] * X + F->rcw[1] * Y) + F->rcw[2] * Z) + F->tcw[0]; /* :282 */
        const float pcy = ((F->rcw[3] * X + F->rcw[4] * Y) + F->rcw[5] * Z) + F->tcw[1];
        const float pcz = ((F->rcw[6] * X + F->rcw[7] * Y) + F->rcw[8] * Z) + F->tcw[2];
        const float pcDist = sqrtf((pcx * pcx + pcy * pcy) + pcz * pcz);
        const float invz = 1.0f / pcz;
        if (pcz < 0.0f) continue; /* :288 */
        float u, v;
        camera_project(F, pcx, pcy, pcz, &u, &v); /* mpCamera->project(Pc), :291 */
        if (u < F->minX || u > F->maxX) continue; /* NaN (pcz == 0 with pcx == 0) passes, as in the reference */
        if (v < F->minY || v > F->maxY) continue;
        o->projX = u; /* :299-300: set before the distance test */
        o->projY = v;
        const float maxD = 1.1f * p->maxDistance, minD = 0.9f * p->minDistance; /* MapPoint.cc:543-553 */
        const float ox = X - F->twc[0], oy = Y - F->twc[1], oz = Z - F->twc[2];
        const float dist = sqrtf((ox * ox + oy * oy) + oz * oz);
        if (dist < minD || dist > maxD) continue;
        /* PredictScale */
        const float ratio = p->maxDistance / dist;
        const float q = orc_spec_logf(ratio) / F->logScaleFactor;
        int nScale;
        if (!(q > 0.0f)) nScale = 0; /* covers q <= 0 and NaN (the reference's int conversion is undefined there) */
        else if (q >= (float)F->nLevels) nScale = F->nLevels - 1;
        else {
            nScale = (int)ceilf(q);
            if (nScale >= F->nLevels) nScale = F->nLevels - 1;
        }
        o->inView = 1;
        o->level = nScale;
        o->viewCos = 1.0f; /* :316: the normal test is disabled in this fork */
        o->trackDepth = pcDist;
        if (projXR) projXR[i] = u - F->mbf * invz;
    }
}

/* ---------------------------------------------------------------------------------------------
 * SURVEY 8f row f2 (first half): the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false)
 * (src/ORBmatcher.cc:678-836): per map point the projection (:722-766, SPEC DECISION S8 arithmetic;
 * the reference applies Tcw through Sophus' quaternion form, here Rcw * p + tcw as in isInFrustum),
 * KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:790-833) and the chi-square gated nearest descriptor
 * (:779-826).  Returns bestIdx / bestDist per map point; the caller applies bestDist <= TH_LOW and the
 * map-point graph edits (:829-849), which stay on the host in list order.
 * pts[i].skip carries "!pMP || pMP->IsInKeyFrame(pKF)", pts[i].bad carries isBad() (:706-721).
 * uRight == NULL means a monocular key frame (mvuRight[i] < 0 for all i).
 * ------------------------------------------------------------------------------------------ */
static void fuse_search_body(const orc_frame_view *KF, const float *invLevelSigma2, const float *uRight,
                             const orc_frustum *F, float th, int M, const orc_world_point *pts, const uint8_t *mpDesc,
                             int chi2Gate, int nRight, int *bestIdxOut, int *bestDistOut)
{
    /* nRight >= 0 is bRight (:684-688, :820): KF describes the LEFT features (KF->n == pKF->NLeft, they fill mGrid and the
     * gates read them), KF->desc is all of mDescriptors (NLeft + nRight rows), the compared row and the returned index
     * are idx + NLeft.  A row beyond the matrix is skipped (the reference would read out of bounds there). */
    cell_t *g = build_grid(KF);
    int *vIndices = (int *)malloc(sizeof(int) * (size_t)(KF->n > 0 ? KF->n : 1));
    for (int i = 0; i < M; i++) {
        const orc_world_point *p = &pts[i];
        bestIdxOut[i] = -1;
        bestDistOut[i] = 256;
        if (p->skip || p->bad) continue;
        const float X = p->x, Y = p->y, Z = p->z;
        const float pcx = ((F->rcw[0] * X + F->rcw[1] * Y) + F->rcw[2] * Z) + F->tcw[0];
        const float pcy = ((F->rcw[3] * X + F->rcw[4] * Y) + F->rcw[5] * Z) + F->tcw[1];
        const float pcz = ((F->rcw[6] * X + F->rcw[7] * Y) + F->rcw[8] * Z) + F->tcw[2];
        if (pcz < 0.0f) continue; /* :725 */
        const float invz = 1 / pcz;
        float u, v;
        camera_project(F, pcx, pcy, pcz, &u, &v);
        if (!(u >= F->minX && u < F->maxX && v >= F->minY && v < F->maxY)) continue; /* KeyFrame::IsInImage */
        const float ur = u - F->mbf * invz;
        const float maxD = 1.1f * p->maxDistance, minD = 0.9f * p->minDistance;
        const float ox = X - F->twc[0], oy = Y - F->twc[1], oz = Z - F->twc[2];
        const float dist3D = sqrtf((ox * ox + oy * oy) + oz * oz);
        if (dist3D < minD || dist3D > maxD) continue; /* :748 */
        const float ratio = p->maxDistance / dist3D; /* PredictScale */
        const float q = orc_spec_logf(ratio) / F->logScaleFactor;
        int lvl;
        if (!(q > 0.0f)) lvl = 0;
        else if (q >= (float)F->nLevels) lvl = F->nLevels - 1;
        else {
            lvl = (int)ceilf(q);
            if (lvl >= F->nLevels) lvl = F->nLevels - 1;
        }
        const float radius = th * KF->scaleFactors[lvl]; /* :766 */
        const int nc = features_in_area(KF, g, u, v, radius, -1, -1, vIndices);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = vIndices[c];
            const orc_keypoint *kp = &KF->kp[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue; /* :787 */
            if (!chi2Gate) {
                /* the Sim3 overload has no reprojection gate (:937-953) */
            } else if (uRight && uRight[idx] >= 0) {
                const float ex = u - kp->x, ey = v - kp->y, er = ur - uRight[idx];
                const float e2 = (ex * ex + ey * ey) + er * er;
                if (e2 * invLevelSigma2[kpLevel] > 7.8) continue; /* float product against a double constant */
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
            }
            int row = idx;
            if (nRight >= 0) {
                if (idx >= nRight) continue;
                row = idx + KF->n; /* :820 */
            }
            const int dist = orc_hamming(mpDesc + (size_t)i * 32, KF->desc + (size_t)row * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = row;
            }
        }
        bestIdxOut[i] = bestIdx;
        bestDistOut[i] = bestDist;
    }
    free(vIndices);
    free_grid(KF, g);
}

void orc_fuse_search(const orc_frame_view *KF, const float *invLevelSigma2, const float *uRight, const orc_frustum *F,
                     float th, int M, const orc_world_point *pts, const uint8_t *mpDesc, int *bestIdxOut,
                     int *bestDistOut)
{
    fuse_search_body(KF, invLevelSigma2, uRight, F, th, M, pts, mpDesc, 1, -1, bestIdxOut, bestDistOut);
}

/* ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = true): F carries GetRightPose / GetRightTranslationInverse / mpCamera2 */
void orc_fuse_search_right(const orc_frame_view *KFleft, int nRight, const float *invLevelSigma2, const float *uRight,
                           const orc_frustum *F, float th, int M, const orc_world_point *pts, const uint8_t *mpDesc,
                           int *bestIdxOut, int *bestDistOut)
{
    fuse_search_body(KFleft, invLevelSigma2, uRight, F, th, M, pts, mpDesc, 1, nRight < 0 ? 0 : nRight, bestIdxOut,
                     bestDistOut);
}

/* The search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:864-975):
 * the same walk as above without the viewing-angle and chi-square tests.  The caller decomposes Scw as the
 * reference does (:867-868): F->rcw = Scw.rotationMatrix(), F->tcw = Scw.translation() / Scw.scale(),
 * F->twc = Tcw.inverse().translation(); pts[i].skip carries "spAlreadyFound.count(pMP)".  bestDist starts at
 * INT_MAX in the reference; 256 here is equivalent (every distance is <= 256 and only "<= TH_LOW" is read). */
void orc_fuse_search_sim3(const orc_frame_view *KF, const orc_frustum *F, float th, int M, const orc_world_point *pts,
                          const uint8_t *mpDesc, int *bestIdxOut, int *bestDistOut)
{
    fuse_search_body(KF, NULL, NULL, F, th, M, pts, mpDesc, 0, -1, bestIdxOut, bestDistOut);
}

/* PredictScale (src/MapPoint.cc:580-612), SPEC DECISION S8 */
static int predict_scale(float maxDistance, float dist, float logScaleFactor, int nLevels)
{
    const float ratio = maxDistance / dist;
    const float q = orc_spec_logf(ratio) / logScaleFactor;
    int lvl;
    if (!(q > 0.0f)) lvl = 0;
    else if (q >= (float)nLevels) lvl = nLevels - 1;
    else {
        lvl = (int)ceilf(q);
        if (lvl >= nLevels) lvl = nLevels - 1;
    }
    return lvl;
}

/* One direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1017-1092 for 1->2, :1094-1170 for 2->1): the map
 * points of the source key frame are taken to its camera frame (D->rcw, D->tcw = source pose), moved by the
 * similarity (D->sr = s*R, D->t), projected with the PINHOLE expression written out in the function itself
 * (:1038-1043: invz = 1/z, u = fx * (x * invz) + cx -- not GeometricCamera::project, and with pKF1's intrinsics
 * in both directions, :979-982) and searched in the target key frame.  SPEC DECISION S8 arithmetic. */
static void sim3_direction(const orc_frame_view *T, const orc_sim3_dir *D, int n, const orc_world_point *pts,
                           const uint8_t *mpDesc, float th, int *vnMatch)
{
    cell_t *g = build_grid(T);
    int *vIndices = (int *)malloc(sizeof(int) * (size_t)(T->n > 0 ? T->n : 1));
    for (int i = 0; i < n; i++) {
        vnMatch[i] = -1;
        const orc_world_point *p = &pts[i];
        if (p->skip || p->bad) continue; /* !pMP || vbAlreadyMatched (:1021), isBad (:1024) */
        const float X = p->x, Y = p->y, Z = p->z;
        const float ax = ((D->rcw[0] * X + D->rcw[1] * Y) + D->rcw[2] * Z) + D->tcw[0]; /* p3Dc1 = T1w * p3Dw */
        const float ay = ((D->rcw[3] * X + D->rcw[4] * Y) + D->rcw[5] * Z) + D->tcw[1];
        const float az = ((D->rcw[6] * X + D->rcw[7] * Y) + D->rcw[8] * Z) + D->tcw[2];
        const float bx = ((D->sr[0] * ax + D->sr[1] * ay) + D->sr[2] * az) + D->t[0]; /* p3Dc2 = S21 * p3Dc1 */
        const float by = ((D->sr[3] * ax + D->sr[4] * ay) + D->sr[5] * az) + D->t[1];
        const float bz = ((D->sr[6] * ax + D->sr[7] * ay) + D->sr[8] * az) + D->t[2];
        if (bz < 0.0f) continue; /* :1032 */
        const float invz = 1.0f / bz; /* 1.0 / z in double, rounded to float: identical to the float quotient */
        const float x = bx * invz, y = by * invz;
        const float u = D->fx * x + D->cx, v = D->fy * y + D->cy;
        if (!(u >= D->minX && u < D->maxX && v >= D->minY && v < D->maxY)) continue; /* KeyFrame::IsInImage */
        const float maxD = 1.1f * p->maxDistance, minD = 0.9f * p->minDistance;
        const float dist3D = sqrtf((bx * bx + by * by) + bz * bz); /* p3Dc2.norm() */
        if (dist3D < minD || dist3D > maxD) continue; /* :1052 */
        const int lvl = predict_scale(p->maxDistance, dist3D, D->logScaleFactor, D->nLevels);
        const float radius = th * T->scaleFactors[lvl]; /* :1060 */
        const int nc = features_in_area(T, g, u, v, radius, -1, -1, vIndices);
        int bestDist = 256 + 1, bestIdx = -1; /* INT_MAX in the reference: any candidate replaces it */
        for (int c = 0; c < nc; c++) {
            const int idx = vIndices[c];
            const int oct = T->kp[idx].octave;
            if (oct < lvl - 1 || oct > lvl) continue; /* :1078 */
            const int dist = orc_hamming(mpDesc + (size_t)i * 32, T->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx; /* :1088 */
    }
    free(vIndices);
    free_grid(T, g);
}

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:977-1200).  mp1 / mp2: one record per key-frame feature
 * (KF->n entries; skip = "no map point" or vbAlreadyMatched of :1001-1012, which the caller derives from
 * vpMatches12 and GetIndexInKeyFrame).  match12Out[i1] = index in key frame 2 whose map point the reference
 * writes into vpMatches12[i1], or -1; returns nFound. */
int orc_search_by_sim3(const orc_frame_view *KF1, const orc_frame_view *KF2, const orc_sim3_dir *d12,
                       const orc_sim3_dir *d21, const orc_world_point *mp1, const uint8_t *mpDesc1,
                       const orc_world_point *mp2, const uint8_t *mpDesc2, float th, int *match12Out)
{
    const int N1 = KF1->n, N2 = KF2->n;
    int *vn1 = (int *)malloc(sizeof(int) * (size_t)(N1 > 0 ? N1 : 1));
    int *vn2 = (int *)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1));
    sim3_direction(KF2, d12, N1, mp1, mpDesc1, th, vn1);
    sim3_direction(KF1, d21, N2, mp2, mpDesc2, th, vn2);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) { /* :1175-1189 */
        match12Out[i1] = -1;
        const int idx2 = vn1[i1];
        if (idx2 >= 0 && vn2[idx2] == i1) {
            match12Out[i1] = idx2;
            nFound++;
        }
    }
    free(vn1);
    free(vn2);
    return nFound;
}

/* ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, checkOrientation)
 * (src/ORBmatcher.cc:1202-1326, relocalisation).  pts[i] / mpDesc / kfAngle: one record per key-frame feature i
 * (skip = "no map point" or "in sAlreadyFound"); frameHasMP[i2] != 0 iff CurrentFrame->mvpMapPoints[i2] is set on
 * entry (NULL == none).  matchOut[i2] = key-frame feature index whose map point ends up in
 * CurrentFrame->mvpMapPoints[i2] through this call, or -1; returns nmatches.
 * Note what the reference does NOT test here: the depth sign (a point behind the camera projects through the
 * pinhole expression like any other) -- reproduced.  A NaN projection passes the bounds test (:1232-1235) and
 * reaches Frame::GetFeaturesInArea, whose float-to-int conversion is then undefined; it yields an empty query on
 * x86 (INT_MIN cell) -- the oracle makes that explicit. */
int orc_search_by_projection_kf(const orc_frame_view *F, const orc_frustum *Fr, int M, const orc_world_point *pts,
                                const uint8_t *mpDesc, const float *kfAngle, const uint8_t *frameHasMP, float th,
                                int checkOrientation, int *matchOut)
{
    int nmatches = 0;
    cell_t *g = build_grid(F);
    const int nF = F->n;
    int *vIndices = (int *)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
    uint8_t *taken = (uint8_t *)malloc((size_t)(nF > 0 ? nF : 1));
    int *rotHist[HISTO_LENGTH];
    int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
        rotN[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nF; i++) {
        matchOut[i] = -1;
        taken[i] = frameHasMP ? (frameHasMP[i] != 0) : 0;
    }
    for (int i = 0; i < M; i++) {
        const orc_world_point *p = &pts[i];
        if (p->skip || p->bad) continue; /* :1221-1225 */
        const float X = p->x, Y = p->y, Z = p->z;
        const float pcx = ((Fr->rcw[0] * X + Fr->rcw[1] * Y) + Fr->rcw[2] * Z) + Fr->tcw[0];
        const float pcy = ((Fr->rcw[3] * X + Fr->rcw[4] * Y) + Fr->rcw[5] * Z) + Fr->tcw[1];
        const float pcz = ((Fr->rcw[6] * X + Fr->rcw[7] * Y) + Fr->rcw[8] * Z) + Fr->tcw[2];
        float u, v;
        camera_project(Fr, pcx, pcy, pcz, &u, &v); /* :1230 */
        if (u < Fr->minX || u > Fr->maxX) continue;
        if (v < Fr->minY || v > Fr->maxY) continue;
        if (u != u || v != v) continue; /* see the header note */
        const float ox = X - Fr->twc[0], oy = Y - Fr->twc[1], oz = Z - Fr->twc[2];
        const float dist3D = sqrtf((ox * ox + oy * oy) + oz * oz);
        const float maxD = 1.1f * p->maxDistance, minD = 0.9f * p->minDistance;
        if (dist3D < minD || dist3D > maxD) continue; /* :1245 */
        const int lvl = predict_scale(p->maxDistance, dist3D, Fr->logScaleFactor, Fr->nLevels);
        const float radius = th * F->scaleFactors[lvl]; /* :1253 */
        const int nc = features_in_area(F, g, u, v, radius, lvl - 1, lvl + 1, vIndices);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = vIndices[c];
            if (taken[i2]) continue; /* :1268 */
            const int dist = orc_hamming(mpDesc + (size_t)i * 32, F->desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) { /* :1282 */
            matchOut[bestIdx2] = i;
            taken[bestIdx2] = 1;
            nmatches++;
            if (checkOrientation) {
                float rot = kfAngle[i] - F->kp[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (checkOrientation) { /* :1304-1323 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                matchOut[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(taken);
    free(vIndices);
    free_grid(F, g);
    return nmatches;
}

/* ---------------------------------------------------------------------------------------------
 * SURVEY 8f row f2 (second half): ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:441-676), one
 * pinhole camera per key frame (mpCamera2 == null).  The caller hands over what the function derives
 * from the two poses with the reference's own expressions -- the epipole ep (:451-454) and the fundamental
 * matrix F12 = K1^-T [t12]x R12 K2^-1 that Pinhole::epipolarConstrain rebuilds for every pair
 * (src/CameraModels/Pinhole.cpp:106-109), row-major -- plus the FeatureVector merge-walk as CSR groups.
 * hasMP* = "GetMapPoint(idx) != null", stereo* = "mvuRight[idx] >= 0" (NULL == monocular).
 * This fork never sets vbMatched2 (:470,531), so key-frame-1 features choose independently.
 * The per-pair float arithmetic follows SPEC DECISION S8 (left to right, no contraction).
 * matches12Out[i1] = index in key frame 2 or -1; returns nmatches.
 * ------------------------------------------------------------------------------------------ */
/* ---- KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220) = TriangulateMatches (:306-370)
 * returning a depth above 0.0001.  SPEC DECISION S10: binary32 one operation at a time as the reference's Eigen / cv
 * expressions read, EXCEPT the null vector of the 4x4 system A (Triangulate, :385-396), which the reference takes from
 * Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV) -- an iteration whose rounding is not reproducible outside Eigen.  The spec
 * takes the eigenvector of the smallest eigenvalue of A^T A, formed in binary64, by eight cyclic Jacobi sweeps in the pair
 * order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3).  tan(theta) of unproject is sin / cos of the S5 sequences.  PARITY UNPINNED
 * against Eigen: tests/test_triangulation.py measures the agreement of the verdicts with a float64 SVD restatement. */
void orc_kb8_unproject(const float cam[8], int model, float precision, float u, float v, float *rx, float *ry)
{
    const float pwx = (u - cam[2]) / cam[0];
    const float pwy = (v - cam[3]) / cam[1];
    *rx = pwx;
    *ry = pwy;
    if (model == 0) return; /* Pinhole::unproject, src/CameraModels/Pinhole.cpp:57-60 */
    /* KannalaBrandt8::unproject, :115-142 */
    float scale = 1.0f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    const float kHalfPi = 0x1.921fb6p+0f;
    theta_d = fminf(fmaxf(-kHalfPi, theta_d), kHalfPi);
    if (theta_d > 1e-8f) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2,
                        theta8 = theta4 * theta4;
            const float k0t2 = cam[4] * theta2, k1t4 = cam[5] * theta4, k2t6 = cam[6] * theta6, k3t8 = cam[7] * theta8;
            const float num = theta * ((((1.0f + k0t2) + k1t4) + k2t6) + k3t8) - theta_d;
            const float den = (((1.0f + 3.0f * k0t2) + 5.0f * k1t4) + 7.0f * k2t6) + 9.0f * k3t8;
            const float fix = num / den;
            theta = theta - fix;
            if (fabsf(fix) < precision) break;
        }
        float c, sn;
        orc_cos_sin_deg(theta * 0x1.ca5dc2p+5f, &c, &sn);
        scale = (sn / c) / theta_d;
    }
    *rx = pwx * scale;
    *ry = pwy * scale;
}

static void sym4_min_eigenvector(double M[4][4], double vOut[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 8; sweep++)
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                const double apq = M[p][q];
                if (apq == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double sn = t * c;
                for (int k = 0; k < 4; k++) {
                    const double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = c * mkp - sn * mkq;
                    M[k][q] = sn * mkp + c * mkq;
                }
                for (int k = 0; k < 4; k++) {
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
    int m = 0;
    for (int i = 1; i < 4; i++)
        if (M[i][i] < M[m][m]) m = i;
    for (int k = 0; k < 4; k++) vOut[k] = V[k][m];
}

static void project_cam(const float cam[8], int model, float x, float y, float z, float *u, float *v)
{
    orc_frustum F;
    memset(&F, 0, sizeof F);
    F.cameraModel = model;
    F.fx = cam[0], F.fy = cam[1], F.cx = cam[2], F.cy = cam[3];
    F.k1 = cam[4], F.k2 = cam[5], F.k3 = cam[6], F.k4 = cam[7];
    camera_project(&F, x, y, z, u, v);
}

int orc_kb8_epipolar_constrain(const orc_tri_cameras *C, float u1, float v1, float u2, float v2, float sigmaLevel,
                               float unc, float xyzOut[3])
{
    float r1x, r1y, r2x, r2y;
    orc_kb8_unproject(C->cam1, C->model1, C->precision, u1, v1, &r1x, &r1y);
    orc_kb8_unproject(C->cam2, C->model2, C->precision, u2, v2, &r2x, &r2y);
    const float *R12 = C->R12, *t12 = C->t12;
    /* :313-319 */
    const float r21x = (R12[0] * r2x + R12[1] * r2y) + R12[2];
    const float r21y = (R12[3] * r2x + R12[4] * r2y) + R12[5];
    const float r21z = (R12[6] * r2x + R12[7] * r2y) + R12[8];
    const float dot = (r1x * r21x + r1y * r21y) + r21z;
    const float n1 = sqrtf((r1x * r1x + r1y * r1y) + 1.0f);
    const float n2 = sqrtf((r21x * r21x + r21y * r21y) + r21z * r21z);
    const float cosParallax = dot / (n1 * n2);
    if ((double)cosParallax > 0.9998) return 0;
    /* :333-336 */
    float R21[9], tc[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) tc[i] = -((R21[3 * i] * t12[0] + R21[3 * i + 1] * t12[1]) + R21[3 * i + 2] * t12[2]);
    /* Triangulate, :385-396, Tcw1 = [I | 0] */
    float A[4][4];
    A[0][0] = -1.0f, A[0][1] = 0.0f, A[0][2] = r1x, A[0][3] = 0.0f;
    A[1][0] = 0.0f, A[1][1] = -1.0f, A[1][2] = r1y, A[1][3] = 0.0f;
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
    if (xyzOut) xyzOut[0] = X, xyzOut[1] = Y, xyzOut[2] = Z;
    if (!(Z > 0.0f)) return 0; /* :343-346 */
    const float z2 = ((R21[6] * X + R21[7] * Y) + R21[8] * Z) + tc[2];
    if (!(z2 > 0.0f)) return 0; /* :348-351 */
    float pu, pv;
    project_cam(C->cam1, C->model1, X, Y, Z, &pu, &pv); /* :354-361 */
    const float e1x = pu - u1, e1y = pv - v1;
    if ((double)(e1x * e1x + e1y * e1y) > 5.991 * (double)sigmaLevel) return 0;
    const float X2 = ((R21[0] * X + R21[1] * Y) + R21[2] * Z) + tc[0];
    const float Y2 = ((R21[3] * X + R21[4] * Y) + R21[5] * Z) + tc[1];
    project_cam(C->cam2, C->model2, X2, Y2, z2, &pu, &pv); /* :363-371 */
    const float e2x = pu - u2, e2y = pv - v2;
    if ((double)(e2x * e2x + e2y * e2y) > 5.991 * (double)unc) return 0;
    return Z > 0.0001f; /* :218-219 */
}

int orc_search_for_triangulation(int G, const int *off1, const int *idx1v, const int *off2, const int *idx2v, int n1,
                                 const orc_keypoint *kp1, const uint8_t *desc1, const uint8_t *hasMP1,
                                 const uint8_t *stereo1, int n2, const orc_keypoint *kp2, const uint8_t *desc2,
                                 const uint8_t *hasMP2, const uint8_t *stereo2, const float *scaleFactors2,
                                 const float *F12, float epx, float epy, int bOnlyStereo, int bCoarse,
                                 int checkOrientation, int *matches12Out)
{
    return orc_search_for_triangulation_cam(G, off1, idx1v, off2, idx2v, n1, kp1, desc1, hasMP1, stereo1, n2, kp2, desc2,
                                            hasMP2, stereo2, scaleFactors2, F12, epx, epy, bOnlyStereo, bCoarse,
                                            checkOrientation, NULL, matches12Out);
}

int orc_search_for_triangulation_cam(int G, const int *off1, const int *idx1v, const int *off2, const int *idx2v, int n1,
                                     const orc_keypoint *kp1, const uint8_t *desc1, const uint8_t *hasMP1,
                                     const uint8_t *stereo1, int n2, const orc_keypoint *kp2, const uint8_t *desc2,
                                     const uint8_t *hasMP2, const uint8_t *stereo2, const float *scaleFactors2,
                                     const float *F12, float epx, float epy, int bOnlyStereo, int bCoarse,
                                     int checkOrientation, const orc_tri_cameras *cams, int *matches12Out)
{
    const int kb8 = cams && cams->model1 == 1;
    const int noEpipoleGate = cams && cams->kf1HasCamera2;
    (void)n2;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12Out[i] = -1;
    int *rotHist[HISTO_LENGTH];
    int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
        rotN[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    for (int g = 0; g < G; g++) {
        for (int i1 = off1[g]; i1 < off1[g + 1]; i1++) {
            const int idx1 = idx1v[i1];
            if (hasMP1[idx1]) continue; /* :506-509 */
            const int bStereo1 = stereo1 && stereo1[idx1];
            if (bOnlyStereo && !bStereo1) continue;
            const orc_keypoint *k1 = &kp1[idx1];
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int i2 = off2[g]; i2 < off2[g + 1]; i2++) {
                const int idx2 = idx2v[i2];
                if (hasMP2[idx2]) continue; /* :531 (vbMatched2 is never set) */
                const int bStereo2 = stereo2 && stereo2[idx2];
                if (bOnlyStereo && !bStereo2) continue;
                const int dist = orc_hamming(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                if (dist > TH_LOW || dist > bestDist) continue; /* :545: ties replace the earlier one */
                const orc_keypoint *k2 = &kp2[idx2];
                if (!bStereo1 && !bStereo2 && !noEpipoleGate) { /* :551-565 */
                    const float distex = epx - k2->x, distey = epy - k2->y;
                    const float err = distex * distex + distey * distey;
                    if (err < 100 * scaleFactors2[k2->octave]) continue;
                }
                /* Pinhole::epipolarConstrain, src/CameraModels/Pinhole.cpp:111-125 */
                const float a = (k1->x * F12[0] + k1->y * F12[3]) + F12[6];
                const float b = (k1->x * F12[1] + k1->y * F12[4]) + F12[7];
                const float c = (k1->x * F12[2] + k1->y * F12[5]) + F12[8];
                const float num = (a * k2->x + b * k2->y) + c;
                const float den = a * a + b * b;
                int ok = 0;
                if (kb8) { /* pCamera1 is the KannalaBrandt8 of pKF1: :603 with sigma = mvLevelSigma2[kp1.octave], unc 1 */
                    if (!bCoarse)
                        ok = orc_kb8_epipolar_constrain(cams, k1->x, k1->y, k2->x, k2->y, cams->sigma2_1[k1->octave], 1.0f,
                                                        NULL);
                } else if (den != 0) {
                    const float dsqr = num * num / den;
                    ok = dsqr < 3.84 * 1.0; /* float against a double constant */
                }
                if (bCoarse || ok) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {
                matches12Out[idx1] = bestIdx2;
                nmatches++;
                if (checkOrientation) {
                    float rot = k1->angle - kp2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin][rotN[bin]++] = idx1;
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                matches12Out[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    return nmatches;
}

/* ---------------------------------------------------------------------------------------------
 * MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:343-416) for a batch of map points: set s holds
 * the observed descriptors desc[setOff[s] .. setOff[s+1]); the representative is the one with the least
 * MEDIAN Hamming distance to the set, where the median of row i is element (size_t)(0.5 * (N - 1)) of its
 * sorted distances (the row includes the zero self-distance), first minimum wins (:396-409).
 * bestIdxOut[s] = index inside the set (-1 for an empty set), bestMedianOut[s] = that median.
 * ------------------------------------------------------------------------------------------ */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

void orc_distinctive_descriptors(int nSets, const int *setOff, const uint8_t *desc, int *bestIdxOut,
                                 int *bestMedianOut)
{
    for (int s = 0; s < nSets; s++) {
        const int N = setOff[s + 1] - setOff[s];
        bestIdxOut[s] = -1;
        if (bestMedianOut) bestMedianOut[s] = 0;
        if (N <= 0) continue;
        const uint8_t *d = desc + (size_t)setOff[s] * 32;
        int *row = (int *)malloc(sizeof(int) * (size_t)N);
        int BestMedian = 0x7fffffff, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = i == j ? 0 : orc_hamming(d + (size_t)i * 32, d + (size_t)j * 32);
            qsort(row, (size_t)N, sizeof(int), cmp_int);
            const int median = row[(size_t)(0.5 * (double)(N - 1))];
            if (median < BestMedian) {
                BestMedian = median;
                BestIdx = i;
            }
        }
        free(row);
        bestIdxOut[s] = BestIdx;
        if (bestMedianOut) bestMedianOut[s] = BestMedian;
    }
}
