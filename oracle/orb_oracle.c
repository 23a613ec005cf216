/*
 * orb_oracle.c -- CPU ORACLE (test infrastructure, see orb_oracle.h for the pinning status).
 *
 * Every function cites the reference lines it restates (paths relative to the reference repo).
 * Build with -O2 -ffp-contract=off (S5: no FMA contraction).
 *
 * SPEC DECISIONS:
 *  S1  pyramid: corner-aligned bilinear, Q11 weights, single rounding; 5x5 separable Gaussian
 *      with Q8 taps {22,62,88,62,22}, BORDER_REFLECT_101, single rounding after both passes.
 *  S2  NMS reads the score map with the level's own stride.
 *  S2b candidates in raster order (y, x); high-threshold list then low-threshold list;
 *      pre-NMS cap and final cap keep the raster-first entries.
 *  S3  patch samples outside the level image use BORDER_REFLECT_101.
 *  S4  sorted-phase node order is the total order (count, UL.x, UL.y, creation order).
 *  S5  atan2 / cos / sin are fixed fp32 polynomial sequences (below), no libm, no FMA.
 *  S6  keypoints stay in level coordinates, size = (int)(31 * invScale).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LEVELS 32
#define PATCH_SIZE 31          /* src/ORBextractor.cc:78 */
#define HALF_PATCH_SIZE 15     /* :79 */
#define EDGE_THRESHOLD 5       /* :80 */

static const int8_t k_pattern[1024] = {
#include "brief_pattern.inc"
};

struct orc_extractor {
    int nfeatures, nFastFeatures, nlevels, iniThFAST, minThFAST, W, H;
    double scaleFactor; /* stored as double like include/ORBextractor.h:108 */
    float mvScaleFactor[ORC_MAX_LEVELS], mvInvScaleFactor[ORC_MAX_LEVELS];
    float mvLevelSigma2[ORC_MAX_LEVELS], mvInvLevelSigma2[ORC_MAX_LEVELS];
    int mnFeaturesPerLevel[ORC_MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    uint8_t *img[ORC_MAX_LEVELS], *blur[ORC_MAX_LEVELS];
    /* per-level candidate lists kept for inspection */
    int16_t *candXY[ORC_MAX_LEVELS];
    int *candResp[ORC_MAX_LEVELS];
    int candN[ORC_MAX_LEVELS], candHigh[ORC_MAX_LEVELS], preHigh[ORC_MAX_LEVELS],
        preLow[ORC_MAX_LEVELS];
};

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* cvRound: round half to even (lrint under the default rounding mode) */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

/* BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba), valid for any offset */
static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

static inline int px_reflect(const uint8_t *img, int w, int h, int pitch, int y, int x)
{
    return img[reflect101(y, h) * pitch + reflect101(x, w)];
}

/* ------------------------------------------------------------------------------------------ */
/* S5 float math                                                                              */
/* ------------------------------------------------------------------------------------------ */
#define F_PI      0x1.921fb6p+1f
#define F_PI_2    0x1.921fb6p+0f
#define F_PI_4    0x1.921fb6p-1f
#define F_TAN_PI_8 0x1.a8279ap-2f
#define F_A0 -0x1.555556p-2f
#define F_A1  0x1.99799ep-3f
#define F_A2 -0x1.1fe904p-3f
#define F_A3  0x1.61e174p-4f
#define F_S0 -0x1.555558p-3f
#define F_S1  0x1.110e32p-7f
#define F_S2 -0x1.9b7856p-13f
#define F_C0  0x1.555554p-5f
#define F_C1 -0x1.6c134p-10f
#define F_C2  0x1.9bfe2ep-16f
#define F_INV90   0x1.6c16c2p-7f
#define F_DEG2RAD 0x1.1df46ap-6f   /* (float)(CV_PI/180.f), src/cuda/Orb_gpu.cu:327 */

/* radians, emulates atan2f(y, x) to ~2 ulp with a fixed op sequence */
float orc_spec_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float t = mn / mx;
    float base = 0.0f;
    if (t > F_TAN_PI_8) {
        t = (t - 1.0f) / (t + 1.0f);
        base = F_PI_4;
    }
    float z = t * t;
    float p = F_A3 * z;
    p = p + F_A2;
    p = p * z;
    p = p + F_A1;
    p = p * z;
    p = p + F_A0;
    float r = p * z;
    r = r * t;
    r = r + t;
    r = base + r;
    if (ay > ax) r = F_PI_2 - r;
    if (x < 0.0f) r = F_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* src/cuda/Angle_gpu.cu:73-75 : atan2f -> +2pi if negative -> degrees */
float orc_atan2_deg(float m01, float m10)
{
#ifdef ORC_LIBM
    /* S5 distance study only (tools/s5_libm_study.py, liborb_oracle_libm.so): the host libm's atan2f in place of the
     * specified polynomial -- glibc is correctly rounded to < 1 ulp, the CUDA libm the reference runs is documented to
     * 2 ulp, so this is ONE admissible outcome of the reference's line, not "the" reference value */
    float kp_dir = atan2f(m01, m10);
#else
    float kp_dir = orc_spec_atan2f(m01, m10);
#endif
    if (kp_dir < 0.0f) kp_dir = kp_dir + 2.0f * F_PI;
    kp_dir = kp_dir * (180.0f / F_PI);
    return kp_dir;
}

/* replaces cosf/sinf(angle * factorPI) of src/cuda/Orb_gpu.cu:327-329 */
void orc_cos_sin_deg(float deg, float *c, float *s)
{
#ifdef ORC_LIBM
    {   /* S5 distance study only: the literal src/cuda/Orb_gpu.cu:327-329 on the host libm */
        const float angle = deg * F_DEG2RAD;
        *c = cosf(angle);
        *s = sinf(angle);
        return;
    }
#endif
    float kf = deg * F_INV90;
    kf = kf + 0.5f;
    int k = (int)kf;
    float r = deg - 90.0f * (float)k;
    float x = r * F_DEG2RAD;
    float z = x * x;
    float p = F_S2 * z;
    p = p + F_S1;
    p = p * z;
    p = p + F_S0;
    float sn = p * z;
    sn = sn * x;
    sn = sn + x;
    float q = F_C2 * z;
    q = q + F_C1;
    q = q * z;
    q = q + F_C0;
    float cs = q * z;
    cs = cs * z;
    float h = 0.5f * z;
    h = 1.0f - h;
    cs = cs + h;
    switch (k & 3) {
    case 0: *c = cs; *s = sn; break;
    case 1: *c = -sn; *s = cs; break;
    case 2: *c = -cs; *s = -sn; break;
    default: *c = sn; *s = -cs; break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* constructor tables: src/ORBextractor.cc:82-149, :587-605                                   */
/* ------------------------------------------------------------------------------------------ */
orc_extractor *orc_create(int nFeatures, int nFastFeatures, float scaleFactor_, int nLevels,
                          int iniThFAST, int minThFAST, int imageWidth, int imageHeight)
{
    if (nLevels < 1 || nLevels > ORC_MAX_LEVELS || imageWidth < 16 || imageHeight < 16) return NULL;
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    e->nfeatures = nFeatures;
    e->nFastFeatures = nFastFeatures;
    e->scaleFactor = scaleFactor_;
    e->nlevels = nLevels;
    e->iniThFAST = iniThFAST;
    e->minThFAST = minThFAST;
    e->W = imageWidth;
    e->H = imageHeight;

    e->mvScaleFactor[0] = 1.0f;
    e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nLevels; i++) {
        e->mvScaleFactor[i] = (float)(e->mvScaleFactor[i - 1] * e->scaleFactor); /* :98 */
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    for (int i = 0; i < nLevels; i++) {
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    /* :112-124 */
    float factor = (float)(1.0f / e->scaleFactor);
    float nDesired = (float)nFeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nLevels));
    int sum = 0;
    for (int level = 0; level < nLevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cv_round_f(nDesired);
        sum += e->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    e->mnFeaturesPerLevel[nLevels - 1] = nFeatures - sum > 0 ? nFeatures - sum : 0;

    /* :126-143 umax */
    int v, v0;
    int vmax = (int)floorf(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceilf(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }

    /* AllocatePyramid :587-605 */
    for (int level = 0; level < nLevels; level++) {
        if (level == 0) {
            e->lw[0] = imageWidth;
            e->lh[0] = imageHeight;
        } else {
            float scale = e->mvInvScaleFactor[level];
            e->lw[level] = cv_round_f(scale * (float)imageWidth);
            e->lh[level] = cv_round_f(scale * (float)imageHeight);
        }
        if (e->lw[level] < 16 || e->lh[level] < 16) {
            orc_destroy(e);
            return NULL;
        }
        size_t px = (size_t)e->lw[level] * e->lh[level];
        e->img[level] = (uint8_t *)malloc(px);
        e->blur[level] = (uint8_t *)malloc(px);
        e->candXY[level] = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(nFastFeatures > 0 ? nFastFeatures : 1));
        e->candResp[level] = (int *)malloc(sizeof(int) * (size_t)(nFastFeatures > 0 ? nFastFeatures : 1));
    }
    return e;
}

void orc_destroy(orc_extractor *e)
{
    if (!e) return;
    for (int l = 0; l < ORC_MAX_LEVELS; l++) {
        free(e->img[l]);
        free(e->blur[l]);
        free(e->candXY[l]);
        free(e->candResp[l]);
    }
    free(e);
}

int orc_get_tables(const orc_extractor *e, float *sf, float *inv, float *s2, float *is2, int *fpl,
                   int *umax16, int *levelW, int *levelH)
{
    for (int i = 0; i < e->nlevels; i++) {
        if (sf) sf[i] = e->mvScaleFactor[i];
        if (inv) inv[i] = e->mvInvScaleFactor[i];
        if (s2) s2[i] = e->mvLevelSigma2[i];
        if (is2) is2[i] = e->mvInvLevelSigma2[i];
        if (fpl) fpl[i] = e->mnFeaturesPerLevel[i];
        if (levelW) levelW[i] = e->lw[i];
        if (levelH) levelH[i] = e->lh[i];
    }
    if (umax16)
        for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax16[i] = e->umax[i];
    return e->nlevels;
}

static int node_cap_level(const orc_extractor *e, int level)
{
    /* nodes of one level: <= N+3, or <= 4*nIni after the unconditional first pass */
    int nIni = (int)roundf((float)e->lw[level] / (float)e->lh[level]);
    int a = e->mnFeaturesPerLevel[level] + 3, b = 4 * nIni;
    return a > b ? a : b;
}

int orc_max_keypoints(const orc_extractor *e)
{
    int t = 0;
    for (int l = 0; l < e->nlevels; l++) t += node_cap_level(e, l);
    return t;
}

const uint8_t *orc_level_image(const orc_extractor *e, int level, int blurred)
{
    if (level < 0 || level >= e->nlevels) return NULL;
    return blurred ? e->blur[level] : e->img[level];
}

int orc_level_candidates(const orc_extractor *e, int level, int16_t *xy, int *resp, int cap,
                         int *nHigh, int *preHigh, int *preLow)
{
    int n = e->candN[level];
    int m = n < cap ? n : cap;
    if (xy) memcpy(xy, e->candXY[level], sizeof(int16_t) * 2 * (size_t)m);
    if (resp) memcpy(resp, e->candResp[level], sizeof(int) * (size_t)m);
    if (nHigh) *nHigh = e->candHigh[level];
    if (preHigh) *preHigh = e->preHigh[level];
    if (preLow) *preLow = e->preLow[level];
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* S1 pyramid (replaces cv::cuda::resize INTER_LINEAR + Gaussian 5x5 s=1.2,                   */
/* src/ORBextractor.cc:145,607-623)                                                           */
/* ------------------------------------------------------------------------------------------ */
void orc_resize_bilinear(const uint8_t *src, int sw, int sh, int spitch, uint8_t *dst, int dw,
                         int dh, int dpitch)
{
    /* corner-aligned mapping src = dst * (srcSize/dstSize) with exact rational position;
     * fraction quantised to Q11 (round to nearest); far neighbour clamped to the last pixel. */
    for (int y = 0; y < dh; y++) {
        int64_t sy = (int64_t)y * sh;
        int y1 = (int)(sy / dh);
        int fy = (int)(sy % dh);
        int wy = (int)(((int64_t)fy * 2048 + dh / 2) / dh);
        int y2 = y1 + 1 < sh ? y1 + 1 : sh - 1;
        for (int x = 0; x < dw; x++) {
            int64_t sx = (int64_t)x * sw;
            int x1 = (int)(sx / dw);
            int fx = (int)(sx % dw);
            int wx = (int)(((int64_t)fx * 2048 + dw / 2) / dw);
            int x2 = x1 + 1 < sw ? x1 + 1 : sw - 1;
            uint32_t a = src[y1 * spitch + x1], b = src[y1 * spitch + x2];
            uint32_t c = src[y2 * spitch + x1], d = src[y2 * spitch + x2];
            uint32_t top = a * (uint32_t)(2048 - wx) + b * (uint32_t)wx;
            uint32_t bot = c * (uint32_t)(2048 - wx) + d * (uint32_t)wx;
            uint32_t v = top * (uint32_t)(2048 - wy) + bot * (uint32_t)wy;
            dst[y * dpitch + x] = (uint8_t)((v + (1u << 21)) >> 22);
        }
    }
}

void orc_gauss5(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch)
{
    static const int k[5] = {22, 62, 88, 62, 22}; /* Q8, sum 256 (sigma 1.2) */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * h);
    int *rx = (int *)malloc(sizeof(int) * (size_t)(w + 4));
    int *ry = (int *)malloc(sizeof(int) * (size_t)(h + 4));
    for (int x = -2; x < w + 2; x++) rx[x + 2] = reflect101(x, w);
    for (int y = -2; y < h + 2; y++) ry[y + 2] = reflect101(y, h);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = src + (size_t)y * spitch;
        for (int x = 0; x < w; x++) {
            int32_t s = 0;
            for (int t = 0; t < 5; t++) s += k[t] * row[rx[x + t]];
            tmp[y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        const int32_t *r0 = tmp + (size_t)ry[y] * w, *r1 = tmp + (size_t)ry[y + 1] * w,
                      *r2 = tmp + (size_t)ry[y + 2] * w, *r3 = tmp + (size_t)ry[y + 3] * w,
                      *r4 = tmp + (size_t)ry[y + 4] * w;
        for (int x = 0; x < w; x++) {
            int32_t s = k[0] * r0[x] + k[1] * r1[x] + k[2] * r2[x] + k[3] * r3[x] + k[4] * r4[x];
            dst[y * dpitch + x] = (uint8_t)((s + 32768) >> 16);
        }
    }
    free(rx);
    free(ry);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* FAST-9/16 + score + NMS: src/cuda/Fast_gpu.cu:55-395                                       */
/* ------------------------------------------------------------------------------------------ */

/* ring bit k <-> (dy,dx), SURVEY appendix B2 derived from Fast_gpu.cu:226-254 + :76-181 */
static const int8_t k_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
static const int8_t k_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};

/* equals the c_table lookup of Fast_gpu.cu:187-191 (incl. the popcount > 8 guard) */
int orc_fast_arc9(int mask)
{
    for (int s = 0; s < 16; s++) {
        int ok = 1;
        for (int j = 0; j < 9 && ok; j++)
            if (!((mask >> ((s + j) & 15)) & 1)) ok = 0;
        if (ok) return 1;
    }
    return 0;
}

/* diffType, Fast_gpu.cu:60-65: bit0 <=> x - v < -th, bit1 <=> x - v > th (x = ring px) */
static void calc_mask(const int ring[16], int v, int th, int *mask1, int *mask2)
{
    /* The reference's early-outs (:79,93,107,121) return partially filled masks that can
     * never hold a 9-arc, so they are equivalent to evaluating all 16 bits. */
    int m1 = 0, m2 = 0;
    for (int k = 0; k < 16; k++) {
        int diff = ring[k] - v;
        if (diff < -th) m1 |= 1 << k;
        if (diff > th) m2 |= 1 << k;
    }
    *mask1 = m1;
    *mask2 = m2;
}

/* same predicate as orc_fast_arc9 (checked exhaustively in tests/test_oracle_kat.py), branch-free */
static inline int arc9_fast(unsigned m)
{
    unsigned m2 = m | (m << 16);
    unsigned r = m2 & (m2 >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m2 >> 8;
    return (r & 0xffffu) != 0;
}

static int is_keypoint(int mask1, int mask2) { return arc9_fast((unsigned)mask1) || arc9_fast((unsigned)mask2); }

/* cornerScore, Fast_gpu.cu:193-216: binary search for the largest threshold still a corner */
static int corner_score(const int ring[16], int v, int threshold)
{
    int min = threshold + 1, max = 255;
    while (min <= max) {
        int mid = (min + max) >> 1;
        int m1, m2;
        calc_mask(ring, v, mid, &m1, &m2);
        if (is_keypoint(m1, m2))
            min = mid + 1;
        else
            max = mid - 1;
    }
    return min - 1;
}

int orc_fast_score(const uint8_t *img, int pitch, int x, int y, int threshold)
{
    int ring[16];
    int v = img[y * pitch + x];
    {
        /* isKeyPoint2's early reject (Fast_gpu.cu:237-245): ring bits 4 and 12 are an opposite
         * pair, every 9-arc holds one of them */
        const int e = img[y * pitch + x + 3] - v, w_ = img[y * pitch + x - 3] - v;
        if (!(e < -threshold || e > threshold || w_ < -threshold || w_ > threshold)) return 0;
    }
    for (int k = 0; k < 16; k++) ring[k] = img[(y + k_ring_dy[k]) * pitch + x + k_ring_dx[k]];
    int m1, m2;
    calc_mask(ring, v, threshold, &m1, &m2);
    if (!is_keypoint(m1, m2)) return 0;
    return corner_score(ring, v, threshold);
}

int orc_fast_detect(const uint8_t *img, int w, int h, int pitch, int threshold, int maxKeypoints,
                    int16_t *xy, int *resp, int *preNmsCount)
{
    /* Fast_gpu.cu:365-368 : minX = border, maxX = max(cols - border, border); strict compares */
    int minX = EDGE_THRESHOLD, minY = EDGE_THRESHOLD;
    int maxX = w - EDGE_THRESHOLD > EDGE_THRESHOLD ? w - EDGE_THRESHOLD : EDGE_THRESHOLD;
    int maxY = h - EDGE_THRESHOLD > EDGE_THRESHOLD ? h - EDGE_THRESHOLD : EDGE_THRESHOLD;
    int32_t *score = (int32_t *)calloc((size_t)w * h, sizeof(int32_t)); /* :355 memset */
    size_t capLoc = (size_t)(maxKeypoints > 0 ? maxKeypoints : 1);
    int16_t *loc = (int16_t *)malloc(sizeof(int16_t) * 2 * capLoc);
    unsigned counter = 0;
    for (int y = minY + 1; y < maxY; y++)
        for (int x = minX + 1; x < maxX; x++) {
            int s = orc_fast_score(img, pitch, x, y, threshold);
            if (s > 0) { /* thresholds >= 1 => a corner's score (>= threshold) is never 0 */
                score[y * w + x] = s; /* S2: level stride (:261) */
                unsigned ind = counter++; /* S2b: raster order instead of atomicAdd order */
                if (ind < (unsigned)maxKeypoints) {
                    loc[2 * ind] = (int16_t)x;
                    loc[2 * ind + 1] = (int16_t)y;
                }
            }
        }
    if (preNmsCount) *preNmsCount = (int)counter;
    unsigned count = counter < (unsigned)maxKeypoints ? counter : (unsigned)maxKeypoints; /* :377 */
    int out = 0;
    for (unsigned i = 0; i < count; i++) { /* nonmaxSuppression :289-319 */
        int x = loc[2 * i], y = loc[2 * i + 1];
        int s = score[y * w + x];
        int ismax = s > score[(y - 1) * w + x - 1] && s > score[(y - 1) * w + x] &&
                    s > score[(y - 1) * w + x + 1] && s > score[y * w + x - 1] &&
                    s > score[y * w + x + 1] && s > score[(y + 1) * w + x - 1] &&
                    s > score[(y + 1) * w + x] && s > score[(y + 1) * w + x + 1];
        if (ismax) {
            xy[2 * out] = (int16_t)x;
            xy[2 * out + 1] = (int16_t)y;
            resp[out] = s;
            out++;
        }
    }
    free(score);
    free(loc);
    return out; /* <= count <= maxKeypoints (:392) */
}

/* ------------------------------------------------------------------------------------------ */
/* DistributeOctTree: src/ORBextractor.cc:151-431 -- literal std::list choreography            */
/* ------------------------------------------------------------------------------------------ */
typedef struct qnode {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int *keys; /* indices into the candidate list, insertion order == vKeys order */
    int nkeys;
    int bNoMore;
    struct qnode *prev, *next;
} qnode;

typedef struct qlist {
    qnode *head, *tail;
    int size;
} qlist;

static void ql_push_front(qlist *l, qnode *n)
{
    n->prev = NULL;
    n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n;
    l->size++;
}
static void ql_push_back(qlist *l, qnode *n)
{
    n->next = NULL;
    n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n;
    l->size++;
}
static qnode *ql_erase(qlist *l, qnode *n)
{
    qnode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    free(n->keys);
    free(n);
    return nx;
}

/* ExtractorNode::DivideNode :151-207 */
static void divide_node(const qnode *p, const int16_t *xy, qnode *c[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int i = 0; i < 4; i++) {
        c[i] = (qnode *)calloc(1, sizeof(qnode));
        c[i]->keys = (int *)malloc(sizeof(int) * (size_t)(p->nkeys > 0 ? p->nkeys : 1));
    }
    qnode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;

    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;

    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;

    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;

    for (int i = 0; i < p->nkeys; i++) {
        int id = p->keys[i];
        float px = (float)xy[2 * id], py = (float)xy[2 * id + 1];
        qnode *t;
        if (px < n1->URx) t = (py < n1->BRy) ? n1 : n3;
        else t = (py < n1->BRy) ? n2 : n4;
        t->keys[t->nkeys++] = id;
    }
    for (int i = 0; i < 4; i++)
        if (c[i]->nkeys == 1) c[i]->bNoMore = 1;
}

typedef struct {
    int size;
    int seq; /* position in vSizeAndPointerToNode == creation order (S4 final tie-break) */
    qnode *node;
} size_node;

/* compareNodes :209-224 extended to the total order of S4 */
static int cmp_size_node(const void *a, const void *b)
{
    const size_node *e1 = (const size_node *)a, *e2 = (const size_node *)b;
    if (e1->size != e2->size) return e1->size < e2->size ? -1 : 1;
    if (e1->node->ULx != e2->node->ULx) return e1->node->ULx < e2->node->ULx ? -1 : 1;
    if (e1->node->ULy != e2->node->ULy) return e1->node->ULy < e2->node->ULy ? -1 : 1;
    if (e1->seq != e2->seq) return e1->seq < e2->seq ? -1 : 1;
    return 0;
}

int orc_distribute(int n, const int16_t *xy, const int *resp, int W, int H, int maxFeatures,
                   int *selIdx, int selCap)
{
    /* called with the full image box (0,W,0,H), ORBextractor.cc:485-486 */
    const int minX = 0, maxX = W, minY = 0, maxY = H;
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) return -1;
    const float hX = (float)(maxX - minX) / (float)nIni;

    qlist L = {0, 0, 0};
    qnode **ini = (qnode **)malloc(sizeof(qnode *) * (size_t)nIni);
    for (int i = 0; i < nIni; i++) {
        qnode *ni = (qnode *)calloc(1, sizeof(qnode));
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        ni->keys = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        ql_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) { /* :255-259 */
        const int kp_x = xy[2 * i];
        int b = (int)((float)kp_x / hX);
        ini[b]->keys[ini[b]->nkeys++] = i;
    }
    free(ini);
    for (qnode *it = L.head; it;) { /* :261-274 */
        if (it->nkeys == 1) { it->bNoMore = 1; it = it->next; }
        else if (it->nkeys == 0) it = ql_erase(&L, it);
        else it = it->next;
    }

    int bFinish = 0;
    size_node *vSize = NULL;
    int vSizeN = 0, vSizeCap = 0;
    while (!bFinish) {
        int prevSize = L.size;
        int nToExpand = 0;
        vSizeN = 0;
        qnode *it = L.head;
        while (it) { /* :290-354 */
            if (it->bNoMore) { it = it->next; continue; }
            qnode *c[4];
            divide_node(it, xy, c);
            for (int k = 0; k < 4; k++) {
                if (c[k]->nkeys > 0) {
                    ql_push_front(&L, c[k]);
                    if (c[k]->nkeys > 1) {
                        nToExpand++;
                        if (vSizeN == vSizeCap) {
                            vSizeCap = vSizeCap ? vSizeCap * 2 : 64;
                            vSize = (size_node *)realloc(vSize, sizeof(size_node) * (size_t)vSizeCap);
                        }
                        vSize[vSizeN].size = c[k]->nkeys;
                        vSize[vSizeN].seq = vSizeN;
                        vSize[vSizeN].node = c[k];
                        vSizeN++;
                    }
                } else {
                    free(c[k]->keys);
                    free(c[k]);
                }
            }
            it = ql_erase(&L, it);
        }
        if (L.size >= maxFeatures || L.size == prevSize) { /* :358-361 */
            bFinish = 1;
        } else if (L.size + nToExpand * 3 > maxFeatures) { /* :362 */
            while (!bFinish) {
                prevSize = L.size;
                int m = vSizeN;
                size_node *prev = (size_node *)malloc(sizeof(size_node) * (size_t)(m > 0 ? m : 1));
                memcpy(prev, vSize, sizeof(size_node) * (size_t)m);
                vSizeN = 0;
                qsort(prev, (size_t)m, sizeof(size_node), cmp_size_node);
                for (int j = m - 1; j >= 0; j--) { /* :374-422 */
                    qnode *c[4];
                    divide_node(prev[j].node, xy, c);
                    for (int k = 0; k < 4; k++) {
                        if (c[k]->nkeys > 0) {
                            ql_push_front(&L, c[k]);
                            if (c[k]->nkeys > 1) {
                                if (vSizeN == vSizeCap) {
                                    vSizeCap = vSizeCap ? vSizeCap * 2 : 64;
                                    vSize = (size_node *)realloc(vSize, sizeof(size_node) * (size_t)vSizeCap);
                                }
                                vSize[vSizeN].size = c[k]->nkeys;
                                vSize[vSizeN].seq = vSizeN;
                                vSize[vSizeN].node = c[k];
                                vSizeN++;
                            }
                        } else {
                            free(c[k]->keys);
                            free(c[k]);
                        }
                    }
                    ql_erase(&L, prev[j].node);
                    if (L.size >= maxFeatures) break;
                }
                free(prev);
                if (L.size >= maxFeatures || L.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vSize);

    /* best point per node, first wins ties: ORBextractor.cc:515-527 */
    int count = 0;
    for (qnode *it = L.head; it; it = it->next) {
        int best = it->keys[0];
        float maxResponse = (float)resp[best];
        for (int k = 1; k < it->nkeys; k++)
            if ((float)resp[it->keys[k]] > maxResponse) {
                best = it->keys[k];
                maxResponse = (float)resp[best];
            }
        if (count < selCap) selIdx[count] = best;
        count++;
    }
    while (L.head) ql_erase(&L, L.head);
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* IC_Angle: src/cuda/Angle_gpu.cu:26-80 ; rBRIEF: src/cuda/Orb_gpu.cu:311-350                 */
/* ------------------------------------------------------------------------------------------ */
static const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

float orc_ic_angle(const uint8_t *img, int w, int h, int pitch, int x, int y)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u)
        m_10 += u * px_reflect(img, w, h, pitch, y, x + u);
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, m_sum = 0;
        const int d = k_umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = px_reflect(img, w, h, pitch, y + v, x + u);
            int val_minus = px_reflect(img, w, h, pitch, y - v, x + u);
            v_sum += (val_plus - val_minus);
            m_sum += u * (val_plus + val_minus);
        }
        m_10 += m_sum;
        m_01 += v * v_sum;
    }
    return orc_atan2_deg((float)m_01, (float)m_10);
}

void orc_brief(const uint8_t *img, int w, int h, int pitch, int x, int y, float angleDeg,
               uint8_t desc[32])
{
    float a, b;
    orc_cos_sin_deg(angleDeg, &a, &b);
    for (int t = 0; t < 32; t++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const int8_t *p = &k_pattern[(16 * t + 2 * j) * 2];
            float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            /* getOrbValue, Orb_gpu.cu:311-315: row = rn(px*b + py*a), col = rn(px*a - py*b) */
            float r0 = x0 * b; float r0b = y0 * a; r0 = r0 + r0b;
            float c0 = x0 * a; float c0b = y0 * b; c0 = c0 - c0b;
            float r1 = x1 * b; float r1b = y1 * a; r1 = r1 + r1b;
            float c1 = x1 * a; float c1b = y1 * b; c1 = c1 - c1b;
            int t0 = px_reflect(img, w, h, pitch, y + (int)lrintf(r0), x + (int)lrintf(c0));
            int t1 = px_reflect(img, w, h, pitch, y + (int)lrintf(r1), x + (int)lrintf(c1));
            val |= (t0 < t1) << j;
        }
        desc[t] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* extractFeatures: src/ORBextractor.cc:433-585                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_extract(orc_extractor *e, const uint8_t *gray, int pitch, orc_keypoint *kpOut,
                uint8_t *descOut, int *perLevel)
{
    /* ComputePyramid :607-623 */
    for (int y = 0; y < e->H; y++) memcpy(e->img[0] + (size_t)y * e->W, gray + (size_t)y * pitch, (size_t)e->W);
    orc_gauss5(e->img[0], e->lw[0], e->lh[0], e->lw[0], e->blur[0], e->lw[0]);
    for (int level = 1; level < e->nlevels; ++level) {
        orc_resize_bilinear(e->img[level - 1], e->lw[level - 1], e->lh[level - 1], e->lw[level - 1],
                            e->img[level], e->lw[level], e->lh[level], e->lw[level]);
        orc_gauss5(e->img[level], e->lw[level], e->lh[level], e->lw[level], e->blur[level], e->lw[level]);
    }

    /* ComputeKeyPointsOctTree :433-541 */
    const double featureThreshold = 0.25 * e->nFastFeatures;
    int total = 0;
    const int nFast = e->nFastFeatures;
    for (int level = 0; level < e->nlevels; ++level) {
        const int width = e->lw[level], height = e->lh[level];
        int16_t *kpLoc = e->candXY[level];
        int *response = e->candResp[level];
        int preH = 0, preL = 0;
        unsigned fastKpCountHigh =
            (unsigned)orc_fast_detect(e->img[level], width, height, width, e->iniThFAST, nFast, kpLoc, response, &preH);
        unsigned fastKpCountTotal = fastKpCountHigh;
        const unsigned diff = (unsigned)nFast - fastKpCountHigh;
        if ((double)diff > featureThreshold) { /* :463-465 */
            int16_t *tmpXY = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(nFast > 0 ? nFast : 1));
            int *tmpR = (int *)malloc(sizeof(int) * (size_t)(nFast > 0 ? nFast : 1));
            unsigned fastKpCountLow =
                (unsigned)orc_fast_detect(e->img[level], width, height, width, e->minThFAST, nFast, tmpXY, tmpR, &preL);
            if (fastKpCountLow > 0) {
                const unsigned tmp = fastKpCountHigh + fastKpCountLow;
                if (tmp > (unsigned)nFast) fastKpCountLow -= tmp - (unsigned)nFast; /* :470-473 */
                fastKpCountTotal += fastKpCountLow;
                memcpy(kpLoc + 2 * fastKpCountHigh, tmpXY, sizeof(int16_t) * 2 * fastKpCountLow);
                memcpy(response + fastKpCountHigh, tmpR, sizeof(int) * fastKpCountLow);
            }
            free(tmpXY);
            free(tmpR);
        }
        e->candN[level] = (int)fastKpCountTotal;
        e->candHigh[level] = (int)fastKpCountHigh;
        e->preHigh[level] = preH;
        e->preLow[level] = preL;

        int nl = 0;
        if (fastKpCountTotal > 0) {
            const int cap = node_cap_level(e, level) + 8;
            int *sel = (int *)malloc(sizeof(int) * (size_t)cap);
            nl = orc_distribute((int)fastKpCountTotal, kpLoc, response, width, height,
                                e->mnFeaturesPerLevel[level], sel, cap);
            if (nl < 0) nl = 0;
            /* :505-537 keypoint fill + orientation on the UNBLURRED level */
            const int scaledPatchSize = (int)(PATCH_SIZE * e->mvInvScaleFactor[level]);
            for (int i = 0; i < nl; i++) {
                orc_keypoint *kp = &kpOut[total + i];
                int id = sel[i];
                kp->x = (float)kpLoc[2 * id];
                kp->y = (float)kpLoc[2 * id + 1];
                kp->response = response[id];
                kp->size = (float)scaledPatchSize;
                kp->octave = level;
                kp->angle = orc_ic_angle(e->img[level], width, height, width, kpLoc[2 * id], kpLoc[2 * id + 1]);
            }
            free(sel);
        }
        if (perLevel) perLevel[level] = nl;
        total += nl;
    }
    if (total == 0) return 0; /* nullopt :494-496 */

    /* descriptors on the BLURRED level :566-581 */
    for (int i = 0; i < total; i++) {
        const orc_keypoint *kp = &kpOut[i];
        const int level = kp->octave;
        orc_brief(e->blur[level], e->lw[level], e->lh[level], e->lw[level], (int)(short)kp->x,
                  (int)(short)kp->y, kp->angle, descOut + (size_t)i * 32);
    }
    return total;
}
