/*
 * orb_oracle.h -- CPU ORACLE for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a single-thread, plain-C restatement of the algorithm in the reference fork
 * (geoeo/ORB_SLAM3_V1.0: src/ORBextractor.cc, src/cuda/{Fast,Angle,Orb}_gpu.cu,
 * src/ORBmatcher.cc, src/Frame.cc).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (orb_slam3_v1.0_amd/csrc) never does.
 *
 * PARITY PINNING STATUS
 *   - The reference has no CPU extractor, no tests and no golden vectors and
 *     cannot be compiled here (needs nvcc + OpenCV-CUDA 4.9 + Eigen/Sophus).  The oracle is
 *     pinned by the data pins that do exist in the reference text: FAST table == 9-contiguous
 *     predicate (SHA-256 of c_table), rBRIEF pattern SHA-256, umax table, matcher constants.
 *   - Pyramid resize / Gaussian arithmetic and last-bit atan2f/cosf/sinf live in third-party
 *     OpenCV-CUDA / CUDA libm => "parity unpinned" at those boundaries; SPEC DECISIONS S1/S5
 *      define them here.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to ORB_SLAM3::KeyPoint (include/KeyPoint.h:7-12), 24 bytes */
typedef struct orc_keypoint {
    float x, y;
    int response;
    float size;
    int octave;
    float angle;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:82-149) */
orc_extractor *orc_create(int nFeatures, int nFastFeatures, float scaleFactor, int nLevels,
                          int iniThFAST, int minThFAST, int imageWidth, int imageHeight);
void orc_destroy(orc_extractor *e);

/* scale tables / per-level feature budget / umax / level sizes.  Any pointer may be NULL. */
int orc_get_tables(const orc_extractor *e, float *scaleFactors, float *invScaleFactors,
                   float *levelSigma2, float *invLevelSigma2, int *featuresPerLevel, int *umax16,
                   int *levelW, int *levelH);
int orc_max_keypoints(const orc_extractor *e);

/* ORBextractor::extractFeatures (src/ORBextractor.cc:543-585).  Returns number of keypoints
 * (0 == the reference's nullopt).  perLevel may be NULL. */
int orc_extract(orc_extractor *e, const uint8_t *gray, int pitch, orc_keypoint *kpOut,
                uint8_t *descOut, int *perLevel);

/* state left behind by the last orc_extract: pyramid level pixels (tight pitch == level width) */
const uint8_t *orc_level_image(const orc_extractor *e, int level, int blurred);
/* combined candidate list handed to DistributeOctTree for `level` (high pass then low pass) */
int orc_level_candidates(const orc_extractor *e, int level, int16_t *xy /*2 per cand*/, int *resp,
                         int cap, int *nHigh, int *preHigh, int *preLow);

/* ---- stand-alone stages (unit tests) ---- */
void orc_resize_bilinear(const uint8_t *src, int sw, int sh, int spitch, uint8_t *dst, int dw,
                         int dh, int dpitch);
void orc_gauss5(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch);
/* one GpuFast::detect call (src/cuda/Fast_gpu.cu:354-395) with S2/S2b: returns post-NMS count */
int orc_fast_detect(const uint8_t *img, int w, int h, int pitch, int threshold, int maxKeypoints,
                    int16_t *xy, int *resp, int *preNmsCount);
/* corner score of a single pixel (0 if not a corner at `threshold`) */
int orc_fast_score(const uint8_t *img, int pitch, int x, int y, int threshold);
/* 16-bit circular mask has >=9 contiguous ones */
int orc_fast_arc9(int mask);
/* DistributeOctTree (src/ORBextractor.cc:226-431) + best-per-node (:505-527).
 * selIdx[k] = index into the candidate list of the point kept for the k-th node of the final
 * list (head -> tail).  Returns the node count. */
int orc_distribute(int n, const int16_t *xy, const int *resp, int W, int H, int maxFeatures,
                   int *selIdx, int selCap);
float orc_ic_angle(const uint8_t *img, int w, int h, int pitch, int x, int y);
void orc_brief(const uint8_t *img, int w, int h, int pitch, int x, int y, float angleDeg,
               uint8_t desc[32]);
float orc_spec_atan2f(float y, float x); /* radians, S5 */
float orc_atan2_deg(float m01, float m10);
void orc_cos_sin_deg(float angleDeg, float *c, float *s);

/* ---- matcher (src/ORBmatcher.cc) ---- */
int orc_hamming(const uint8_t *a, const uint8_t *b);

typedef struct orc_frame_view {
    int n;                  /* keypoints in the frame */
    const orc_keypoint *kp; /* mvKeysUn */
    const uint8_t *desc;    /* n x 32 */
    int gridCols, gridRows; /* mFrameGridCols / Rows */
    float minX, minY;       /* mnMinX / mnMinY */
    float gridInvW, gridInvH; /* mfGridElementWidthInv / HeightInv */
    int nLevels;
    const float *scaleFactors; /* mvScaleFactors */
} orc_frame_view;

typedef struct orc_map_point {
    float projX, projY; /* mTrackProjX / mTrackProjY */
    float viewCos;      /* mTrackViewCos */
    float trackDepth;   /* mTrackDepth */
    int level;          /* mnTrackScaleLevel */
    int inView;         /* mbTrackInView */
    int bad;            /* isBad() */
    int observations;   /* Observations() */
} orc_map_point;

/* ORBmatcher::SearchByProjection(Frame, MapPoints, ...) (src/ORBmatcher.cc:31-123), mono case.
 * initObs[i]  : -1 if keypoint i holds no map point on entry, else Observations() of it.
 * matchOut[i] : index of the map point written into F->mvpMapPoints[i] by this call, else -1. */
int orc_search_by_projection(const orc_frame_view *F, int M, const orc_map_point *mps,
                             const uint8_t *mpDesc, const int *initObs, float th, int bFarPoints,
                             float thFarPoints, float nnRatio, int *matchOut);

/* ORBmatcher::SearchByBoW (src/ORBmatcher.cc:133-327), mono case.  The two FeatureVectors are
 * given as the already merge-walked list of shared vocabulary nodes in ascending NodeId order:
 * group g holds KF feature indices kfIdx[kfOff[g]..kfOff[g+1]) and frame feature indices
 * fIdx[fOff[g]..fOff[g+1]).  kfHasMP[i] != 0 iff KF feature i has a non-bad map point.
 * matchOut[j] = KF feature index whose map point is assigned to frame feature j, or -1. */
int orc_search_by_bow_rig(int G, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx,
                          int nKF, const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfHasMP,
                          int nF, const uint8_t *fDesc, const float *fAngle, int nLeft, float nnRatio,
                          int checkOrientation, int *matchOut);
int orc_search_by_bow(int G, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx,
                      int nKF, const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfHasMP,
                      int nF, const uint8_t *fDesc, const float *fAngle, float nnRatio,
                      int checkOrientation, int *matchOut);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:329-439), mono.  Only level-0 keypoints of
 * F1 take part; candidates come from F2's grid.  matches12Out[i1] (F1->n ints) = index in F2 or -1. */
int orc_search_for_initialization(const orc_frame_view *F1, const orc_frame_view *F2, int windowSize,
                                  float nnRatio, int checkOrientation, int *matches12Out);

/* TemplatedVocabulary::transform(feature, ...) (Thirdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1227-1270)
 * for n descriptors; tree as CSR children lists, node 0 = root, leaf = node without children. */
void orc_vocab_transform(int nNodes, const int *childOff, const int *childIdx, const uint8_t *nodeDesc,
                         const int *wordId, const double *weight, int L, const uint8_t *desc, int n,
                         int levelsup, int *wordOut, int *nodeOut, double *weightOut);

/* Frame::isInFrustum for a batch of map points (src/Frame.cc:272-331), SPEC DECISION S8 */
typedef struct {
    float rcw[9], tcw[3], twc[3];
    float minX, maxX, minY, maxY;
    float fx, fy, cx, cy;
    float k1, k2, k3, k4; /* KannalaBrandt8 */
    float mbf, logScaleFactor;
    int nLevels, cameraModel;
} orc_frustum;
typedef struct {
    float x, y, z, minDistance, maxDistance;
    int bad, observations, skip;
} orc_world_point;
float orc_spec_logf(float x);
void orc_is_in_frustum(const orc_frustum *F, int n, const orc_world_point *pts, orc_map_point *out, float *projXR);

/* the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:678-836) */
void orc_fuse_search(const orc_frame_view *KF, const float *invLevelSigma2, const float *uRight, const orc_frustum *F,
                     float th, int M, const orc_world_point *pts, const uint8_t *mpDesc, int *bestIdxOut,
                     int *bestDistOut);

/* the same with bRight = true (:684-688, :820) */
void orc_fuse_search_right(const orc_frame_view *KFleft, int nRight, const float *invLevelSigma2, const float *uRight,
                           const orc_frustum *F, float th, int M, const orc_world_point *pts, const uint8_t *mpDesc,
                           int *bestIdxOut, int *bestDistOut);

/* the search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:864-975) */
void orc_fuse_search_sim3(const orc_frame_view *KF, const orc_frustum *F, float th, int M, const orc_world_point *pts,
                          const uint8_t *mpDesc, int *bestIdxOut, int *bestDistOut);

/* one search direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:977-1200) */
typedef struct {
    float rcw[9], tcw[3]; /* pose of the key frame that owns the map points */
    float sr[9], t[3];    /* similarity into the other key frame: s * R and t */
    float fx, fy, cx, cy; /* pKF1's intrinsics, both directions (:979-982) */
    float minX, maxX, minY, maxY; /* image bounds of the target key frame */
    float logScaleFactor;
    int nLevels;          /* of the target key frame */
} orc_sim3_dir;
int orc_search_by_sim3(const orc_frame_view *KF1, const orc_frame_view *KF2, const orc_sim3_dir *d12,
                       const orc_sim3_dir *d21, const orc_world_point *mp1, const uint8_t *mpDesc1,
                       const orc_world_point *mp2, const uint8_t *mpDesc2, float th, int *match12Out);

/* ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, checkOrientation) (src/ORBmatcher.cc:1202-1326) */
int orc_search_by_projection_kf(const orc_frame_view *F, const orc_frustum *Fr, int M, const orc_world_point *pts,
                                const uint8_t *mpDesc, const float *kfAngle, const uint8_t *frameHasMP, float th,
                                int checkOrientation, int *matchOut);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:441-676), pinhole, one camera per key frame */
int orc_search_for_triangulation(int G, const int *off1, const int *idx1v, const int *off2, const int *idx2v, int n1,
                                 const orc_keypoint *kp1, const uint8_t *desc1, const uint8_t *hasMP1,
                                 const uint8_t *stereo1, int n2, const orc_keypoint *kp2, const uint8_t *desc2,
                                 const uint8_t *hasMP2, const uint8_t *stereo2, const float *scaleFactors2,
                                 const float *F12, float epx, float epy, int bOnlyStereo, int bCoarse,
                                 int checkOrientation, int *matches12Out);

/* the same with the camera models of the two key frames: KannalaBrandt8 for pKF1 selects
 * KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220,306-370), SPEC DECISION S10 */
typedef struct {
    int model1, model2;     /* 0 pinhole, 1 KannalaBrandt8 */
    float cam1[8], cam2[8]; /* fx fy cx cy k1 k2 k3 k4 */
    float precision;        /* KannalaBrandt8::precision */
    float R12[9], t12[3];   /* T12 = T1w * Tw2 (src/ORBmatcher.cc:466-468) */
    float sigma2_1[32]; /* pKF1->mvLevelSigma2 (ORC_MAX_LEVELS entries) */
    int kf1HasCamera2;      /* pKF1->mpCamera2: no epipole gate (:551) */
} orc_tri_cameras;
int orc_kb8_epipolar_constrain(const orc_tri_cameras *C, float u1, float v1, float u2, float v2, float sigmaLevel,
                               float unc, float xyzOut[3]);
void orc_kb8_unproject(const float cam[8], int model, float precision, float u, float v, float *rx, float *ry);
int orc_search_for_triangulation_cam(int G, const int *off1, const int *idx1v, const int *off2, const int *idx2v, int n1,
                                     const orc_keypoint *kp1, const uint8_t *desc1, const uint8_t *hasMP1,
                                     const uint8_t *stereo1, int n2, const orc_keypoint *kp2, const uint8_t *desc2,
                                     const uint8_t *hasMP2, const uint8_t *stereo2, const float *scaleFactors2,
                                     const float *F12, float epx, float epy, int bOnlyStereo, int bCoarse,
                                     int checkOrientation, const orc_tri_cameras *cams, int *matches12Out);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:343-416) for a batch of descriptor sets */
void orc_distinctive_descriptors(int nSets, const int *setOff, const uint8_t *desc, int *bestIdxOut,
                                 int *bestMedianOut);

/* node-side image preparation (image_grabber.hpp:96-110), SPEC DECISION S9 -- see prep_oracle.c */
void orc_prep_remap_pixel(const uint8_t *bgr, int pitch, int srcW, int srcH, float x, float y, uint8_t out[3]);
float orc_prep_scale(int srcN, int dstN);
void orc_prepare_image(const uint8_t *bgr, int pitch, int srcW, int srcH, const float *map1, const float *map2, int dstW,
                       int dstH, uint8_t *grey, int greyPitch, uint8_t *und);

/* Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:157-176,470-480): linear cell per kp or -1 */
void orc_assign_grid(const orc_frame_view *F, int *cellOut);

#ifdef __cplusplus
}
#endif
#endif
