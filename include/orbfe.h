/*
 * orbfe.h -- C ABI of the MI355X-native ORB front-end (liborbfe.so).
 *
 * This is the drop-in boundary for ONE path of geoeo/ORB_SLAM3_V1.0: the per-frame ORB
 * extractor + Hamming matchers.  Each entry point names the reference interface it replaces
 * (paths relative to the reference repo).  Plain pointers and sizes only; no C++/torch types.
 * The reference-signature C++ wrapper lives in include/orbfe_adaptor.hpp; the binding a
 * maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Error behaviour: every function returns an orbfe_status (0 == ORBFE_OK); nothing in the
 * library calls exit()/abort() (the reference does: include/cuda/HelperCuda.h:44-50).
 * A frame with zero keypoints is NOT an error: n == 0 (the reference returns std::nullopt,
 * src/ORBextractor.cc:494-496); the adaptor maps it back to nullopt.
 *
 * Threading: one handle == one HIP stream == one caller at a time for the extract calls (same
 * as the reference).  The matcher calls take their own scratch from the
 * handle and are serialised per handle; use one handle per calling thread.
 *
 * Streams: the *_device entry points are asynchronous on the caller's stream but work in scratch the
 * handle owns (pyramid, candidate lists; the matcher arena).  The library orders consecutive users of
 * that scratch itself: it records an event behind every enqueue and makes the next enqueue on a
 * DIFFERENT stream wait for it, so mixing streams (or a *_device call followed by a host-pointer
 * call) is safe; it does not make two calls run concurrently.
 *
 * Input contract of the image entry points: rows `pitch` bytes apart, pitch < 2^24 and
 * pitch * (height - 1) + ((width + 3) & ~3) < 0x7ffffff0 (frames are addressed with 32-bit byte
 * offsets; larger ones are refused with ORBFE_ERR_INVALID_ARG); when the
 * base address, pitch and frame stride are multiples of 4 the kernels read whole dwords, i.e. up
 * to the 4-byte-rounded end of every row -- the buffer must extend to
 * pitch * (height - 1) + ((width + 3) & ~3) bytes per frame (any pitch >= that rounded width, or a
 * tight width that is a multiple of 4, satisfies it).
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_MAX_LEVELS 32
#define ORBFE_DESC_BYTES 32

typedef enum orbfe_status {
    ORBFE_OK = 0,
    ORBFE_ERR_INVALID_ARG = 1,   /* NULL pointer, size out of range, batch > max_batch ... */
    ORBFE_ERR_UNSUPPORTED = 2,   /* configuration outside what the kernels were built for */
    ORBFE_ERR_NO_DEVICE = 3,     /* no gfx950 device / HIP runtime failure at create */
    ORBFE_ERR_HIP = 4,           /* a HIP call failed; see orbfe_last_error() */
    ORBFE_ERR_OUT_OF_MEMORY = 5,
    ORBFE_ERR_INTERNAL = 6,      /* device-side guard tripped (capacity / round limit) */
    ORBFE_ERR_BUSY = 7           /* orbfe_stream_submit: every slot of the ring is in flight -- collect first */
} orbfe_status;

/* Layout-identical to ORB_SLAM3::KeyPoint (include/KeyPoint.h:7-12): 24 bytes, memcpy-able. */
typedef struct orbfe_keypoint {
    float x, y;     /* LEVEL pixel coordinates (the reference never rescales, S6) */
    int response;   /* FAST corner score (src/cuda/Fast_gpu.cu:193-216) */
    float size;     /* (int)(31 * invScaleFactor[octave]) (src/ORBextractor.cc:511,531) */
    int octave;
    float angle;    /* degrees in [0, 360] */
} orbfe_keypoint;

/* The 8 constructor arguments of ORBextractor::ORBextractor (include/ORBextractor.h:55-56,
 * src/ORBextractor.cc:82-149) plus placement / batching knobs that have no reference analogue. */
typedef struct orbfe_params {
    int n_features;        /* nFeatures      */
    int n_fast_features;   /* nFastFeatures  (per-level FAST candidate cap) */
    float scale_factor;    /* scaleFactor    */
    int n_levels;          /* nlevels        */
    int ini_th_fast;       /* iniThFAST      */
    int min_th_fast;       /* minThFAST      */
    int image_width;       /* imageWidth     */
    int image_height;      /* imageHeight    */
    int device_id;         /* HIP device ordinal (0 for one-process-per-GPU) */
    int max_batch;         /* frames per orbfe_extract_batch* call (>= 1); sizes the workspace */
} orbfe_params;

typedef struct orbfe_handle orbfe_handle;

/* -------------------------------------------------------------------------------------------
 * Extractor
 * ---------------------------------------------------------------------------------------- */

/* replaces ORBextractor::ORBextractor (src/ORBextractor.cc:82-149): scale tables, per-level
 * feature budget, pyramid + scratch allocation (AllocatePyramid :587-605, GpuFast::GpuFast
 * src/cuda/Fast_gpu.cu:321-332).  All device memory is allocated here, none per frame. */
int orbfe_create(const orbfe_params *params, orbfe_handle **out);
void orbfe_destroy(orbfe_handle *h);

/* replaces GetLevels / GetScaleFactor (include/ORBextractor.h:64-72) */
int orbfe_get_levels(const orbfe_handle *h);
float orbfe_get_scale_factor(const orbfe_handle *h);
/* replaces GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:74-92).  Each array holds n_levels floats;
 * any pointer may be NULL. */
int orbfe_get_scale_tables(const orbfe_handle *h, float *scale_factors, float *inv_scale_factors,
                           float *level_sigma2, float *inv_level_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:112-124) and pyramid level sizes (:594-604) */
int orbfe_get_level_info(const orbfe_handle *h, int *features_per_level, int *level_width,
                         int *level_height);
/* upper bound on keypoints per frame: sum over levels of max(N_level + 3, 4 * nIni).  Output
 * arrays of the extract calls are sized with this stride. */
int orbfe_max_keypoints(const orbfe_handle *h);

/* replaces ORBextractor::extractFeatures(const cv::cuda::HostMem&) (include/ORBextractor.h:62,
 * src/ORBextractor.cc:543-585).  `gray` is a host pointer to an 8-bit image of the size given at
 * create, `pitch` bytes per row.  kp_out / desc_out hold orbfe_max_keypoints() entries; *n_out
 * receives the count (0 == the reference's nullopt); per_level_counts (n_levels ints) may be
 * NULL.  One H2D copy, one D2H copy, one host sync. */
int orbfe_extract(orbfe_handle *h, const uint8_t *gray, int pitch, orbfe_keypoint *kp_out,
                  uint8_t *desc_out, int *n_out, int *per_level_counts);

/* Batched many-frame mode (BASELINE config 4; no reference analogue).  Host pointers; frame b's
 * results start at kp_out + b*cap, desc_out + b*cap*32, per_level_counts + b*n_levels with
 * cap = orbfe_max_keypoints(). */
int orbfe_extract_batch(orbfe_handle *h, const uint8_t *const *grays, int pitch, int batch,
                        orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out,
                        int *per_level_counts);

/* Same, with everything already resident in HBM: d_gray points at `batch` frames
 * `frame_stride` bytes apart in device memory; all outputs are device pointers with the strides
 * above.  Asynchronous on `stream` (a hipStream_t; NULL == the handle's own stream); no host
 * sync is performed.  This is the entry bench.py times. */
int orbfe_extract_batch_device(orbfe_handle *h, const uint8_t *d_gray, size_t frame_stride,
                               int pitch, int batch, orbfe_keypoint *d_kp_out,
                               uint8_t *d_desc_out, int *d_n_out, int *d_per_level_counts,
                               void *stream);

/* Device-side guard flags of the LAST extract call (any entry point, any stream): waits for that
 * call, ORs the per-(frame, level) status words (candidate-array / node-table overflow, round
 * limit) into *flags_out (may be NULL) and returns ORBFE_ERR_INTERNAL when any is set.  The
 * host-pointer entry points check this themselves; callers of orbfe_extract_batch_device do it
 * here, outside their timed region. */
int orbfe_get_device_status(orbfe_handle *h, unsigned *flags_out);

/* -------------------------------------------------------------------------------------------
 * Pipelined host-pointer extraction (the reference's call shape -- a host image per frame,
 * src/Frame.cc:178-189 -- at stream rate): a ring of `slots` (>= 2) pinned + device buffers of
 * `slot_frames` (<= max_batch) frames each.  orbfe_stream_submit() enqueues upload, kernels and
 * result download of one slot on three HIP streams and returns without waiting;
 * orbfe_stream_collect() waits for the OLDEST submitted slot only and hands its results over.
 * Upload of slot i+1, kernels of slot i and download of slot i-1 overlap.  Pinned sources with a
 * 4-byte-aligned pitch are copied by the DMA engine straight from the caller's buffer; pageable
 * ones are re-pitched into the slot's pinned block by a small thread pool first.
 * Results are byte-identical to orbfe_extract_batch / orbfe_extract_batch_device.
 * Source lifetime: a pinned source may still be read by the DMA engine AFTER submit has returned
 * (pageable sources have been copied by then, but the caller cannot rely on which path ran):
 * every source frame must stay valid and UNCHANGED until the submission that carried it has been
 * collected -- a capture buffer recycled earlier gives torn frames and no error.
 * Threading: ONE producer thread may submit while ONE consumer thread collects (the reference feeds frames from a grabber
 * thread into the tracking thread, ros2_ws/src/mono-inertial/src/mono_inertial_node.cpp:207-210): the ring counters are
 * atomic, a slot belongs to its submission until that has been collected, and the copy pool is serialised between the
 * two.  Two threads submitting (or two collecting) at the same time, and destroy / orbfe_stream_enable_track while
 * another call is running, are not supported.
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_stream orbfe_stream;
int orbfe_stream_create(orbfe_handle *h, int slots, int slot_frames, orbfe_stream **out);
void orbfe_stream_destroy(orbfe_stream *s);
/* enqueue `n` (1..slot_frames) frames; ORBFE_ERR_BUSY when `slots` submissions are uncollected */
int orbfe_stream_submit(orbfe_stream *s, const uint8_t *const *grays, int pitch, int n);
/* wait for the oldest submission; outputs as orbfe_extract_batch (strides cap / cap*32 / n_levels);
 * *n_frames receives how many frames that submission held.  ORBFE_ERR_INVALID_ARG when nothing
 * is in flight. */
int orbfe_stream_collect(orbfe_stream *s, orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out,
                         int *per_level_counts, int *n_frames);
/* the same without the copy: pointers into the slot's pinned result block (frame b at the strides
 * above), valid until the next orbfe_stream_submit() that reuses the slot (`slots` submissions
 * later) */
int orbfe_stream_collect_view(orbfe_stream *s, const orbfe_keypoint **kp, const uint8_t **desc,
                              const int **n, const int **per_level_counts, int *n_frames);
int orbfe_stream_in_flight(const orbfe_stream *s);

/* replaces the public members mvImagePyramid / mvBlurredImagePyramid
 * (include/ORBextractor.h:94-95): copies level `level` of frame `frame` of the LAST extract call
 * to host memory (out_pitch bytes per row).  Synchronises. */
int orbfe_get_pyramid_level(orbfe_handle *h, int frame, int level, int blurred, uint8_t *out,
                            int out_pitch);

/* Diagnostics for parity tests: FAST candidates of (frame, level) of the last extract call as
 * packed words (score << 24 | y << 12 | x), unordered; counters[4] = {survivors(low pass),
 * survivors with score >= iniThFAST, pre-NMS corners >= minThFAST, pre-NMS corners >= iniThFAST}.
 * Returns the number of words written (<= cap) through *n_out. */
int orbfe_debug_get_candidates(orbfe_handle *h, int frame, int level, uint32_t *packed, int cap,
                               int *n_out, int counters[4]);

/* Stage timing (HIP events on the launch stream).  While enabled, every extract call records
 * one event per stage boundary (a pool of 128 event sets; enabling resets the pool).
 * orbfe_get_stage_ms() synchronises on the recorded events and returns, per stage in the order of
 * orbfe_stage_name(), the SUM of elapsed milliseconds over the recorded calls; *n_calls receives
 * how many calls were summed (<= 128).  The last stage is the whole chain ("total"). */
#define ORBFE_NUM_STAGES 5
int orbfe_set_stage_timing(orbfe_handle *h, int enabled);
int orbfe_get_stage_ms(orbfe_handle *h, float ms[ORBFE_NUM_STAGES], int *n_calls);
const char *orbfe_stage_name(int stage);

/* -------------------------------------------------------------------------------------------
 * Matcher
 * ---------------------------------------------------------------------------------------- */

#define ORBFE_TH_LOW 30        /* ORBmatcher::TH_LOW  (include/ORBmatcher.h:73) */
#define ORBFE_TH_HIGH 100      /* ORBmatcher::TH_HIGH (:74) */
#define ORBFE_HISTO_LENGTH 30  /* ORBmatcher::HISTO_LENGTH (:75) */

/* replaces ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1375-1391); host, no GPU. */
int orbfe_hamming(const uint8_t *a, const uint8_t *b);

/* What SearchByProjection reads from a Frame (include/Frame.h): keypoints, descriptors, the
 * static grid geometry (src/Frame.cc:101-105) and mvScaleFactors. */
typedef struct orbfe_frame_view {
    int n;                        /* mNumKeypoints */
    const orbfe_keypoint *kp;     /* mvKeysUn */
    const uint8_t *desc;          /* mDescriptors, n x 32 */
    int grid_cols, grid_rows;     /* mFrameGridCols / mFrameGridRows */
    float min_x, min_y;           /* mnMinX / mnMinY */
    float grid_inv_w, grid_inv_h; /* mfGridElementWidthInv / mfGridElementHeightInv */
    int n_levels;
    const float *scale_factors;   /* mvScaleFactors */
} orbfe_frame_view;

/* The MapPoint fields SearchByProjection reads (src/ORBmatcher.cc:36-59). */
typedef struct orbfe_map_point {
    float proj_x, proj_y;  /* mTrackProjX / mTrackProjY */
    float view_cos;        /* mTrackViewCos */
    float track_depth;     /* mTrackDepth */
    int level;             /* mnTrackScaleLevel */
    int in_view;           /* mbTrackInView */
    int bad;               /* isBad() */
    int observations;      /* Observations() */
} orbfe_map_point;

/* replaces ORBmatcher::SearchByProjection(Frame, vector<MapPoint>, th, bFarPoints, thFarPoints,
 * nnRatio, checkOrientation) (src/ORBmatcher.cc:31-123; callers src/Tracking.cc:1115), mono.
 * All pointers are HOST pointers.  init_obs[i] = -1 if F->mvpMapPoints[i] is empty on entry, else
 * the Observations() of the point it holds (may be NULL == all empty).  match_out[i] = index of
 * the map point this call writes into F->mvpMapPoints[i], or -1.  *n_matches = return value of
 * the reference function.  The greedy, order-dependent claim semantics are reproduced exactly. */
int orbfe_match_projection(orbfe_handle *h, const orbfe_frame_view *frame, int n_map_points,
                           const orbfe_map_point *map_points, const uint8_t *mp_desc,
                           const int *init_obs, float th, int far_points, float th_far_points,
                           float nn_ratio, int *match_out, int *n_matches);

/* Batched, HBM-resident form of the same function (no reference analogue; BASELINE configs 2/3):
 * frame b's keypoints / descriptors / counts are the outputs of orbfe_extract_batch_device
 * (stride kp_stride == orbfe_max_keypoints()), its map points are d_map_points[b*n_map_points ..],
 * scale factors are the handle's own mvScaleFactor.  All d_* pointers are device pointers;
 * d_init_obs may be NULL.  Fully asynchronous on `stream` (a hipStream_t; NULL == the handle's own
 * stream): three kernels are queued and no host synchronisation takes place; results are complete when
 * the stream reaches that point. */
int orbfe_match_projection_batch_device(orbfe_handle *h, int batch, const orbfe_keypoint *d_kp,
                                        const uint8_t *d_desc, const int *d_n, int kp_stride,
                                        int grid_cols, int grid_rows, float min_x, float min_y,
                                        float grid_inv_w, float grid_inv_h, int n_map_points,
                                        const orbfe_map_point *d_map_points, const uint8_t *d_mp_desc,
                                        const int *d_init_obs, float th, int far_points,
                                        float th_far_points, float nn_ratio, int *d_match_out,
                                        int *d_n_matches, void *stream);

/* replaces ORBmatcher::SearchByBoW(KeyFrame, Frame, matches, nnRatio, checkOrientation)
 * (src/ORBmatcher.cc:133-327; caller src/Tracking.cc:835), mono.  The merge-walk over the two
 * DBoW2::FeatureVector maps (:161-163,289-301) is done by the caller/adaptor and handed over as
 * CSR groups in ascending NodeId order: group g = KF features kf_idx[kf_off[g]..kf_off[g+1]) and
 * frame features f_idx[f_off[g]..f_off[g+1]).  kf_has_mp[i] != 0 iff KF feature i has a non-bad
 * map point.  match_out[j] (n_f ints) = KF feature whose map point goes to frame feature j, or
 * -1; *n_matches = the reference's return value (after the rotation-histogram filter). */
int orbfe_match_bow(orbfe_handle *h, int n_groups, const int *kf_off, const int *kf_idx,
                    const int *f_off, const int *f_idx, int n_kf, const uint8_t *kf_desc,
                    const float *kf_angle, const uint8_t *kf_has_mp, int n_f,
                    const uint8_t *f_desc, const float *f_angle, float nn_ratio,
                    int check_orientation, int *match_out, int *n_matches);

/* the same for a frame of a two-camera rig (F->Nleft != -1, src/ORBmatcher.cc:205-233,263-286): frame features
 * >= n_left belong to the right camera and keep their own best / second best; the right best is accepted whenever the
 * LEFT best passes TH_LOW and its own distance does too (the reference's ratio test there reads "|| true").
 * n_left == -1 is orbfe_match_bow. */
int orbfe_match_bow_rig(orbfe_handle *h, int n_groups, const int *kf_off, const int *kf_idx,
                    const int *f_off, const int *f_idx, int n_kf, const uint8_t *kf_desc,
                    const float *kf_angle, const uint8_t *kf_has_mp, int n_f,
                    const uint8_t *f_desc, const float *f_angle, int n_left, float nn_ratio,
                    int check_orientation, int *match_out, int *n_matches);

/* replaces ORBmatcher::SearchForInitialization(F1, F2, windowSize, nnRatio, checkOrientation)
 * (src/ORBmatcher.cc:329-439; caller src/Tracking.cc:605-607), mono.  HOST pointers.  Frame 1
 * contributes keypoints + descriptors; frame 2 additionally its grid (GetFeaturesInArea runs on
 * frame 2).  Only level-0 keypoints of frame 1 take part (:346-347).  matches12_out (f1->n ints)
 * receives vnMatches12 (index in frame 2 or -1), *n_matches the function's return value. */
int orbfe_match_initialization(orbfe_handle *h, const orbfe_frame_view *f1, const orbfe_frame_view *f2,
                               int window_size, float nn_ratio, int check_orientation,
                               int *matches12_out, int *n_matches);

/* -------------------------------------------------------------------------------------------
 * Map-point projection: Frame::isInFrustum for a batch
 * ---------------------------------------------------------------------------------------- */
#define ORBFE_CAMERA_PINHOLE 0
#define ORBFE_CAMERA_KANNALA_BRANDT8 1

/* what Frame::isInFrustum reads from the frame (src/Frame.cc:272-331) */
typedef struct orbfe_frustum {
    float rcw[9];             /* GetRcw(), row-major */
    float tcw[3];             /* GetTcw() */
    float twc[3];             /* GetTwc() (camera centre) */
    float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    float fx, fy, cx, cy;     /* mvParameters[0..3] (src/CameraModels/Pinhole.cpp:41-47, KannalaBrandt8.cpp:66-83) */
    float k1, k2, k3, k4;     /* KannalaBrandt8 mvParameters[4..7]; ignored by the pinhole model */
    float mbf;                /* stereo baseline * fx (mTrackProjXR) */
    float log_scale_factor;   /* mfLogScaleFactor (src/Frame.cc:75) */
    int n_levels;             /* mnScaleLevels */
    int camera_model;         /* ORBFE_CAMERA_PINHOLE or ORBFE_CAMERA_KANNALA_BRANDT8 */
} orbfe_frustum;

/* what it reads from a MapPoint, plus the two skip conditions of Tracking::SearchLocalPoints
 * (src/Tracking.cc:1066-1069) */
typedef struct orbfe_world_point {
    float x, y, z;            /* GetWorldPos() */
    float min_distance;       /* mfMinDistance (GetMinDistanceInvariance() == 0.9f * this) */
    float max_distance;       /* mfMaxDistance (GetMaxDistanceInvariance() == 1.1f * this) */
    int bad;                  /* isBad() */
    int observations;         /* Observations() (passed through to the matcher record) */
    int skip;                 /* mnLastFrameSeen == current frame id */
} orbfe_world_point;

/* replaces the isInFrustum loop of Tracking::SearchLocalPoints (src/Tracking.cc:1059-1077): out[i] holds
 * the fields the reference writes into the MapPoint (mTrackProjX/Y, mTrackViewCos, mTrackDepth,
 * mnTrackScaleLevel, mbTrackInView), i.e. the input records of orbfe_match_projection; proj_xr
 * (mTrackProjXR) may be NULL.  HOST pointers.  Points that fail keep in_view == 0; proj_x / proj_y are -1 unless
 * the projection landed inside the image bounds (as the reference leaves them). */
int orbfe_project_map_points(orbfe_handle *h, const orbfe_frustum *frustum, int n,
                             const orbfe_world_point *points, orbfe_map_point *out, float *proj_xr);
/* Same with DEVICE pointers, asynchronous on `stream` (NULL == the handle's stream): the output feeds
 * orbfe_match_projection_batch_device directly. */
int orbfe_project_map_points_device(orbfe_handle *h, const orbfe_frustum *frustum, int n,
                                    const orbfe_world_point *d_points, orbfe_map_point *d_out,
                                    float *d_proj_xr, void *stream);

/* -------------------------------------------------------------------------------------------
 * The tracking thread's per-frame chain as ONE submission
 * ---------------------------------------------------------------------------------------- */

/* the frame statics and call parameters SearchByProjection reads besides the map points (src/Frame.cc:101-105,
 * src/Tracking.cc:1108-1115); versioned by its size like orbfe_tri_params */
typedef struct orbfe_track_params {
    int struct_size;               /* sizeof(orbfe_track_params) at the caller's compile time */
    int grid_cols, grid_rows;      /* mFrameGridCols / mFrameGridRows */
    float min_x, min_y;            /* mnMinX / mnMinY */
    float grid_inv_w, grid_inv_h;  /* mfGridElementWidthInv / mfGridElementHeightInv */
    float th;                      /* 20 before the IMU is initialised, 40 after (src/Tracking.cc:1108-1113) */
    float nn_ratio;                /* 0.85 / 0.75 */
    int far_points;                /* bFarPoints */
    float th_far_points;           /* thFarPoints */
} orbfe_track_params;
#define ORBFE_TRACK_PARAMS_INIT {(int)sizeof(orbfe_track_params)}

/* What the tracking thread of this fork does with every frame once the IMU is initialised, in one call:
 *   Frame::Frame -> ExtractORB -> ORBextractor::extractFeatures          (src/Tracking.cc:152-173, src/Frame.cc:178-189)
 *   Tracking::SearchLocalPoints: isInFrustum for every local map point   (src/Tracking.cc:1059-1077, src/Frame.cc:272-333)
 *   ORBmatcher::SearchByProjection(mCurrentFrame, mvpLocalMapPoints, ...) (src/Tracking.cc:1108-1115)
 * The pose that isInFrustum needs does not depend on the frame's features here: TrackWithMotionModel is the IMU
 * prediction alone (src/Tracking.cc:908-923, PredictStateIMU :293), and the local map comes from the LAST frame's
 * matches (:1190), so image, predicted pose and local map points are all known when the frame arrives.
 * Equivalent to orbfe_extract + orbfe_project_map_points + orbfe_match_projection (frame view = the fresh keypoints
 * with the grid of `tp` and the extractor's own mvScaleFactors; init_obs == NULL: mvpMapPoints is empty at that point),
 * bit for bit -- but as ONE captured hipGraph: the host frame and one block [frustum | points | descriptors] go up,
 * memset + extraction kernels + frustum kernel + three matcher kernels run back to back with no host in between, ONE
 * result block comes back, one host synchronisation.
 * gray / pitch as orbfe_extract; points / mp_desc: n_points local map points (HOST pointers; mp_desc n_points x 32);
 * frustum->n_levels must not exceed the extractor's level count.
 * kp_out / desc_out / n_out / per_level_counts as orbfe_extract; mp_out (n_points records, may be NULL) and proj_xr_out
 * (may be NULL) as orbfe_project_map_points; match_out (orbfe_max_keypoints() ints; the first *n_out are meaningful)
 * and n_matches as orbfe_match_projection.  A frame without keypoints gives *n_out == 0 and *n_matches == 0.
 * Graphs are cached per (n_points rounded up to a bucket, input pitch, tp): a steady stream replays one graph. */
int orbfe_track_frame(orbfe_handle *h, const uint8_t *gray, int pitch, const orbfe_frustum *frustum,
                      const orbfe_track_params *tp, int n_points, const orbfe_world_point *points,
                      const uint8_t *mp_desc, orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out,
                      int *per_level_counts, orbfe_map_point *mp_out, float *proj_xr_out, int *match_out,
                      int *n_matches);

/* what ORBmatcher::SearchForTriangulation derives from the two key-frame poses (src/ORBmatcher.cc:448-465).
 * ABI: the block is versioned by its size.  struct_size must hold sizeof(orbfe_tri_params) of the header the CALLER was
 * compiled against (`orbfe_tri_params p = ORBFE_TRI_PARAMS_INIT;` zero-initialises the rest); the library refuses any
 * other value with ORBFE_ERR_INVALID_ARG instead of reading past a shorter block (the struct grew a camera-model tail in
 * round 2). */
typedef struct orbfe_tri_params {
    int struct_size;        /* sizeof(orbfe_tri_params) at the caller's compile time */
    float f12[9];           /* F12 = K1^-T [t12]x R12 K2^-1, row-major: the matrix Pinhole::epipolarConstrain
                             * rebuilds for every pair (src/CameraModels/Pinhole.cpp:106-109) */
    float ep_x, ep_y;       /* epipole: pKF2->mpCamera->project(T2w * Cw) (:451-454) */
    int only_stereo;        /* bOnlyStereo */
    int coarse;             /* bCoarse */
    int check_orientation;  /* checkOrientation */
    /* ---- camera models (zero-initialised == one pinhole camera per key frame, the fields above suffice) ---- */
    int camera_model1;      /* ORBFE_CAMERA_* of pKF1->mpCamera: KANNALA_BRANDT8 selects KannalaBrandt8::epipolarConstrain
                             * (src/CameraModels/KannalaBrandt8.cpp:216-220,306-370: triangulate the pair, both depths positive,
                             * both reprojection errors inside 5.991 sigma^2) instead of the F12 line test */
    int camera_model2;      /* ORBFE_CAMERA_* of pKF2->mpCamera (unproject / project of the second view) */
    float cam1[8], cam2[8]; /* mvParameters of the two cameras: fx fy cx cy k1 k2 k3 k4 */
    float kb_precision;     /* KannalaBrandt8::precision (Newton stop of unproject; the reference's default is 1e-6) */
    float r12[9], t12[3];   /* T12 = T1w * Tw2 (:466-468; in this fork also when the key frames carry a second camera:
                             * bRight1 / bRight2 are constants, :520,:549): rotation row-major, translation */
    float level_sigma2_1[ORBFE_MAX_LEVELS]; /* pKF1->mvLevelSigma2 (sigmaLevel of the first view; unc is 1.0, :603) */
    int kf1_has_camera2;    /* pKF1->mpCamera2 is set: the epipole gate (:551) is skipped */
} orbfe_tri_params;
#define ORBFE_TRI_PARAMS_INIT {(int)sizeof(orbfe_tri_params)}

/* replaces ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse,
 * checkOrientation) (src/ORBmatcher.cc:441-676; caller src/LocalMapping.cc:488), one pinhole camera per key
 * frame, or KannalaBrandt8 cameras (camera_model1 / camera_model2 of the parameter block: SPEC DECISION S10 -- the
 * reference triangulates with an Eigen JacobiSVD in binary32, which is not reproducible bit for bit; here the null vector
 * of the 4x4 system comes from a fixed cyclic-Jacobi sequence on A^T A in binary64, so the verdicts agree with the
 * reference except for pairs whose depth or reprojection error sits within rounding distance of a threshold).
 * The merge-walk over the two FeatureVectors (:489-617) is handed over as CSR groups in ascending
 * NodeId order (as for orbfe_match_bow).  has_mp* [i] != 0 iff GetMapPoint(i) is set; stereo* [i] != 0 iff
 * mvuRight[i] >= 0 (NULL == monocular); scale_factors2 = pKF2->mvScaleFactors (n_levels2 floats).
 * matches12_out[i1] (n1 ints) = index in key frame 2 or -1, i.e. vMatchedPairs = {(i1, matches12_out[i1])} in
 * ascending i1; *n_matches = the return value.  HOST pointers. */
int orbfe_match_triangulation(orbfe_handle *h, int n_groups, const int *kf1_off, const int *kf1_idx,
                              const int *kf2_off, const int *kf2_idx, int n1, const orbfe_keypoint *kp1,
                              const uint8_t *desc1, const uint8_t *has_mp1, const uint8_t *stereo1, int n2,
                              const orbfe_keypoint *kp2, const uint8_t *desc2, const uint8_t *has_mp2,
                              const uint8_t *stereo2, const float *scale_factors2, int n_levels2,
                              const orbfe_tri_params *params, int *matches12_out, int *n_matches);

/* -------------------------------------------------------------------------------------------
 * Map points resident in HBM + the extract-and-match form of the ring
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_map orbfe_map;

/* A device-resident table of what isInFrustum and SearchByProjection read from a MapPoint (GetWorldPos, mfMinDistance,
 * mfMaxDistance, isBad, Observations: orbfe_world_point; GetDescriptor: 32 bytes), indexed by a caller-chosen id in
 * [0, capacity).  The local map of consecutive frames is nearly the same set of points (src/Tracking.cc:1117-1230), so a
 * frame names its points by id (4 bytes each) instead of shipping 64 bytes per point per frame across PCIe.
 * orbfe_map_update (HOST pointers) writes entries ids[0..n) -- new points, and points whose position / descriptor /
 * observation count changed; it is ordered behind everything submitted before it and returns when the table is updated.
 * The `skip` member of the records is ignored (it is per frame: see orbfe_stream_submit_track).
 * Lifetime: orbfe_map_destroy waits for the handle's stream, drops the graphs of orbfe_track_frame_map and detaches the map
 * from every ring it was given to (orbfe_stream_enable_track): that ring refuses later orbfe_stream_submit_track calls with
 * ORBFE_ERR_INVALID_ARG (collect what is in flight BEFORE destroying the map).  orbfe_destroy releases the device memory of the
 * maps and rings created from the handle; destroying them afterwards is safe and only frees the shell, any other call on them
 * returns ORBFE_ERR_INVALID_ARG. */
int orbfe_map_create(orbfe_handle *h, int capacity, orbfe_map **out);
void orbfe_map_destroy(orbfe_map *m);
int orbfe_map_update(orbfe_handle *h, orbfe_map *m, int n, const int *ids, const orbfe_world_point *points,
                     const uint8_t *desc);

/* Gives the ring what it needs to run the whole per-frame chain of orbfe_track_frame per slot: extraction, isInFrustum
 * of the frame's local map points against the frame's own pose, SearchByProjection -- H2D || extract + project + match ||
 * D2H.  max_points = the largest n_points a submission will carry.  Call once, before the first submission (a call that
 * fails with ORBFE_ERR_OUT_OF_MEMORY leaves the ring as it was and may be repeated). */
/* orbfe_track_frame with the local map points named by id out of the resident map (ids as orbfe_stream_submit_track:
 * id >= 0, ~id = skipped for this frame, outside the map = no point): 8 KB instead of 128 KB go up per frame at 2000
 * points.  mp_out / proj_xr_out / match_out index the id list.  Same results as orbfe_track_frame on the same points. */
int orbfe_track_frame_map(orbfe_handle *h, const uint8_t *gray, int pitch, const orbfe_frustum *frustum,
                          const orbfe_track_params *tp, const orbfe_map *map, int n_points, const int *ids,
                          orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out, int *per_level_counts,
                          orbfe_map_point *mp_out, float *proj_xr_out, int *match_out, int *n_matches);

int orbfe_stream_enable_track(orbfe_stream *s, orbfe_map *map, int max_points);
/* orbfe_stream_submit + per frame b: frusta[b] (pose, bounds, camera of THAT frame) and the ids of its n_points local
 * map points, ids[b * n_points + i]: id >= 0 = entry of the resident map; ~id (negative) = the same entry with
 * "mnLastFrameSeen == current frame" (src/Tracking.cc:1066, skipped); an id outside the map = no point.  tp as
 * orbfe_track_frame.  HOST pointers; everything is copied before the call returns except pinned frames (see above). */
int orbfe_stream_submit_track(orbfe_stream *s, const uint8_t *const *grays, int pitch, int n, const orbfe_track_params *tp,
                              const orbfe_frustum *frusta, int n_points, const int *ids);
/* orbfe_stream_collect + the matches of a submission made with orbfe_stream_submit_track: match_out[b * cap + i] =
 * position in frame b's id list of the map point written to mvpMapPoints[i], or -1 (cap = orbfe_max_keypoints());
 * n_matches[b] = the return value of SearchByProjection.  Byte-identical to orbfe_track_frame per frame. */
int orbfe_stream_collect_track(orbfe_stream *s, orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out, int *per_level_counts,
                               int *match_out, int *n_matches, int *n_frames);

/* -------------------------------------------------------------------------------------------
 * Key frames resident in HBM + the mapping thread's batched SearchForTriangulation
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_keyframe orbfe_keyframe;

/* Uploads what the key-frame matchers read from a KeyFrame and what never changes after its construction
 * (src/KeyFrame.cc:33-80): mvKeysUn (n keypoints), mDescriptors (n x 32), mFeatVec as node_id[i] = the
 * DBoW2::FeatureVector key of feature i (the vocabulary node `levelsup` levels above its word, src/Frame.cc:483-495;
 * -1 = the feature is in no node: DBoW2 adds a feature to the FeatureVector only when its word's weight is > 0,
 * TemplatedVocabulary.h:1168-1172, so features on stopped words get -1), stereo[i] != 0 iff mvuRight[i] >= 0 (NULL == monocular), mvScaleFactors.
 * The features of a node are walked in ascending feature index, the order DBoW2 stores them in
 * (Thirdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1157-1170).  HOST pointers; one upload, then the key frame stays
 * on the device until orbfe_keyframe_destroy.  Keypoint octaves must lie in [0, n_levels). */
int orbfe_keyframe_create(orbfe_handle *h, int n, const orbfe_keypoint *kp, const uint8_t *desc, const int *node_id,
                          const uint8_t *stereo, const float *scale_factors, int n_levels, orbfe_keyframe **out);
void orbfe_keyframe_destroy(orbfe_keyframe *kf);
int orbfe_keyframe_size(const orbfe_keyframe *kf);

/* replaces the K calls ORBmatcher::SearchForTriangulation(mpCurrentKeyFrame, pKF2, vMatchedIndices, false, bCoarse, true)
 * of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:455-488, up to 30 neighbours per new key frame) by ONE
 * launch: kf1 against kf2[0..K), params[k] as orbfe_match_triangulation's for the pair (kf1, kf2[k]),
 * has_mp1[i] / has_mp2[k][i] != 0 iff GetMapPoint(i) is set WHEN THE CALL IS MADE.
 * Between two neighbours the reference turns matches into new map points (:500-700); a feature of key frame 1 that
 * received one is skipped for the later neighbours (src/ORBmatcher.cc:506-509) and drops out of their rotation
 * histograms.  So this call returns, per neighbour k and feature i1 of key frame 1, the RAW partner
 * raw_match12[k * n1 + i1] (index in kf2[k] or -1) and its rotation bin raw_bin[k * n1 + i1] BEFORE the orientation
 * filter; the caller walks the neighbours in order and calls orbfe_triangulation_select with the has_mp1 flags as they
 * stand at that point.  That reproduces the K sequential calls exactly: vbMatched2 is never set in this fork
 * (src/ORBmatcher.cc:485,537), so every feature of key frame 1 chooses its partner independently of the others, and the
 * only state shared between neighbours is "idx1 already has a map point".  HOST pointers except the key frames. */
int orbfe_match_triangulation_batch(orbfe_handle *h, const orbfe_keyframe *kf1, const uint8_t *has_mp1, int K,
                                    const orbfe_keyframe *const *kf2, const uint8_t *const *has_mp2,
                                    const orbfe_tri_params *params, int *raw_match12, uint8_t *raw_bin);
/* host-only: vMatchedPairs / the return value of neighbour k's SearchForTriangulation call from the raw results of the
 * batch call and the CURRENT flags of key frame 1: features with has_mp1_now[i1] != 0 drop out (:506-509), then the
 * rotation histogram and ComputeThreeMaxima (:633-661, :1328-1370) when check_orientation.  raw_match12 / raw_bin point
 * at neighbour k's n1 entries.  matches12_out[i1] = index in key frame 2 or -1. */
int orbfe_triangulation_select(int n1, const int *raw_match12, const uint8_t *raw_bin, const uint8_t *has_mp1_now,
                               int check_orientation, int *matches12_out, int *n_matches);

/* replaces the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false)
 * (src/ORBmatcher.cc:678-836; callers src/LocalMapping.cc:822,852): per map point the projection into the
 * key frame, KeyFrame::IsInImage, PredictScale, KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:790-833), the
 * chi-square gate and the nearest descriptor.  points[i].skip carries "!pMP || pMP->IsInKeyFrame(pKF)".
 * inv_level_sigma2 = pKF->mvInvLevelSigma2 (KF->n_levels floats); u_right = pKF->mvuRight or NULL (monocular).
 * best_idx_out[i] / best_dist_out[i] = bestIdx / bestDist of :776-826 (-1 / 256 when nothing qualified); the
 * caller applies bestDist <= TH_LOW and the map-point graph edits (:829-849) in list order.  HOST pointers. */
int orbfe_fuse_search(orbfe_handle *h, const orbfe_frame_view *KF, const float *inv_level_sigma2,
                      const float *u_right, const orbfe_frustum *frustum, float th, int M,
                      const orbfe_world_point *points, const uint8_t *mp_desc, int *best_idx_out,
                      int *best_dist_out);

/* What the projection searches INTO a resident key frame read besides orbfe_keyframe_create's arrays: mGrid (the geometry of
 * orbfe_frame_view: grid_cols x grid_rows cells from (min_x, min_y) with the inverse cell sizes, src/KeyFrame.cc:33-80 copies them
 * from the Frame), mvInvLevelSigma2 (n_levels floats) and mvuRight (n floats, or NULL == monocular).  The per-level cell
 * tables are built here, ONCE, and stay with the key frame; calling it again replaces them.  HOST pointers; synchronises. */
int orbfe_keyframe_set_grid(orbfe_handle *h, orbfe_keyframe *kf, int grid_cols, int grid_rows, float min_x, float min_y,
                            float grid_inv_w, float grid_inv_h, const float *inv_level_sigma2, const float *u_right);

/* orbfe_fuse_search on RESIDENT data -- LocalMapping::SearchInNeighbors (src/LocalMapping.cc:764-860) calls
 * ORBmatcher::Fuse(pKFi, vpMapPointMatches) for every neighbour and Fuse(mpCurrentKeyFrame, vpFuseCandidates) once
 * (:822,:852): up to 2 x 30 calls per new key frame, every one of which re-uploaded its target key frame (56 B per feature)
 * and its map points (64 B each) through orbfe_fuse_search although both already sit in HBM.  Here the target is a key
 * frame of orbfe_keyframe_create with orbfe_keyframe_set_grid done, the map points are entries of an orbfe_map named by id:
 * ids[i] >= 0 = the entry; ~id (negative) = the entry with "!pMP || pMP->IsInKeyFrame(pKF)" for this call (skipped,
 * src/ORBmatcher.cc:706-721); an id outside the map = no point.  Up go the frustum and 4 bytes per map point, down come
 * best_idx_out[i] / best_dist_out[i] as orbfe_fuse_search (-1 / 256 when nothing qualified).  The calls of one
 * SearchInNeighbors stay sequential -- between two of them the caller replaces / adds map points (:829-849); descriptor
 * and position changes reach the map through orbfe_map_update, which is ordered in front of the next call.
 * Same results as orbfe_fuse_search on the same key frame, map points and flags (monocular / stereo left camera; the bRight
 * overload has no resident form). */
int orbfe_fuse_search_keyframe(orbfe_handle *h, const orbfe_keyframe *kf, const orbfe_map *map, int M, const int *ids,
                               const orbfe_frustum *frustum, float th, int *best_idx_out, int *best_dist_out);

/* the same with bRight = true (src/ORBmatcher.cc:684-688,:820; callers src/LocalMapping.cc:824,854 when the key frame
 * has a second camera): frustum carries pKF->GetRightPose() / GetRightTranslationInverse() / mpCamera2; KF_left
 * describes the LEFT features (KF_left->n == pKF->NLeft -- they fill mGrid, and the level and chi-square gates read them,
 * as in the reference) while KF_left->desc is all of pKF->mDescriptors: NLeft + n_right rows.  The compared descriptor
 * row and best_idx_out are idx + NLeft (:820).  Where the reference would read a row beyond the matrix (idx >= n_right)
 * the candidate is skipped. */
int orbfe_fuse_search_right(orbfe_handle *h, const orbfe_frame_view *KF_left, int n_right, const float *inv_level_sigma2,
                            const float *u_right, const orbfe_frustum *frustum, float th, int M,
                            const orbfe_world_point *points, const uint8_t *mp_desc, int *best_idx_out,
                            int *best_dist_out);

/* replaces the search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:864-975;
 * no caller in this fork -- the loop-closing thread is gone -- but part of the class, include/ORBmatcher.h:69): the
 * same walk as orbfe_fuse_search without the chi-square gate.  The caller decomposes Scw as the reference does
 * (:867-868): frustum->rcw = Scw.rotationMatrix(), tcw = Scw.translation() / Scw.scale(), twc =
 * Tcw.inverse().translation(); points[i].skip carries "spAlreadyFound.count(pMP)".  The caller applies
 * bestDist <= TH_LOW, vpReplacePoint and the graph edits (:955-971) in list order.  HOST pointers. */
int orbfe_fuse_search_sim3(orbfe_handle *h, const orbfe_frame_view *KF, const orbfe_frustum *frustum, float th, int M,
                           const orbfe_world_point *points, const uint8_t *mp_desc, int *best_idx_out,
                           int *best_dist_out);

/* one search direction of ORBmatcher::SearchBySim3 */
typedef struct orbfe_sim3_view {
    float rcw[9], tcw[3];     /* pose of the key frame that owns the map points (T1w for 1->2, :986-987) */
    float sr[9], t[3];        /* the similarity into the other key frame as s*R (row-major) and t (S21 for 1->2) */
    float fx, fy, cx, cy;     /* pKF1's intrinsics -- the reference uses them for both directions (:979-982) */
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY of the TARGET key frame (KeyFrame::IsInImage) */
    float log_scale_factor;   /* mfLogScaleFactor of the target key frame */
    int n_levels;             /* mnScaleLevels of the target key frame */
} orbfe_sim3_view;

/* replaces ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:977-1200; no caller in this
 * fork).  mp1 / mp_desc1: one record per feature of key frame 1 (KF1->n entries; skip = "no map point" or
 * vbAlreadyMatched1 of :1001-1012), mp2 / mp_desc2 likewise for key frame 2.  dir12 moves key frame 1's points
 * into key frame 2 (T1w, S21), dir21 the other way (T2w, S12).  match12_out[i1] (KF1->n ints) = feature of key
 * frame 2 whose map point the reference stores in vpMatches12[i1], or -1; *n_found = the return value.
 * HOST pointers. */
int orbfe_search_by_sim3(orbfe_handle *h, const orbfe_frame_view *KF1, const orbfe_frame_view *KF2,
                         const orbfe_sim3_view *dir12, const orbfe_sim3_view *dir21, const orbfe_world_point *mp1,
                         const uint8_t *mp_desc1, const orbfe_world_point *mp2, const uint8_t *mp_desc2, float th,
                         int *match12_out, int *n_found);

/* replaces ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, checkOrientation)
 * (src/ORBmatcher.cc:1202-1326, the relocalisation overload; no caller in this fork).  points / mp_desc / kf_angle:
 * one record per feature i of the key frame (skip = "no map point" or "in sAlreadyFound"; kf_angle[i] =
 * pKF->mvKeysUn[i].angle, may be NULL when check_orientation == 0); frustum = the current frame's pose, bounds,
 * camera and scale pyramid; frame_has_mp[i2] != 0 iff CurrentFrame->mvpMapPoints[i2] is set on entry (NULL == none).
 * match_out[i2] (frame->n ints) = key-frame feature whose map point this call leaves in
 * CurrentFrame->mvpMapPoints[i2], or -1; *n_matches = the return value.  The greedy order-dependent claims and the
 * rotation-histogram filter are reproduced exactly.  HOST pointers. */
int orbfe_match_projection_keyframe(orbfe_handle *h, const orbfe_frame_view *frame, const orbfe_frustum *frustum,
                                    int n_points, const orbfe_world_point *points, const uint8_t *mp_desc,
                                    const float *kf_angle, const uint8_t *frame_has_mp, float th,
                                    int check_orientation, int *match_out, int *n_matches);

/* replaces the selection loop of MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:343-416; called after every
 * AddObservation / Fuse, src/LocalMapping.cc) for a batch of map points: set s holds the observed descriptors
 * desc[set_off[s] .. set_off[s+1]) (32 bytes each, in the order the reference collects them); best_idx_out[s] =
 * index inside the set of the descriptor with the least median distance to the rest (-1 for an empty set, the
 * reference returns early); best_median_out may be NULL.  HOST pointers. */
int orbfe_distinctive_descriptors(orbfe_handle *h, int n_sets, const int *set_off, const uint8_t *desc,
                                  int *best_idx_out, int *best_median_out);

/* -------------------------------------------------------------------------------------------
 * Node-side image preparation (honourable mention): undistort + resize + grey
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_prep orbfe_prep;

/* Uploads the undistortion maps the node builds with cv::fisheye::initUndistortRectifyMap(..., CV_32F, map1, map2)
 * (ros2_ws/src/mono-inertial/src/mono_inertial_node.cpp:61-71): map1 / map2 hold src_h x src_w floats (x and y source
 * coordinates per undistorted pixel, row-major, no padding).  dst_w x dst_h is the size ImageGrabber resizes to
 * (m_width x m_height, image_grabber.hpp:102). */
int orbfe_prep_create(orbfe_handle *h, int src_w, int src_h, const float *map1, const float *map2, int dst_w,
                      int dst_h, orbfe_prep **out);
void orbfe_prep_destroy(orbfe_prep *p);

/* replaces ImageGrabber::ConvertImageToGPU (ros2_ws/src/mono-inertial/include/image_grabber.hpp:96-110):
 * cv::cuda::remap(INTER_CUBIC, BORDER_CONSTANT 0) + cv::cuda::resize(INTER_LINEAR) + cv::cuda::cvtColor(BGR2GRAY) in
 * one kernel.  bgr: src_h rows of src_w BGR pixels (3 bytes each), `pitch` bytes per row, HOST pointer (pinned buffers
 * are read by the DMA engine directly); gray_out: dst_h rows of dst_w bytes, HOST pointer.  Arithmetic: SPEC DECISION S9
 *  -- parity against OpenCV-CUDA is unpinned (third-party arithmetic, absent here). */
int orbfe_prepare_image(orbfe_handle *h, orbfe_prep *p, const uint8_t *bgr, int pitch, uint8_t *gray_out,
                        int gray_pitch);
/* Same with DEVICE pointers, asynchronous on `stream` (NULL == the handle's stream). */
int orbfe_prepare_image_device(orbfe_handle *h, orbfe_prep *p, const uint8_t *d_bgr, int pitch, uint8_t *d_gray,
                               int gray_pitch, void *stream);
/* ConvertImageToGPU followed by ORBextractor::extractFeatures (what the node and Frame::ExtractORB do back to back,
 * image_grabber.hpp:160 -> src/Frame.cc:178-189) without the grey image leaving the device: the prepared frame is
 * written straight into the extractor's level-0 rows.  The handle must have been created for dst_w x dst_h images.
 * gray_out (optional, may be NULL) receives the grey frame as well (the tracker keeps it for the viewer). */
int orbfe_prepare_and_extract(orbfe_handle *h, orbfe_prep *p, const uint8_t *bgr, int pitch, orbfe_keypoint *kp_out,
                              uint8_t *desc_out, int *n_out, int *per_level, uint8_t *gray_out, int gray_pitch);

/* -------------------------------------------------------------------------------------------
 * Vocabulary tree: the per-feature part of Frame::ComputeBoW
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_vocab orbfe_vocab;

/* Uploads a DBoW2 vocabulary tree (TemplatedVocabulary::m_nodes,
 * Thirdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h): node 0 is the root, child_idx[child_off[i] ..
 * child_off[i+1]) are the children of node i in DBoW2 order (a node without children is a leaf),
 * node_desc n_nodes x 32 bytes, word_id / weight per node (read at leaves), L = m_L. */
int orbfe_vocab_create(orbfe_handle *h, int n_nodes, const int *child_off, const int *child_idx,
                       const uint8_t *node_desc, const int *word_id, const double *weight, int L,
                       orbfe_vocab **out);
void orbfe_vocab_destroy(orbfe_vocab *v);
/* replaces TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (:1227-1270) for n
 * descriptors at once (HOST pointers): per feature the word id, its weight and the id of the node at
 * level L - levelsup.  The adaptor assembles BowVector / FeatureVector from these (:1136-1204). */
int orbfe_bow_transform(orbfe_handle *h, orbfe_vocab *v, const uint8_t *desc, int n, int levelsup,
                        int *word_id_out, int *node_id_out, double *weight_out);

/* -------------------------------------------------------------------------------------------
 * The tracking thread's chain of a frame tracked against its reference key frame, as ONE submission
 * ---------------------------------------------------------------------------------------- */
/* replaces, for one frame (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:825-835 -- every frame while the map's IMU
 * is not initialised, :454-458):
 *   Frame::Frame -> ExtractORB -> ORBextractor::extractFeatures                    (src/Frame.cc:178-189)
 *   mCurrentFrame->ComputeBoW(): TemplatedVocabulary::transform per feature        (src/Frame.cc:483-495)
 *   ORBmatcher::SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches, nnRatio, true)   (src/ORBmatcher.cc:133-327)
 * gray / pitch / kp_out / desc_out / n_out / per_level_counts as orbfe_extract.  word_id_out / node_id_out / weight_out
 * (capacity orbfe_max_keypoints(); weight_out may be NULL) as orbfe_bow_transform on the frame's descriptors: the caller
 * assembles mBowVec / mFeatVec from them.  kf: the reference key frame, resident (orbfe_keyframe_create, whose node_id
 * came from the same vocabulary and levelsup); kf_has_mp[j] != 0 iff key-frame feature j has a map point that is not bad
 * as of this call (:182-187), orbfe_keyframe_size(kf) entries.  match_out[i] (capacity orbfe_max_keypoints()) = the
 * key-frame feature whose map point is written to vpMapPointMatches[i], or -1; n_matches = the return value.
 * The frame's descriptors never leave the device between extraction, descent and matching; nothing is decided on the
 * host in between.  A frame feature whose word has weight 0 (a stopped word) is in no node of mFeatVec
 * (TemplatedVocabulary.h:1168-1172,1196-1200) and takes no part in the matching, although its word_id_out / node_id_out /
 * weight_out (== 0) are still returned.  Results are byte-identical to orbfe_extract -> orbfe_bow_transform ->
 * orbfe_match_bow with the FeatureVectors of both sides (features with weight > 0 only).  HOST pointers; gray may be pinned (read in place) or pageable. */
int orbfe_track_reference_keyframe(orbfe_handle *h, const uint8_t *gray, int pitch, const orbfe_vocab *vocab, int levelsup,
                                   const orbfe_keyframe *kf, const uint8_t *kf_has_mp, float nn_ratio,
                                   int check_orientation, orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out,
                                   int *per_level_counts, int *word_id_out, int *node_id_out, double *weight_out,
                                   int *match_out, int *n_matches);

/* -------------------------------------------------------------------------------------------
 * The tracking thread's chain of a frame during monocular initialisation, as ONE submission
 * ---------------------------------------------------------------------------------------- */
typedef struct orbfe_init_frame orbfe_init_frame;

/* mInitialFrame of Tracking::MonocularInitialization (src/Tracking.cc:569-586: the first frame with more than FEAT_INIT_COUNT
 * keypoints becomes the reference every following frame is matched against until the map is initialised or the attempt is
 * reset, :588-602), resident in HBM: its mvKeysUn (n keypoints) and mDescriptors (n x 32).  HOST pointers; one upload.
 * The frame stays on the device until orbfe_init_frame_destroy (which waits for the device). */
int orbfe_init_frame_create(orbfe_handle *h, int n, const orbfe_keypoint *kp, const uint8_t *desc, orbfe_init_frame **out);
void orbfe_init_frame_destroy(orbfe_init_frame *f);
int orbfe_init_frame_size(const orbfe_init_frame *f);

/* replaces, for one frame while the map is being initialised (Tracking::MonocularInitialization, src/Tracking.cc:566-607):
 *   Frame::Frame -> ExtractORB -> ORBextractor::extractFeatures                                          (src/Frame.cc:178-189)
 *   ORBmatcher::SearchForInitialization(mInitialFrame, mCurrentFrame, windowSize, nnRatio, true)         (src/ORBmatcher.cc:329-439;
 *                                                                                          the call: src/Tracking.cc:603-607, 40 / 0.45)
 * gray / pitch / kp_out / desc_out / n_out / per_level_counts as orbfe_extract; tp: the CURRENT frame's grid statics
 * (GetFeaturesInArea runs on frame 2; th / nn_ratio / far_points of the block are not read); matches12_out
 * (orbfe_init_frame_size(f1) ints) = vnMatches12 (index in the current frame or -1), *n_matches = the function's return value.
 * The thresholds around the call (FEAT_INIT_COUNT, the 2 s time-out, :579,:588-598) stay with the caller.
 * One captured hipGraph per (initial frame, input pitch, parameters): extraction kernels, the matcher's kernels on the fresh
 * keypoints, ONE result block back, one synchronisation.  Byte-identical to orbfe_extract followed by
 * orbfe_match_initialization(f1, current frame).  HOST pointers; gray may be pinned (read in place) or pageable. */
int orbfe_track_initialization(orbfe_handle *h, const uint8_t *gray, int pitch, const orbfe_init_frame *f1,
                               const orbfe_track_params *tp, int window_size, float nn_ratio, int check_orientation,
                               orbfe_keypoint *kp_out, uint8_t *desc_out, int *n_out, int *per_level_counts,
                               int *matches12_out, int *n_matches);

/* -------------------------------------------------------------------------------------------
 * Misc
 * ---------------------------------------------------------------------------------------- */
const char *orbfe_status_string(int status);
/* The host-pointer entry points capture their launch sequence into a hipGraph the first time a shape is seen (on a throw-away
 * stream) and replay it afterwards.  On ROCm 7.2 a NULL-stream operation of ANOTHER thread of the process (a plain hipMemcpy /
 * hipMemset of the application) that meets a capture in flight is refused by the runtime with hipErrorStreamCaptureImplicit
 * and kills the capture: the library call then runs on plain launches and is unaffected, but the application's call has
 * failed.  An application that uses the NULL stream from other threads can switch capturing off (enable = 0: plain launches,
 * about 0.04 ms more per single-frame call) or make its first call of every shape before those threads start.  The
 * library itself never uses the NULL stream. */
int orbfe_set_graph_capture(orbfe_handle *h, int enable);
/* Priority of the handle's own HIP stream (the one every host-pointer entry point and every `stream == NULL` call runs on).
 * The reference runs tracking and local mapping on two threads (src/System.cc); with one handle each, high = 1 on the tracking
 * thread's handle lets its kernels be dispatched ahead of the mapping thread's whenever both have work queued (running waves
 * are not pre-empted).  The stream is re-created: call it while the handle is idle (nothing submitted and not yet collected),
 * typically right after orbfe_create.  high = 0 selects the lowest priority of the device's range. */
int orbfe_set_stream_priority(orbfe_handle *h, int high);
/* diagnostics: graphs this handle has captured so far, and captures that failed (e.g. invalidated by another thread's NULL-stream
 * call: the affected call ran on plain launches, its result is unaffected; after 8 failures IN A ROW a handle stops capturing
 * until orbfe_set_graph_capture(h, 1)) */
int orbfe_debug_graph_stats(orbfe_handle *h, int *captured, int *failed);
/* diagnostics (bench.py's `sustained.sclk_mhz`): one wave on `stream` (NULL == the handle's stream; asynchronous) stamps the shader-cycle
 * counter and the constant 100 MHz counter, sleeps for spin_us microseconds and stamps again: d_out[0] = shader cycles, d_out[1] =
 * 100 MHz ticks that went by (DEVICE pointer to two 64-bit words), i.e. the clock the chip held meanwhile is d_out[0] / d_out[1] x
 * 100 MHz.  Launched on a stream of its own beside a running workload it reads the clock THAT workload runs at. */
int orbfe_debug_clock_probe(orbfe_handle *h, int spin_us, unsigned long long *d_out, void *stream);
/* message of the last failing HIP call on this handle ("" if none) */
const char *orbfe_last_error(const orbfe_handle *h);
/* library / build identification, e.g. "orbfe 0.1 gfx950" */
const char *orbfe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
