// match.h -- host entry points of the Hamming matchers (kernels_match.hip).
#pragma once
#include <string>

#include "orbfe_internal.h"

namespace orbfe {

// device + pinned scratch reused across matcher calls (grown on demand, never shrunk)
struct MatchScratch {
    void* d = nullptr;      // device arena
    size_t dBytes = 0;
    void* hpin = nullptr;   // pinned host arena
    size_t hBytes = 0;
    hipEvent_t busy = nullptr;  // recorded behind the last launch that used the arenas (owned by the handle)
};

void match_scratch_free(MatchScratch& m);

int match_projection_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F, int M,
                         const orbfe_map_point* mps, const uint8_t* mpDesc, const int* initObs, float th,
                         int farPoints, float thFar, float nnRatio, int* matchOut, int* nMatches,
                         std::string& err);

int match_projection_batch_device(MatchScratch& m, hipStream_t s, int B, const orbfe_keypoint* dKp, const uint8_t* dDesc,
                                  const int* dN, int kpStride, int gridCols, int gridRows, float minX, float minY,
                                  float invW, float invH, const float* dScaleFactors, int nLevels, int M,
                                  const orbfe_map_point* dMps, const uint8_t* dMpDesc, const int* dInitObs, float th,
                                  int farPoints, float thFar, float nnRatio, int* dMatchOut, int* dNMatches,
                                  std::string& err);

int match_initialization_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F1, const orbfe_frame_view* F2,
                             int windowSize, float nnRatio, int checkOrientation, int* matches12Out, int* nMatches,
                             std::string& err);

// the same matcher inside a captured per-frame chain (orbfe_track_initialization): frame 1 = a resident initial frame (n1
// keypoints, n0 of them on level 0, listed in index order in dList0), frame 2 = the fresh keypoints of the current frame on the
// device (count in *dN2, <= cap); `scratch` = init_track_scratch_bytes(n1, n0, cap) bytes of device memory
size_t init_track_scratch_bytes(int n1, int n0, int cap);
int init_track_launch(hipStream_t s, int n1, int n0, const orbfe_keypoint* dKp1, const uint8_t* dDesc1, const int* dList0,
                      const orbfe_keypoint* dKp2, const uint8_t* dDesc2, const int* dN2, int cap, int gridCols, int gridRows,
                      float minX, float minY, float invW, float invH, int windowSize, float nnRatio, int checkOrientation,
                      int* dMatches12, int* dNMatches, uint8_t* scratch, std::string& err);

int match_bow_run(MatchScratch& m, hipStream_t s, int G, const int* kfOff, const int* kfIdx, const int* fOff,
                  const int* fIdx, int nKF, const uint8_t* kfDesc, const float* kfAngle, const uint8_t* kfHasMP,
                  int nF, const uint8_t* fDesc, const float* fAngle, int nLeft, float nnRatio, int checkOrientation,
                  int* matchOut, int* nMatches, std::string& err);

// the search part of ORBmatcher::Fuse (kernels_match_kf.hip)
int fuse_search_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* KF, const float* invLevelSigma2,
                    const float* uRight, const orbfe_frustum* F, float th, int M, const orbfe_world_point* pts,
                    const uint8_t* mpDesc, int chi2Gate, int nRight, int* bestIdxOut, int* bestDistOut, std::string& err);
// ORBmatcher::SearchBySim3 and the relocalisation SearchByProjection overload (kernels_match_kf.hip)
int search_by_sim3_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* KF1, const orbfe_frame_view* KF2,
                       const orbfe_sim3_view* d12, const orbfe_sim3_view* d21, const orbfe_world_point* mp1,
                       const uint8_t* mpDesc1, const orbfe_world_point* mp2, const uint8_t* mpDesc2, float th,
                       int* match12Out, int* nFound, std::string& err);
int match_projection_kf_run(MatchScratch& m, hipStream_t s, const orbfe_frame_view* F, const orbfe_frustum* Fr, int M,
                            const orbfe_world_point* pts, const uint8_t* mpDesc, const float* kfAngle,
                            const uint8_t* frameHasMP, float th, int checkOrientation, int* matchOut, int* nMatches,
                            std::string& err);
// kernels_match_tri.hip
int match_triangulation_run(MatchScratch& m, hipStream_t s, int G, const int* off1, const int* idx1, const int* off2,
                            const int* idx2, int n1, const orbfe_keypoint* kp1, const uint8_t* desc1, const uint8_t* hasMP1,
                            const uint8_t* stereo1, int n2, const orbfe_keypoint* kp2, const uint8_t* desc2,
                            const uint8_t* hasMP2, const uint8_t* stereo2, const float* sf2, int nLevels2,
                            const orbfe_tri_params* P, int* matches12, int* nMatches, std::string& err);
// kernels_distinct.hip (MapPoint::ComputeDistinctiveDescriptors for a batch)
int distinctive_run(MatchScratch& m, hipStream_t s, int nSets, const int* setOff, const uint8_t* desc, int* bestIdx,
                    int* bestMedian, std::string& err);
// kernels_frustum.hip
int frustum_validate(const orbfe_frustum* F);
int frustum_launch(hipStream_t s, const orbfe_frustum* F, int n, const orbfe_world_point* dPts, orbfe_map_point* dOut,
                   float* dProjXR, std::string& err);
// the same with the frustum block resident in HBM (orbfe_track_frame: the launch is part of a captured hipGraph)
int frustum_launch_dev(hipStream_t s, const orbfe_frustum* dF, int n, const orbfe_world_point* dPts, orbfe_map_point* dOut,
                       float* dProjXR, std::string& err);

// resident map points (orbfe_map): gather by id + isInFrustum for B frames with their own frusta; scatter of updated entries
int frustum_gather_launch(hipStream_t s, int B, const orbfe_frustum* dF, const int* dIds, int M, int mapCap,
                          const orbfe_world_point* mapPts, const uint8_t* mapDesc, orbfe_map_point* dOut, uint8_t* dDescOut,
                          float* dProjXR, std::string& err);
int map_scatter_launch(hipStream_t s, int n, const int* dIds, const orbfe_world_point* dPts, const uint8_t* dDesc, int mapCap,
                       orbfe_world_point* mapPts, uint8_t* mapDesc, std::string& err);

}  // namespace orbfe
