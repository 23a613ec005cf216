// keyframe.h -- key frames resident in HBM (kernels_match_tri.hip).
//
// What the key-frame matchers read from a KeyFrame and what never changes after its construction (src/KeyFrame.cc:33-80):
// mvKeysUn, mDescriptors, mFeatVec, mvuRight >= 0, mvScaleFactors.  Uploaded once per key frame (56 B per feature + the
// FeatureVector as sorted arrays: a 1000-feature key frame is 70 KB, ten thousand of them 0.7 GB of the 288 GB), so the
// up to 30 SearchForTriangulation calls a new key frame triggers (src/LocalMapping.cc:455-488) move only the flags and the
// per-pair geometry across PCIe.
#pragma once
#include <string>

#include "match.h"
#include "match_proj.h"

namespace orbfe {

struct KeyFrameDev {
    void* block = nullptr;  // one allocation; the pointers below point into it
    size_t bytes = 0;
    int n = 0, nLevels = 0, G = 0;
    bool hasStereo = false;
    const orbfe_keypoint* kp = nullptr;
    const uint8_t* desc = nullptr;
    const int* node = nullptr;       // [n] vocabulary node of every feature (FeatureVector key), -1 = none
    const uint8_t* stereo = nullptr; // [n] mvuRight >= 0, or null (monocular)
    const float* sf = nullptr;       // mvScaleFactors
    const int* order = nullptr;      // features with a node, sorted by (node, index)
    const int* nodeList = nullptr;   // [G] distinct nodes, ascending
    const int* nodeOff = nullptr;    // [G + 1] ranges of `order`
    // ---- what the projection searches into this key frame read (Fuse: KeyFrame::GetFeaturesInArea on mGrid,
    //      src/KeyFrame.cc:790-833), built ONCE by keyframe_set_grid: the per-level cell tables of kernels_match_proj.hip
    //      (storage order, cell-column starts, descriptors in storage order) plus mvInvLevelSigma2 and mvuRight ----
    bool hasGrid = false;
    MatchScratch gridMem;            // owns the tables (device arena)
    proj::ProjArgs grid{};           // bound to gridMem + the key frame's own keypoints / descriptors / scale factors
    const float* invLevelSigma2 = nullptr;  // [nLevels]
    const float* uRight = nullptr;          // [n] or null (monocular)
};

int keyframe_create(int n, const orbfe_keypoint* kp, const uint8_t* desc, const int* nodeId, const uint8_t* stereo,
                    const float* sf, int nLevels, hipStream_t s, KeyFrameDev** out, std::string& err);
void keyframe_destroy(KeyFrameDev* K);

// mGrid of the key frame (grid geometry as orbfe_frame_view), mvInvLevelSigma2 (nLevels floats) and mvuRight (n floats or
// null): one launch of the grid kernel, tables kept with the key frame.  Synchronises.
int keyframe_set_grid(KeyFrameDev* K, hipStream_t s, int gridCols, int gridRows, float minX, float minY, float invW, float invH,
                      const float* invLevelSigma2, const float* uRight, std::string& err);
// the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th) against a RESIDENT key frame (keyframe_set_grid done) with the map
// points named by id out of a resident map: ids[i] >= 0 entry of the map, ~id (negative) = "!pMP || pMP->IsInKeyFrame(pKF)"
// for this call (skipped), outside the map = no point.  Up: the ids and the frustum; down: (bestIdx, bestDist) per id.
int fuse_search_keyframe_run(MatchScratch& m, hipStream_t s, const KeyFrameDev* K, int mapCap, const orbfe_world_point* mapPts,
                             const uint8_t* mapDesc, int M, const int* ids, const orbfe_frustum* F, float th, int* bestIdxOut,
                             int* bestDistOut, std::string& err);

// SearchForTriangulation of key frame 1 against K neighbours in one launch: raw matches + rotation bins per (neighbour,
// feature of key frame 1); the per-neighbour selection runs on the host (orbfe_triangulation_select)
int match_triangulation_batch_run(MatchScratch& m, hipStream_t s, const KeyFrameDev* kf1, const uint8_t* hasMP1, int K,
                                  const KeyFrameDev* const* kf2, const uint8_t* const* hasMP2, const orbfe_tri_params* P,
                                  int* rawMatch, uint8_t* rawBin, std::string& err);

// SearchByBoW of a frame whose features are still on the device (kernels_match_bow.hip, orbfe_track_reference_keyframe).
// The key frame is named by a record in device memory, uploaded with the call's inputs, so a captured graph of the chain
// serves every reference key frame.
struct BowKfRef {
    const uint8_t* desc;
    const orbfe_keypoint* kp;
    const int* order;
    const int* nodeList;
    const int* nodeOff;
    int G, n;
};
struct BowTrackArgs {
    const BowKfRef* ref;        // device
    const uint8_t* kfHasMP;     // device [ref->n]: the key-frame feature has a map point that is not bad (:182-187)
    const orbfe_keypoint* fKp;  // frame keypoints (orientation), descriptors, (word, node) of every feature, count
    const uint8_t* fDesc;
    const int* fBow;            // [cap][2]
    const int* fLeaf;           // [cap] the leaf (word) every feature reached, and the vocabulary's node weights: a feature whose
    const double* weight;       // word has weight 0 is in no node of mFeatVec (addFeature runs only for w > 0: TemplatedVocabulary.h:1168-1172)
    const int* nF;
    int cap;
    float nnRatio;
    int checkOrientation;
    int* matchOut;              // [cap] key-frame feature matched to frame feature i, -1 = none
    int* binOf;                 // [cap] scratch
    int* nMatches;
};
int bow_track_launch(hipStream_t s, const BowTrackArgs& A, std::string& err);

}  // namespace orbfe
