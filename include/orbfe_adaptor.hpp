// orbfe_adaptor.hpp -- header-only C++17 adaptor that keeps the reference's public signatures on
// top of the C ABI (include/orbfe.h), so src/Frame.cc / src/Tracking.cc call the MI355X front-end
// unchanged.
//
//   ORB_SLAM3::ORBextractor   mirrors include/ORBextractor.h:52-130 (ctor :55-56, extractFeatures
//                             :62, the six getters :64-92)
//   ORB_SLAM3::ORBmatcher     mirrors include/ORBmatcher.h:36-84 for the functions on the hot path
//                             (DescriptorDistance :41, SearchByProjection :45, SearchByBoW :55)
//
// Without OpenCV (this repository's build) the image / descriptor containers are plain views and
// std::vector; define ORBFE_WITH_OPENCV in a tree that has OpenCV-CUDA to get the exact reference
// types (cv::cuda::HostMem in, HostMem N x 32 out).  The matcher wrappers are templates over the
// Frame / KeyFrame / MapPoint types: they only name the members the reference functions read
// (mvKeysUn, mDescriptors, mvpMapPoints, mbTrackInView, ...), so they instantiate against the real
// classes in the reference tree and against light mock types in tests/cpp/test_adaptor.cpp.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "orbfe.h"

#ifdef ORBFE_WITH_OPENCV
#include <opencv2/core/cuda.hpp>
#include <KeyPoint.h>  // the reference's ORB_SLAM3::KeyPoint
#endif

namespace ORB_SLAM3 {

#ifndef ORBFE_WITH_OPENCV
// include/KeyPoint.h:7-12 without cv::Point2f (same 24-byte layout: pt.x, pt.y, response, size, octave, angle)
struct KeyPoint {
    struct { float x, y; } pt;
    int response;
    float size;
    int octave;
    float angle;
};
#endif
static_assert(sizeof(KeyPoint) == sizeof(orbfe_keypoint), "KeyPoint must stay memcpy-compatible with orbfe_keypoint");

namespace orbfe_detail {
inline void check(int rc, const orbfe_handle* h, const char* where)
{
    // The reference aborts the process on device errors (include/cuda/HelperCuda.h:44-50); the adaptor
    // throws instead so the caller can decide.
    if (rc != ORBFE_OK)
        throw std::runtime_error(std::string(where) + ": " + orbfe_status_string(rc) + " " + (h ? orbfe_last_error(h) : ""));
}
}  // namespace orbfe_detail

struct GrayImageView {  // stand-in for cv::cuda::HostMem when OpenCV is absent
    const uint8_t* data;
    int pitch;
};

class ORBextractor {
public:
    // include/ORBextractor.h:55-56
    ORBextractor(int nFeatures, int nFastFeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int imageWidth, int imageHeight, int deviceId = 0)
    {
        orbfe_params p{nFeatures, nFastFeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth, imageHeight, deviceId, 1};
        orbfe_detail::check(orbfe_create(&p, &h_), nullptr, "orbfe_create");
        const int n = nlevels;
        mvScaleFactor.resize(n);
        mvInvScaleFactor.resize(n);
        mvLevelSigma2.resize(n);
        mvInvLevelSigma2.resize(n);
        orbfe_detail::check(orbfe_get_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                                   mvInvLevelSigma2.data()), h_, "orbfe_get_scale_tables");
        cap_ = orbfe_max_keypoints(h_);
        width_ = imageWidth;
        height_ = imageHeight;
    }
    ~ORBextractor() { orbfe_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // include/ORBextractor.h:62 -- nullopt when no keypoint was found (src/ORBextractor.cc:494-496)
    std::optional<std::tuple<std::shared_ptr<std::vector<KeyPoint>>, std::vector<uint8_t>>> extractFeatures(const GrayImageView& im)
    {
        auto keys = std::make_shared<std::vector<KeyPoint>>(cap_);
        std::vector<uint8_t> desc((size_t)cap_ * ORBFE_DESC_BYTES);
        int n = 0;
        orbfe_detail::check(orbfe_extract(h_, im.data, im.pitch, reinterpret_cast<orbfe_keypoint*>(keys->data()), desc.data(), &n,
                                          nullptr), h_, "orbfe_extract");
        if (n == 0) return std::nullopt;
        keys->resize(n);
        desc.resize((size_t)n * ORBFE_DESC_BYTES);
        return {{keys, std::move(desc)}};
    }

#ifdef ORBFE_WITH_OPENCV
    // the exact reference signature
    std::optional<std::tuple<std::shared_ptr<std::vector<KeyPoint>>, cv::cuda::HostMem>> extractFeatures(const cv::cuda::HostMem& im_managed)
    {
        const cv::Mat im = im_managed.createMatHeader();
        auto r = extractFeatures(GrayImageView{im.data, (int)im.step});
        if (!r) return std::nullopt;
        auto& [keys, desc] = *r;
        cv::cuda::HostMem d((int)keys->size(), 32, CV_8UC1, cv::cuda::HostMem::AllocType::SHARED);
        cv::Mat dm = d.createMatHeader();
        for (int i = 0; i < dm.rows; i++) std::memcpy(dm.ptr(i), desc.data() + (size_t)i * 32, 32);
        return {{keys, d}};
    }
#endif

    // Upstream-style call operator (UZ-SLAMLab ORB_SLAM3: ORBextractor::operator()(image, mask, keypoints, descriptors,
    // vLappingArea)).  This fork has no such member (no referent) -- it is offered for callers
    // written against upstream and forwards to the same C ABI call.  Returns the number of keypoints.  Keypoints stay in
    // LEVEL pixels like everywhere in this fork (SPEC DECISION S6) unless toLevel0 is set, which multiplies pt by
    // mvScaleFactor[octave] as upstream does -- a clearly non-reference convenience.
    int operator()(const GrayImageView& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors, bool toLevel0 = false)
    {
        auto r = extractFeatures(image);
        keypoints.clear();
        descriptors.clear();
        if (!r) return 0;
        keypoints = *std::get<0>(*r);
        descriptors = std::move(std::get<1>(*r));
        if (toLevel0)
            for (auto& k : keypoints) {
                k.pt.x *= mvScaleFactor[k.octave];
                k.pt.y *= mvScaleFactor[k.octave];
            }
        return (int)keypoints.size();
    }

    // include/ORBextractor.h:64-92
    int GetLevels() { return orbfe_get_levels(h_); }
    float GetScaleFactor() { return orbfe_get_scale_factor(h_); }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid / mvBlurredImagePyramid (include/ORBextractor.h:94-95): copy of one level of the last frame
    std::vector<uint8_t> PyramidLevel(int level, bool blurred, int* w = nullptr, int* h = nullptr)
    {
        std::vector<int> lw(GetLevels()), lh(GetLevels());
        orbfe_detail::check(orbfe_get_level_info(h_, nullptr, lw.data(), lh.data()), h_, "orbfe_get_level_info");
        std::vector<uint8_t> out((size_t)lw[level] * lh[level]);
        orbfe_detail::check(orbfe_get_pyramid_level(h_, 0, level, blurred, out.data(), lw[level]), h_, "orbfe_get_pyramid_level");
        if (w) *w = lw[level];
        if (h) *h = lh[level];
        return out;
    }

    orbfe_handle* handle() { return h_; }
    int maxKeypoints() const { return cap_; }

private:
    orbfe_handle* h_ = nullptr;
    int cap_ = 0, width_ = 0, height_ = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// The image half of ORB_SLAM3::ImageGrabber (ros2_ws/src/mono-inertial/include/image_grabber.hpp:38-49,96-110): the
// undistortion maps are uploaded once (the node passes two cv::cuda::GpuMat built from
// cv::fisheye::initUndistortRectifyMap, mono_inertial_node.cpp:61-71), ConvertImageToGPU turns a BGR camera frame into
// the grey frame the tracker consumes.  extractFeatures() chains the extractor without the grey frame leaving the device.
class ImagePreparer {
public:
    ImagePreparer(ORBextractor& extractor, int srcWidth, int srcHeight, const float* map1, const float* map2, int width, int height)
        : h_(extractor.handle()), cap_(extractor.maxKeypoints()), w_(width), h2_(height)
    {
        orbfe_detail::check(orbfe_prep_create(h_, srcWidth, srcHeight, map1, map2, width, height, &p_), h_, "orbfe_prep_create");
    }
    ~ImagePreparer() { orbfe_prep_destroy(p_); }
    ImagePreparer(const ImagePreparer&) = delete;
    ImagePreparer& operator=(const ImagePreparer&) = delete;

    // image_grabber.hpp:96-110; grey: height rows of `width` bytes
    std::vector<uint8_t> ConvertImageToGPU(const uint8_t* bgr, int pitch)
    {
        std::vector<uint8_t> grey((size_t)w_ * h2_);
        orbfe_detail::check(orbfe_prepare_image(h_, p_, bgr, pitch, grey.data(), w_), h_, "orbfe_prepare_image");
        return grey;
    }

    // ConvertImageToGPU + ORBextractor::extractFeatures (include/ORBextractor.h:62); `greyOut` (optional) receives the frame
    std::optional<std::tuple<std::shared_ptr<std::vector<KeyPoint>>, std::vector<uint8_t>>> extractFeatures(
        const uint8_t* bgr, int pitch, std::vector<uint8_t>* greyOut = nullptr)
    {
        auto keys = std::make_shared<std::vector<KeyPoint>>(cap_);
        std::vector<uint8_t> desc((size_t)cap_ * ORBFE_DESC_BYTES);
        if (greyOut) greyOut->resize((size_t)w_ * h2_);
        int n = 0;
        orbfe_detail::check(orbfe_prepare_and_extract(h_, p_, bgr, pitch, reinterpret_cast<orbfe_keypoint*>(keys->data()), desc.data(),
                                                      &n, nullptr, greyOut ? greyOut->data() : nullptr, w_), h_,
                            "orbfe_prepare_and_extract");
        if (n == 0) return std::nullopt;
        keys->resize(n);
        desc.resize((size_t)n * ORBFE_DESC_BYTES);
        return {{keys, std::move(desc)}};
    }

#ifdef ORBFE_WITH_OPENCV
    // the reference signature: sensor image in, managed grey HostMem out
    cv::cuda::HostMem ConvertImageToGPU(const cv::Mat& cv_im)
    {
        cv::cuda::HostMem grey(h2_, w_, CV_8UC1, cv::cuda::HostMem::AllocType::SHARED);
        cv::Mat g = grey.createMatHeader();
        orbfe_detail::check(orbfe_prepare_image(h_, p_, cv_im.data, (int)cv_im.step, g.data, (int)g.step), h_, "orbfe_prepare_image");
        return grey;
    }
#endif

private:
    orbfe_handle* h_ = nullptr;
    orbfe_prep* p_ = nullptr;
    int cap_ = 0, w_ = 0, h2_ = 0;
};

// All-static like the reference (include/ORBmatcher.h:40-75); the GPU handle is the extractor's.
class ORBmatcher {
public:
    static constexpr int TH_LOW = ORBFE_TH_LOW;
    static constexpr int TH_HIGH = ORBFE_TH_HIGH;
    static constexpr size_t HISTO_LENGTH = ORBFE_HISTO_LENGTH;

    // src/ORBmatcher.cc:1375-1391 (rows of the descriptor matrices)
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return orbfe_hamming(a, b); }

    // src/ORBmatcher.cc:31-123.  `F` needs: mNumKeypoints, mvKeysUn (shared_ptr<vector<KeyPoint>>), descriptor
    // rows via descPtr(F), mvpMapPoints (vector<shared_ptr<MapPoint>>), mvScaleFactors, the static grid members
    // mnMinX, mnMinY, mfGridElementWidthInv/HeightInv, getFrameGridCols()/Rows().  `MapPoint` needs: mbTrackInView,
    // mTrackDepth, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX/Y, GetDescriptor-like descPtr(), Observations().
    template <class FramePtr, class MapPointPtr, class DescOfFrame, class DescOfMapPoint>
    static int SearchByProjection(orbfe_handle* h, FramePtr F, const std::vector<MapPointPtr>& vpMapPoints, const float th,
                                  const bool bFarPoints, const float thFarPoints, const float nnRatio,
                                  const bool /*checkOrientation: unused by the reference, :31*/, DescOfFrame frameDesc,
                                  DescOfMapPoint mpDesc)
    {
        const int n = F->mNumKeypoints;
        const int M = (int)vpMapPoints.size();
        std::vector<orbfe_map_point> mps(M);
        std::vector<uint8_t> mpd((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            const auto& p = vpMapPoints[i];
            // mbTrackInViewR only exists for the stereo-fisheye case; mono: inView == mbTrackInView (:40-41,49)
            mps[i] = orbfe_map_point{p->mTrackProjX, p->mTrackProjY, p->mTrackViewCos, p->mTrackDepth, p->mnTrackScaleLevel,
                                     p->mbTrackInView ? 1 : 0, p->isBad() ? 1 : 0, p->Observations()};
            std::memcpy(&mpd[(size_t)i * 32], mpDesc(p), 32);
        }
        std::vector<int> initObs(n, -1);
        for (int i = 0; i < n; i++)
            if (F->mvpMapPoints[i]) initObs[i] = F->mvpMapPoints[i]->Observations();
        orbfe_frame_view fv{n, reinterpret_cast<const orbfe_keypoint*>(F->mvKeysUn->data()), frameDesc(F),
                            F->getFrameGridCols(), F->getFrameGridRows(), F->mnMinX, F->mnMinY, F->mfGridElementWidthInv,
                            F->mfGridElementHeightInv, (int)F->mvScaleFactors.size(), F->mvScaleFactors.data()};
        std::vector<int> match(n > 0 ? n : 1);
        int nmatches = 0;
        orbfe_detail::check(orbfe_match_projection(h, &fv, M, mps.data(), mpd.data(), initObs.data(), th, bFarPoints, thFarPoints,
                                                   nnRatio, match.data(), &nmatches), h, "orbfe_match_projection");
        for (int i = 0; i < n; i++)
            if (match[i] >= 0) F->mvpMapPoints[i] = vpMapPoints[match[i]];  // :113
        return nmatches;
    }

    // src/ORBmatcher.cc:329-439 (include/ORBmatcher.h:58): returns {nmatches, vnMatches12}
    template <class FramePtr, class DescOfFrame>
    static std::pair<int, std::vector<int>> SearchForInitialization(orbfe_handle* h, FramePtr F1, FramePtr F2, int windowSize,
                                                                    const float nnRatio, const bool checkOrientation,
                                                                    DescOfFrame frameDesc)
    {
        auto view = [&](const FramePtr& F) {
            return orbfe_frame_view{(int)F->mvKeysUn->size(), reinterpret_cast<const orbfe_keypoint*>(F->mvKeysUn->data()),
                                    frameDesc(F), F->getFrameGridCols(), F->getFrameGridRows(), F->mnMinX, F->mnMinY,
                                    F->mfGridElementWidthInv, F->mfGridElementHeightInv, (int)F->mvScaleFactors.size(),
                                    F->mvScaleFactors.data()};
        };
        const orbfe_frame_view v1 = view(F1), v2 = view(F2);
        std::vector<int> m12(v1.n > 0 ? v1.n : 1, -1);
        int nmatches = 0;
        orbfe_detail::check(orbfe_match_initialization(h, &v1, &v2, windowSize, nnRatio, checkOrientation, m12.data(), &nmatches),
                            h, "orbfe_match_initialization");
        m12.resize(v1.n);
        return {nmatches, m12};
    }

    // src/ORBmatcher.cc:133-327.  FeatureVector = std::map<NodeId, std::vector<unsigned>> (DBoW2).  The merge-walk
    // over the two maps (:161-163,289-301) happens here on the host and is handed over as CSR groups.
    template <class KeyFramePtr, class FramePtr, class MapPointPtr, class FeatureVector, class DescOfKF, class DescOfFrame>
    static int SearchByBoW(orbfe_handle* h, KeyFramePtr pKF, FramePtr F, std::vector<MapPointPtr>& vpMapPointMatches,
                           const FeatureVector& vFeatVecKF, const FeatureVector& vFeatVecF, const float nnRatio,
                           const bool checkOrientation, DescOfKF kfDesc, DescOfFrame frameDesc)
    {
        const auto vpMapPointsKF = pKF->GetMapPointMatches();
        const int nKF = (int)vpMapPointsKF.size();
        const int nF = F->mNumKeypoints;
        vpMapPointMatches.assign(nF, MapPointPtr());
        std::vector<int> kfOff{0}, kfIdx, fOff{0}, fIdx;
        auto KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
        auto Fit = vFeatVecF.begin(), Fend = vFeatVecF.end();
        while (KFit != KFend && Fit != Fend) {
            if (KFit->first == Fit->first) {
                for (unsigned v : KFit->second) kfIdx.push_back((int)v);
                for (unsigned v : Fit->second) fIdx.push_back((int)v);
                kfOff.push_back((int)kfIdx.size());
                fOff.push_back((int)fIdx.size());
                ++KFit;
                ++Fit;
            } else if (KFit->first < Fit->first) {
                KFit = vFeatVecKF.lower_bound(Fit->first);
            } else {
                Fit = vFeatVecF.lower_bound(KFit->first);
            }
        }
        std::vector<uint8_t> hasMP(nKF > 0 ? nKF : 1, 0);
        std::vector<float> kfAngle(nKF > 0 ? nKF : 1), fAngle(nF > 0 ? nF : 1);
        for (int i = 0; i < nKF; i++) {
            hasMP[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :172-176
            kfAngle[i] = (*pKF->mvKeysUn)[i].angle;
        }
        for (int i = 0; i < nF; i++) fAngle[i] = (*F->mvKeysUn)[i].angle;
        std::vector<int> match(nF > 0 ? nF : 1);
        int nmatches = 0;
        // F->Nleft != -1 (two-camera rig): features >= Nleft keep their own best / second best (:205-233,263-286)
        orbfe_detail::check(orbfe_match_bow_rig(h, (int)kfOff.size() - 1, kfOff.data(), kfIdx.data(), fOff.data(), fIdx.data(), nKF,
                                                kfDesc(pKF), kfAngle.data(), hasMP.data(), nF, frameDesc(F), fAngle.data(), F->Nleft,
                                                nnRatio, checkOrientation, match.data(), &nmatches), h, "orbfe_match_bow_rig");
        for (int j = 0; j < nF; j++)
            if (match[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[match[j]];  // :241
        return nmatches;
    }
};

// ORBmatcher::SearchForTriangulation / ORBmatcher::Fuse (src/ORBmatcher.cc:441-676, 678-851): the data-parallel
// search runs on the GPU, the std::map merge-walk and the map-point graph edits stay on the host.
struct KeyFrameMatcher {
    // src/ORBmatcher.cc:441-676.  `KF` needs: N, mvKeysUn, mFeatVec (std::map<NodeId, vector<unsigned>>), GetMapPoint(i),
    // mvuRight, mvScaleFactors; descriptors through descOf(pKF) (N x 32 bytes).  F12 / ep are computed by the caller
    // with the reference's own expressions (Pinhole.cpp:106-109, ORBmatcher.cc:451-454) and passed in `prm`; for
    // KannalaBrandt8 key frames the caller also fills prm's camera block (T12 of :466-468, both parameter vectors,
    // mvLevelSigma2 of pKF1, whether pKF1->mpCamera2 is set) and the test becomes KannalaBrandt8::epipolarConstrain.
    template <class KeyFramePtr, class DescOf>
    static int SearchForTriangulation(orbfe_handle* h, KeyFramePtr pKF1, KeyFramePtr pKF2, const orbfe_tri_params& prm,
                                      std::vector<std::pair<size_t, size_t>>& vMatchedPairs, DescOf descOf)
    {
        std::vector<int> off1{0}, idx1, off2{0}, idx2;
        auto f1 = pKF1->mFeatVec.begin(), f2 = pKF2->mFeatVec.begin();
        while (f1 != pKF1->mFeatVec.end() && f2 != pKF2->mFeatVec.end()) {  // :489-617
            if (f1->first == f2->first) {
                idx1.insert(idx1.end(), f1->second.begin(), f1->second.end());
                idx2.insert(idx2.end(), f2->second.begin(), f2->second.end());
                off1.push_back((int)idx1.size());
                off2.push_back((int)idx2.size());
                ++f1;
                ++f2;
            } else if (f1->first < f2->first) {
                f1 = pKF1->mFeatVec.lower_bound(f2->first);
            } else {
                f2 = pKF2->mFeatVec.lower_bound(f1->first);
            }
        }
        const int n1 = pKF1->N, n2 = pKF2->N;
        std::vector<uint8_t> has1(n1 > 0 ? n1 : 1), has2(n2 > 0 ? n2 : 1), st1(n1 > 0 ? n1 : 1), st2(n2 > 0 ? n2 : 1);
        for (int i = 0; i < n1; i++) {
            has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;
            st1[i] = pKF1->mvuRight[i] >= 0 ? 1 : 0;
        }
        for (int i = 0; i < n2; i++) {
            has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;
            st2[i] = pKF2->mvuRight[i] >= 0 ? 1 : 0;
        }
        std::vector<int> m12(n1 > 0 ? n1 : 1);
        int nmatches = 0;
        orbfe_detail::check(orbfe_match_triangulation(
            h, (int)off1.size() - 1, off1.data(), idx1.data(), off2.data(), idx2.data(), n1,
            reinterpret_cast<const orbfe_keypoint*>(pKF1->mvKeysUn->data()), descOf(pKF1), has1.data(), st1.data(), n2,
            reinterpret_cast<const orbfe_keypoint*>(pKF2->mvKeysUn->data()), descOf(pKF2), has2.data(), st2.data(),
            pKF2->mvScaleFactors.data(), (int)pKF2->mvScaleFactors.size(), &prm, m12.data(), &nmatches), h,
            "orbfe_match_triangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches);
        for (int i = 0; i < n1; i++)
            if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);  // :664-673
        return nmatches;
    }

    // src/ORBmatcher.cc:678-851.  bRight (the key frame of a two-camera rig, src/LocalMapping.cc:824,854): `frustum`
    // carries GetRightPose / GetRightTranslationInverse / mpCamera2, kfView describes the NLeft left features with
    // desc = all of mDescriptors, and the indices written to the graph are idx + NLeft (:820).
    // The search result of every map point is computed first; the
    // loop below then replays :699-849 in list order with the CURRENT graph state (isBad / IsInKeyFrame may have
    // changed through an earlier Replace), exactly like the reference.  `frustum` carries the key frame's pose and
    // pinhole intrinsics (GetPose / GetTranslationInverse), mbf, mfLogScaleFactor, mnScaleLevels, image bounds.
    template <class KeyFramePtr, class MapPointPtr, class DescOfKF, class DescOfMP>
    static int Fuse(orbfe_handle* h, KeyFramePtr pKF, const std::vector<MapPointPtr>& vpMapPoints, const float th,
                    const orbfe_frustum& frustum, const orbfe_frame_view& kfView, DescOfKF, DescOfMP descOfMP,
                    const bool bRight = false)
    {
        const int M = (int)vpMapPoints.size();
        std::vector<orbfe_world_point> pts(M > 0 ? M : 1);
        std::vector<uint8_t> mpd((size_t)(M > 0 ? M : 1) * 32);
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpMapPoints[i];
            orbfe_world_point& p = pts[i];
            p = orbfe_world_point{0, 0, 0, 0, 0, 0, 0, 1};
            if (!pMP) continue;
            const auto P = pMP->GetWorldPos();
            p = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                  pMP->Observations(), pMP->IsInKeyFrame(pKF) ? 1 : 0};
            std::memcpy(&mpd[(size_t)i * 32], descOfMP(pMP), 32);
        }
        std::vector<int> bestIdx(M > 0 ? M : 1), bestDist(M > 0 ? M : 1);
        if (bRight)
            orbfe_detail::check(orbfe_fuse_search_right(h, &kfView, pKF->NRight, pKF->mvInvLevelSigma2.data(), pKF->mvuRight.data(),
                                                        &frustum, th, M, pts.data(), mpd.data(), bestIdx.data(), bestDist.data()),
                                h, "orbfe_fuse_search_right");
        else
            orbfe_detail::check(orbfe_fuse_search(h, &kfView, pKF->mvInvLevelSigma2.data(), pKF->mvuRight.data(), &frustum, th, M,
                                                  pts.data(), mpd.data(), bestIdx.data(), bestDist.data()), h, "orbfe_fuse_search");
        int nFused = 0;
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // :701-720, re-evaluated in order
            if (bestDist[i] > ORBFE_TH_LOW) continue;                       // :829
            auto pMPinKF = pKF->GetMapPoint(bestIdx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx[i]);
                pKF->AddMapPoint(pMP, bestIdx[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // src/ORBmatcher.cc:864-975: Fuse(pKF, Scw, vpPoints, th, vpReplacePoint).  `frustum` carries the decomposition
    // the reference makes at :867-868 (rcw = Scw.rotationMatrix(), tcw = Scw.translation() / Scw.scale(), twc =
    // Tcw.inverse().translation()) plus the key frame's camera, bounds and scale pyramid.  spAlreadyFound is taken
    // once on entry like the reference (:871); the replay loop applies :955-971 in list order on the live graph.
    template <class KeyFramePtr, class MapPointPtr, class DescOfMP>
    static int Fuse(orbfe_handle* h, KeyFramePtr pKF, const orbfe_frustum& frustum, const std::vector<MapPointPtr>& vpPoints,
                    float th, std::vector<MapPointPtr>& vpReplacePoint, const orbfe_frame_view& kfView, DescOfMP descOfMP)
    {
        const int M = (int)vpPoints.size();
        const auto spAlreadyFound = pKF->GetMapPoints();
        std::vector<orbfe_world_point> pts(M > 0 ? M : 1);
        std::vector<uint8_t> mpd((size_t)(M > 0 ? M : 1) * 32);
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpPoints[i];
            const auto P = pMP->GetWorldPos();
            pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                       pMP->Observations(), spAlreadyFound.count(pMP) ? 1 : 0};
            std::memcpy(&mpd[(size_t)i * 32], descOfMP(pMP), 32);
        }
        std::vector<int> bestIdx(M > 0 ? M : 1), bestDist(M > 0 ? M : 1);
        orbfe_detail::check(orbfe_fuse_search_sim3(h, &kfView, &frustum, th, M, pts.data(), mpd.data(), bestIdx.data(),
                                                   bestDist.data()), h, "orbfe_fuse_search_sim3");
        int nFused = 0;
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpPoints[i];
            if (pts[i].bad || pts[i].skip) continue;       // :882 (nothing in this loop changes isBad of a later point)
            if (bestDist[i] > ORBFE_TH_LOW) continue;       // :955
            auto pMPinKF = pKF->GetMapPoint(bestIdx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx[i]);
                pKF->AddMapPoint(pMP, bestIdx[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // src/ORBmatcher.cc:977-1200.  dir12 / dir21 carry the poses and the similarity (T1w + S21, T2w + S12), pKF1's
    // intrinsics and the target key frame's bounds / pyramid; `KF` needs GetMapPointMatches() and the map points
    // GetIndexInKeyFrame(pKF) (a tuple whose first element is the left index, as in the reference, :1007).
    template <class KeyFramePtr, class MapPointPtr, class DescOfMP>
    static int SearchBySim3(orbfe_handle* h, KeyFramePtr pKF1, KeyFramePtr pKF2, std::vector<MapPointPtr>& vpMatches12,
                            const orbfe_sim3_view& dir12, const orbfe_sim3_view& dir21, const float th,
                            const orbfe_frame_view& view1, const orbfe_frame_view& view2, DescOfMP descOfMP)
    {
        const auto vpMapPoints1 = pKF1->GetMapPointMatches();
        const auto vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<uint8_t> already1(N1 > 0 ? N1 : 1, 0), already2(N2 > 0 ? N2 : 1, 0);
        for (int i = 0; i < N1; i++) {  // :1001-1012
            const auto& pMP = vpMatches12[i];
            if (!pMP) continue;
            already1[i] = 1;
            const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
            if (idx2 >= 0 && idx2 < N2) already2[idx2] = 1;
        }
        auto pack = [&](const std::vector<MapPointPtr>& v, const std::vector<uint8_t>& already, std::vector<orbfe_world_point>& pts,
                        std::vector<uint8_t>& d) {
            const int n = (int)v.size();
            pts.assign(n > 0 ? n : 1, orbfe_world_point{0, 0, 0, 0, 0, 0, 0, 1});
            d.assign((size_t)(n > 0 ? n : 1) * 32, 0);
            for (int i = 0; i < n; i++) {
                const auto& pMP = v[i];
                if (!pMP || already[i]) continue;
                const auto P = pMP->GetWorldPos();
                pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                           pMP->Observations(), 0};
                std::memcpy(&d[(size_t)i * 32], descOfMP(pMP), 32);
            }
        };
        std::vector<orbfe_world_point> p1, p2;
        std::vector<uint8_t> d1, d2;
        pack(vpMapPoints1, already1, p1, d1);
        pack(vpMapPoints2, already2, p2, d2);
        std::vector<int> m12(N1 > 0 ? N1 : 1, -1);
        int nFound = 0;
        orbfe_detail::check(orbfe_search_by_sim3(h, &view1, &view2, &dir12, &dir21, p1.data(), d1.data(), p2.data(), d2.data(),
                                                 th, m12.data(), &nFound), h, "orbfe_search_by_sim3");
        for (int i1 = 0; i1 < N1; i1++)
            if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];  // :1184
        return nFound;
    }

    // src/ORBmatcher.cc:1202-1326: the relocalisation overload SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th,
    // checkOrientation).  `frustum` = the current frame's pose (GetPose / Tcw.inverse().translation()), bounds, camera
    // and pyramid; `frameView` its keypoints, descriptors and grid.
    template <class FramePtr, class KeyFramePtr, class MapPointSet, class DescOfMP>
    static int SearchByProjection(orbfe_handle* h, FramePtr CurrentFrame, KeyFramePtr pKF, const MapPointSet& sAlreadyFound,
                                  const float th, const bool checkOrientation, const orbfe_frustum& frustum,
                                  const orbfe_frame_view& frameView, DescOfMP descOfMP)
    {
        const auto vpMPs = pKF->GetMapPointMatches();
        const int M = (int)vpMPs.size(), n = frameView.n;
        std::vector<orbfe_world_point> pts(M > 0 ? M : 1, orbfe_world_point{0, 0, 0, 0, 0, 0, 0, 1});
        std::vector<uint8_t> mpd((size_t)(M > 0 ? M : 1) * 32);
        std::vector<float> ang(M > 0 ? M : 1);
        for (int i = 0; i < M; i++) {
            ang[i] = (*pKF->mvKeysUn)[i].angle;
            const auto& pMP = vpMPs[i];
            if (!pMP) continue;
            const auto P = pMP->GetWorldPos();
            pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                       pMP->Observations(), sAlreadyFound.count(pMP) ? 1 : 0};
            std::memcpy(&mpd[(size_t)i * 32], descOfMP(pMP), 32);
        }
        std::vector<uint8_t> has(n > 0 ? n : 1);
        for (int i = 0; i < n; i++) has[i] = CurrentFrame->mvpMapPoints[i] ? 1 : 0;
        std::vector<int> match(n > 0 ? n : 1, -1);
        int nmatches = 0;
        orbfe_detail::check(orbfe_match_projection_keyframe(h, &frameView, &frustum, M, pts.data(), mpd.data(), ang.data(),
                                                            has.data(), th, checkOrientation ? 1 : 0, match.data(), &nmatches),
                            h, "orbfe_match_projection_keyframe");
        for (int i = 0; i < n; i++)
            if (match[i] >= 0) CurrentFrame->mvpMapPoints[i] = vpMPs[match[i]];  // :1284 (entries the histogram removed stay as they were: empty)
        return nmatches;
    }
};

// A key frame's immutable matcher inputs resident on the GPU (orbfe_keyframe_*): create it where the reference constructs
// the KeyFrame (after ComputeBoW, src/LocalMapping.cc:258), keep it for the key frame's lifetime.  `KF` needs N, mvKeysUn,
// mFeatVec, mvuRight, mvScaleFactors; descriptors through descOf(pKF).
class ResidentKeyFrame {
public:
    template <class KeyFramePtr, class DescOf>
    ResidentKeyFrame(orbfe_handle* h, KeyFramePtr pKF, DescOf descOf)
    {
        const int n = pKF->N;
        std::vector<int> node(n > 0 ? n : 1, -1);
        for (const auto& e : pKF->mFeatVec)
            for (unsigned i : e.second) node[i] = (int)e.first;  // DBoW2::FeatureVector: node -> features, ascending index
        std::vector<uint8_t> st(n > 0 ? n : 1, 0);
        bool anyStereo = false;
        for (int i = 0; i < n && i < (int)pKF->mvuRight.size(); i++) {
            st[i] = pKF->mvuRight[i] >= 0 ? 1 : 0;
            anyStereo = anyStereo || st[i];
        }
        orbfe_detail::check(orbfe_keyframe_create(h, n, reinterpret_cast<const orbfe_keypoint*>(pKF->mvKeysUn->data()), descOf(pKF),
                                                  node.data(), anyStereo ? st.data() : nullptr, pKF->mvScaleFactors.data(),
                                                  (int)pKF->mvScaleFactors.size(), &kf_), h, "orbfe_keyframe_create");
    }
    ~ResidentKeyFrame() { orbfe_keyframe_destroy(kf_); }
    ResidentKeyFrame(const ResidentKeyFrame&) = delete;
    ResidentKeyFrame& operator=(const ResidentKeyFrame&) = delete;
    const orbfe_keyframe* get() const { return kf_; }

    // mGrid of the key frame (src/KeyFrame.cc:33-80 copies the geometry from the Frame), mvInvLevelSigma2 and mvuRight: what the
    // projection searches INTO this key frame read.  Call it once, right after the constructor; KeyFrameMatcher::FuseResident
    // needs it.  `KF` needs mnGridCols, mnGridRows, mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv.
    template <class KeyFramePtr>
    void SetGrid(orbfe_handle* h, KeyFramePtr pKF)
    {
        orbfe_detail::check(orbfe_keyframe_set_grid(h, kf_, pKF->mnGridCols, pKF->mnGridRows, pKF->mnMinX, pKF->mnMinY,
                                                    pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, pKF->mvInvLevelSigma2.data(),
                                                    pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data()), h, "orbfe_keyframe_set_grid");
    }

private:
    orbfe_keyframe* kf_ = nullptr;
};

// Map points resident on the GPU (orbfe_map_*): what isInFrustum, SearchByProjection and Fuse read from a MapPoint, indexed by
// an id the caller assigns (MapPoint::mnId modulo the capacity, or a slot from a free list).  Update(pMP, id) where the
// reference changes a point: the constructors, SetWorldPos, UpdateNormalAndDepth (mfMin/MaxDistance),
// ComputeDistinctiveDescriptors, SetBadFlag / Replace (src/MapPoint.cc).
class ResidentMap {
public:
    ResidentMap(orbfe_handle* h, int capacity) : h_(h)
    {
        orbfe_detail::check(orbfe_map_create(h, capacity, &m_), h, "orbfe_map_create");
    }
    ~ResidentMap() { orbfe_map_destroy(m_); }
    ResidentMap(const ResidentMap&) = delete;
    ResidentMap& operator=(const ResidentMap&) = delete;
    template <class MapPointPtr, class DescOfMP>
    void Update(const std::vector<MapPointPtr>& vpMPs, const std::vector<int>& ids, DescOfMP descOfMP)
    {
        const int n = (int)vpMPs.size();
        std::vector<orbfe_world_point> pts(n > 0 ? n : 1);
        std::vector<uint8_t> d((size_t)(n > 0 ? n : 1) * 32);
        for (int i = 0; i < n; i++) {
            const auto& pMP = vpMPs[i];
            const auto P = pMP->GetWorldPos();
            pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0, pMP->Observations(), 0};
            std::memcpy(&d[(size_t)i * 32], descOfMP(pMP), 32);
        }
        orbfe_detail::check(orbfe_map_update(h_, m_, n, ids.data(), pts.data(), d.data()), h_, "orbfe_map_update");
    }
    const orbfe_map* get() const { return m_; }

private:
    orbfe_handle* h_;
    orbfe_map* m_ = nullptr;
};

// ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:678-851) for LocalMapping::SearchInNeighbors
// (src/LocalMapping.cc:764-860, calls at :822,:852) on RESIDENT data: the target key frame is a ResidentKeyFrame with SetGrid
// done, the map points are entries of a ResidentMap (ids[i] = the entry of vpMapPoints[i]); a frustum and 4 bytes per map point
// go up instead of the key frame and 64 bytes per point.  The replay loop is KeyFrameMatcher::Fuse's: :699-849 in list order
// on the live graph.  The caller pushes what the loop changed (Replace: descriptor / observations of the surviving point) to
// the ResidentMap before the next neighbour's call, as it would call ComputeDistinctiveDescriptors in the reference (:836-848).
struct ResidentFuse {
    template <class KeyFramePtr, class MapPointPtr>
    static int Fuse(orbfe_handle* h, KeyFramePtr pKF, const ResidentKeyFrame& resident, const ResidentMap& map,
                    const std::vector<MapPointPtr>& vpMapPoints, const std::vector<int>& ids, const float th,
                    const orbfe_frustum& frustum)
    {
        const int M = (int)vpMapPoints.size();
        std::vector<int> call(M > 0 ? M : 1), bestIdx(M > 0 ? M : 1), bestDist(M > 0 ? M : 1);
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpMapPoints[i];
            // "!pMP || pMP->IsInKeyFrame(pKF)" travels with the id (~id); isBad() is the resident entry's own flag (:706-721)
            call[i] = (!pMP || pMP->IsInKeyFrame(pKF)) ? ~ids[i] : ids[i];
        }
        orbfe_detail::check(orbfe_fuse_search_keyframe(h, resident.get(), map.get(), M, call.data(), &frustum, th, bestIdx.data(),
                                                       bestDist.data()), h, "orbfe_fuse_search_keyframe");
        int nFused = 0;
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // :701-720, re-evaluated in order
            if (bestDist[i] > ORBFE_TH_LOW) continue;                       // :829
            auto pMPinKF = pKF->GetMapPoint(bestIdx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx[i]);
                pKF->AddMapPoint(pMP, bestIdx[i]);
            }
            nFused++;
        }
        return nFused;
    }
};

// The SearchForTriangulation loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:455-488) with ONE GPU launch for
// all neighbours.  Construct it before the loop (prm[k] = the F12 / epipole / camera block of the pair (pKF1, neighbour
// k), computed with the reference's own expressions); inside the loop, where the reference calls
// ORBmatcher::SearchForTriangulation(mpCurrentKeyFrame, pKF2, vMatchedIndices, ...), call Matches(k, pKF1, ...): it takes
// the map points pKF1 has NOW (those the loop created for earlier neighbours included, :506-509) and returns exactly what
// the k-th sequential call would return.
class TriangulationBatch {
public:
    template <class KeyFramePtr>
    TriangulationBatch(orbfe_handle* h, KeyFramePtr pKF1, const ResidentKeyFrame& r1, const std::vector<KeyFramePtr>& vpNeighKFs,
                       const std::vector<const ResidentKeyFrame*>& r2, const std::vector<orbfe_tri_params>& prm)
        : n1_(pKF1->N), check_(prm.empty() ? 1 : prm[0].check_orientation)
    {
        const int K = (int)vpNeighKFs.size();
        std::vector<uint8_t> has1(n1_ > 0 ? n1_ : 1);
        for (int i = 0; i < n1_; i++) has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;
        std::vector<std::vector<uint8_t>> has2(K);
        std::vector<const uint8_t*> has2p(K);
        std::vector<const orbfe_keyframe*> kf2(K);
        for (int k = 0; k < K; k++) {
            const int n2 = vpNeighKFs[k]->N;
            has2[k].resize(n2 > 0 ? n2 : 1);
            for (int i = 0; i < n2; i++) has2[k][i] = vpNeighKFs[k]->GetMapPoint(i) ? 1 : 0;
            has2p[k] = has2[k].data();
            kf2[k] = r2[k]->get();
        }
        raw_.assign((size_t)(K > 0 ? K : 1) * (n1_ > 0 ? n1_ : 1), -1);
        bin_.assign(raw_.size(), 0);
        orbfe_detail::check(orbfe_match_triangulation_batch(h, r1.get(), has1.data(), K, kf2.data(), has2p.data(), prm.data(), raw_.data(),
                                                            bin_.data()), h, "orbfe_match_triangulation_batch");
    }

    template <class KeyFramePtr>
    int Matches(int k, KeyFramePtr pKF1, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) const
    {
        std::vector<uint8_t> now(n1_ > 0 ? n1_ : 1);
        for (int i = 0; i < n1_; i++) now[i] = pKF1->GetMapPoint(i) ? 1 : 0;
        std::vector<int> m12(n1_ > 0 ? n1_ : 1);
        int nmatches = 0;
        orbfe_detail::check(orbfe_triangulation_select(n1_, raw_.data() + (size_t)k * n1_, bin_.data() + (size_t)k * n1_, now.data(), check_,
                                                       m12.data(), &nmatches), nullptr, "orbfe_triangulation_select");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches);
        for (int i = 0; i < n1_; i++)
            if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);  // src/ORBmatcher.cc:664-673
        return nmatches;
    }

private:
    int n1_, check_;
    std::vector<int> raw_;
    std::vector<uint8_t> bin_;
};

// The isInFrustum loop of Tracking::SearchLocalPoints (src/Tracking.cc:1059-1077) for all local map points in one
// launch.  `frustum` carries what Frame::isInFrustum reads from the frame (GetRcw / GetTcw / GetTwc, image bounds,
// pinhole intrinsics, mbf, mfLogScaleFactor, mnScaleLevels); `MapPoint` needs GetWorldPos() (indexable [0..2]),
// mfMinDistance / mfMaxDistance (the RAW values: the public getters return them scaled by 0.9 / 1.1), isBad(),
// Observations(), mnLastFrameSeen, and the mTrack* fields the reference writes.  Returns nToMatch.
struct LocalPointProjector {
    template <class MapPointPtr>
    static int ProjectLocalMapPoints(orbfe_handle* h, const orbfe_frustum& frustum, long unsigned int currentFrameId,
                                     const std::vector<MapPointPtr>& vpLocalMapPoints)
    {
        const int n = (int)vpLocalMapPoints.size();
        std::vector<orbfe_world_point> pts(n > 0 ? n : 1);
        for (int i = 0; i < n; i++) {
            const auto& pMP = vpLocalMapPoints[i];
            const auto P = pMP->GetWorldPos();
            pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                       pMP->Observations(), pMP->mnLastFrameSeen == currentFrameId ? 1 : 0};
        }
        std::vector<orbfe_map_point> out(n > 0 ? n : 1);
        std::vector<float> xr(n > 0 ? n : 1);
        orbfe_detail::check(orbfe_project_map_points(h, &frustum, n, pts.data(), out.data(), xr.data()), h,
                            "orbfe_project_map_points");
        int nToMatch = 0;
        for (int i = 0; i < n; i++) {
            if (pts[i].skip || pts[i].bad) continue;  // :1066-1069: untouched
            const auto& pMP = vpLocalMapPoints[i];
            pMP->mbTrackInView = out[i].in_view != 0;  // src/Frame.cc:274-276
            pMP->mTrackProjX = out[i].proj_x;
            pMP->mTrackProjY = out[i].proj_y;
            if (out[i].in_view) {  // :319-328
                pMP->mTrackProjXR = xr[i];
                pMP->mTrackDepth = out[i].track_depth;
                pMP->mnTrackScaleLevel = out[i].level;
                pMP->mTrackViewCos = out[i].view_cos;
                nToMatch++;  // the caller still does pMP->IncreaseVisible() (:1073)
            }
        }
        return nToMatch;
    }
};

// The tracking thread's per-frame chain as ONE submission (orbfe_track_frame): what Tracking::GrabImageMonocular does with
// a frame once the IMU is initialised -- Frame::Frame -> ExtractORB (src/Tracking.cc:152-173, src/Frame.cc:178-189), then
// Track() -> TrackWithMotionModel (IMU prediction only, :908-923) -> TrackLocalMap -> SearchLocalPoints: the isInFrustum
// loop (:1059-1077) and ORBmatcher::SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, false, thFarPoints, nnRatio)
// (:1108-1115).  The caller builds `frustum` from the PREDICTED pose (it does not depend on the frame's features) and
// passes the local map of UpdateLocalMap (:930).  On return `F` carries the fresh keypoints and descriptors and the
// matches in mvpMapPoints, the map points carry their mTrack* fields (LocalPointProjector's contract), *nToMatch is the
// reference's counter.  `F` needs: mNumKeypoints, mvKeysUn, mvpMapPoints, the grid statics (as SearchByProjection above);
// the descriptor rows go to the frame through setDesc(F, rows, n) (cv::cuda::HostMem in the reference, a vector here).
// Returns the match count, or -1 when the frame has no keypoints (the reference returns before Track(), :158-159).
struct FrameTracker {
    template <class FramePtr, class MapPointPtr, class DescOfMapPoint, class SetFrameDesc>
    static int ExtractAndSearchLocalPoints(ORBextractor& extractor, const GrayImageView& im, FramePtr F, const orbfe_frustum& frustum,
                                           long unsigned int currentFrameId, const std::vector<MapPointPtr>& vpLocalMapPoints,
                                           const float th, const bool bFarPoints, const float thFarPoints, const float nnRatio,
                                           DescOfMapPoint mpDesc, SetFrameDesc setDesc, int* nToMatch = nullptr)
    {
        orbfe_handle* h = extractor.handle();
        const int cap = extractor.maxKeypoints();
        const int M = (int)vpLocalMapPoints.size();
        std::vector<orbfe_world_point> pts(M > 0 ? M : 1);
        std::vector<uint8_t> mpd((size_t)(M > 0 ? M : 1) * 32);
        for (int i = 0; i < M; i++) {
            const auto& pMP = vpLocalMapPoints[i];
            const auto P = pMP->GetWorldPos();
            pts[i] = orbfe_world_point{P[0], P[1], P[2], pMP->mfMinDistance, pMP->mfMaxDistance, pMP->isBad() ? 1 : 0,
                                       pMP->Observations(), pMP->mnLastFrameSeen == currentFrameId ? 1 : 0};
            std::memcpy(&mpd[(size_t)i * 32], mpDesc(pMP), 32);
        }
        orbfe_track_params tp = ORBFE_TRACK_PARAMS_INIT;
        tp.grid_cols = F->getFrameGridCols();
        tp.grid_rows = F->getFrameGridRows();
        tp.min_x = F->mnMinX;
        tp.min_y = F->mnMinY;
        tp.grid_inv_w = F->mfGridElementWidthInv;
        tp.grid_inv_h = F->mfGridElementHeightInv;
        tp.th = th;
        tp.nn_ratio = nnRatio;
        tp.far_points = bFarPoints ? 1 : 0;
        tp.th_far_points = thFarPoints;
        auto keys = std::make_shared<std::vector<KeyPoint>>(cap);
        std::vector<uint8_t> desc((size_t)cap * ORBFE_DESC_BYTES);
        std::vector<int> match(cap);
        std::vector<orbfe_map_point> out(M > 0 ? M : 1);
        std::vector<float> xr(M > 0 ? M : 1);
        int n = 0, nmatches = 0;
        orbfe_detail::check(orbfe_track_frame(h, im.data, im.pitch, &frustum, &tp, M, pts.data(), mpd.data(),
                                              reinterpret_cast<orbfe_keypoint*>(keys->data()), desc.data(), &n, nullptr, out.data(),
                                              xr.data(), match.data(), &nmatches), h, "orbfe_track_frame");
        keys->resize(n);
        F->mNumKeypoints = n;
        F->mvKeysUn = keys;
        setDesc(F, desc.data(), n);
        F->mvpMapPoints.assign(n, nullptr);
        int toMatch = 0;
        for (int i = 0; i < M; i++) {  // what isInFrustum leaves in the map points (src/Frame.cc:274-276,319-328)
            if (pts[i].skip || pts[i].bad) continue;
            const auto& pMP = vpLocalMapPoints[i];
            pMP->mbTrackInView = out[i].in_view != 0;
            pMP->mTrackProjX = out[i].proj_x;
            pMP->mTrackProjY = out[i].proj_y;
            if (out[i].in_view) {
                pMP->mTrackProjXR = xr[i];
                pMP->mTrackDepth = out[i].track_depth;
                pMP->mnTrackScaleLevel = out[i].level;
                pMP->mTrackViewCos = out[i].view_cos;
                toMatch++;
            }
        }
        if (nToMatch) *nToMatch = toMatch;
        if (n == 0) return -1;
        for (int i = 0; i < n; i++)
            if (match[i] >= 0) F->mvpMapPoints[i] = vpLocalMapPoints[match[i]];  // src/ORBmatcher.cc:113
        return nmatches;
    }
};

// ORB_SLAM3::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ORBVocabulary.h:29-30), the
// part Frame::ComputeBoW / KeyFrame::ComputeBoW use (src/Frame.cc:483-495): load the text vocabulary, transform a
// frame's descriptors into BowVector + FeatureVector.  The tree descent of every feature runs on the GPU
// (orbfe_bow_transform); the <= N map inserts and the normalisation stay on the host, in the reference's order, so
// the doubles are bit-identical.  `BowVector` is any std::map<unsigned, double>-shaped type (DBoW2::BowVector),
// `FeatureVector` any std::map<unsigned, std::vector<unsigned>>-shaped type (DBoW2::FeatureVector).
class ORBVocabulary {
public:
    enum WeightingType { TF_IDF, TF, IDF, BINARY };                                        // BowVector.h:26-32
    enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };     // BowVector.h:35-43

    explicit ORBVocabulary(orbfe_handle* h) : h_(h) {}
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;
    ~ORBVocabulary() { orbfe_vocab_destroy(v_); }

    // TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1349-1436); blank lines are skipped.
    bool loadFromTextFile(const std::string& filename)
    {
        FILE* f = fopen(filename.c_str(), "r");
        if (!f) return false;
        int n1 = 0, n2 = 0;
        if (fscanf(f, "%d %d %d %d", &k_, &L_, &n1, &n2) != 4 || k_ < 0 || k_ > 20 || L_ < 1 || L_ > 10 || n1 < 0 || n1 > 5 ||
            n2 < 0 || n2 > 3) {
            fclose(f);
            return false;
        }
        scoring_ = (ScoringType)n1;
        weighting_ = (WeightingType)n2;
        std::vector<int> parent{0}, leaf{0};
        std::vector<uint8_t> desc(32, 0);
        std::vector<double> weight{0.0};
        for (;;) {
            int pid, isLeaf;
            if (fscanf(f, "%d %d", &pid, &isLeaf) != 2) break;
            uint8_t d[32];
            bool ok = true;
            for (int i = 0; i < 32; i++) {
                int v;
                ok = ok && fscanf(f, "%d", &v) == 1;
                d[i] = (uint8_t)v;
            }
            double w;
            if (!ok || fscanf(f, "%lf", &w) != 1 || pid < 0 || pid >= (int)parent.size()) {
                fclose(f);
                return false;
            }
            parent.push_back(pid);
            leaf.push_back(isLeaf);
            desc.insert(desc.end(), d, d + 32);
            weight.push_back(w);
        }
        fclose(f);
        const int n = (int)parent.size();
        std::vector<int> childOff(n + 1, 0), childIdx(n > 1 ? n - 1 : 1), wordId(n, 0);
        for (int i = 1; i < n; i++) childOff[parent[i] + 1]++;
        for (int i = 0; i < n; i++) childOff[i + 1] += childOff[i];
        std::vector<int> fill(childOff.begin(), childOff.end() - 1);
        for (int i = 1; i < n; i++) childIdx[fill[parent[i]]++] = i;  // ascending id == push_back order (:1403)
        nWords_ = 0;
        for (int i = 1; i < n; i++)
            if (leaf[i] > 0) wordId[i] = nWords_++;  // :1419-1426
        return create(n, childOff.data(), childIdx.data(), desc.data(), wordId.data(), weight.data(), L_);
    }

    // direct construction from DBoW2's node table (for callers that already hold one)
    bool create(int nNodes, const int* childOff, const int* childIdx, const uint8_t* nodeDesc, const int* wordId,
                const double* weight, int L)
    {
        orbfe_vocab_destroy(v_);
        v_ = nullptr;
        L_ = L;
        return orbfe_vocab_create(h_, nNodes, childOff, childIdx, nodeDesc, wordId, weight, L, &v_) == ORBFE_OK;
    }

    bool empty() const { return v_ == nullptr; }
    unsigned size() const { return (unsigned)nWords_; }
    int getBranchingFactor() const { return k_; }
    int getDepthLevels() const { return L_; }
    void setScoringType(ScoringType t) { scoring_ = t; }
    void setWeightingType(WeightingType t) { weighting_ = t; }

    // TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1136-1204); `desc` is the
    // frame's N x 32 descriptor matrix (what Converter::toDescriptorVector splits into rows, src/Frame.cc:487).
    template <class BowVector, class FeatureVector>
    void transform(const uint8_t* desc, int n, BowVector& v, FeatureVector& fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        if (empty() || n <= 0) return;
        std::vector<int> word(n), node(n);
        std::vector<double> w(n);
        orbfe_detail::check(orbfe_bow_transform(h_, v_, desc, n, levelsup, word.data(), node.data(), w.data()), h_,
                            "orbfe_bow_transform");
        assemble(word.data(), node.data(), w.data(), n, v, fv);
    }

    // the host half of transform(): BowVector / FeatureVector from the per-feature (word, node, weight) triples, in the
    // reference's insertion order (TemplatedVocabulary.h:1157-1204), so the doubles are bit-identical
    template <class BowVector, class FeatureVector>
    void assemble(const int* word, const int* node, const double* w, int n, BowVector& v, FeatureVector& fv) const
    {
        v.clear();
        fv.clear();
        const bool must = scoring_ != DOT_PRODUCT;  // ScoringObject.h:74-89
        const bool tf = weighting_ == TF || weighting_ == TF_IDF;
        for (int i = 0; i < n; i++) {
            if (!(w[i] > 0)) continue;  // stopped word
            auto it = v.lower_bound(word[i]);
            if (it != v.end() && !(v.key_comp()(word[i], it->first))) {
                if (tf) it->second += w[i];  // BowVector::addWeight; addIfNotExist leaves it
            } else {
                v.insert(it, typename BowVector::value_type(word[i], w[i]));
            }
            fv[node[i]].push_back(i);  // FeatureVector::addFeature
        }
        if (tf && !v.empty() && !must) {
            const double nd = (double)v.size();
            for (auto& e : v) e.second /= nd;
        }
        if (must) {  // BowVector::normalize, src/DBoW2/BowVector.cpp:62-84
            double norm = 0.0;
            if (scoring_ == L2_NORM) {
                for (auto& e : v) norm += e.second * e.second;
                norm = std::sqrt(norm);
            } else {
                for (auto& e : v) norm += std::fabs(e.second);
            }
            if (norm > 0.0)
                for (auto& e : v) e.second /= norm;
        }
    }

    const orbfe_vocab* get() const { return v_; }

private:
    orbfe_handle* h_;
    orbfe_vocab* v_ = nullptr;
    int k_ = 0, L_ = 0, nWords_ = 0;
    ScoringType scoring_ = L1_NORM;
    WeightingType weighting_ = TF_IDF;
};

// The tracking thread's chain of a frame tracked against its reference key frame, as ONE submission
// (orbfe_track_reference_keyframe): what Tracking::GrabImageMonocular + Tracking::TrackReferenceKeyFrame do up to the pose
// solver -- Frame::Frame -> ExtractORB (src/Frame.cc:178-189), mCurrentFrame->ComputeBoW() (src/Tracking.cc:829,
// src/Frame.cc:483-495), ORBmatcher::SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches, nnRatio, true) (:835).
// `resident` is the reference key frame's ResidentKeyFrame (made when the key frame was created); its map points are read
// as they stand now.  On return `F` carries the fresh keypoints, descriptors (through setDesc(F, rows, n)), mBowVec and
// mFeatVec; vpMapPointMatches is what SearchByBoW would have filled.  Returns the match count, or -1 when the frame has no
// keypoints (the reference returns before Track(), src/Tracking.cc:158-159).
struct ReferenceKeyFrameTracker {
    template <class FramePtr, class KeyFramePtr, class MapPointPtr, class SetFrameDesc>
    static int ExtractAndSearchByBoW(ORBextractor& extractor, const GrayImageView& im, FramePtr F, const ORBVocabulary& voc,
                                     KeyFramePtr pKF, const ResidentKeyFrame& resident, std::vector<MapPointPtr>& vpMapPointMatches,
                                     const float nnRatio, const bool checkOrientation, SetFrameDesc setDesc, const int levelsup = 4)
    {
        orbfe_handle* h = extractor.handle();
        const int cap = extractor.maxKeypoints();
        const auto vpMapPointsKF = pKF->GetMapPointMatches();
        const int nKF = (int)vpMapPointsKF.size();
        std::vector<uint8_t> hasMP(nKF > 0 ? nKF : 1, 0);
        for (int i = 0; i < nKF; i++) hasMP[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // src/ORBmatcher.cc:182-187
        auto keys = std::make_shared<std::vector<KeyPoint>>(cap);
        std::vector<uint8_t> desc((size_t)cap * ORBFE_DESC_BYTES);
        std::vector<int> word(cap), node(cap), match(cap);
        std::vector<double> w(cap);
        int n = 0, nmatches = 0;
        orbfe_detail::check(orbfe_track_reference_keyframe(h, im.data, im.pitch, voc.get(), levelsup, resident.get(), hasMP.data(), nnRatio,
                                                           checkOrientation ? 1 : 0, reinterpret_cast<orbfe_keypoint*>(keys->data()),
                                                           desc.data(), &n, nullptr, word.data(), node.data(), w.data(), match.data(),
                                                           &nmatches), h, "orbfe_track_reference_keyframe");
        keys->resize(n);
        F->mNumKeypoints = n;
        F->mvKeysUn = keys;
        setDesc(F, desc.data(), n);
        voc.assemble(word.data(), node.data(), w.data(), n, F->mBowVec, F->mFeatVec);
        vpMapPointMatches.assign(n, MapPointPtr());
        if (n == 0) return -1;
        for (int i = 0; i < n; i++)
            if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];  // src/ORBmatcher.cc:241
        return nmatches;
    }
};

// The tracking thread's chain of a frame while the map is being initialised, as ONE submission (orbfe_track_initialization):
// what Tracking::GrabImageMonocular + Tracking::MonocularInitialization do up to ReconstructWithTwoViews -- Frame::Frame ->
// ExtractORB (src/Frame.cc:178-189), ORBmatcher::SearchForInitialization(mInitialFrame, mCurrentFrame, 40, 0.45, true)
// (src/Tracking.cc:603-607).  Seed(F) where the reference sets mInitialFrame = mCurrentFrame (:569-586) -- the frame's
// keypoints and descriptors go to the GPU once; ExtractAndSearch(...) for every following frame; Reset() where the reference
// clears mbReadyToInitializate (:588-602).  The thresholds around the calls (FEAT_INIT_COUNT, the 2 s time-out) stay in
// Tracking.cc.
class InitializationTracker {
public:
    explicit InitializationTracker(ORBextractor& extractor) : ex_(extractor) {}
    ~InitializationTracker() { orbfe_init_frame_destroy(f1_); }
    InitializationTracker(const InitializationTracker&) = delete;
    InitializationTracker& operator=(const InitializationTracker&) = delete;
    bool Ready() const { return f1_ != nullptr; }
    void Reset()
    {
        orbfe_init_frame_destroy(f1_);
        f1_ = nullptr;
    }
    template <class FramePtr, class DescOf>
    void Seed(FramePtr F, DescOf descOf)
    {
        Reset();
        orbfe_detail::check(orbfe_init_frame_create(ex_.handle(), (int)F->mvKeysUn->size(),
                                                    reinterpret_cast<const orbfe_keypoint*>(F->mvKeysUn->data()), descOf(F), &f1_),
                            ex_.handle(), "orbfe_init_frame_create");
    }
    // -> {nmatches, vnMatches12} as SearchForInitialization; F receives the fresh keypoints and descriptors (setDesc(F, rows, n));
    // nmatches == -1 when the frame has no keypoints (the reference returns before Track(), src/Tracking.cc:158-159)
    template <class FramePtr, class SetFrameDesc>
    std::pair<int, std::vector<int>> ExtractAndSearch(const GrayImageView& im, FramePtr F, SetFrameDesc setDesc, int windowSize = 40,
                                                      float nnRatio = 0.45f, bool checkOrientation = true)
    {
        orbfe_handle* h = ex_.handle();
        const int cap = ex_.maxKeypoints(), n1 = orbfe_init_frame_size(f1_);
        auto keys = std::make_shared<std::vector<KeyPoint>>(cap);
        std::vector<uint8_t> desc((size_t)cap * ORBFE_DESC_BYTES);
        std::vector<int> m12(n1 > 0 ? n1 : 1, -1);
        orbfe_track_params tp = ORBFE_TRACK_PARAMS_INIT;
        tp.grid_cols = F->getFrameGridCols(); tp.grid_rows = F->getFrameGridRows();
        tp.min_x = F->mnMinX; tp.min_y = F->mnMinY;
        tp.grid_inv_w = F->mfGridElementWidthInv; tp.grid_inv_h = F->mfGridElementHeightInv;
        int n = 0, nmatches = 0;
        orbfe_detail::check(orbfe_track_initialization(h, im.data, im.pitch, f1_, &tp, windowSize, nnRatio, checkOrientation ? 1 : 0,
                                                       reinterpret_cast<orbfe_keypoint*>(keys->data()), desc.data(), &n, nullptr, m12.data(),
                                                       &nmatches), h, "orbfe_track_initialization");
        keys->resize(n);
        F->mNumKeypoints = n;
        F->mvKeysUn = keys;
        setDesc(F, desc.data(), n);
        m12.resize(n1);
        return {n == 0 ? -1 : nmatches, std::move(m12)};
    }

private:
    ORBextractor& ex_;
    orbfe_init_frame* f1_ = nullptr;
};

}  // namespace ORB_SLAM3
