// TEST-ONLY stand-in for the reference's include/KeyPoint.h (a 24-byte POD: cv::Point2f pt, int response, float size,
// int octave, float angle), so that the ORBFE_WITH_OPENCV branch of the adaptor compiles here.
#pragma once
#include <opencv2/core/cuda.hpp>

namespace ORB_SLAM3 {
struct KeyPoint {
    cv::Point2f pt;
    int response;
    float size;
    int octave;
    float angle;
};
}  // namespace ORB_SLAM3
