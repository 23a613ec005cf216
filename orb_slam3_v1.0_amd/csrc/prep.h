// prep.h -- node-side image preparation (kernels_prep.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace orbfe {

struct PrepArgs {
    const uint8_t* src;   // BGR, 3 bytes per pixel
    int srcPitch;         // bytes
    size_t srcFrameStride;
    int srcW, srcH;
    const float* map1;    // srcH x srcW, x coordinates (cv::fisheye::initUndistortRectifyMap, CV_32F)
    const float* map2;    // y coordinates
    uint8_t* dst;         // grey
    int dstPitch;
    size_t dstFrameStride;
    int dstW, dstH;
    float fx, fy;         // prep_scale(srcW, dstW), prep_scale(srcH, dstH)
};

float prep_scale(int srcN, int dstN);
void prep_launch(hipStream_t s, const PrepArgs& P, int batch);

}  // namespace orbfe
