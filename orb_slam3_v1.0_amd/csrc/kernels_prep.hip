// kernels_prep.hip -- node-side image preparation on gfx950 (honourable mention).
//
// Replaces ImageGrabber::ConvertImageToGPU (ros2_ws/src/mono-inertial/include/image_grabber.hpp:96-110):
// cv::cuda::remap(INTER_CUBIC, BORDER_CONSTANT 0) -> cv::cuda::resize(INTER_LINEAR) -> cv::cuda::cvtColor(BGR2GRAY),
// three library launches with two full-size intermediates (2048x1536x3 and 614x460x3 in the node's configuration,
// mono_inertial_node.cpp:20,59-71).  Here it is ONE kernel: a grey output pixel needs the four undistorted pixels
// around (dx * fx, dy * fy); with the node's resize factor 0.3 those are 4 of every ~11 pixels of the undistorted
// image, so only they are interpolated (2.8x fewer cubic evaluations) and neither intermediate touches HBM.
// The 8-bit rounding of both intermediates is part of the reference's result and is kept.
// Arithmetic: SPEC DECISION S9 (oracle/prep_oracle.c header) -- binary32, one rounding per operation, no contraction;
// PARITY UNPINNED against OpenCV-CUDA (third-party, absent, CUDA_FAST_MATH build).
#include <hip/hip_runtime.h>

#include "prep.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

__device__ __forceinline__ float cubic_coeff(float t)
{
    t = fabsf(t);
    if (t <= 1.0f) return t * t * (1.5f * t - 2.5f) + 1.0f;
    if (t < 2.0f) return t * (t * (-0.5f * t + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

__device__ __forceinline__ int sat_u8(float v)
{
    if (!(v > 0.0f)) return 0;  // also NaN
    if (v >= 255.0f) return 255;
    return __float2int_rn(v);  // round-half-even
}

// one pixel of the undistorted image, packed b | g<<8 | r<<16
__device__ __forceinline__ unsigned remap_pixel(const uint8_t* __restrict__ bgr, int pitch, int srcW, int srcH, float x, float y)
{
    if (!(x > -3.0f && x < (float)srcW + 2.0f && y > -3.0f && y < (float)srcH + 2.0f)) return 0u;  // no tap inside the image
    const float xmin = ceilf(x - 2.0f), xmax = floorf(x + 2.0f);
    const float ymin = ceilf(y - 2.0f), ymax = floorf(y + 2.0f);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, wsum = 0.0f;
    for (float cy = ymin; cy <= ymax; cy += 1.0f) {
        const int iy = (int)cy;
        const float wy = cubic_coeff(y - cy);
        const bool rowIn = iy >= 0 && iy < srcH;
        const uint8_t* row = bgr + (size_t)(rowIn ? iy : 0) * pitch;
        for (float cx = xmin; cx <= xmax; cx += 1.0f) {
            const float w = cubic_coeff(x - cx) * wy;
            const int ix = (int)cx;
            if (rowIn && ix >= 0 && ix < srcW) {
                const uint8_t* p = row + (size_t)ix * 3;
                s0 = s0 + w * (float)p[0];
                s1 = s1 + w * (float)p[1];
                s2 = s2 + w * (float)p[2];
            }
            wsum = wsum + w;
        }
    }
    if (wsum == 0.0f) return 0u;
    return (unsigned)sat_u8(s0 / wsum) | ((unsigned)sat_u8(s1 / wsum) << 8) | ((unsigned)sat_u8(s2 / wsum) << 16);
}

__global__ __launch_bounds__(256) void prep_kernel(PrepArgs P)
{
    const int dx = blockIdx.x * 32 + (threadIdx.x & 31);
    const int dy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (dx >= P.dstW || dy >= P.dstH) return;
    const uint8_t* bgr = P.src + (size_t)blockIdx.z * P.srcFrameStride;
    const float sx = (float)dx * P.fx, sy = (float)dy * P.fy;
    const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const int xr[2] = {min(x1, P.srcW - 1), min(x2, P.srcW - 1)};
    const int yr[2] = {min(y1, P.srcH - 1), min(y2, P.srcH - 1)};
    unsigned px[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t m = (size_t)yr[k >> 1] * P.srcW + xr[k & 1];
        px[k] = remap_pixel(bgr, P.srcPitch, P.srcW, P.srcH, P.map1[m], P.map2[m]);
    }
    const float w11 = ((float)x2 - sx) * ((float)y2 - sy), w12 = (sx - (float)x1) * ((float)y2 - sy);
    const float w21 = ((float)x2 - sx) * (sy - (float)y1), w22 = (sx - (float)x1) * (sy - (float)y1);
    unsigned c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float o = 0.0f;
        o = o + (float)((px[0] >> (8 * ch)) & 255u) * w11;
        o = o + (float)((px[1] >> (8 * ch)) & 255u) * w12;
        o = o + (float)((px[2] >> (8 * ch)) & 255u) * w21;
        o = o + (float)((px[3] >> (8 * ch)) & 255u) * w22;
        c[ch] = (unsigned)sat_u8(o);
    }
    P.dst[(size_t)blockIdx.z * P.dstFrameStride + (size_t)dy * P.dstPitch + dx] =
        (uint8_t)((c[0] * 1868u + c[1] * 9617u + c[2] * 4899u + 8192u) >> 14);
}

}  // namespace

float prep_scale(int srcN, int dstN) { return (float)(1.0 / ((double)dstN / (double)srcN)); }

void prep_launch(hipStream_t s, const PrepArgs& P, int batch)
{
    hipLaunchKernelGGL(prep_kernel, dim3((P.dstW + 31) / 32, (P.dstH + 7) / 8, batch), dim3(256), 0, s, P);
}

}  // namespace orbfe
