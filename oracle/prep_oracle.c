/* prep_oracle.c -- CPU restatement of the node-side image preparation (TEST INFRASTRUCTURE ONLY).
 *
 * Follows ImageGrabber::ConvertImageToGPU (ros2_ws/src/mono-inertial/include/image_grabber.hpp:96-110):
 *   cv::cuda::remap(bgr, undistorted, map1, map2, INTER_CUBIC, BORDER_CONSTANT, Scalar())      (:103)
 *   cv::cuda::resize(undistorted, resized, Size(m_width, m_height), 0, 0, INTER_LINEAR)        (:105)
 *   cv::cuda::cvtColor(resized, grey, COLOR_BGR2GRAY)                                          (:107)
 * with the maps of cv::fisheye::initUndistortRectifyMap(..., CV_32F, ...) (mono_inertial_node.cpp:61-71).
 *
 * PARITY UNPINNED: all three operators live in OpenCV 4.9 + opencv_contrib (cudawarping / cudaimgproc / cudev),
 * which is not under /root/reference and is built there with CUDA_FAST_MATH; no test or golden vector of the
 * reference pins a pixel.  SPEC DECISION S9 fixes the arithmetic to the operators' published form in
 * binary32, one rounding per operation, no contraction:
 *   cubic:    taps cx in [ceil(x-2), floor(x+2)], cy likewise (cy outer, cx inner); weight w = c(x-cx) * c(y-cy) with
 *             c(t) = |t|<=1 ? t*t*(1.5*t - 2.5) + 1 : |t|<2 ? t*(t*(-0.5*t + 2.5) - 4) + 2 : 0   (t = |t| first);
 *             sum_c += w * src_c (0 outside the image), wsum += w; pixel_c = wsum == 0 ? 0 : sum_c / wsum;
 *             stored as u8 by round-half-even + clamp (the intermediate image IS 8-bit in the reference).
 *   bilinear: sx = dx * fx, fx = (float)(1.0 / ((double)dst_w / src_w)); x1 = floor(sx), x2 = x1 + 1 (read clamped to
 *             cols-1); out = p11*((x2-sx)*(y2-sy)) + p12*((sx-x1)*(y2-sy)) + p21*((x2-sx)*(sy-y1)) + p22*((sx-x1)*(sy-y1)),
 *             accumulated in that order from 0; u8 by round-half-even + clamp.  No half-pixel centre shift.
 *   grey:     (B*1868 + G*9617 + R*4899 + 8192) >> 14.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "orb_oracle.h"

static float cubic_coeff(float t)
{
    t = fabsf(t);
    if (t <= 1.0f) return t * t * (1.5f * t - 2.5f) + 1.0f;
    if (t < 2.0f) return t * (t * (-0.5f * t + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

static uint8_t sat_u8(float v)
{
    if (!(v > 0.0f)) return 0; /* also NaN */
    if (v >= 255.0f) return 255;
    return (uint8_t)lrintf(v); /* round-half-even in the default rounding mode */
}

/* one undistorted pixel (3 channels) */
void orc_prep_remap_pixel(const uint8_t *bgr, int pitch, int srcW, int srcH, float x, float y, uint8_t out[3])
{
    /* every tap outside the image: sum == 0, so the pixel is 0 whatever wsum is (also the case for NaN / infinite /
     * huge map entries, where the tap loop itself would be ill-defined) */
    if (!(x > -3.0f && x < (float)srcW + 2.0f && y > -3.0f && y < (float)srcH + 2.0f)) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    const float xmin = ceilf(x - 2.0f), xmax = floorf(x + 2.0f);
    const float ymin = ceilf(y - 2.0f), ymax = floorf(y + 2.0f);
    float sum[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
    for (float cy = ymin; cy <= ymax; cy += 1.0f)
        for (float cx = xmin; cx <= xmax; cx += 1.0f) {
            const float w = cubic_coeff(x - cx) * cubic_coeff(y - cy);
            const int ix = (int)cx, iy = (int)cy;
            if (ix >= 0 && ix < srcW && iy >= 0 && iy < srcH) {
                const uint8_t *p = bgr + (size_t)iy * pitch + (size_t)ix * 3;
                sum[0] = sum[0] + w * (float)p[0];
                sum[1] = sum[1] + w * (float)p[1];
                sum[2] = sum[2] + w * (float)p[2];
            }
            wsum = wsum + w;
        }
    for (int c = 0; c < 3; c++) out[c] = wsum == 0.0f ? 0 : sat_u8(sum[c] / wsum);
}

float orc_prep_scale(int srcN, int dstN) { return (float)(1.0 / ((double)dstN / (double)srcN)); }

/* full chain; und (srcH x srcW x 3, may be NULL) receives the intermediate undistorted image for the tests */
void orc_prepare_image(const uint8_t *bgr, int pitch, int srcW, int srcH, const float *map1, const float *map2, int dstW,
                       int dstH, uint8_t *grey, int greyPitch, uint8_t *und)
{
    uint8_t *U = und ? und : (uint8_t *)malloc((size_t)srcW * srcH * 3);
    for (int y = 0; y < srcH; y++)
        for (int x = 0; x < srcW; x++)
            orc_prep_remap_pixel(bgr, pitch, srcW, srcH, map1[(size_t)y * srcW + x], map2[(size_t)y * srcW + x],
                                 U + ((size_t)y * srcW + x) * 3);
    const float fx = orc_prep_scale(srcW, dstW), fy = orc_prep_scale(srcH, dstH);
    for (int dy = 0; dy < dstH; dy++)
        for (int dx = 0; dx < dstW; dx++) {
            const float sx = (float)dx * fx, sy = (float)dy * fy;
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = x2 < srcW - 1 ? x2 : srcW - 1, y2r = y2 < srcH - 1 ? y2 : srcH - 1;
            const int x1r = x1 < srcW - 1 ? x1 : srcW - 1, y1r = y1 < srcH - 1 ? y1 : srcH - 1; /* dst larger than src */
            const float w11 = ((float)x2 - sx) * ((float)y2 - sy), w12 = (sx - (float)x1) * ((float)y2 - sy);
            const float w21 = ((float)x2 - sx) * (sy - (float)y1), w22 = (sx - (float)x1) * (sy - (float)y1);
            uint8_t px[3];
            for (int c = 0; c < 3; c++) {
                float o = 0.0f;
                o = o + (float)U[((size_t)y1r * srcW + x1r) * 3 + c] * w11;
                o = o + (float)U[((size_t)y1r * srcW + x2r) * 3 + c] * w12;
                o = o + (float)U[((size_t)y2r * srcW + x1r) * 3 + c] * w21;
                o = o + (float)U[((size_t)y2r * srcW + x2r) * 3 + c] * w22;
                px[c] = sat_u8(o);
            }
            grey[(size_t)dy * greyPitch + dx] = (uint8_t)(((unsigned)px[0] * 1868u + (unsigned)px[1] * 9617u + (unsigned)px[2] * 4899u + 8192u) >> 14);
        }
    if (!und) free(U);
}
