// device_math.h -- the pinned fp32 sequences of SPEC DECISION S5 for gfx950.
// Replaces atan2f (src/cuda/Angle_gpu.cu:73) and cosf/sinf (src/cuda/Orb_gpu.cu:329) of the
// reference, whose last bits depend on the CUDA math library.  Every operation is a single
// IEEE-754 fp32 op; the translation unit is compiled with -ffp-contract=off so no FMA is formed.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace orbfe {

__device__ __forceinline__ int reflect101(int i, int n)
{
    // BORDER_REFLECT_101 for any offset (period 2n-2); n >= 2 on every pyramid level
    const int p = 2 * n - 2;
    i = i % p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// REFLECT_101 for overshoots smaller than n (one bounce, no modulo), clamped so that the result is
// always a valid index even in a caller's don't-care region
__device__ __forceinline__ int reflect_near(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float spec_atan2f(float y, float x)
{
    const float kPi = 0x1.921fb6p+1f, kPi2 = 0x1.921fb6p+0f, kPi4 = 0x1.921fb6p-1f;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float t = mn / mx;  // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
    float base = 0.0f;
    if (t > 0x1.a8279ap-2f) {
        t = (t - 1.0f) / (t + 1.0f);
        base = kPi4;
    }
    const float z = t * t;
    float p = 0x1.61e174p-4f * z;
    p = p + -0x1.1fe904p-3f;
    p = p * z;
    p = p + 0x1.99799ep-3f;
    p = p * z;
    p = p + -0x1.555556p-2f;
    float r = p * z;
    r = r * t;
    r = r + t;
    r = base + r;
    if (ay > ax) r = kPi2 - r;
    if (x < 0.0f) r = kPi - r;
    if (y < 0.0f) r = -r;
    return r;
}

// src/cuda/Angle_gpu.cu:73-75
__device__ __forceinline__ float atan2_deg(float m01, float m10)
{
    const float kPi = 0x1.921fb6p+1f;
    float d = spec_atan2f(m01, m10);
    if (d < 0.0f) d = d + 2.0f * kPi;
    d = d * (180.0f / kPi);
    return d;
}

// cos/sin of an angle given in degrees (src/cuda/Orb_gpu.cu:327-329)
__device__ __forceinline__ void cos_sin_deg(float deg, float& c, float& s)
{
    float kf = deg * 0x1.6c16c2p-7f;
    kf = kf + 0.5f;
    const int k = (int)kf;
    const float r = deg - 90.0f * (float)k;
    const float x = r * 0x1.1df46ap-6f;
    const float z = x * x;
    float p = -0x1.9b7856p-13f * z;
    p = p + 0x1.110e32p-7f;
    p = p * z;
    p = p + -0x1.555558p-3f;
    float sn = p * z;
    sn = sn * x;
    sn = sn + x;
    float q = 0x1.9bfe2ep-16f * z;
    q = q + -0x1.6c134p-10f;
    q = q * z;
    q = q + 0x1.555554p-5f;
    float cs = q * z;
    cs = cs * z;
    float h = 0.5f * z;
    h = 1.0f - h;
    cs = cs + h;
    switch (k & 3) {
    case 0: c = cs; s = sn; break;
    case 1: c = -sn; s = cs; break;
    case 2: c = -cs; s = -sn; break;
    default: c = sn; s = -cs; break;
    }
}

// SPEC DECISION S8: natural logarithm of MapPoint::PredictScale (src/MapPoint.cc:580), same operation
// sequence as oracle/match_oracle.c orc_spec_logf: x = m * 2^e, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s)
// with s = (m-1)/(m+1) as a degree-4 polynomial in s^2; contraction is off for this translation unit.
__device__ __forceinline__ float spec_logf(float x)
{
    if (!(x > 0.0f)) return -INFINITY;
    if (x > 3.0e38f) return INFINITY;
    int e;
    float m = frexpf(x, &e);
    if (m < 0x1.6a09e6p-1f) {
        m = m * 2.0f;
        e -= 1;
    }
    const float s = __fdiv_rn(m - 1.0f, m + 1.0f);
    const float z = s * s;
    float p = 0x1.c71c72p-4f;
    p = p * z + 0x1.24924ap-3f;
    p = p * z + 0x1.99999ap-3f;
    p = p * z + 0x1.555556p-2f;
    p = p * z;
    const float t = s + s;
    const float r = t + t * p;
    const float ef = (float)e;
    return ef * 0x1.62ep-1f + (r + ef * 0x1.0bfbe8p-15f);
}

// GeometricCamera::project of the two camera models (src/CameraModels/Pinhole.cpp:41-47,
// src/CameraModels/KannalaBrandt8.cpp:66-83), same operation sequence as oracle/match_oracle.c camera_project.
// `Frustum` is orbfe_frustum (include/orbfe.h); a template keeps this header free of the C ABI include.
template <class Frustum>
__device__ __forceinline__ void camera_project(const Frustum& F, float x, float y, float z, float& u, float& v)
{
    if (F.camera_model == 0) {
        u = F.fx * x / z + F.cx;
        v = F.fy * y / z + F.cy;
        return;
    }
    const float x2_plus_y2 = x * x + y * y;
    const float theta = spec_atan2f(sqrtf(x2_plus_y2), z);
    const float psi = spec_atan2f(y, x);
    const float theta2 = theta * theta;
    const float theta3 = theta * theta2;
    const float theta5 = theta3 * theta2;
    const float theta7 = theta5 * theta2;
    const float theta9 = theta7 * theta2;
    const float r = (((theta + F.k1 * theta3) + F.k2 * theta5) + F.k3 * theta7) + F.k4 * theta9;
    float deg = psi * 0x1.ca5dc2p+5f;  // 180 / pi
    if (deg < 0.0f) deg = deg + 360.0f;
    float c, s;
    cos_sin_deg(deg, c, s);
    u = F.fx * r * c + F.cx;
    v = F.fy * r * s + F.cy;
}

}  // namespace orbfe
