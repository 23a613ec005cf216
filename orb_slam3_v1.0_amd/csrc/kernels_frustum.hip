// kernels_frustum.hip -- Frame::isInFrustum for all local map points of a frame in one launch
// (SURVEY.md section 8f, row f3).
//
// Replaces the per-point loop of Tracking::SearchLocalPoints (src/Tracking.cc:1059-1077) ->
// Frame::isInFrustum (src/Frame.cc:272-331) -> Pinhole::project (src/CameraModels/Pinhole.cpp:41-47) and
// MapPoint::PredictScale (src/MapPoint.cc:572-587).  The output records are exactly what
// SearchByProjection reads (orbfe_map_point), so projection and matching chain on the device with no
// host round trip in between.  One thread per map point; 32 B read + 32 B written per point.
// SPEC DECISION S8: binary32, left-to-right, no contraction, sqrtf((x*x + y*y) + z*z), spec_logf.
// (sqrtf and "/" are correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; the
// __fsqrt_rn intrinsic is NOT -- it maps to the native approximate square root.)
#include <string>

#include "device_math.h"
#include "match.h"

#pragma clang fp contract(off)

namespace orbfe {

namespace {

__device__ __forceinline__ void frustum_point(const orbfe_frustum& F, int i, const orbfe_world_point* __restrict__ pts,
                                              orbfe_map_point* __restrict__ out, float* __restrict__ projXR)
{
    const orbfe_world_point p = pts[i];
    orbfe_map_point o;
    o.proj_x = -1.0f;  // :275-276
    o.proj_y = -1.0f;
    o.view_cos = 0.0f;
    o.track_depth = 0.0f;
    o.level = 0;
    o.in_view = 0;
    o.bad = p.bad;
    o.observations = p.observations;
    float xr = 0.0f;
    do {
        if (p.skip || p.bad) break;  // src/Tracking.cc:1066-1069
        const float X = p.x, Y = p.y, Z = p.z;
        const float pcx = ((F.rcw[0] * X + F.rcw[1] * Y) + F.rcw[2] * Z) + F.tcw[0];  // :282
        const float pcy = ((F.rcw[3] * X + F.rcw[4] * Y) + F.rcw[5] * Z) + F.tcw[1];
        const float pcz = ((F.rcw[6] * X + F.rcw[7] * Y) + F.rcw[8] * Z) + F.tcw[2];
        const float pcDist = sqrtf((pcx * pcx + pcy * pcy) + pcz * pcz);
        const float invz = __fdiv_rn(1.0f, pcz);
        if (pcz < 0.0f) break;  // :288
        float u, v;
        camera_project(F, pcx, pcy, pcz, u, v);  // mpCamera->project(Pc), :291
        if (u < F.min_x || u > F.max_x) break;
        if (v < F.min_y || v > F.max_y) break;
        o.proj_x = u;  // :299-300: set before the distance test
        o.proj_y = v;
        const float maxD = 1.1f * p.max_distance, minD = 0.9f * p.min_distance;  // MapPoint.cc:543-553
        const float ox = X - F.twc[0], oy = Y - F.twc[1], oz = Z - F.twc[2];
        const float dist = sqrtf((ox * ox + oy * oy) + oz * oz);
        if (dist < minD || dist > maxD) break;
        const float ratio = __fdiv_rn(p.max_distance, dist);
        const float q = __fdiv_rn(spec_logf(ratio), F.log_scale_factor);
        int nScale;
        if (!(q > 0.0f)) nScale = 0;
        else if (q >= (float)F.n_levels) nScale = F.n_levels - 1;
        else {
            nScale = (int)ceilf(q);
            if (nScale >= F.n_levels) nScale = F.n_levels - 1;
        }
        o.in_view = 1;
        o.level = nScale;
        o.view_cos = 1.0f;  // :316: the normal test is disabled in this fork
        o.track_depth = pcDist;
        xr = u - F.mbf * invz;
    } while (false);
    out[i] = o;
    if (projXR) projXR[i] = xr;
}

__global__ __launch_bounds__(256) void frustum_kernel(orbfe_frustum F, int n, const orbfe_world_point* __restrict__ pts,
                                                      orbfe_map_point* __restrict__ out, float* __restrict__ projXR)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) frustum_point(F, i, pts, out, projXR);
}

// the frame's pose and camera read from HBM: a captured hipGraph (orbfe_track_frame) freezes kernel arguments, so whatever
// changes from frame to frame has to arrive through memory
__global__ __launch_bounds__(256) void frustum_dev_kernel(const orbfe_frustum* __restrict__ dF, int n,
                                                          const orbfe_world_point* __restrict__ pts,
                                                          orbfe_map_point* __restrict__ out, float* __restrict__ projXR)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const orbfe_frustum F = *dF;  // block-uniform: scalar loads
    frustum_point(F, i, pts, out, projXR);
}

// Streaming / resident-map form (orbfe_map, orbfe_stream_submit_track): frame b = blockIdx.y reads ITS frustum and ITS list
// of map-point ids; id >= 0 names an entry of the resident map, ~id (negative) the same entry with "mnLastFrameSeen ==
// current frame" (src/Tracking.cc:1066: skipped), ids outside the map give a bad record.  Besides the record
// SearchByProjection reads, the descriptor is copied next to it so that the matcher finds frame b's descriptors contiguous.
__global__ __launch_bounds__(256) void frustum_gather_kernel(const orbfe_frustum* __restrict__ dF, const int* __restrict__ ids, int M,
                                                             int mapCap, const orbfe_world_point* __restrict__ mapPts,
                                                             const uint8_t* __restrict__ mapDesc, orbfe_map_point* __restrict__ out,
                                                             uint8_t* __restrict__ descOut, float* __restrict__ projXR)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int b = blockIdx.y;
    const orbfe_frustum F = dF[b];  // block-uniform: scalar loads
    const int raw = ids[(size_t)b * M + i];
    const int id = raw < 0 ? ~raw : raw;
    orbfe_world_point p{};
    uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
    if (id < mapCap) {
        p = mapPts[id];
        const uint4* dp = reinterpret_cast<const uint4*>(mapDesc + (size_t)id * 32);
        d0 = dp[0];
        d1 = dp[1];
        p.skip = raw < 0 ? 1 : 0;
    } else {
        p.bad = 1;
        p.skip = 1;
    }
    frustum_point(F, 0, &p, out + (size_t)b * M + i, projXR ? projXR + (size_t)b * M + i : nullptr);
    uint4* dd = reinterpret_cast<uint4*>(descOut + ((size_t)b * M + i) * 32);
    dd[0] = d0;
    dd[1] = d1;
}

__global__ __launch_bounds__(256) void map_scatter_kernel(int n, const int* __restrict__ ids, const orbfe_world_point* __restrict__ pts,
                                                          const uint8_t* __restrict__ desc, int mapCap,
                                                          orbfe_world_point* __restrict__ mapPts, uint8_t* __restrict__ mapDesc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int id = ids[i];
    if (id < 0 || id >= mapCap) return;  // validated on the host; never written
    mapPts[id] = pts[i];
    const uint4* sp = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
    uint4* dp = reinterpret_cast<uint4*>(mapDesc + (size_t)id * 32);
    dp[0] = sp[0];
    dp[1] = sp[1];
}

}  // namespace

int frustum_validate(const orbfe_frustum* F)
{
    if (!F) return ORBFE_ERR_INVALID_ARG;
    if (F->camera_model != ORBFE_CAMERA_PINHOLE && F->camera_model != ORBFE_CAMERA_KANNALA_BRANDT8) return ORBFE_ERR_UNSUPPORTED;
    if (F->n_levels < 1 || !(F->log_scale_factor > 0.0f)) return ORBFE_ERR_INVALID_ARG;
    return ORBFE_OK;
}

int frustum_launch(hipStream_t s, const orbfe_frustum* F, int n, const orbfe_world_point* dPts, orbfe_map_point* dOut,
                   float* dProjXR, std::string& err)
{
    if (n == 0) return ORBFE_OK;
    hipLaunchKernelGGL(frustum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *F, n, dPts, dOut, dProjXR);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        err = std::string("frustum_kernel: ") + hipGetErrorString(e);
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

int frustum_launch_dev(hipStream_t s, const orbfe_frustum* dF, int n, const orbfe_world_point* dPts, orbfe_map_point* dOut,
                       float* dProjXR, std::string& err)
{
    if (n == 0) return ORBFE_OK;
    hipLaunchKernelGGL(frustum_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dF, n, dPts, dOut, dProjXR);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        err = std::string("frustum_dev_kernel: ") + hipGetErrorString(e);
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

int frustum_gather_launch(hipStream_t s, int B, const orbfe_frustum* dF, const int* dIds, int M, int mapCap,
                          const orbfe_world_point* mapPts, const uint8_t* mapDesc, orbfe_map_point* dOut, uint8_t* dDescOut,
                          float* dProjXR, std::string& err)
{
    if (B == 0 || M == 0) return ORBFE_OK;
    hipLaunchKernelGGL(frustum_gather_kernel, dim3((M + 255) / 256, B), dim3(256), 0, s, dF, dIds, M, mapCap, mapPts, mapDesc, dOut, dDescOut,
                       dProjXR);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        err = std::string("frustum_gather_kernel: ") + hipGetErrorString(e);
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

int map_scatter_launch(hipStream_t s, int n, const int* dIds, const orbfe_world_point* dPts, const uint8_t* dDesc, int mapCap,
                       orbfe_world_point* mapPts, uint8_t* mapDesc, std::string& err)
{
    if (n == 0) return ORBFE_OK;
    hipLaunchKernelGGL(map_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, dIds, dPts, dDesc, mapCap, mapPts, mapDesc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        err = std::string("map_scatter_kernel: ") + hipGetErrorString(e);
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

}  // namespace orbfe
