// kernels_desc.hip -- intensity-centroid orientation + 256-bit steered BRIEF, gfx950.
//
// Replaces IC_Angle_kernel (src/cuda/Angle_gpu.cu:26-80, launched per level :82-88) and
// calcOrb_kernel / getOrbValue (src/cuda/Orb_gpu.cu:311-350, launched per level :372-379), plus
// the KeyPoint fill of ComputeKeyPointsOctTree (src/ORBextractor.cc:505-533).  ONE launch covers
// every level of every frame: one 64-lane wavefront per keypoint (the reference uses 32 threads
// per keypoint, half a CDNA wave).
//
//  * orientation on the UNBLURRED level: lanes 0..30 take column u = lane-15 of row +v, lanes
//    32..62 the same column of row -v; integer moments, wave butterfly reduction (exact).
//  * descriptor on the BLURRED level: lane i evaluates pattern pairs i, 64+i, 128+i, 192+i; each
//    __ballot() is 8 descriptor bytes (bit k of byte t == pair 8t+k, Orb_gpu.cu:331-349).
//  * samples outside the level image follow BORDER_REFLECT_101 (SPEC DECISION S3).
//  * atan2 / cos / sin: SPEC DECISION S5 (device_math.h); rounding of the rotated sample offsets is
//    round-half-even (__float2int_rn == rintf, Orb_gpu.cu:313-314), no FMA contraction.
#include "launch.h"
#include "device_math.h"

#pragma clang fp contract(off)

namespace orbfe {

__constant__ int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(256) void orient_brief_kernel(const PipelineDesc* __restrict__ P,
                                                           const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                           int gray0Pitch, const uint8_t* __restrict__ ws,
                                                           const uint32_t* __restrict__ counters,
                                                           const uint32_t* __restrict__ lvlKp,
                                                           orbfe_keypoint* __restrict__ kpOut,
                                                           uint8_t* __restrict__ descOut, int* __restrict__ nOut,
                                                           int* __restrict__ perLevelOut, int* __restrict__ statusOut)
{
    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);  // output index within the frame
    const int nL = P->nLevels;
    const uint32_t* cnt = counters + (size_t)f * nL * kCntWords;

    // locate (level, j): levels are concatenated in order (keypointsAcc, ORBextractor.cc:499-500)
    int l = 0, base = 0, total = 0;
    bool found = false;
    int lv = 0, j = 0;
    for (l = 0; l < nL; l++) {
        const int c = (int)cnt[l * kCntWords + kCntKp];
        if (!found && slot < total + c) {
            found = true;
            lv = l;
            j = slot - total;
        }
        total += c;
    }
    (void)base;
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        nOut[f] = total;
        if (perLevelOut)
            for (int q = 0; q < nL; q++) perLevelOut[(size_t)f * nL + q] = (int)cnt[q * kCntWords + kCntKp];
        if (statusOut) {  // device-side guard flags of all levels, so the host path needs no copy of the counters
            uint32_t st = 0;
            for (int q = 0; q < nL; q++) st |= cnt[q * kCntWords + kCntStatus];
            statusOut[f] = (int)st;
        }
    }
    if (!found) return;

    const LevelDesc& L = P->lv[lv];
    const uint32_t kw = lvlKp[(size_t)f * P->kpCapFrame + L.kpBase + j];
    const int x = cand_x(kw), y = cand_y(kw), resp = cand_score(kw);
    const int w = L.w, h = L.h;

    const uint8_t* img;
    int ipitch;
    if (lv == 0) {
        img = gray0 + (size_t)f * gray0FrameStride;
        ipitch = gray0Pitch;
    } else {
        img = ws + L.imgOff + (size_t)f * L.imgFrameStride;
        ipitch = L.pitch;
    }
    const uint8_t* blur = ws + L.blurOff + (size_t)f * L.blurFrameStride;
    const int bpitch = L.pitch;

    // ---- IC angle ----
    const int u = (lane & 31) - kHalfPatch;   // -15..16 (16 == idle lane 31/63)
    const int sgn = (lane < 32) ? 1 : -1;
    // every level is >= 16 px and keypoints sit >= 6 px inside, so one reflection suffices
    const int cx = reflect_near(x + u, w);
    int vals[kHalfPatch + 1];
#pragma unroll
    for (int v = 0; v <= kHalfPatch; v++)  // 16 independent loads in flight
        vals[v] = img[(size_t)reflect_near(y + sgn * v, h) * ipitch + cx];
    int m10 = 0, m01 = 0;
    const int au = abs(u);
#pragma unroll
    for (int v = 0; v <= kHalfPatch; v++) {
        const bool act = (au <= c_umax[v]) && !(v == 0 && sgn < 0);  // c_umax <= 15 masks the idle lane (u == 16)
        const int val = act ? vals[v] : 0;
        m10 += u * val;
        m01 += sgn * v * val;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = atan2_deg((float)m01, (float)m10);

    // ---- steered BRIEF ----
    float a, b;
    cos_sin_deg(angle, a, b);
    int t0[4], t1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {  // 8 independent sample loads in flight (rotated reach <= 18.4 px)
        const int pair = q * 64 + lane;
        const int8_t* pt = &c_pattern[pair * 4];
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        float r0 = x0 * b; const float r0b = y0 * a; r0 = r0 + r0b;
        float c0 = x0 * a; const float c0b = y0 * b; c0 = c0 - c0b;
        float r1 = x1 * b; const float r1b = y1 * a; r1 = r1 + r1b;
        float c1 = x1 * a; const float c1b = y1 * b; c1 = c1 - c1b;
        const int ya = reflect_near(y + __float2int_rn(r0), h), xa = reflect_near(x + __float2int_rn(c0), w);
        const int yb = reflect_near(y + __float2int_rn(r1), h), xb = reflect_near(x + __float2int_rn(c1), w);
        t0[q] = blur[(size_t)ya * bpitch + xa];
        t1[q] = blur[(size_t)yb * bpitch + xb];
    }
    unsigned long long bits[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bits[q] = __ballot(t0[q] < t1[q]);

    // ---- outputs: 24-byte keypoint (6 dwords) + 32-byte descriptor (4 qwords) ----
    orbfe_keypoint* ko = kpOut + (size_t)f * P->kpCapFrame + slot;
    if (lane < 6) {
        uint32_t wv;
        switch (lane) {
        case 0: wv = __float_as_uint((float)x); break;
        case 1: wv = __float_as_uint((float)y); break;
        case 2: wv = (uint32_t)resp; break;
        case 3: wv = __float_as_uint((float)L.scaledPatch); break;
        case 4: wv = (uint32_t)lv; break;
        default: wv = __float_as_uint(angle); break;
        }
        reinterpret_cast<uint32_t*>(ko)[lane] = wv;
    }
    if (lane >= 8 && lane < 12) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(descOut + ((size_t)f * P->kpCapFrame + slot) * 32);
        const int q = lane - 8;
        d[q] = q == 0 ? bits[0] : q == 1 ? bits[1] : q == 2 ? bits[2] : bits[3];
    }
}

void launch_orient_brief(hipStream_t s, int frames, int kpCapFrame, const PipelineDesc* dP, const uint8_t* gray0,
                         size_t gray0FrameStride, int gray0Pitch, const uint8_t* ws, const uint32_t* counters,
                         const uint32_t* lvlKp, orbfe_keypoint* kpOut, uint8_t* descOut, int* nOut,
                         int* perLevelOut, int* statusOut)
{
    dim3 block(256);
    dim3 grid(frames, (kpCapFrame + 3) / 4);
    hipLaunchKernelGGL(orient_brief_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch, ws,
                       counters, lvlKp, kpOut, descOut, nOut, perLevelOut, statusOut);
}

}  // namespace orbfe
