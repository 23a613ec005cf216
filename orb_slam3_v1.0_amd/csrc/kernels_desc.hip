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
//    __ballot() is 8 descriptor bytes (bit k of byte t == pair 8t+k, Orb_gpu.cu:331-349).  The 39 x 39
//    sample window of an interior keypoint is staged in a wave-private LDS tile first.
//  * samples outside the level image follow BORDER_REFLECT_101 (SPEC DECISION S3).
//  * atan2 / cos / sin: SPEC DECISION S5 (device_math.h); rounding of the rotated sample offsets is
//    round-half-even (__float2int_rn == rintf, Orb_gpu.cu:313-314), no FMA contraction.
#include "launch.h"
#include "device_math.h"

#pragma clang fp contract(off)

namespace orbfe {

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// wave64 sum with DPP row operations (one VALU instruction per step instead of an LDS permute + add):
// quad swaps, row half-mirror and mirror leave every lane of a 16-lane row with the row total; row_bcast 15 / 31
// accumulate the rows into lane 63, which is read back as a scalar.
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// first storage slot of every level in the per-frame level-keypoint array (LevelDesc::kpBase), by value in the
// kernel arguments so that a wave knows its level without touching memory
struct KpBaseTab {
    int base[kMaxLevels + 1];
};

__global__ __launch_bounds__(256) void orient_brief_kernel(const PipelineDesc* __restrict__ P,
                                                           const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                           int gray0Pitch, const uint8_t* __restrict__ ws,
                                                           const uint32_t* __restrict__ counters,
                                                           const uint32_t* __restrict__ lvlKp,
                                                           orbfe_keypoint* __restrict__ kpOut,
                                                           uint8_t* __restrict__ descOut, int* __restrict__ nOut,
                                                           int* __restrict__ perLevelOut, int* __restrict__ statusOut,
                                                           int frames, int slotBlocks, KpBaseTab tab)
{
    // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, so XCD x gets linear ids x, x+8, ...;
    // give it the frames x, x+8, ... ONE AFTER THE OTHER (all keypoint blocks of a frame are consecutive on its
    // XCD): the two level images a frame's patches gather from (about 2 MB) then stay in that XCD's 4 MB L2 while
    // its ~1000 keypoints are processed, instead of 32 frames thrashing it.
    // (fewer than 8 frames: plain frame-major order, no idle blocks)
    const int G = frames >= 8 ? 8 : 1;
    const int xcd = G == 8 ? (int)(blockIdx.x & 7) : 0;
    const int q8 = G == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int fgrp = q8 / slotBlocks;
    const int sb = q8 - fgrp * slotBlocks;
    const int f = fgrp * G + xcd;
    if (f >= frames) return;
    const int lane = threadIdx.x & 63;
    // One wave per STORAGE slot of the per-frame level-keypoint array: its level is static (table in the kernel
    // arguments), so the keypoint word is requested at once, in parallel with the per-level counts; the count of its
    // own level says whether the slot is occupied, the counts of the lower levels give the output position
    // (levels are concatenated in order: keypointsAcc, ORBextractor.cc:499-500).  Two dependent memory round trips
    // lead to the pixel loads instead of four.
    const int sslot = __builtin_amdgcn_readfirstlane(sb * 4 + (int)(threadIdx.x >> 6));
    const int nL = P->nLevels;
    const uint32_t kw = lvlKp[(size_t)f * P->kpCapFrame + min(sslot, P->kpCapFrame - 1)];
    const uint32_t* cnt = counters + (size_t)f * nL * kCntWords;
    // the first 8 levels with constant indices: one s_load for the table, 8 independent scalar loads for the counts
    // (a runtime-bounded loop would wait for every load in turn); deeper pyramids finish in the loops below
    int lv = 0;
#pragma unroll
    for (int l = 1; l < 8; l++) lv = (l < nL && sslot >= tab.base[l]) ? l : lv;
    for (int l = 8; l < nL; l++) lv = sslot >= tab.base[l] ? l : lv;
    int c8[8];
#pragma unroll
    for (int l = 0; l < 8; l++) c8[l] = l < nL ? (int)cnt[l * kCntWords + kCntKp] : 0;
    int before = 0, total = 0, mine = 0, baseLv = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
        before += l < lv ? c8[l] : 0;
        mine = l == lv ? c8[l] : mine;
        baseLv = l == lv ? tab.base[l] : baseLv;
        total += c8[l];
    }
    for (int l = 8; l < nL; l++) {
        const int c = (int)cnt[l * kCntWords + kCntKp];
        before += l < lv ? c : 0;
        mine = l == lv ? c : mine;
        baseLv = l == lv ? tab.base[l] : baseLv;
        total += c;
    }
    const int j = sslot - baseLv;
    if (sb == 0 && threadIdx.x == 0) {
        nOut[f] = total;
        if (perLevelOut)
            for (int q = 0; q < nL; q++) perLevelOut[(size_t)f * nL + q] = (int)cnt[q * kCntWords + kCntKp];
        if (statusOut) {  // device-side guard flags of all levels, so the host path needs no copy of the counters
            uint32_t st = 0;
            for (int q = 0; q < nL; q++) st |= cnt[q * kCntWords + kCntStatus];
            statusOut[f] = (int)st;
        }
    }
    if (sslot >= P->kpCapFrame || j >= mine) return;
    const int slot = before + j;  // output index within the frame

    // the four pattern dwords of this lane do not depend on the keypoint: request them now
    int32_t pwq[4];
    {
        const int32_t* pat32 = reinterpret_cast<const int32_t*>(c_pattern);
#pragma unroll
        for (int q = 0; q < 4; q++) pwq[q] = pat32[q * 64 + lane];
    }
    const LevelDesc& L = P->lv[lv];
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

    // Keypoints at least 19 px inside the level (the rotated BRIEF pattern reaches 18.4 px, the orientation patch
    // 16) never touch the border: their sample addresses need no reflection.  One keypoint per wave, so the choice
    // is wave-uniform.
    const bool inner = x >= 19 && x < w - 19 && y >= 19 && y < h - 19;

    // the 39 x 39 BRIEF window does not depend on the angle: its rows are requested here, together with the
    // orientation patch, and land in LDS after the angle is known
    constexpr int kPatchRows = 39, kPatchDw = 12, kStageIters = (kPatchRows * kPatchDw + 63) / 64;
    __shared__ uint32_t sPatch[4][kPatchRows * kPatchDw];
    const int wvb = threadIdx.x >> 6;
    const int xs = (x - 19) & ~3;                       // dword-aligned left edge of the staged rows
    const bool staged = inner && xs + 4 * kPatchDw <= bpitch;  // the blurred levels live in the workspace: 64-B aligned rows
    uint32_t stg[kStageIters];
    if (staged) {
        // 32-bit unsigned offsets from the (wave-uniform) level base (rows and pitches are < 2^24, a level is < 2^32
        // bytes): scalar-base + 32-bit-offset loads, no 64-bit address arithmetic.  Element e = it * 64 + lane of the
        // 39 x 12 dword window sits at (row e / 12, dword e % 12); e + 64 is five rows and four dwords further, so the
        // (row, dword) pair and the offset advance incrementally instead of a division per iteration.
        int r = (int)(__umul24((unsigned)lane, 43691u) >> 19);  // lane / 12
        int d = lane - r * kPatchDw;
        uint32_t off = __umul24((unsigned)(y - 19 + r), (unsigned)bpitch) + (unsigned)(xs + 4 * d);
        const uint32_t rowStep = 5u * (unsigned)bpitch + 16u, wrapStep = (unsigned)bpitch - 48u;
#pragma unroll
        for (int it = 0; it < kStageIters; it++) {
            stg[it] = 0;
            if (it * 64 + lane < kPatchRows * kPatchDw) stg[it] = *reinterpret_cast<const uint32_t*>(blur + off);
            d += 4;
            off += rowStep;
            if (d >= kPatchDw) {
                d -= kPatchDw;
                off += wrapStep;
            }
        }
    }

    // ---- IC angle ----
    const int u = (lane & 31) - kHalfPatch;   // -15..16 (16 == idle lane 31/63)
    const int sgn = (lane < 32) ? 1 : -1;
    int vals[kHalfPatch + 1];
    if (inner) {
        uint32_t idx = __umul24((unsigned)y, (unsigned)ipitch) + (unsigned)(x + u);  // byte offset inside the level
        const uint32_t step = (uint32_t)(sgn * ipitch);                             // +-pitch, modulo 2^32
#pragma unroll
        for (int v = 0; v <= kHalfPatch; v++) {  // 16 independent loads in flight
            vals[v] = img[idx];
            idx += step;
        }
    } else {
        // every level is >= 16 px and keypoints sit >= 6 px inside, so one reflection suffices
        const int cx = reflect_near(x + u, w);
#pragma unroll
        for (int v = 0; v <= kHalfPatch; v++)
            vals[v] = img[(size_t)reflect_near(y + sgn * v, h) * ipitch + cx];
    }
    // the patch is a disc: column u is active in rows v <= c_umax[|u|] (the table is symmetric by construction,
    // src/ORBextractor.cc:126-147); lane 31/63 (u == 16) is idle, and row 0 belongs to the upper half only
    const int au = abs(u);
    const int vmax = au <= kHalfPatch ? c_umax[au] : -1;
    // row 0 (upper half only) contributes to the column sum but not to m01; rows >= 1 need the disc test alone
    int sumv = (sgn > 0 && vmax >= 0) ? vals[0] : 0, m01 = 0;
#pragma unroll
    for (int v = 1; v <= kHalfPatch; v++) {
        const int val = v <= vmax ? vals[v] : 0;
        sumv += val;
        m01 += v * val;
    }
    const int m10 = wave_sum(u * sumv);
    m01 = wave_sum(sgn * m01);
    const float angle = atan2_deg((float)m01, (float)m10);

    // ---- steered BRIEF ----
    // The 512 rotated sample positions of a keypoint fall in a 39 x 39 window (reach <= 18.4 px).  For interior
    // keypoints the window is staged into a wave-private LDS tile with eight coalesced dword loads (39 rows x 12
    // dwords) and the 8 samples per lane become LDS byte reads: a 64-lane byte gather from global memory touches
    // ~30 cache lines per instruction, the staged rows ~6.  Border keypoints keep the direct (reflected) path.
    float a, b;
    cos_sin_deg(angle, a, b);
    if (staged) {
#pragma unroll
        for (int it = 0; it < kStageIters; it++) {
            const int e = it * 64 + lane;
            if (e < kPatchRows * kPatchDw) sPatch[wvb][e] = stg[it];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // wave-private tile: LDS ops of a wave run in order
    }
    const uint8_t* pbytes = reinterpret_cast<const uint8_t*>(sPatch[wvb]);
    const int cOff = (x - 19) - xs + 19;                // LDS column of the keypoint
    int t0[4], t1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {  // 8 independent sample loads in flight
        const int32_t pw = pwq[q];  // x0, y0, x1, y1 of pattern pair q*64+lane as one dword (requested at kernel entry)
        const float x0 = (float)(int8_t)(pw & 0xff), y0 = (float)(int8_t)((pw >> 8) & 0xff);
        const float x1 = (float)(int8_t)((pw >> 16) & 0xff), y1 = (float)(int8_t)(pw >> 24);
        float r0 = x0 * b; const float r0b = y0 * a; r0 = r0 + r0b;
        float c0 = x0 * a; const float c0b = y0 * b; c0 = c0 - c0b;
        float r1 = x1 * b; const float r1b = y1 * a; r1 = r1 + r1b;
        float c1 = x1 * a; const float c1b = y1 * b; c1 = c1 - c1b;
        const int dya = __float2int_rn(r0), dxa = __float2int_rn(c0);
        const int dyb = __float2int_rn(r1), dxb = __float2int_rn(c1);
        if (staged) {
            t0[q] = pbytes[(dya + 19) * (4 * kPatchDw) + dxa + cOff];
            t1[q] = pbytes[(dyb + 19) * (4 * kPatchDw) + dxb + cOff];
        } else {
            int ya = y + dya, xa = x + dxa, yb = y + dyb, xb = x + dxb;
            if (!inner) {
                ya = reflect_near(ya, h); xa = reflect_near(xa, w);
                yb = reflect_near(yb, h); xb = reflect_near(xb, w);
            }
            t0[q] = blur[ya * bpitch + xa];  // level pixels < 2^24: 32-bit offsets
            t1[q] = blur[yb * bpitch + xb];
        }
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
                         int* perLevelOut, int* statusOut, const int* kpBase, int nLevels)
{
    KpBaseTab tab{};
    for (int l = 0; l < nLevels; l++) tab.base[l] = kpBase[l];
    dim3 block(256);
    const int slotBlocks = (kpCapFrame + 3) / 4;
    dim3 grid((unsigned)((frames >= 8 ? ((frames + 7) / 8) * 8 : frames) * slotBlocks));
    hipLaunchKernelGGL(orient_brief_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch, ws,
                       counters, lvlKp, kpOut, descOut, nOut, perLevelOut, statusOut, frames, slotBlocks, tab);
}

}  // namespace orbfe
