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

// The kernel is bound by VALU issue (431 vector instructions per keypoint in its one-wave-per-keypoint form, ~94 % of the
// measured issue rate), so it is organised around the instruction count:
//   * a wave owns kKpPerWave = 16 consecutive storage slots of a frame.  Everything that does not depend on the keypoint
//     -- the lane's four pattern pairs as floats, the disc weights of its orientation column -- is set up once per wave;
//   * the S5 sequences (atan2, degrees, cos / sin: ~80 instructions) run ONCE for the 16 keypoints, lane k working on
//     keypoint k, instead of 64 identical copies per keypoint;
//   * orientation moments: the lane's 16 column samples are packed four to a dword and reduced with v_dot4_u32_u8
//     against per-lane weight bytes (1 / 0 for the column sum, v / 0 for m01: the disc and the row-0 rule live in the
//     weights) -- 8 dot products instead of 48 compare / select / multiply-add instructions;
//   * rotation of the sample offsets as float PAIRS (both points of a pattern pair per v_pk_mul / v_pk_add: every
//     element is still one IEEE binary32 operation, no contraction), rounding half-to-even by adding 1.5 * 2^23 (the
//     integer lands in the low mantissa bits; exactly rintf for |v| < 2^22), LDS tile with a 64-byte pitch.
constexpr int kKpPerWave = 8;
constexpr int kTileRows = 39, kTileDw = 12, kTilePitch = 64;  // staged BRIEF window: 39 rows x 48 bytes, LDS pitch 64 B
constexpr int kStageIters = (kTileRows * kTileDw + 63) / 64;

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
    __shared__ __attribute__((aligned(16))) uint8_t sTile[4][kTileRows * kTilePitch];
    __shared__ __attribute__((aligned(16))) uint8_t sPat[4][31 * 40 + 24];  // orientation patches (wave-private)

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
    const int wvb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nL = P->nLevels;
    const int capF = P->kpCapFrame;
    const uint32_t* cnt = counters + (size_t)f * nL * kCntWords;

    // ---- slots of this wave: lane k < 16 describes storage slot s0 + k (level, position among the kept keypoints) ----
    // The level of a storage slot is static (table in the kernel arguments), the count of its own level says whether the
    // slot is occupied, the counts of the lower levels give the output position (levels are concatenated in order:
    // keypointsAcc, ORBextractor.cc:499-500).
    const int s0 = (sb * 4 + wvb) * kKpPerWave;
    const int sslot = s0 + (lane & (kKpPerWave - 1));
    const uint32_t kwv = lvlKp[(size_t)f * capF + min(sslot, capF - 1)];
    int lvv = 0;
    for (int l = 1; l < nL; l++) lvv = sslot >= tab.base[l] ? l : lvv;
    // per-level keypoint counts: ONE vector load (lane l holds level l), then broadcasts -- a loop of scalar loads is a
    // chain of nL dependent memory round trips in front of every wave's work
    const int cvec = lane < nL ? (int)cnt[lane * kCntWords + kCntKp] : 0;
    int before = 0, total = 0, mine = 0, baseLv = 0;
    for (int l = 0; l < nL; l++) {
        const int c = __builtin_amdgcn_readlane(cvec, l);
        before += l < lvv ? c : 0;
        mine = l == lvv ? c : mine;
        baseLv = l == lvv ? tab.base[l] : baseLv;
        total += c;
    }
    const bool occupied = sslot < capF && sslot - baseLv < mine;
    const int outIdx = before + (sslot - baseLv);  // output index within the frame
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
    const unsigned long long occMask =
        __builtin_amdgcn_ballot_w64(occupied) & (kKpPerWave == 64 ? ~0ull : ((1ull << (kKpPerWave & 63)) - 1ull));
    if (occMask == 0) return;  // wave-uniform

    // ---- per-lane constants ----
    // pattern pairs q*64 + lane, q = 0..3: (x0, x1) and (y0, y1) as float pairs
    f32x2 px[4], py[4];
    {
        const int32_t* pat32 = reinterpret_cast<const int32_t*>(c_pattern);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int32_t pw = pat32[q * 64 + lane];
            px[q] = f32x2{(float)(int8_t)(pw & 0xff), (float)(int8_t)((pw >> 16) & 0xff)};
            py[q] = f32x2{(float)(int8_t)((pw >> 8) & 0xff), (float)(int8_t)(pw >> 24)};
        }
    }
    // orientation column of the lane: lanes 0..30 take column u = lane - 15 of rows +v, lanes 32..62 the same column of rows
    // -v; the patch is a disc: column u is active in rows v <= c_umax[|u|] (symmetric by construction,
    // src/ORBextractor.cc:126-147); lane 31 / 63 (u == 16) is idle, and row 0 belongs to the upper half only
    const int u = (lane & 31) - kHalfPatch;
    const int sgn = (lane < 32) ? 1 : -1;
    const int vmax = abs(u) <= kHalfPatch ? c_umax[abs(u)] : -1;
    uint32_t wSum[4], wMom[4];  // weight bytes of rows 4g .. 4g+3
#pragma unroll
    for (int g = 0; g < 4; g++) {
        wSum[g] = wMom[g] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = 4 * g + i;
            const bool in = v <= vmax && !(v == 0 && sgn < 0);
            wSum[g] |= (in ? 1u : 0u) << (8 * i);
            wMom[g] |= (in ? (uint32_t)v : 0u) << (8 * i);
        }
    }

    // level parameters are wave-uniform and change at most a few times over 16 consecutive slots: (re)loaded on change
    int curLv = -1, w = 0, h = 0, ipitch = 0, bpitch = 0, scaledPatch = 0;
    const uint8_t* img = nullptr;
    const uint8_t* blur = nullptr;
    auto set_level = [&](int lv) {
        const LevelDesc& L = P->lv[lv];
        w = L.w;
        h = L.h;
        bpitch = L.pitch;
        scaledPatch = L.scaledPatch;
        if (lv == 0) {
            img = gray0 + (size_t)f * gray0FrameStride;
            ipitch = gray0Pitch;
        } else {
            img = ws + L.imgOff + (size_t)f * L.imgFrameStride;
            ipitch = L.pitch;
        }
        blur = ws + L.blurOff + (size_t)f * L.blurFrameStride;
        curLv = lv;
    };

    // ---- phase 1: integer moments of the 16 keypoints; lane k keeps (m01, m10) of keypoint k ----
    // The loads of the NEXT occupied slot are requested before the current one is reduced: a wave is a serial chain of 16
    // keypoints, and without the look-ahead every keypoint would wait for its own memory round trip.
    int m01v = 0, m10v = 0;
    // Interior keypoints: the 31 x 31 patch (rows y-15 .. y+15, the dword-aligned 40 bytes from column (x-15) & ~3) is
    // fetched with FIVE coalesced dword loads and read back from a wave-private LDS tile as bytes; sixteen byte loads per
    // lane straight from global memory keep the texture path busy four times as long (the kernel ran at the L1's access
    // rate).  Border keypoints take the direct, reflected byte loads (S3).
    constexpr int kPatRows = 31, kPatDw = 10, kPatPitch = 4 * kPatDw, kPatIters = (kPatRows * kPatDw + 63) / 64;
    struct Patch {
        uint32_t stg[kPatIters];
        uint32_t vals[kHalfPatch + 1];
        int colOff;  // LDS byte of (row 15, the lane's column)
        bool inner;
    };
    auto request_patch = [&](int k, Patch& Q) {
        const uint32_t kw = (uint32_t)__builtin_amdgcn_readlane((int)kwv, k);
        const int lv = __builtin_amdgcn_readlane(lvv, k);
        if (lv != curLv) set_level(lv);
        const int x = cand_x(kw), y = cand_y(kw);
        // Keypoints at least 19 px inside the level (the rotated BRIEF pattern reaches 18.4 px, the orientation patch
        // 16) never touch the border: their sample addresses need no reflection.
        Q.inner = x >= 19 && x < w - 19 && y >= 19 && y < h - 19;  // (the 40-byte rows may run a few bytes into the next row: rows y+16.. exist)
        if (Q.inner) {
            const int xs = (x - 15) & ~3;
            Q.colOff = kHalfPatch * kPatPitch + (x - xs) + u;
            // element e = it * 64 + lane -> (row e / 10, dword e % 10); e + 64 is six rows and four dwords further
            int r = (int)(__umul24((unsigned)lane, 52429u) >> 19);  // lane / 10
            int d = lane - r * kPatDw;
            uint32_t off = __umul24((unsigned)(y - kHalfPatch + r), (unsigned)ipitch) + (unsigned)(xs + 4 * d);
            const uint32_t rowStep = 6u * (unsigned)ipitch + 16u, wrapStep = (unsigned)ipitch - 40u;
            const bool al = (ipitch & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3u) == 0;  // level 0 may be handed over unaligned
#pragma unroll
            for (int it = 0; it < kPatIters; it++) {
                Q.stg[it] = 0;
                if (it * 64 + lane < kPatRows * kPatDw) {
                    if (al) Q.stg[it] = *reinterpret_cast<const uint32_t*>(img + off);
                    else Q.stg[it] = (uint32_t)img[off] | ((uint32_t)img[off + 1] << 8) | ((uint32_t)img[off + 2] << 16) | ((uint32_t)img[off + 3] << 24);
                }
                d += 4;
                off += rowStep;
                if (d >= kPatDw) {
                    d -= kPatDw;
                    off += wrapStep;
                }
            }
        } else {
            // every level is >= 16 px and keypoints sit >= 6 px inside, so one reflection suffices (S3)
            const int cx = reflect_near(x + u, w);
#pragma unroll
            for (int v = 0; v <= kHalfPatch; v++) Q.vals[v] = img[(size_t)reflect_near(y + sgn * v, h) * ipitch + cx];
        }
    };
    uint32_t patOff[kPatIters];  // LDS offsets of the lane's staged dwords (the same for every keypoint)
    {
        int r = (int)(__umul24((unsigned)lane, 52429u) >> 19);
        int d = lane - r * kPatDw;
        uint32_t lo = (uint32_t)(r * kPatPitch + 4 * d);
#pragma unroll
        for (int it = 0; it < kPatIters; it++) {
            patOff[it] = lo;
            d += 4;
            lo += 6 * kPatPitch + 16;
            if (d >= kPatDw) {
                d -= kPatDw;
                lo += kPatPitch - 40;
            }
        }
    }
    uint8_t* const ptile = sPat[wvb];
    {
        unsigned long long todo = occMask;
        Patch next;
        request_patch(__builtin_ctzll(todo), next);
#pragma unroll 1
        while (todo) {  // wave-uniform
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            const Patch cur = next;
            if (cur.inner) {
#pragma unroll
                for (int it = 0; it < kPatIters; it++)
                    if (it * 64 + lane < kPatRows * kPatDw) *reinterpret_cast<uint32_t*>(ptile + patOff[it]) = cur.stg[it];
            }
            if (todo) request_patch(__builtin_ctzll(todo), next);
            uint32_t vals[kHalfPatch + 1];
            if (cur.inner) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: LDS ops of a wave run in order
                const uint8_t* col = ptile + cur.colOff;
#pragma unroll
                for (int v = 0; v <= kHalfPatch; v++) vals[v] = col[sgn * v * kPatPitch];
            } else {
#pragma unroll
                for (int v = 0; v <= kHalfPatch; v++) vals[v] = cur.vals[v];
            }
            uint32_t sumv = 0, mom = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t pk = vals[4 * g] | (vals[4 * g + 1] << 8) | (vals[4 * g + 2] << 16) | (vals[4 * g + 3] << 24);
                sumv = __builtin_amdgcn_udot4(pk, wSum[g], sumv, false);
                mom = __builtin_amdgcn_udot4(pk, wMom[g], mom, false);
            }
            const int m10 = wave_sum(u * (int)sumv);
            const int m01 = wave_sum(sgn * (int)mom);
            m10v = lane == k ? m10 : m10v;
            m01v = lane == k ? m01 : m01v;
        }
    }

    // ---- phase 2: S5 once for the wave -- lane k: angle, cos, sin of keypoint k ----
    const float anglev = atan2_deg((float)m01v, (float)m10v);
    float av, bv;
    cos_sin_deg(anglev, av, bv);

    // ---- phase 3: steered BRIEF per keypoint; lane k collects the 8 descriptor dwords of keypoint k ----
    // The 512 rotated sample positions of a keypoint fall in a 39 x 39 window (reach <= 18.4 px).  For interior
    // keypoints the window is staged into a wave-private LDS tile with eight coalesced dword loads (39 rows x 12
    // dwords) and the 8 samples per lane become LDS byte reads: a 64-lane byte gather from global memory touches
    // ~30 cache lines per instruction, the staged rows ~6.  Border keypoints keep the direct (reflected) path.
    uint32_t dsc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t* const tile = sTile[wvb];
    const float kMagic = 12582912.0f;  // 1.5 * 2^23: v + kMagic has rint(v) in its low mantissa bits (|v| < 2^22)
    // window of slot k: requested one keypoint ahead (registers), written to the wave's LDS tile when its turn comes
    struct Win {
        uint32_t stg[kStageIters];
        int x, y, xs;
        bool inner, staged;
    };
    auto request = [&](int k, Win& W) {
        const uint32_t kw = (uint32_t)__builtin_amdgcn_readlane((int)kwv, k);
        const int lv = __builtin_amdgcn_readlane(lvv, k);
        if (lv != curLv) set_level(lv);
        W.x = cand_x(kw);
        W.y = cand_y(kw);
        W.inner = W.x >= 19 && W.x < w - 19 && W.y >= 19 && W.y < h - 19;
        W.xs = (W.x - 19) & ~3;                                       // dword-aligned left edge of the staged rows
        W.staged = W.inner && W.xs + 4 * kTileDw <= bpitch;           // the blurred levels live in the workspace: 64-B aligned rows
        if (W.staged) {
            // Element e = it * 64 + lane of the 39 x 12 dword window sits at (row e / 12, dword e % 12); e + 64 is five rows
            // and four dwords further, so the (row, dword) pair and the offset advance incrementally.  32-bit unsigned
            // offsets from the (wave-uniform) level base: scalar-base + 32-bit-offset loads.
            int r = (int)(__umul24((unsigned)lane, 43691u) >> 19);  // lane / 12
            int d = lane - r * kTileDw;
            uint32_t off = __umul24((unsigned)(W.y - 19 + r), (unsigned)bpitch) + (unsigned)(W.xs + 4 * d);
            const uint32_t rowStep = 5u * (unsigned)bpitch + 16u, wrapStep = (unsigned)bpitch - 48u;
#pragma unroll
            for (int it = 0; it < kStageIters; it++) {
                W.stg[it] = 0;
                if (it * 64 + lane < kTileRows * kTileDw) W.stg[it] = *reinterpret_cast<const uint32_t*>(blur + off);
                d += 4;
                off += rowStep;
                if (d >= kTileDw) {
                    d -= kTileDw;
                    off += wrapStep;
                }
            }
        }
    };
    // LDS offsets of the lane's staged dwords (the same for every keypoint)
    uint32_t ldsOff[kStageIters];
    {
        int r = (int)(__umul24((unsigned)lane, 43691u) >> 19);
        int d = lane - r * kTileDw;
        uint32_t lo = (uint32_t)(r * kTilePitch + 4 * d);
#pragma unroll
        for (int it = 0; it < kStageIters; it++) {
            ldsOff[it] = lo;
            d += 4;
            lo += 5 * kTilePitch + 16;
            if (d >= kTileDw) {
                d -= kTileDw;
                lo += kTilePitch - 48;
            }
        }
    }
    {
        unsigned long long todo = occMask;
        Win next;
        request(__builtin_ctzll(todo), next);
#pragma unroll 1
        while (todo) {  // wave-uniform
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            const Win cur = next;
            // the blurred level this keypoint samples (border path reads it directly): remember before the look-ahead moves on
            const uint8_t* const blurK = blur;
            const int bpitchK = bpitch, wK = w, hK = h;
            if (cur.staged) {
#pragma unroll
                for (int it = 0; it < kStageIters; it++)
                    if (it * 64 + lane < kTileRows * kTileDw) *reinterpret_cast<uint32_t*>(tile + ldsOff[it]) = cur.stg[it];
            }
            if (todo) request(__builtin_ctzll(todo), next);
            if (cur.staged) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: LDS ops of a wave run in order
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, av), k));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), k));
            const int cOff = (cur.x - 19) - cur.xs + 19;  // LDS column of the keypoint
            const f32x2 a2 = f32x2{a, a}, b2 = f32x2{b, b}, m2 = f32x2{kMagic, kMagic};
            uint32_t t0[4], t1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {  // 8 independent sample loads in flight
                // row = rint(x * b + y * a), column = rint(x * a - y * b) (Orb_gpu.cu:313-314), both points of the pair at once
                const f32x2 rb = px[q] * b2, ra = py[q] * a2;
                const f32x2 ca = px[q] * a2, cb = py[q] * b2;
                const f32x2 rr = rb + ra, cc = ca - cb;
                const f32x2 rm = rr + m2, cm = cc + m2;  // low mantissa bits: the rounded integers (two's complement)
                // 0x4B400000 + n: the low 16 bits are n = rint(v) in two's complement (|n| <= 19)
                const float rmx = rm.x, rmy = rm.y, cmx = cm.x, cmy = cm.y;
                const int dya = (int)(short)(__float_as_uint(rmx) & 0xffffu), dyb = (int)(short)(__float_as_uint(rmy) & 0xffffu);
                const int dxa = (int)(short)(__float_as_uint(cmx) & 0xffffu), dxb = (int)(short)(__float_as_uint(cmy) & 0xffffu);
                if (cur.staged) {
                    t0[q] = tile[(dya + 19) * kTilePitch + dxa + cOff];
                    t1[q] = tile[(dyb + 19) * kTilePitch + dxb + cOff];
                } else {
                    int ya = cur.y + dya, xa = cur.x + dxa, yb = cur.y + dyb, xb = cur.x + dxb;
                    if (!cur.inner) {
                        ya = reflect_near(ya, hK); xa = reflect_near(xa, wK);
                        yb = reflect_near(yb, hK); xb = reflect_near(xb, wK);
                    }
                    t0[q] = blurK[ya * bpitchK + xa];  // level pixels < 2^24: 32-bit offsets
                    t1[q] = blurK[yb * bpitchK + xb];
                }
            }
            // bit k of byte t == pair 8t + k (Orb_gpu.cu:331-349): each ballot is 8 descriptor bytes
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long bits = __builtin_amdgcn_ballot_w64(t0[q] < t1[q]);
                dsc[2 * q] = lane == k ? (uint32_t)bits : dsc[2 * q];
                dsc[2 * q + 1] = lane == k ? (uint32_t)(bits >> 32) : dsc[2 * q + 1];
            }
        }
    }

    // ---- outputs: lane k writes the 24-byte keypoint (6 dwords) and the 32-byte descriptor of its keypoint ----
    if (lane < kKpPerWave && occupied) {
        const LevelDesc& Lk = P->lv[lvv];
        uint32_t* ko = reinterpret_cast<uint32_t*>(kpOut + (size_t)f * capF + outIdx);
        ko[0] = __float_as_uint((float)cand_x(kwv));
        ko[1] = __float_as_uint((float)cand_y(kwv));
        ko[2] = (uint32_t)cand_score(kwv);
        ko[3] = __float_as_uint((float)Lk.scaledPatch);
        ko[4] = (uint32_t)lvv;
        ko[5] = __float_as_uint(anglev);
        uint4* d = reinterpret_cast<uint4*>(descOut + ((size_t)f * capF + outIdx) * 32);
        d[0] = make_uint4(dsc[0], dsc[1], dsc[2], dsc[3]);
        d[1] = make_uint4(dsc[4], dsc[5], dsc[6], dsc[7]);
    }
    (void)scaledPatch;
}

void launch_orient_brief(hipStream_t s, int frames, int kpCapFrame, const PipelineDesc* dP, const uint8_t* gray0,
                         size_t gray0FrameStride, int gray0Pitch, const uint8_t* ws, const uint32_t* counters,
                         const uint32_t* lvlKp, orbfe_keypoint* kpOut, uint8_t* descOut, int* nOut,
                         int* perLevelOut, int* statusOut, const int* kpBase, int nLevels)
{
    KpBaseTab tab{};
    for (int l = 0; l < nLevels; l++) tab.base[l] = kpBase[l];
    dim3 block(256);
    const int slotBlocks = (kpCapFrame + 4 * kKpPerWave - 1) / (4 * kKpPerWave);  // a block = 4 waves x 16 storage slots
    dim3 grid((unsigned)((frames >= 8 ? ((frames + 7) / 8) * 8 : frames) * slotBlocks));
    hipLaunchKernelGGL(orient_brief_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch, ws,
                       counters, lvlKp, kpOut, descOut, nOut, perLevelOut, statusOut, frames, slotBlocks, tab);
}

}  // namespace orbfe
