// kernels_pyramid.hip -- bilinear pyramid downscale for gfx950 (SPEC DECISION S1).
//
// Replaces cv::cuda::resize(level l-1 -> l, INTER_LINEAR) of ORBextractor::ComputePyramid
// (src/ORBextractor.cc:607-623).  Pure integer: corner-aligned source position, Q11 weights, one
// rounding; defined in oracle/orb_oracle.c (orc_resize_bilinear) and reproduced bit for bit here.
// (The Gaussian of each level is fused into the FAST kernel, kernels_fast.hip.)
//
// Launch geometry: blockIdx.x = frame (fastest-varying so that frames f and f+8 share an XCD and
// every tile of one frame hits the same L2), blockIdx.y/z = tile.
#include <algorithm>
#include <cstdlib>

#include "launch.h"

namespace orbfe {

// Output tile 128 x 16 per 256-thread block (pyramid levels are a few hundred pixels wide: wide tiles waste
// a quarter of their lanes on the right edge).  The source footprint (<= 40 rows x 304 px, i.e. scale
// factors up to ~2.2) is staged in LDS with dword loads whose addresses come from integer arithmetic
// (no dependent table load in front of the staging); the weight tables are fetched meanwhile.
// Every thread produces 2 x 4 pixels and stores two dwords.  xtab/ytab hold (x1 | Q11 weight << 16),
// 16-byte aligned per level.  Algorithmic bytes per output pixel: 1 written + scale^2 read.
constexpr int kRsTW = 128, kRsTH = 16;
constexpr int kRsMaxRows = 40, kRsMaxCols = 304;

// KPASS = staging passes of 4 rows a thread issues (6 covers footprints up to 24 rows, i.e. level ratios up to 1.375;
// the host picks 10 for larger ratios)
template <int KPASS>
__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, size_t srcFrameStride,
                                                     int sw, int sh, int spitch, int srcAligned4,
                                                     uint8_t* __restrict__ dst, size_t dstFrameStride,
                                                     int dw, int dh, int dpitch,
                                                     const uint32_t* __restrict__ xtab,
                                                     const uint32_t* __restrict__ ytab)
{
    __shared__ __attribute__((aligned(16))) uint8_t sSrc[kRsMaxRows][kRsMaxCols];

    const int f = blockIdx.x;
    const int tx0 = blockIdx.y * kRsTW;
    const int ty0 = blockIdx.z * kRsTH;
    const int tid = threadIdx.x;
    const uint8_t* s = src + (size_t)f * srcFrameStride;

    // this thread's outputs: 4 pixels at x0 in rows y and y + 8; table entries requested up front
    const int x0 = tx0 + (tid & 31) * 4;
    const int yA = ty0 + (tid >> 5), yB = yA + 8;
    const bool colOk = x0 < dw;
    uint4 xt4 = make_uint4(0, 0, 0, 0);
    uint32_t ytA = 0, ytB = 0;
    if (colOk) xt4 = *reinterpret_cast<const uint4*>(xtab + x0);  // table padded to a multiple of 4
    if (yA < dh) ytA = ytab[yA];
    if (yB < dh) ytB = ytab[yB];

    // source footprint of the tile
    const int txLast = min(tx0 + kRsTW, dw) - 1;
    const int tyLast = min(ty0 + kRsTH, dh) - 1;
    // (the low half of a table entry IS floor(x * sw / dw): four scalar loads instead of four 64-bit divisions)
    const int sx0 = (int)(xtab[tx0] & 0xffffu) & ~3;                            // dword-aligned left edge
    const int sx1 = min((int)(xtab[txLast] & 0xffffu) + 1, sw - 1);
    const int sy0 = (int)(ytab[ty0] & 0xffffu);
    const int sy1 = min((int)(ytab[tyLast] & 0xffffu) + 1, sh - 1);
    const int nrows = sy1 - sy0 + 1;
    const int ndw = (sx1 - sx0) / 4 + 1;                                          // dwords per staged row
    const bool staged = nrows <= 4 * KPASS && ndw * 4 <= kRsMaxCols;

    if (staged) {
        // thread (c4, r) grid over the footprint: 64 dword columns x 4 rows per pass, no division
        const int c4l = tid & 63, rl = tid >> 6;
        if (srcAligned4 && ndw <= 64) {
            // aligned source (always true for the workspace levels): whole dwords, and ALL loads of a thread are
            // issued before the first LDS store (a load / store pair per loop iteration would serialise the
            // global-memory latency up to ten times).  Bytes past the row end lie inside the pitch and are never
            // addressed by the tables.
            constexpr int kPass = KPASS;
            const bool cok = c4l < ndw;
            // 32-bit offsets from the block-uniform frame base (rows, pitches < 2^24; a level < 2^32 bytes): the loads
            // take the scalar-base + 32-bit-offset form instead of 64-bit address arithmetic per load
            uint32_t off = __umul24((unsigned)(sy0 + rl), (unsigned)spitch) + (unsigned)(sx0 + 4 * c4l);
            const uint32_t step = 4u * (unsigned)spitch;
            uint32_t wv[kPass];
#pragma unroll
            for (int k = 0; k < kPass; k++) {
                wv[k] = 0;
                if (cok && rl + 4 * k < nrows) wv[k] = *reinterpret_cast<const uint32_t*>(s + off);
                off += step;
            }
#pragma unroll
            for (int k = 0; k < kPass; k++)
                if (cok && rl + 4 * k < nrows) *reinterpret_cast<uint32_t*>(&sSrc[rl + 4 * k][4 * c4l]) = wv[k];
        } else {
            for (int r = rl; r < nrows; r += 4) {
                const uint8_t* row = s + (size_t)(sy0 + r) * spitch;
                for (int c4 = c4l; c4 < ndw; c4 += 64) {
                    const int gx = sx0 + 4 * c4;
                    uint32_t wv = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (gx + i < sw) wv |= (uint32_t)row[gx + i] << (8 * i);
                    *reinterpret_cast<uint32_t*>(&sSrc[r][4 * c4]) = wv;
                }
            }
        }
        __syncthreads();
    }
    if (!colOk) return;

    const uint32_t xt[4] = {xt4.x, xt4.y, xt4.z, xt4.w};
    const int nvalid = min(4, dw - x0);
    // two separately compiled bodies: the LDS one must index sSrc directly (a pointer merged with the
    // global fallback would degrade every LDS read to a flat load)
    auto body = [&](auto fetch) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int y = half ? yB : yA;
            if (y >= dh) continue;
            const uint32_t yt = half ? ytB : ytA;
            const int y1 = (int)(yt & 0xffffu);
            const uint32_t wy = yt >> 16;
            const int y2 = min(y1 + 1, sh - 1);
            uint32_t outw = 0;
            const uint32_t wyc = 2048u - wy;
            // all four pixels are computed unconditionally (the x table is padded to a multiple of 4 with valid
            // entries); every product has operands below 2^24, so the 24-bit multiplier (full rate) is exact:
            // top, bot <= 255 * 2048 < 2^19, weights <= 2048
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x1 = (int)(xt[i] & 0xffffu);
                const uint32_t wx = xt[i] >> 16;
                const uint32_t wxc = 2048u - wx;
                const int x2 = min(x1 + 1, sw - 1);
                const uint32_t a = fetch(y1, x1), b = fetch(y1, x2), c = fetch(y2, x1), e = fetch(y2, x2);
                const uint32_t top = __umul24(a, wxc) + __umul24(b, wx);
                const uint32_t bot = __umul24(c, wxc) + __umul24(e, wx);
                const uint32_t v = __umul24(top, wyc) + __umul24(bot, wy) + (1u << 21);
                outw |= (v >> 22) << (8 * i);
            }
            uint8_t* d = dst + (size_t)f * dstFrameStride + __umul24((unsigned)y, (unsigned)dpitch);
            if (nvalid == 4) {
                *reinterpret_cast<uint32_t*>(d + x0) = outw;  // dpitch % 64 == 0 and x0 % 4 == 0
            } else {
                for (int i = 0; i < nvalid; i++) d[x0 + i] = (uint8_t)(outw >> (8 * i));
            }
        }
    };
    if (staged)
        body([&](int yy, int xx) -> uint32_t {  // 24-bit multiply: the row offset is below 40 * 304
            return (&sSrc[0][0])[__umul24((uint32_t)(yy - sy0), (uint32_t)kRsMaxCols) + (uint32_t)(xx - sx0)];
        });
    else
        body([&](int yy, int xx) -> uint32_t { return s[(size_t)yy * spitch + xx]; });
}


// ---------------------------------------------------------------------------------------------------------------
// pyramid_kernel -- level l from the unblurred level l-1 (src/ORBextractor.cc:613-621), one launch per level, every
// wave an independent (strip of output rows) x (256 output columns) task.
//
// The arithmetic is the same (S1) but organised for few vector instructions (the table-driven tile kernel above
// spends ~29 per output pixel, this one ~11):
//   * a lane owns 4 adjacent output columns; per step of four output rows the wave copies the (at most kPyrRows)
//     source rows they touch into a wave-private LDS ring with COALESCED dword loads -- row tables, row addresses and
//     the vertical weights are scalars.  (Letting every lane fetch its own 12-byte window from global memory costs
//     ~15 L1 accesses per wave instruction -- neighbouring windows overlap -- and ran at the L1's access rate.)
//   * horizontal pass per source row: the 12 source bytes a lane's 4 outputs can touch come out of LDS as three
//     dwords, v_perm_b32 with a per-lane selector puts the two taps of an output into the halves of a dword and ONE
//     v_dot2_u32_u16 against (2048 - wx | wx << 16) gives a * (2048 - wx) + b * wx (<= 255 * 2048);
//   * vertical pass: top * 4(2048 - wy) + bot * 4 wy + 2^23 with two 24-bit multiply-adds -- the quotient
//     (v + 2^21) >> 22 is the TOP BYTE of that word, so v_perm_b32 packs four results without shifts.
// The host checks per level that every tap pair of a lane lies inside its 12-byte window, that four output rows never
// span more than kPyrRows source rows, that a 256-column chunk spans at most kPyrSegDw source dwords (level ratios up
// to ~1.45) and that x1 + 1 / y1 + 1 never need the clamp; other levels take the tile kernel above.
// Measured and rejected: ALL levels of a frame in one launch (one 512- or 1024-thread block per frame, block barrier
// between levels): a block needs 0.2 ms per frame whatever the batch -- 0.32 ms per 512 frames at 45 % of the VALU
// issue rate, and slower than the per-level launches below 256 frames.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPyrWaves = 4;    // waves (independent tasks) per block

constexpr int kPyrRows = 7;     // source rows staged per step of 4 output rows
constexpr int kPyrSegDw = 96;   // dwords per staged row segment (256 output columns)
constexpr int kPyrStride = 128; // dwords between staged rows in LDS: both halves of a row are stored by all 64 lanes


// a * w + c with the 24-bit multiplier (a < 2^19, w <= 8192 scalar); hipcc splits the expression into two multiplies
// and a three-input add
__device__ __forceinline__ uint32_t mad24_vsv(uint32_t a, uint32_t w, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(w), "v"(c));
    return r;
}

__device__ __forceinline__ void pyr_hpass(uint32_t d0, uint32_t d1, uint32_t d2, const uint32_t (&sel)[4], const uint32_t (&wp)[4],
                                          uint32_t (&h)[4])
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // columns 0..2 take their taps from window bytes 0..7, column 3 from bytes 3..10
    const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, 3), e1 = __builtin_amdgcn_alignbyte(d2, d1, 3);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t pair = i < 3 ? __builtin_amdgcn_perm(d1, d0, sel[i]) : __builtin_amdgcn_perm(e1, e0, sel[i]);
        h[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pair), __builtin_bit_cast(us2, wp[i]), 0u, false);
    }
}

__global__ __launch_bounds__(kPyrWaves * 64) void pyramid_kernel(const PipelineDesc* __restrict__ P, int l, int R,
                                                                 const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                                 int gray0Pitch, uint8_t* __restrict__ ws,
                                                                 const uint32_t* __restrict__ tabs)
{
    __shared__ uint32_t sRows[kPyrWaves][kPyrRows][kPyrStride];  // wave-private: no block barrier

    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* const ring = &sRows[wave][0][0];

    {
        const LevelDesc& S = P->lv[l - 1];
        const LevelDesc& D = P->lv[l];
        const int sw = S.w, sh = S.h, dw = D.w, dh = D.h, dpitch = D.pitch;
        const uint8_t* src = l == 1 ? gray0 + (size_t)f * gray0FrameStride : ws + S.imgOff + (size_t)f * S.imgFrameStride;
        const int spitch = l == 1 ? gray0Pitch : S.pitch;
        uint8_t* dst = ws + D.imgOff + (size_t)f * D.imgFrameStride;
        const uint32_t* xtab = tabs + D.xtabOff;
        const uint32_t* ytab = tabs + D.ytabOff;
        const int strips = (dh + R - 1) / R, chunks = (dw + 255) / 256;
        // source level through a buffer descriptor: rows 0..sh-1, the last one up to its 4-byte-rounded end (the aligned
        // level-0 contract, orbfe.h); lanes whose dword would start past a row's rounded width use kNowhere (below), so
        // the descriptor's range check is only the mechanism that drops those lanes, never the guard of the frame's end
        const __amdgpu_buffer_rsrc_t srsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (sh - 1) * spitch + ((sw + 3) & ~3), 0x00020000);
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dh * dpitch, 0x00020000);

        const int task = (int)blockIdx.y * kPyrWaves + wave;
        if (task < strips * chunks) {
            const int strip = task / chunks, chunk = task - strip * chunks;
            const int x0 = chunk * 256 + lane * 4;
            const bool colOk = x0 < dw;
            // per-lane column constants (lanes past the level width copy the last group: their results are not stored)
            const int xg = min(x0, (dw - 1) & ~3);
            const uint4 xt4 = *reinterpret_cast<const uint4*>(xtab + xg);  // table padded to a multiple of 4
            const uint32_t xt[4] = {xt4.x, xt4.y, xt4.z, xt4.w};
            const uint32_t start = (xt[0] & 0xffffu) & ~3u;
            const uint32_t segBase = (xtab[chunk * 256] & 0xffffu) & ~3u;  // scalar: first source byte of the chunk
            const uint32_t winOff = start - segBase;                        // byte offset of the lane's window in a staged row
            uint32_t sel[4], wp[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t o = (xt[i] & 0xffffu) - start - (i == 3 ? 3u : 0u);
                sel[i] = 0x0c000c00u | o | ((o + 1u) << 16);
                const uint32_t wx = xt[i] >> 16;
                wp[i] = (2048u - wx) | (wx << 16);
            }
            const int nvalid = min(4, dw - x0);
            // The scalar unit is this kernel's busiest issue port, so nothing below is predicated per lane: lanes without a
            // second dword (and, for the stores, lanes past the level width) use a byte offset beyond the buffer -- the
            // range check drops the access -- instead of an EXEC mask and a branch around every load and store.
            constexpr uint32_t kNowhere = 0x7ffffff0u;
            // the lane's two dwords of a staged row.  The row offset travels in the instruction's SGPR offset, which the
            // buffer range check may not see (only the per-lane offset is guaranteed to be tested), so a dword that would
            // start past the row's 4-byte-rounded width is sent nowhere HERE: with the row clamped to sh - 1 every load
            // then ends inside pitch * (sh - 1) + round4(sw), the input contract of orbfe.h, whatever the descriptor does
            const uint32_t rowEnd = (uint32_t)((sw + 3) & ~3);
            const uint32_t colA = segBase + 4u * (uint32_t)lane;
            const uint32_t ldA = colA + 4u <= rowEnd ? colA : kNowhere;
            const uint32_t ldB = (lane < kPyrSegDw - 64 && colA + 260u <= rowEnd) ? colA + 256u : kNowhere;
            const uint32_t stX = colOk ? (uint32_t)x0 : kNowhere;
            const bool partial = (dw & 3) != 0;  // the level's last column group is narrower than a dword (wave-uniform)

            const int ys = strip * R, ye = min(ys + R, dh);
            const uint32_t roMax = (uint32_t)((sh - 1) * spitch);
            uint32_t kRound = 1u << 23;
            asm("" : "+v"(kRound));  // in a vector register: v_mad_u32_u24 takes one scalar operand (the weight)
            // software pipeline: the global loads (and row-table entries) of step n + 1 are requested before step n is
            // computed, so a wave always has kPyrRows x 2 dwords in flight behind its arithmetic
            uint32_t ytN[4];
            uint32_t ga[kPyrRows], gb[kPyrRows];
            auto request = [&](int yb) {
                // four row-table entries with one scalar load: strips start at multiples of four rows and the table is
                // padded to a multiple of four with copies of its last entry (rows past the strip are computed, not stored)
                const uint4 y4 = *reinterpret_cast<const uint4*>(ytab + yb);
                ytN[0] = y4.x; ytN[1] = y4.y; ytN[2] = y4.z; ytN[3] = y4.w;
                uint32_t ro = (ytN[0] & 0xffffu) * (uint32_t)spitch;
#pragma unroll
                for (int j = 0; j < kPyrRows; j++) {
                    const uint32_t roc = min(ro, roMax);
                    ga[j] = __builtin_amdgcn_raw_buffer_load_b32(srsrc, ldA, roc, 0);
                    gb[j] = __builtin_amdgcn_raw_buffer_load_b32(srsrc, ldB, roc, 0);
                    ro += (uint32_t)spitch;
                }
            };
            request(ys);
            uint32_t stRow = (uint32_t)(ys * dpitch);
#pragma unroll 1
            for (int yb = ys; yb < ye; yb += 4) {
                uint32_t yt[4];
#pragma unroll
                for (int k = 0; k < 4; k++) yt[k] = ytN[k];
                const int rA = (int)(yt[0] & 0xffffu);  // first staged source row
#pragma unroll
                for (int j = 0; j < kPyrRows; j++) {
                    ring[j * kPyrStride + lane] = ga[j];
                    ring[j * kPyrStride + 64 + lane] = gb[j];
                }
                if (yb + 4 < ye) request(yb + 4);  // wave-uniform
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int jt = (int)(yt[k] & 0xffffu) - rA;  // staged row of the top taps (scalar), bottom = jt + 1
                    const uint32_t* wt = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(ring + jt * kPyrStride) + winOff);
                    const uint32_t* wb = wt + kPyrStride;
                    uint32_t hT[4], hB[4];
                    pyr_hpass(wt[0], wt[1], wt[2], sel, wp, hT);
                    pyr_hpass(wb[0], wb[1], wb[2], sel, wp, hB);
                    const uint32_t wy4 = (yt[k] >> 16) * 4u, wyc4 = 8192u - wy4;
                    uint32_t v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = mad24_vsv(hB[i], wy4, mad24_vsv(hT[i], wyc4, kRound));
                    // result byte = bits 31..24 of each v
                    const uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0703u);
                    const uint32_t hi = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0703u);
                    const uint32_t outw = lo | (hi << 16);
                    if (yb + k < ye) {  // wave-uniform
                        if (!partial) {
                            __builtin_amdgcn_raw_buffer_store_b32(outw, drsrc, stX, stRow, 0);  // 4-aligned
                        } else if (colOk) {
                            if (nvalid == 4) {
                                __builtin_amdgcn_raw_buffer_store_b32(outw, drsrc, (uint32_t)x0, stRow, 0);
                            } else {
                                for (int i = 0; i < nvalid; i++)
                                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(outw >> (8 * i)), drsrc, (uint32_t)(x0 + i), stRow, 0);
                            }
                        }
                    }
                    stRow += (uint32_t)dpitch;
                }
            }
        }
    }
}

void launch_resize(hipStream_t s, int frames, const uint8_t* src, size_t srcFrameStride, int sw, int sh,
                   int spitch, int srcAligned4, uint8_t* dst, size_t dstFrameStride, int dw, int dh, int dpitch,
                   const uint32_t* xtab, const uint32_t* ytab)
{
    dim3 block(256);
    dim3 grid(frames, (dw + kRsTW - 1) / kRsTW, (dh + kRsTH - 1) / kRsTH);
    // largest source footprint of a 16-row tile: 16 output rows span (15 * sh) / dh source rows, + 2 for the taps
    const int maxRows = (int)(((long long)(kRsTH - 1) * sh + dh - 1) / dh) + 2;
    if (maxRows <= 24)
        hipLaunchKernelGGL(resize_kernel<6>, grid, block, 0, s, src, srcFrameStride, sw, sh, spitch, srcAligned4, dst,
                           dstFrameStride, dw, dh, dpitch, xtab, ytab);
    else
        hipLaunchKernelGGL(resize_kernel<kRsMaxRows / 4>, grid, block, 0, s, src, srcFrameStride, sw, sh, spitch, srcAligned4,
                           dst, dstFrameStride, dw, dh, dpitch, xtab, ytab);
}


// Which levels the one-launch kernel can produce: every tap pair of a 4-column group inside the group's 12-byte
// window in the places pyr_hpass expects, and no clamped far tap (true for level ratios between 1 and ~1.45).
bool pyramid_level_fits(const uint32_t* xtab, const uint32_t* ytab, int sw, int sh, int dw, int dh)
{
    for (int y = 0; y < dh; y++) {
        if ((int)(ytab[y] & 0xffffu) + 1 > sh - 1) return false;
        // four output rows starting anywhere (strips start at arbitrary rows) must fit the staged ring
        const int yl = std::min(y + 3, dh - 1);
        if ((int)(ytab[yl] & 0xffffu) + 1 - (int)(ytab[y] & 0xffffu) + 1 > kPyrRows) return false;
    }
    for (int c0 = 0; c0 < dw; c0 += 256) {
        // last window of the chunk ends within the staged segment
        const uint32_t segBase = (xtab[c0] & 0xffffu) & ~3u;
        const int xl = (std::min(c0 + 256, dw) - 1) & ~3;
        const uint32_t startL = (xtab[xl] & 0xffffu) & ~3u;
        if (startL + 12 - segBase > 4u * kPyrSegDw) return false;
    }
    for (int x0 = 0; x0 < dw; x0 += 4) {
        const uint32_t start = (xtab[x0] & 0xffffu) & ~3u;
        for (int i = 0; i < 4 && x0 + i < dw; i++) {
            const uint32_t x1 = xtab[x0 + i] & 0xffffu;
            if ((int)x1 + 1 > sw - 1 || x1 < start) return false;
            const uint32_t o = x1 - start;
            if (i < 3 ? o + 1 > 7 : (o < 3 || o - 3 + 1 > 7)) return false;
        }
    }
    return true;
}

void launch_pyramid_level(hipStream_t s, int frames, const PipelineDesc* dP, int level, int dw, int dh, const uint8_t* gray0,
                          size_t gray0FrameStride, int gray0Pitch, uint8_t* ws, const uint32_t* tabs)
{
    // output rows per wave task: long strips amortise the per-task column setup when the launch fills the chip anyway,
    // one 4-row step per task keeps a small batch short (a task is a serial chain of steps)
#ifdef ORBFE_DIAG
    static const int envR = getenv("ORBFE_PYR_ROWS") ? atoi(getenv("ORBFE_PYR_ROWS")) : 0;  // tuning experiments (liborbfe_diag.so)
#else
    constexpr int envR = 0;
#endif
    const int R = envR > 0 ? (envR + 3) & ~3 : frames >= 128 ? 8 : 4;  // strips start at multiples of four rows (x4 row-table loads)
    const int tasks = ((dh + R - 1) / R) * ((dw + 255) / 256);
    hipLaunchKernelGGL(pyramid_kernel, dim3(frames, (tasks + kPyrWaves - 1) / kPyrWaves), dim3(kPyrWaves * 64), 0, s, dP, level, R, gray0,
                       gray0FrameStride, gray0Pitch, ws, tabs);
}


// ---------------------------------------------------------------------------------------------------------------------
// Small launches (one to eight frames): several levels per launch.
// A level is resized from the level below it, so the pyramid of a single frame is a chain of seven dependent launches,
// each a few microseconds of latency on an otherwise idle chip (tools/stage_b1.py: 28 us for the seven, as much as FAST).
// Here one launch produces up to three consecutive levels l0+1 .. l0+depth from level l0: a pixel of level l0+k is
// computed from level l0 directly, re-evaluating the (2 x 2)^(k-1) pixels of the levels in between that it depends on
// with exactly the arithmetic of resize_kernel (same tables, same 11-bit weights, same rounding: the intermediate values
// are the bytes the other threads store for those levels).  The recomputation -- a factor 4 per skipped level -- is
// irrelevant for a few frames and would be waste for many: the batched path keeps one row-streaming launch per level.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct ChainCtx {
    const uint8_t* src0;  // level l0 of this frame
    int pitch0, w0, h0;
    const uint32_t* xtab[3];
    const uint32_t* ytab[3];
    int w[3], h[3];       // sizes of levels l0+1 .. l0+3
};

// pixel (x, y) of level l0 + K
template <int K>
__device__ __forceinline__ uint32_t chain_px(const ChainCtx& c, int x, int y)
{
    const uint32_t xt = c.xtab[K - 1][x], yt = c.ytab[K - 1][y];
    const int sw = K == 1 ? c.w0 : c.w[K - 2], sh = K == 1 ? c.h0 : c.h[K - 2];
    const int x1 = (int)(xt & 0xffffu), y1 = (int)(yt & 0xffffu);
    const uint32_t wx = xt >> 16, wy = yt >> 16;
    const uint32_t wxc = 2048u - wx, wyc = 2048u - wy;
    const int x2 = min(x1 + 1, sw - 1), y2 = min(y1 + 1, sh - 1);
    uint32_t a, b, cc, e;
    if constexpr (K == 1) {
        const uint8_t* r1 = c.src0 + (size_t)y1 * c.pitch0;
        const uint8_t* r2 = c.src0 + (size_t)y2 * c.pitch0;
        a = r1[x1]; b = r1[x2]; cc = r2[x1]; e = r2[x2];
    } else {
        a = chain_px<K - 1>(c, x1, y1); b = chain_px<K - 1>(c, x2, y1);
        cc = chain_px<K - 1>(c, x1, y2); e = chain_px<K - 1>(c, x2, y2);
    }
    const uint32_t top = __umul24(a, wxc) + __umul24(b, wx);
    const uint32_t bot = __umul24(cc, wxc) + __umul24(e, wx);
    const uint32_t v = __umul24(top, wyc) + __umul24(bot, wy) + (1u << 21);
    return v >> 22;
}

// thread -> one pixel of one of the `depth` output levels (adjacent lanes store adjacent bytes); grid.x covers the pixels of
// all of them (grid.x is the dimension without the 65535 limit: two levels of a 4095 x 4095 frame are 77 k blocks), grid.y
// the frames.  One pixel per thread keeps the dependent chain of a thread at `depth` rounds of loads.
__global__ __launch_bounds__(256) void pyramid_chain_kernel(const PipelineDesc* __restrict__ P, int l0, int depth,
                                                            const uint8_t* __restrict__ gray0, size_t gray0FrameStride, int gray0Pitch,
                                                            uint8_t* __restrict__ ws, const uint32_t* __restrict__ tabs,
                                                            uint32_t* __restrict__ zero, int nZeroPerFrame)
{
    const int f = blockIdx.y;
    // the first launch of a chain also clears the frame's per-level counters (FAST, two launches later at the earliest, is
    // their first user): one node less in the chain than a separate memset
    if (zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < nZeroPerFrame; i += 256) zero[(size_t)f * nZeroPerFrame + i] = 0;
    ChainCtx c;
    const LevelDesc& S = P->lv[l0];
    c.src0 = l0 == 0 ? gray0 + (size_t)f * gray0FrameStride : ws + S.imgOff + (size_t)f * S.imgFrameStride;
    c.pitch0 = l0 == 0 ? gray0Pitch : S.pitch;
    c.w0 = S.w;
    c.h0 = S.h;
    int np[3] = {0, 0, 0};  // pixels per level
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const LevelDesc& D = P->lv[min(l0 + 1 + k, P->nLevels - 1)];
        c.xtab[k] = tabs + D.xtabOff;
        c.ytab[k] = tabs + D.ytabOff;
        c.w[k] = D.w;
        c.h[k] = D.h;
        np[k] = k < depth ? D.w * D.h : 0;
    }
    int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    int k = 0;
    if (idx >= np[0]) { idx -= np[0]; k = 1; if (idx >= np[1]) { idx -= np[1]; k = 2; if (idx >= np[2]) return; } }
    const LevelDesc& D = P->lv[l0 + 1 + k];
    const int y = idx / D.w, x = idx - y * D.w;
    uint32_t out;
    if (k == 0) out = chain_px<1>(c, x, y);
    else if (k == 1) out = chain_px<2>(c, x, y);
    else out = chain_px<3>(c, x, y);
    ws[D.imgOff + (size_t)f * D.imgFrameStride + (size_t)y * D.pitch + x] = (uint8_t)out;
}

}  // namespace

// levels l0+1 .. l0+depth (depth 1..3) of `frames` frames in one launch; hostP is the host copy of *dP
void launch_pyramid_chain(hipStream_t s, int frames, const PipelineDesc* dP, const PipelineDesc& hostP, int l0, int depth,
                          const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, uint8_t* ws, const uint32_t* tabs,
                          uint32_t* zero, int nZeroPerFrame)
{
    long long pixels = 0;
    for (int k = 0; k < depth; k++) {
        const LevelDesc& D = hostP.lv[l0 + 1 + k];
        pixels += (long long)D.w * D.h;
    }
    hipLaunchKernelGGL(pyramid_chain_kernel, dim3((unsigned)((pixels + 255) / 256), frames), dim3(256), 0, s, dP, l0, depth, gray0,
                       gray0FrameStride, gray0Pitch, ws, tabs, zero, nZeroPerFrame);
}

}  // namespace orbfe
