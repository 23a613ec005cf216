// kernels_pyramid.hip -- bilinear pyramid downscale for gfx950 (SPEC DECISION S1).
//
// Replaces cv::cuda::resize(level l-1 -> l, INTER_LINEAR) of ORBextractor::ComputePyramid
// (src/ORBextractor.cc:607-623).  Pure integer: corner-aligned source position, Q11 weights, one
// rounding; defined in oracle/orb_oracle.c (orc_resize_bilinear) and reproduced bit for bit here.
// (The Gaussian of each level is fused into the FAST kernel, kernels_fast.hip.)
//
// Launch geometry: blockIdx.x = frame (fastest-varying so that frames f and f+8 share an XCD and
// every tile of one frame hits the same L2), blockIdx.y/z = tile.
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

}  // namespace orbfe
