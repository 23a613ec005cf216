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

// Output tile 256 x 4 per 256-thread block; the source footprint (<= 8 rows x 520 px for scale
// factors up to 2) is staged in LDS with dword loads, then every thread produces 4 pixels and
// stores one dword.  xtab/ytab hold (x1 | Q11 weight << 16), 16-byte aligned per level.
// Algorithmic bytes per output pixel: 1 written + scale^2 read.
constexpr int kRsTW = 256, kRsTH = 4;
constexpr int kRsMaxRows = 8, kRsMaxCols = 528;

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

    // source footprint of this tile (uniform)
    const int txLast = min(tx0 + kRsTW, dw) - 1;
    const int tyLast = min(ty0 + kRsTH, dh) - 1;
    const int sx0 = (int)(xtab[tx0] & 0xffffu) & ~3;               // dword-aligned left edge
    const int sx1 = min((int)(xtab[txLast] & 0xffffu) + 1, sw - 1);
    const int sy0 = (int)(ytab[ty0] & 0xffffu);
    const int sy1 = min((int)(ytab[tyLast] & 0xffffu) + 1, sh - 1);
    const int nrows = sy1 - sy0 + 1;
    const int ndw = (sx1 - sx0) / 4 + 1;                            // dwords per staged row
    const bool staged = nrows <= kRsMaxRows && ndw * 4 <= kRsMaxCols;

    if (staged) {
        for (int e = tid; e < nrows * ndw; e += 256) {
            const int r = e / ndw;
            const int c4 = e - r * ndw;
            const int gx = sx0 + 4 * c4;
            const uint8_t* row = s + (size_t)(sy0 + r) * spitch;
            uint32_t wv;
            if (srcAligned4 && gx + 3 < sw) {
                wv = *reinterpret_cast<const uint32_t*>(row + gx);
            } else {
                wv = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (gx + i < sw) wv |= (uint32_t)row[gx + i] << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(&sSrc[r][4 * c4]) = wv;
        }
        __syncthreads();
    }

    const int x0 = tx0 + (tid & 63) * 4;
    const int y = ty0 + (tid >> 6);
    if (x0 >= dw || y >= dh) return;
    uint8_t* d = dst + (size_t)f * dstFrameStride + (size_t)y * dpitch;

    const uint32_t yt = ytab[y];
    const int y1 = (int)(yt & 0xffffu);
    const uint32_t wy = yt >> 16;
    const int y2 = min(y1 + 1, sh - 1);
    const uint4 xt4 = *reinterpret_cast<const uint4*>(xtab + x0);  // table padded to a multiple of 4
    const uint32_t xt[4] = {xt4.x, xt4.y, xt4.z, xt4.w};

    uint32_t outw = 0;
    const int nvalid = min(4, dw - x0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nvalid) {
            const int x1 = (int)(xt[i] & 0xffffu);
            const uint32_t wx = xt[i] >> 16;
            const int x2 = min(x1 + 1, sw - 1);
            uint32_t a, b, c, e;
            if (staged) {
                const uint8_t* r1 = sSrc[y1 - sy0];
                const uint8_t* r2 = sSrc[y2 - sy0];
                a = r1[x1 - sx0]; b = r1[x2 - sx0]; c = r2[x1 - sx0]; e = r2[x2 - sx0];
            } else {
                const uint8_t* r1 = s + (size_t)y1 * spitch;
                const uint8_t* r2 = s + (size_t)y2 * spitch;
                a = r1[x1]; b = r1[x2]; c = r2[x1]; e = r2[x2];
            }
            const uint32_t top = a * (2048u - wx) + b * wx;
            const uint32_t bot = c * (2048u - wx) + e * wx;
            const uint32_t v = top * (2048u - wy) + bot * wy;
            outw |= ((v + (1u << 21)) >> 22) << (8 * i);
        }
    }
    if (nvalid == 4) {
        *reinterpret_cast<uint32_t*>(d + x0) = outw;  // dpitch % 64 == 0 and x0 % 4 == 0
    } else {
        for (int i = 0; i < nvalid; i++) d[x0 + i] = (uint8_t)(outw >> (8 * i));
    }
}

void launch_resize(hipStream_t s, int frames, const uint8_t* src, size_t srcFrameStride, int sw, int sh,
                   int spitch, int srcAligned4, uint8_t* dst, size_t dstFrameStride, int dw, int dh, int dpitch,
                   const uint32_t* xtab, const uint32_t* ytab)
{
    dim3 block(256);
    dim3 grid(frames, (dw + kRsTW - 1) / kRsTW, (dh + kRsTH - 1) / kRsTH);
    hipLaunchKernelGGL(resize_kernel, grid, block, 0, s, src, srcFrameStride, sw, sh, spitch, srcAligned4, dst,
                       dstFrameStride, dw, dh, dpitch, xtab, ytab);
}

}  // namespace orbfe
