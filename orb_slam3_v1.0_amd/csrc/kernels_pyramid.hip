// kernels_pyramid.hip -- image pyramid for gfx950 (SPEC DECISION S1).
//
// Replaces the two library calls of ORBextractor::ComputePyramid (src/ORBextractor.cc:607-623):
//   cv::cuda::resize(level l-1 -> l, INTER_LINEAR)            -> resize_kernel
//   cv::cuda::Filter::apply (Gaussian 5x5, sigma 1.2, REFLECT_101) -> blur_kernel
// Both are pure integer; the arithmetic is defined in oracle/orb_oracle.c (orc_resize_bilinear,
// orc_gauss5) and reproduced bit for bit here.
//
// Launch geometry: blockIdx.x = frame (fastest-varying so that frames f and f+8 share an XCD and
// every tile of one frame hits the same L2), blockIdx.y/z = tile.
#include "launch.h"
#include "device_math.h"

namespace orbfe {

// ---------------------------------------------------------------------------------------------
// resize: 4 output pixels per thread, one dword store.  xtab/ytab hold (x1 | weight << 16).
// Algorithmic bytes per output pixel: 1 written + ~1.44 read (the source level is 1.2^2 larger).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, size_t srcFrameStride,
                                                     int sw, int sh, int spitch,
                                                     uint8_t* __restrict__ dst, size_t dstFrameStride,
                                                     int dw, int dh, int dpitch,
                                                     const uint32_t* __restrict__ xtab,
                                                     const uint32_t* __restrict__ ytab)
{
    const int f = blockIdx.x;
    const int x0 = (blockIdx.y * 64 + threadIdx.x) * 4;
    const int y = blockIdx.z * 4 + threadIdx.y;
    if (x0 >= dw || y >= dh) return;
    const uint8_t* s = src + (size_t)f * srcFrameStride;
    uint8_t* d = dst + (size_t)f * dstFrameStride + (size_t)y * dpitch;

    const uint32_t yt = ytab[y];
    const int y1 = (int)(yt & 0xffffu);
    const uint32_t wy = yt >> 16;
    const int y2 = min(y1 + 1, sh - 1);
    const uint8_t* r1 = s + (size_t)y1 * spitch;
    const uint8_t* r2 = s + (size_t)y2 * spitch;

    uint32_t outw = 0;
    const int nvalid = min(4, dw - x0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nvalid) {
            const uint32_t xt = xtab[x0 + i];
            const int x1 = (int)(xt & 0xffffu);
            const uint32_t wx = xt >> 16;
            const int x2 = min(x1 + 1, sw - 1);
            const uint32_t a = r1[x1], b = r1[x2], c = r2[x1], e = r2[x2];
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

// ---------------------------------------------------------------------------------------------
// blur: separable Q8 {22,62,88,62,22}, one rounding after both passes.
// Tile 128 x 16 output pixels per 256-thread block; source tile (132 x 20) staged in LDS with
// REFLECT_101 applied at load, horizontal pass -> u16 LDS, vertical pass -> packed dword stores.
// Algorithmic bytes per pixel: 1 read + 1 written.
// ---------------------------------------------------------------------------------------------
constexpr int kBlurTW = 128, kBlurTH = 16;

__global__ __launch_bounds__(256) void blur_kernel(const PipelineDesc* __restrict__ P,
                                                   const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                   int gray0Pitch, uint8_t* __restrict__ ws,
                                                   const int* __restrict__ blurTileBase)
{
    __shared__ uint8_t sSrc[kBlurTH + 4][kBlurTW + 8];
    __shared__ uint16_t sTmp[kBlurTH + 4][kBlurTW];

    const int f = blockIdx.x;
    // locate level from the flattened tile index (uniform -> scalar loop)
    const int tile = blockIdx.y;
    int l = 0;
    const int nL = P->nLevels;
    while (l + 1 < nL && tile >= blurTileBase[l + 1]) l++;
    const LevelDesc& L = P->lv[l];
    const int w = L.w, h = L.h;
    const int tilesX = (w + kBlurTW - 1) / kBlurTW;
    const int t = tile - blurTileBase[l];
    const int tx0 = (t % tilesX) * kBlurTW;
    const int ty0 = (t / tilesX) * kBlurTH;

    const uint8_t* src;
    int spitch;
    if (l == 0) {
        src = gray0 + (size_t)f * gray0FrameStride;
        spitch = gray0Pitch;
    } else {
        src = ws + L.imgOff + (size_t)f * L.imgFrameStride;
        spitch = L.pitch;
    }
    uint8_t* dst = ws + L.blurOff + (size_t)f * L.blurFrameStride;
    const int dpitch = L.pitch;

    const int tid = threadIdx.x;
    // stage 0: load (kBlurTW+4) x (kBlurTH+4) with reflection
    for (int e = tid; e < (kBlurTH + 4) * (kBlurTW + 4); e += 256) {
        const int r = e / (kBlurTW + 4);
        const int c = e - r * (kBlurTW + 4);
        const int gy = reflect101(ty0 - 2 + r, h);
        const int gx = reflect101(tx0 - 2 + c, w);
        sSrc[r][c] = src[(size_t)gy * spitch + gx];
    }
    __syncthreads();
    // stage 1: horizontal
    for (int e = tid; e < (kBlurTH + 4) * kBlurTW; e += 256) {
        const int r = e / kBlurTW;
        const int c = e - r * kBlurTW;
        const uint32_t v = 22u * sSrc[r][c] + 62u * sSrc[r][c + 1] + 88u * sSrc[r][c + 2] +
                           62u * sSrc[r][c + 3] + 22u * sSrc[r][c + 4];
        sTmp[r][c] = (uint16_t)v;  // <= 255 * 256 = 65280
    }
    __syncthreads();
    // stage 2: vertical, 4 px per thread, 2 rows per thread
    const int xq = (tid & 31) * 4;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int ry = (tid >> 5) + rr * 8;
        const int gy = ty0 + ry;
        const int gx = tx0 + xq;
        if (gy < h && gx < w) {
            uint32_t outw = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t v = 22u * sTmp[ry][xq + i] + 62u * sTmp[ry + 1][xq + i] +
                                   88u * sTmp[ry + 2][xq + i] + 62u * sTmp[ry + 3][xq + i] +
                                   22u * sTmp[ry + 4][xq + i];
                outw |= ((v + 32768u) >> 16) << (8 * i);
            }
            uint8_t* drow = dst + (size_t)gy * dpitch;
            if (gx + 3 < w) {
                *reinterpret_cast<uint32_t*>(drow + gx) = outw;
            } else {
                for (int i = 0; gx + i < w; i++) drow[gx + i] = (uint8_t)(outw >> (8 * i));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
void launch_resize(hipStream_t s, int frames, const uint8_t* src, size_t srcFrameStride, int sw, int sh,
                   int spitch, uint8_t* dst, size_t dstFrameStride, int dw, int dh, int dpitch,
                   const uint32_t* xtab, const uint32_t* ytab)
{
    dim3 block(64, 4);
    dim3 grid(frames, (dw + 255) / 256, (dh + 3) / 4);
    hipLaunchKernelGGL(resize_kernel, grid, block, 0, s, src, srcFrameStride, sw, sh, spitch, dst,
                       dstFrameStride, dw, dh, dpitch, xtab, ytab);
}

int blur_tiles_for(int w, int h) { return ((w + kBlurTW - 1) / kBlurTW) * ((h + kBlurTH - 1) / kBlurTH); }

void launch_blur(hipStream_t s, int frames, int totalTiles, const PipelineDesc* dP, const uint8_t* gray0,
                 size_t gray0FrameStride, int gray0Pitch, uint8_t* ws, const int* dBlurTileBase)
{
    dim3 block(256);
    dim3 grid(frames, totalTiles);
    hipLaunchKernelGGL(blur_kernel, grid, block, 0, s, dP, gray0, gray0FrameStride, gray0Pitch, ws,
                       dBlurTileBase);
}

}  // namespace orbfe
