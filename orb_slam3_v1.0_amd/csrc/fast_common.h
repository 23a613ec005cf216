// fast_common.h -- FAST-9/16 helpers shared by the fused tile kernel (kernels_fast.hip) and the
// rare exact-cap path of the quadtree kernel (kernels_quadtree.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbfe {

constexpr int kFastTW = 64, kFastTH = 32;  // FAST tile (pixels)

// 16-bit circular mask contains >= 9 contiguous ones (== c_table lookup, src/cuda/Fast_gpu.cu:187-191)
__device__ __forceinline__ bool arc9(uint32_t m)
{
    uint32_t m2 = m | (m << 16);
    uint32_t r = m2 & (m2 >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m2 >> 8;
    return (r & 0xffffu) != 0;
}

// Segment test of the pixel (x, y) straight from global memory at threshold th.  Ring bit k <->
// (dy, dx) as in SURVEY.md appendix B2 (derived from Fast_gpu.cu:226-254).
__device__ __forceinline__ bool corner_at_global(const uint8_t* __restrict__ img, int pitch, int x, int y, int th)
{
    const int8_t dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    const int8_t dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int v = img[(size_t)y * pitch + x];
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int d = (int)img[(size_t)(y + dy[k]) * pitch + x + dx[k]] - v;
        mb |= (uint32_t)(d > th) << k;
        md |= (uint32_t)(d < -th) << k;
    }
    return arc9(mb) || arc9(md);
}

}  // namespace orbfe
