// launch.h -- host-callable launch wrappers of the gfx950 kernels (one per stage).
#pragma once
#include "orbfe_internal.h"

namespace orbfe {

// kernels_pyramid.hip
void launch_resize(hipStream_t s, int frames, const uint8_t* src, size_t srcFrameStride, int sw, int sh,
                   int spitch, int srcAligned4, uint8_t* dst, size_t dstFrameStride, int dw, int dh, int dpitch,
                   const uint32_t* xtab, const uint32_t* ytab);

// row-streaming level kernel (levels whose tables pass pyramid_level_fits; the others take launch_resize)
bool pyramid_level_fits(const uint32_t* xtab, const uint32_t* ytab, int sw, int sh, int dw, int dh);
void launch_pyramid_level(hipStream_t s, int frames, const PipelineDesc* dP, int level, int dw, int dh, const uint8_t* gray0,
                          size_t gray0FrameStride, int gray0Pitch, uint8_t* ws, const uint32_t* tabs);

// small launches: levels l0+1 .. l0+depth (depth <= 3) from level l0 in ONE launch (intermediate levels re-evaluated per pixel)
void launch_pyramid_chain(hipStream_t s, int frames, const PipelineDesc* dP, const PipelineDesc& hostP, int l0, int depth,
                          const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, uint8_t* ws, const uint32_t* tabs,
                          uint32_t* zero = nullptr, int nZeroPerFrame = 0);  // optional: words to clear per frame (the level counters)

// kernels_fast.hip (FAST + NMS + compaction fused with the Gaussian blur of the same tile)
void fast_tiles_for(int w, int h, int* tx, int* ty);
uint32_t fast_tile_info(int level, int tileX, int tileY);  // entry of the per-tile table (level << 24 | ty << 12 | tx)
void launch_fast_blur(hipStream_t s, int frames, int totalTiles, const PipelineDesc* dP, const uint32_t* dTileInfo,
                      const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, int gray0Aligned4, uint8_t* ws,
                      uint32_t* cand, uint32_t* counters, uint16_t* tileRows);

// kernels_quadtree.hip
int quadtree_node_capacity(int maxNodeCap);
size_t quadtree_scratch_bytes_per_block(int maxNodeCap);
void launch_quadtree(hipStream_t s, int frames, int nLevels, int maxNodeCap, const PipelineDesc* dP,
                     const uint32_t* cand, uint16_t* nodeOf, uint32_t* counters, uint32_t* lvlKp,
                     const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, const uint8_t* ws,
                     const uint16_t* tileRows, uint8_t* scratch);

// kernels_desc.hip
void launch_orient_brief(hipStream_t s, int frames, int kpCapFrame, const PipelineDesc* dP, const uint8_t* gray0,
                         size_t gray0FrameStride, int gray0Pitch, const uint8_t* ws, const uint32_t* counters,
                         const uint32_t* lvlKp, orbfe_keypoint* kpOut, uint8_t* descOut, int* nOut,
                         int* perLevelOut, int* statusOut, const int* kpBase, int nLevels);

// kernels_probe.hip: one wave writes {shader cycles, 100 MHz ticks} spent while `ticks` of the constant clock went by
void launch_clock_probe(hipStream_t s, unsigned long long* dOut2, unsigned ticks);

}  // namespace orbfe
