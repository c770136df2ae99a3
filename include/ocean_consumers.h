/*
 * include/ocean_consumers.h -- SURVEY.md 8f ranks 3-4: what consumes the two maps on the device (vertex stage, cascades, mip chain).
 * Part of the C ABI of libocean_hip.so (include/ocean.h is the drop-in boundary; this header declares more of the same library's exports).
 */
#ifndef OCEAN_CONSUMERS_H_
#define OCEAN_CONSUMERS_H_

#include "ocean.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- vertex-stage consumer ----------------------------------
 * What the reference's vertex shader does with the two maps
 * (src/shaders/WaterSurfaceMesh.vert:24-41) for the grid its mesh generator builds
 * (WaterSurfaceMesh::CreateGridVertices, WaterSurfaceMesh.cpp:500-533), on the device:
 * vertex (x, y), x, y = -grid_size/2 .. grid_size/2, sits at (x, 0, y) * vertex_distance with
 * uv = (x + half, y + half) / grid_size; both maps are sampled at uv * uv_scale with the
 * reference's sampler (LINEAR, REPEAT: vulkan/Sampler.cpp:60-66);
 *   position = inPos + (D.x, D.y * A, D.z), w = D.w          (A = amplitude of the frame)
 *   normal   = normalize(-s.x / (1 + choppy*s.z), 1, -s.y / (1 + choppy*s.w)), w = 0
 * for the most recent frame of `tile`, ordered on its stream.  choppy is the value the
 * reference feeds (GetDisplacementLambda(), WaterSurfaceMesh.cpp:172).  Results stay in
 * device buffers owned by the context ((grid_size+1)^2 float4 each): ocean_read_grid copies
 * them out (synchronises), ocean_device_grid hands out the pointers.                  */
int ocean_displace_grid(ocean_t* ctx, uint32_t tile, uint32_t grid_size, float vertex_distance,
                        float uv_scale, float choppy);
/* Cascades (the reference's to-do "Endless - solving the tiling artifacts", README.md:37-44): the
 * tiles first_tile .. first_tile+count-1 of the batch (count <= 8) -- independent oceans with their own tile length,
 * wind, seed -- are summed by the consumer, tile c sampled at uv * uv_scales[c]:
 *   position = inPos + sum_c (D_c.x, D_c.y * A_c, D_c.z),  w = min_c D_c.w
 *   normal   = normalize(-S.x / (1 + choppy*S.z), 1, -S.y / (1 + choppy*S.w)),  S = sum_c normal-map sample of tile c
 * With incommensurate scales the surface no longer repeats with the period of one tile.  Same output buffers
 * and read-out as ocean_displace_grid.                                                                          */
int ocean_displace_grid_cascades(ocean_t* ctx, uint32_t first_tile, uint32_t count, uint32_t grid_size,
                                 float vertex_distance, const float* uv_scales /* count */, float choppy);
int ocean_read_grid(ocean_t* ctx, float* positions, float* normals);
int ocean_device_grid(ocean_t* ctx, void** d_positions, void** d_normals, uint32_t* vertices);

/* Mip chain of one tile's maps: the reference's LOD hook.  Its map textures are created and filled with a
 * `mipmapping` flag (s_kUseMipMapping, WaterSurfaceMesh.h:216, passed at WaterSurfaceMesh.cpp:611-618,652-690; off in the
 * shipped build, "LOD. anti-aliasing" on its to-do list, README.md:37-44); when set, Texture2D::GenerateMipmaps
 * (vulkan/Texture2D.cpp:228-330) blits level i-1 into level i at half the extent with VK_FILTER_LINEAR,
 * floor(log2(N)) + 1 levels in all.  ocean_build_mips does the same for both maps of `tile` behind the frame that wrote
 * them: levels 1 .. log2(N) (level 0 is the map itself), each texel the 2 x 2 mean of the level above, tightly packed
 * one after another -- level l starts at texel sum_{k=1}^{l-1} (N >> k)^2, ocean_mip_texels(N) = (N^2 - 1) / 3 texels
 * of RGBA32F per map.  Results stay in device buffers owned by the context: ocean_read_mips copies them out
 * (synchronises), ocean_device_mips hands out the pointers and the number of levels.                                  */
size_t ocean_mip_texels(uint32_t tile_size);
int ocean_build_mips(ocean_t* ctx, uint32_t tile);
int ocean_read_mips(ocean_t* ctx, float* disp_mips, float* nrm_mips);
int ocean_device_mips(ocean_t* ctx, void** d_disp_mips, void** d_nrm_mips, uint32_t* levels);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_CONSUMERS_H_ */
