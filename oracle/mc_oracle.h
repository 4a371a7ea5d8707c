/*
 * mc_oracle.h -- CPU restatement of the reference's marching-cubes extraction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (volumetricterrain_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / the CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" against an execution of the reference.  The reference's
 * extractor is three Unity HLSL compute shaders driven from C#; neither toolchain exists in the
 * build container and the reference holds no tests, golden vectors or fixtures.
 * The restatement is pinned by: (i) SHA-256 digests of the three lookup tables
 * (VoxelTerrain.cs:489-794), (ii) analytic plane / sphere / empty / full answers whose density
 * formulas are the reference's own (TerrainModifier.cs:59-62, :79-82), (iii) block-decomposition
 * invariance.  See tests/test_oracle_*.py.
 *
 * Paths below are relative to /root/reference/Unity-Project/Assets/.
 */
#ifndef VTMC_MC_ORACLE_H
#define VTMC_MC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Wire format of one triangle: Scripts/VoxelTerrain.cs:23-37, Shaders/MarchingCube.compute:18-27.
 * 76 bytes, packed, little-endian. Positions are block-local cell units in [0,8]. */
typedef struct {
    float position0[3];
    float position1[3];
    float position2[3];
    float normal0[3];
    float normal1[3];
    float normal2[3];
    int32_t block;
} vto_triangle;

/* Expanded int32 tables exactly as Scripts/VoxelTerrain.cs:489-507, :511-529, :536-794. */
void vto_tables(int32_t edge[256], int32_t tri_num[256], int32_t vert[256 * 15]);

/* Scripts/VoxelTerrain.cs:337-361 -- gather one 10x10x10 tile per block, x fastest in the output.
 * grid is addressed as grid[x*sx + y*sy + z*sz] (element strides); block_list holds B (bx,by,bz). */
void vto_gather_tiles(const float *grid, int64_t sx, int64_t sy, int64_t sz,
                      const int32_t *block_list, int32_t n_blocks, float *samples);

/* Shaders/SampleNormal.compute:23-34 -- normals[(729*b + x + 9y + 81z)*3 + c]. */
void vto_sample_normal(const float *samples, int32_t n_blocks, float *normals);

/* Shaders/CollectTriNum.compute:23-64 -- corner_flags[512*b + x + 8y + 64z]; returns _TriNum[0]. */
uint32_t vto_collect_tri_num(const float *samples, int32_t n_blocks, uint32_t *corner_flags);

/* Shaders/MarchingCube.compute:101-165 in the canonical order (block, cell x+8y+64z, triangle i).
 * block_tri_offsets (n_blocks+1 entries, may be NULL) receives the exclusive prefix of per-block
 * triangle counts.  Returns the number of triangles written. */
int64_t vto_marching_cube(const float *samples, const float *normals, const uint32_t *corner_flags,
                          int32_t n_blocks, vto_triangle *meshes, int32_t *block_tri_offsets);

/* Scripts/VoxelTerrain.cs:430-446 -- bin triangles by _block, scaling positions by voxel_scale.
 * vertices / normals receive 3 float3 per triangle grouped by block (block b occupies
 * [3*block_tri_offsets[b], 3*block_tri_offsets[b+1]) vertices).  Works for any input order;
 * within one block the input order is kept (as List.Add does). */
void vto_bin_triangles(const vto_triangle *tris, int64_t n_tris, int32_t n_blocks, float voxel_scale,
                       float *vertices, float *normals, int32_t *block_tri_offsets);

/* Whole path of VoxelTerrain.BatchUpdate (Scripts/VoxelTerrain.cs:330-427) on a grid in place:
 * per block gather -> normals -> classify -> emit, canonical order.  meshes may be NULL (count only).
 * n_threads > 1 uses OpenMP over blocks (count pass, prefix, emit pass).  Returns T, or -1 if
 * capacity < T. */
int64_t vto_extract_grid(const float *grid, int64_t sx, int64_t sy, int64_t sz,
                         const int32_t *block_list, int32_t n_blocks,
                         vto_triangle *meshes, int64_t capacity, int32_t *block_tri_offsets,
                         uint8_t *cases_u8, int32_t n_threads);

int32_t vto_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
