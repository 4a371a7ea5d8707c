/*
 * mc_oracle.c -- CPU restatement (plain C, FP32, contraction off) of the reference extractor.
 *
 * TEST INFRASTRUCTURE ONLY (see mc_oracle.h): the checker for the HIP path and the CPU baseline
 * of bench.py.  "parity unpinned" against an execution of the reference (no C#/HLSL toolchain);
 * pinned by table digests + analytic known answers (tests/test_oracle_*.py).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference/Unity-Project/Assets/.  Build with -ffp-contract=off so each expression is one
 * fixed IEEE-754 binary32 answer.
 */
#include "mc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../volumetricterrain_amd/csrc/mc_tables_packed.h"

/* Sizes: VoxelTerrain.cs:54 (blockSize), SampleNormal.compute:11-17, CollectTriNum.compute:11-15. */
enum { BS = 8, SS = 9, SS2 = 81, SS3 = 729, IS = 10, IS2 = 100, IS3 = 1000 };

static const uint64_t k_packed[VTMC_MC_TABLE_WORDS] = VTMC_MC_TABLE_INIT;

/* MarchingCube.compute:40-43 */
static const int k_edge_conn[12][2] = {
    {0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
/* MarchingCube.compute:46-50 */
static const int k_vert_off[8][3] = {
    {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

static int32_t g_edge[256], g_trinum[256], g_vert[256 * 15];
static int g_tables_ready = 0;

static void build_tables(void)
{
    if (g_tables_ready) return;
    for (int c = 0; c < 256; ++c) {
        uint64_t w = k_packed[c];
        int32_t mask = 0;
        for (int k = 0; k < 15; ++k) {
            int v = (int)((w >> (4 * k)) & 0xF);
            g_vert[15 * c + k] = (v == 0xF) ? -1 : v;
            if (v != 0xF) mask |= 1 << v;
        }
        g_edge[c] = mask;
        g_trinum[c] = (int32_t)(w >> 60);
    }
    g_tables_ready = 1;
}

void vto_tables(int32_t edge[256], int32_t tri_num[256], int32_t vert[256 * 15])
{
    build_tables();
    memcpy(edge, g_edge, sizeof g_edge);
    memcpy(tri_num, g_trinum, sizeof g_trinum);
    memcpy(vert, g_vert, sizeof g_vert);
}

int32_t vto_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* VoxelTerrain.cs:341-361: samples[ix + 10*iy + 100*iz + 1000*blockNum] = voxel[8x+ix, 8y+iy, 8z+iz] */
static void gather_one(const float *grid, int64_t sx, int64_t sy, int64_t sz, const int32_t *b3, float *tile)
{
    for (int ix = 0; ix < IS; ++ix)
        for (int iy = 0; iy < IS; ++iy)
            for (int iz = 0; iz < IS; ++iz) {
                int64_t qx = (int64_t)b3[0] * BS + ix;
                int64_t qy = (int64_t)b3[1] * BS + iy;
                int64_t qz = (int64_t)b3[2] * BS + iz;
                tile[ix + iy * IS + iz * IS2] = grid[qx * sx + qy * sy + qz * sz];
            }
}

void vto_gather_tiles(const float *grid, int64_t sx, int64_t sy, int64_t sz,
                      const int32_t *block_list, int32_t n_blocks, float *samples)
{
    for (int32_t b = 0; b < n_blocks; ++b)
        gather_one(grid, sx, sy, sz, block_list + 3 * (int64_t)b, samples + (int64_t)b * IS3);
}

/* SampleNormal.compute:23-34.  HLSL normalize(v) = v * rsqrt(dot(v,v)); restated with a correctly
 * rounded divide by sqrt -- the reason positions/normals are compared at 1e-5. */
static void normals_one(const float *tile, float *nrm)
{
    for (int z = 0; z < SS; ++z)
        for (int y = 0; y < SS; ++y)
            for (int x = 0; x < SS; ++x) {
                int in = x + y * IS + z * IS2;
                float value = tile[in];
                float dx = value - tile[in + 1];
                float dy = value - tile[in + IS];
                float dz = value - tile[in + IS2];
                float len2 = dx * dx + dy * dy + dz * dz;
                float len = sqrtf(len2);
                float *o = nrm + 3 * (x + y * SS + z * SS2);
                o[0] = dx / len;
                o[1] = dy / len;
                o[2] = dz / len;
            }
}

void vto_sample_normal(const float *samples, int32_t n_blocks, float *normals)
{
    for (int32_t b = 0; b < n_blocks; ++b)
        normals_one(samples + (int64_t)b * IS3, normals + (int64_t)b * SS3 * 3);
}

/* CollectTriNum.compute:23-38 == MarchingCube.compute:52-67 */
static void fill_cube(const float *tile, int cx, int cy, int cz, float cube[8])
{
    int start = cx + cy * IS + cz * IS * IS;
    cube[0] = tile[start];
    cube[1] = tile[start + 1];
    cube[2] = tile[start + 1 + IS];
    cube[3] = tile[start + IS];
    cube[4] = tile[start + IS * IS];
    cube[5] = tile[start + 1 + IS * IS];
    cube[6] = tile[start + 1 + IS + IS * IS];
    cube[7] = tile[start + IS + IS * IS];
}

/* CollectTriNum.compute:41-64: strict '>' (NaN => outside), flags stored x + 8y + 64z */
static uint32_t classify_one(const float *tile, uint32_t *flags)
{
    uint32_t total = 0;
    for (int z = 0; z < BS; ++z)
        for (int y = 0; y < BS; ++y)
            for (int x = 0; x < BS; ++x) {
                float cube[8];
                fill_cube(tile, x, y, z, cube);
                int flag = 0;
                for (int i = 0; i < 8; ++i)
                    if (cube[i] > 0) flag |= 1 << i;
                total += (uint32_t)g_trinum[flag];
                flags[x + y * BS + z * BS * BS] = (uint32_t)flag;
            }
    return total;
}

uint32_t vto_collect_tri_num(const float *samples, int32_t n_blocks, uint32_t *corner_flags)
{
    build_tables();
    uint32_t total = 0;
    for (int32_t b = 0; b < n_blocks; ++b)
        total += classify_one(samples + (int64_t)b * IS3, corner_flags + (int64_t)b * 512);
    return total;
}

static inline float lerpf(float u, float v, float t) { return u + t * (v - u); } /* HLSL lerp */

/* MarchingCube.compute:69-99 -- coord0 = floor, coord1 = ceil, t = coord - coord0; x, then y, then z */
static void normal_trilinear(const float *nrm, const float coord[3], float out[3])
{
    int c0[3], c1[3];
    float t[3];
    for (int k = 0; k < 3; ++k) {
        float f = floorf(coord[k]);
        c0[k] = (int)f;
        t[k] = coord[k] - (float)c0[k];
        c1[k] = (int)ceilf(coord[k]);
    }
#define NRM(ix, iy, iz) (nrm + 3 * ((ix) + (iy) * SS + (iz) * SS2))
    const float *c000 = NRM(c0[0], c0[1], c0[2]);
    const float *c100 = NRM(c1[0], c0[1], c0[2]);
    const float *c010 = NRM(c0[0], c1[1], c0[2]);
    const float *c001 = NRM(c0[0], c0[1], c1[2]);
    const float *c110 = NRM(c1[0], c1[1], c0[2]);
    const float *c011 = NRM(c0[0], c1[1], c1[2]);
    const float *c101 = NRM(c1[0], c0[1], c1[2]);
    const float *c111 = NRM(c1[0], c1[1], c1[2]);
#undef NRM
    for (int k = 0; k < 3; ++k) {
        float c00 = lerpf(c000[k], c100[k], t[0]);
        float c10 = lerpf(c010[k], c110[k], t[0]);
        float c01 = lerpf(c001[k], c101[k], t[0]);
        float c11 = lerpf(c011[k], c111[k], t[0]);
        float c0v = lerpf(c00, c10, t[1]);
        float c1v = lerpf(c01, c11, t[1]);
        out[k] = lerpf(c0v, c1v, t[2]);
    }
}

/* MarchingCube.compute:101-165 for one block, canonical order instead of the atomic append. */
static int32_t march_one(const float *tile, const float *nrm, const uint32_t *flags, int32_t block_id,
                         vto_triangle *out)
{
    int32_t n_out = 0;
    for (int cell = 0; cell < 512; ++cell) {
        int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
        int flag = (int)flags[cell];
        int edge_flag = g_edge[flag];
        if (edge_flag == 0) continue;
        float cube[8];
        fill_cube(tile, cx, cy, cz, cube);
        float mv[12][3];
        float cell_min[3] = {(float)cx, (float)cy, (float)cz};
        for (int e = 0; e < 12; ++e) {
            if ((edge_flag & (1 << e)) == 0) continue;
            int a = k_edge_conn[e][0], b = k_edge_conn[e][1];
            float t = (-cube[a]) / (cube[b] - cube[a]);
            for (int k = 0; k < 3; ++k) {
                float u = cell_min[k] + (float)k_vert_off[a][k];
                float v = cell_min[k] + (float)k_vert_off[b][k];
                mv[e][k] = lerpf(u, v, t);
            }
        }
        for (int i = 0; i < 5; ++i) {
            const int32_t *row = g_vert + flag * 15 + i * 3;
            if (row[0] < 0) continue; /* MarchingCube.compute:141: 'if', not 'break' */
            if (out) {
                vto_triangle *tri = out + n_out;
                const float *p0 = mv[row[0]];
                const float *p1 = mv[row[2]]; /* winding swap, MarchingCube.compute:151 */
                const float *p2 = mv[row[1]];
                memcpy(tri->position0, p0, 12);
                memcpy(tri->position1, p1, 12);
                memcpy(tri->position2, p2, 12);
                normal_trilinear(nrm, p0, tri->normal0);
                normal_trilinear(nrm, p1, tri->normal1);
                normal_trilinear(nrm, p2, tri->normal2);
                tri->block = block_id;
            }
            ++n_out;
        }
    }
    return n_out;
}

int64_t vto_marching_cube(const float *samples, const float *normals, const uint32_t *corner_flags,
                          int32_t n_blocks, vto_triangle *meshes, int32_t *block_tri_offsets)
{
    build_tables();
    int64_t total = 0;
    for (int32_t b = 0; b < n_blocks; ++b) {
        if (block_tri_offsets) block_tri_offsets[b] = (int32_t)total;
        total += march_one(samples + (int64_t)b * IS3, normals + (int64_t)b * SS3 * 3,
                           corner_flags + (int64_t)b * 512, b, meshes ? meshes + total : NULL);
    }
    if (block_tri_offsets) block_tri_offsets[n_blocks] = (int32_t)total;
    return total;
}

/* VoxelTerrain.cs:430-446 */
void vto_bin_triangles(const vto_triangle *tris, int64_t n_tris, int32_t n_blocks, float voxel_scale,
                       float *vertices, float *normals, int32_t *block_tri_offsets)
{
    int32_t *fill = (int32_t *)calloc((size_t)n_blocks + 1, sizeof(int32_t));
    for (int64_t i = 0; i < n_tris; ++i) fill[tris[i].block + 1]++;
    block_tri_offsets[0] = 0;
    for (int32_t b = 0; b < n_blocks; ++b) block_tri_offsets[b + 1] = block_tri_offsets[b] + fill[b + 1];
    memset(fill, 0, ((size_t)n_blocks + 1) * sizeof(int32_t));
    for (int64_t i = 0; i < n_tris; ++i) {
        const vto_triangle *t = tris + i;
        int64_t slot = (int64_t)block_tri_offsets[t->block] + fill[t->block]++;
        float *v = vertices + slot * 9, *n = normals + slot * 9;
        for (int k = 0; k < 3; ++k) {
            v[k] = t->position0[k] * voxel_scale;
            v[3 + k] = t->position1[k] * voxel_scale;
            v[6 + k] = t->position2[k] * voxel_scale;
            n[k] = t->normal0[k];
            n[3 + k] = t->normal1[k];
            n[6 + k] = t->normal2[k];
        }
    }
    free(fill);
}

/* VoxelTerrain.cs:330-427 end to end, one block at a time so no B-sized intermediates are kept. */
int64_t vto_extract_grid(const float *grid, int64_t sx, int64_t sy, int64_t sz,
                         const int32_t *block_list, int32_t n_blocks,
                         vto_triangle *meshes, int64_t capacity, int32_t *block_tri_offsets,
                         uint8_t *cases_u8, int32_t n_threads)
{
    build_tables();
    if (n_threads < 1) n_threads = 1;
    int32_t *offs = block_tri_offsets;
    int32_t *own = NULL;
    if (!offs) offs = own = (int32_t *)malloc(((size_t)n_blocks + 1) * sizeof(int32_t));

    /* pass 1: classify + count (CollectTriNum) */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) if (n_threads > 1)
    for (int32_t b = 0; b < n_blocks; ++b) {
        float tile[IS3];
        uint32_t flags[512];
        gather_one(grid, sx, sy, sz, block_list + 3 * (int64_t)b, tile);
        offs[b + 1] = (int32_t)classify_one(tile, flags);
        if (cases_u8)
            for (int c = 0; c < 512; ++c) cases_u8[(int64_t)b * 512 + c] = (uint8_t)flags[c];
    }
    offs[0] = 0;
    for (int32_t b = 0; b < n_blocks; ++b) offs[b + 1] += offs[b];
    int64_t total = offs[n_blocks];

    if (meshes) {
        if (capacity < total) {
            free(own);
            return -1;
        }
        /* pass 2: normals + emit (SampleNormal + MarchingCube), only where triangles exist */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) if (n_threads > 1)
        for (int32_t b = 0; b < n_blocks; ++b) {
            if (offs[b + 1] == offs[b]) continue;
            float tile[IS3];
            float nrm[SS3 * 3];
            uint32_t flags[512];
            gather_one(grid, sx, sy, sz, block_list + 3 * (int64_t)b, tile);
            classify_one(tile, flags);
            normals_one(tile, nrm);
            march_one(tile, nrm, flags, b, meshes + offs[b]);
        }
    }
    free(own);
    return total;
}

/* ---------------------------------------------------------------------------------------------
 * Indexed (welded) form of the same meshes -- the checker of the HIP library's VTMC_OUTPUT_INDEXED
 * mode.  The reference has no such format (it welds later with Mesh.Optimize(), VoxelTerrain.cs:460):
 * the welding RULE below is the build's own, everything it evaluates is the reference's arithmetic.
 *   vertex  = one per lattice edge of the block's 9^3 lattice whose endpoints differ in sign class
 *             (CollectTriNum.compute:50, strict '>').  Every such edge is a cube edge of exactly one
 *             OWNER cell: the cell whose corner 0 is the edge's low point -- cube edges 0 (x), 3 (y),
 *             8 (z), MarchingCube.compute:40-43 -- or, where that cell would lie outside the block
 *             (low point on the x = 8 / y = 8 / z = 8 face), the boundary cell next to it, as its cube
 *             edge 1, 9 (x = 8), 2, 11 (y = 8), 4, 7 (z = 8), 10, 5, 6 (two faces).  Vertices are ordered
 *             by owner cell x + 8y + 64z, then by cube edge id: the order falls out of the cells'
 *             cases (edge mask & ownership mask), so a wave numbers 64 cells with one prefix sum;
 *             position = MarchingCube.compute:128-133 evaluated from the edge's LOW endpoint
 *             (a = low, b = high: t = -cube[a] / (cube[b] - cube[a]), lerp(a, b, t)), normal =
 *             SampleNormalTrilinear (MarchingCube.compute:69-99) at that position;
 *   indices = for every triangle of the canonical order, the block-local ids of the vertices on its
 *             three edges, with the reference's winding swap (MarchingCube.compute:147-157).
 * Returns T; *n_vertices receives V.  vertices / indices may be NULL (count only).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float position[3];
    float normal[3];
} vto_vertex;

int64_t vto_extract_grid_indexed(const float *grid, int64_t sx, int64_t sy, int64_t sz, const int32_t *block_list,
                                 int32_t n_blocks, vto_vertex *vertices, int64_t vertex_capacity, int32_t *indices,
                                 int64_t tri_capacity, int32_t *block_vertex_offsets, int32_t *block_tri_offsets,
                                 int64_t *n_vertices)
{
    build_tables();
    int64_t T = 0, V = 0;
    static const int step[3] = {1, IS, IS2};
    for (int32_t b = 0; b < n_blocks; ++b) {
        float tile[IS3];
        float nrm[SS3 * 3];
        uint32_t flags[512];
        int32_t vid[SS3][3];
        gather_one(grid, sx, sy, sz, block_list + 3 * (int64_t)b, tile);
        if (block_vertex_offsets) block_vertex_offsets[b] = (int32_t)V;
        if (block_tri_offsets) block_tri_offsets[b] = (int32_t)T;
        if (classify_one(tile, flags) == 0) continue;
        normals_one(tile, nrm);
        int32_t nv = 0;
        for (int p = 0; p < SS3; ++p) vid[p][0] = vid[p][1] = vid[p][2] = -1;
        for (int cell = 0; cell < 512; ++cell) {
            const int cc[3] = {cell & 7, (cell >> 3) & 7, cell >> 6};
            const int mask = g_edge[flags[cell]];   /* cube edges with a sign change, VoxelTerrain.cs:489-507 */
            for (int e = 0; e < 12 && mask; ++e) {
                if (!((mask >> e) & 1)) continue;
                const int a = k_edge_conn[e][0], bb = k_edge_conn[e][1];
                int lo[3], axis = 0, owned = 1;
                for (int q = 0; q < 3; ++q) {
                    const int oa = k_vert_off[a][q], ob = k_vert_off[bb][q];
                    if (oa != ob) axis = q;
                    lo[q] = oa < ob ? oa : ob;   /* offset of the edge's low point from the cell's corner 0 */
                }
                for (int q = 0; q < 3; ++q)      /* owned: low point at corner 0, or beyond the block on that axis */
                    if (q != axis && lo[q] == 1 && cc[q] != BS - 1) owned = 0;
                if (!owned) continue;
                const int c[3] = {cc[0] + lo[0], cc[1] + lo[1], cc[2] + lo[2]};
                const int p = c[0] + c[1] * SS + c[2] * SS2, ti = c[0] + c[1] * IS + c[2] * IS2;
                vid[p][axis] = nv;
                if (vertices) {
                    if (V + nv >= vertex_capacity) return -1;
                    vto_vertex *o = vertices + V + nv;
                    const float va = tile[ti], vb = tile[ti + step[axis]];
                    const float t = (-va) / (vb - va);
                    for (int k = 0; k < 3; ++k) o->position[k] = lerpf((float)c[k], (float)(c[k] + (k == axis)), t);
                    normal_trilinear(nrm, o->position, o->normal);
                }
                ++nv;
            }
        }
        for (int cell = 0; cell < 512; ++cell) {
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const int flag = (int)flags[cell];
            for (int i = 0; i < 5; ++i) {
                const int32_t *row = g_vert + flag * 15 + i * 3;
                if (row[0] < 0) continue;
                if (indices) {
                    if (T >= tri_capacity) return -1;
                    const int order[3] = {row[0], row[2], row[1]};
                    for (int k = 0; k < 3; ++k) {
                        const int e = order[k], a = k_edge_conn[e][0], bb = k_edge_conn[e][1];
                        int lo[3], axis = 0;
                        for (int q = 0; q < 3; ++q) {
                            const int oa = k_vert_off[a][q], ob = k_vert_off[bb][q];
                            if (oa != ob) axis = q;
                            lo[q] = (q == 0 ? cx : (q == 1 ? cy : cz)) + (oa < ob ? oa : ob);
                        }
                        indices[3 * T + k] = vid[lo[0] + lo[1] * SS + lo[2] * SS2][axis];
                    }
                }
                ++T;
            }
        }
        V += nv;
    }
    if (block_vertex_offsets) block_vertex_offsets[n_blocks] = (int32_t)V;
    if (block_tri_offsets) block_tri_offsets[n_blocks] = (int32_t)T;
    if (n_vertices) *n_vertices = V;
    return T;
}
