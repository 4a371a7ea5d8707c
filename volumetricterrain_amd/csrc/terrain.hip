// terrain.hip -- device-resident density grid with the reference's CSG write semantics: the
// "density-field sampler" stage of the path (hand-written gfx950 / CDNA4).
//
// Replaces (paths relative to /root/reference/Unity-Project/Assets/Scripts/):
//   VoxelTerrain.cs:145-149  Init: grid filled with "void" values            -> terrain_fill_kernel
//   VoxelTerrain.cs:284-305  Update: per-sample QueryDensity + clamp + max / min -> terrain_modify_kernel
//   TerrainModifier.cs:59-62 (plane), :79-82 (sphere), :143-149 (cylinder),
//   IslandModifier.cs:45-73 (bilinear heightmap, the modifier of the world build TerrainEngine.cs:87) -> query_density
// One lane per sample of the modifier's AABB, x fastest (the grid is x fastest), so a wave reads and
// writes contiguous 256-byte row segments.  HBM-bound: 8 bytes per touched sample, a handful of
// FP32 operations in the reference's order (library built with -ffp-contract=off; sqrt is the
// correctly rounded one, as Mathf.Sqrt = (float)Math.Sqrt is).
//
// voidDensity / fullDensity (VoxelTerrain.cs:50-51) are FRESH random numbers in [-2,-1] / [1,2] on
// every read in the reference (UnityEngine.Random: a stream nobody can replay).  Here they are a
// counter-based hash of (seed, event, sample index, draw index) -- same ranges, same number of
// draws per sample (2 per add, 4 per erode, 1 per Init sample), deterministic; the CPU oracle
// restates the same hash (oracle/terrain_ref.c).
#include "vtmc_internal.h"

namespace vtmc {

__host__ __device__ __forceinline__ float terrain_uniform(uint64_t seed, uint32_t event, uint64_t sample, uint32_t draw)
{
    uint64_t z = (seed ^ ((uint64_t)event << 40) ^ (sample << 2) ^ (uint64_t)draw) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;  // 24 bits * 2^-24: exact, in [0,1)
}

__device__ __forceinline__ float clampf(float v, float lo, float hi)  // Mathf.Clamp
{
    if (v < lo) v = lo;
    else if (v > hi) v = hi;
    return v;
}

// Mathf.Clamp(v, voidDensity, fullDensity) with void = draw k, full = draw k+1.  void lies in [-2,-1)
// and full in [1,2), so a value in [-1,1] is never clamped and neither draw is evaluated for it; the
// result is the same as drawing both (each draw is a pure function of its counter).
__device__ __forceinline__ float clamp_drawn(float v, uint64_t seed, uint32_t event, uint64_t sample, uint32_t k)
{
    if (v < -1.0f) {
        const float lo = terrain_uniform(seed, event, sample, k) - 2.0f;
        if (v < lo) v = lo;
    } else if (v > 1.0f) {
        const float hi = terrain_uniform(seed, event, sample, k + 1u) + 1.0f;
        if (v > hi) v = hi;
    }
    return v;
}

__device__ __forceinline__ float lerp_unity(float a, float b, float t)  // Mathf.Lerp: a + (b - a) * Clamp01(t)
{
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    return a + (b - a) * t;
}

__device__ __forceinline__ float query_density(const TerrainModifierArgs &m, float px, float py, float pz)
{
    if (m.kind == 0) return m.p[0] - py;  // PlaneModifier: _height - pos.y
    if (m.kind == 1) {                    // SphereModifier: _radius - (pos - _center).magnitude
        const float dx = px - m.p[0], dy = py - m.p[1], dz = pz - m.p[2];
        return m.p[3] - __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    }
    // CylinderModifier: Min(projLength, _axisLength - projLength, _radius - Sqrt(|start2pos|^2 - projLength^2))
    const float sx = px - m.p[0], sy = py - m.p[1], sz = pz - m.p[2];
    const float proj = sx * m.p[3] + sy * m.p[4] + sz * m.p[5];
    const float sq = sx * sx + sy * sy + sz * sz;
    const float c = m.p[7] - __builtin_sqrtf(sq - proj * proj);
    float r = proj;  // Mathf.Min(params): `if (v < min) min = v`, so a NaN candidate is skipped
    const float b = m.p[6] - proj;
    if (b < r) r = b;
    if (c < r) r = c;
    return r;
}

__global__ __launch_bounds__(256) void terrain_fill_kernel(float *__restrict__ grid, long long n, uint64_t seed)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        grid[i] = terrain_uniform(seed, 0u, (uint64_t)i, 0u) - 2.0f;  // voidDensity
}

// What of a modifier's density does not depend on y (the heightmap's bilinear fetch): evaluated once
// per (x, z) column and reused for the kYRun samples a thread walks.
__device__ __forceinline__ float column_term(const TerrainModifierArgs &m, float px, float pz)
{
    if (m.kind != 3) return 0.0f;
    // IslandModifier.cs:45-73: bilinear interpolation of _heightmap[u, v]
    const float wm1 = (float)(m.dims0 - 1), hm1 = (float)(m.dims1 - 1);
    float u = clampf(px, 0.0f, m.p[0]);
    u = u / m.p[0] * wm1;
    u = clampf(u, 0.0f, wm1);
    float v = clampf(pz, 0.0f, m.p[1]);
    v = v / m.p[1] * hm1;
    v = clampf(v, 0.0f, hm1);
    const int u0 = (int)floorf(u), u1 = (int)ceilf(u), v0 = (int)floorf(v), v1 = (int)ceilf(v);
    const float h00 = m.data[(size_t)u0 * m.dims1 + v0], h10 = m.data[(size_t)u1 * m.dims1 + v0];
    const float h01 = m.data[(size_t)u0 * m.dims1 + v1], h11 = m.data[(size_t)u1 * m.dims1 + v1];
    const float h0 = lerp_unity(h00, h01, v - (float)v0);
    const float h1 = lerp_unity(h10, h11, v - (float)v0);
    return lerp_unity(h0, h1, u - (float)u0);
}

constexpr int kYRun = 16;  // samples along y per thread

// launch shape: 64 x 4 threads = 64 samples along x (the stride-1 axis) of 4 z-planes; a thread walks
// kYRun samples along y; grid = (x segments, z quads, y runs)
__global__ __launch_bounds__(256) void terrain_modify_kernel(float *__restrict__ grid, TerrainShape sh, TerrainModifierArgs m)
{
    const int ix = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y, iy0 = blockIdx.z * kYRun;
    if (ix >= m.dx || iz >= m.dz) return;
    const int x = m.lx + ix, z = m.lz + iz;
    // worldPos = new Vector3(x, y, z) * _voxelScale + TerrainOrigin (VoxelTerrain.cs:290)
    const float px = (float)x * sh.scale + sh.origin[0];
    const float pz = (float)z * sh.scale + sh.origin[2];
    const float col = column_term(m, px, pz);
    const int iy1 = iy0 + kYRun < m.dy ? iy0 + kYRun : m.dy;
    for (int iy = iy0; iy < iy1; ++iy) {
        const int y = m.ly + iy;
        const float py = (float)y * sh.scale + sh.origin[1];
        const uint64_t sample = (uint64_t)x + (uint64_t)sh.dim_x * ((uint64_t)y + (uint64_t)sh.dim_y * (uint64_t)z);
        const float q = m.kind == 3 ? col - py : query_density(m, px, py, pz);  // IslandModifier: elevation - pos.y
        const float md = clamp_drawn(q, sh.seed, m.event, sample, 0u);
        const float s = grid[sample];
        float r;
        if (m.add_or_erode) {
            r = s > md ? s : md;  // Mathf.Max(S, md)
        } else {
            const float minus_md = -md;
            r = clamp_drawn(s < minus_md ? s : minus_md, sh.seed, m.event, sample, 2u);  // Clamp(Min(S, -md), void, full)
        }
        grid[sample] = r;
    }
}

hipError_t launch_terrain_fill(float *grid, long long n, uint64_t seed, int n_cus, hipStream_t stream)
{
    long long wgs = (n + 255) / 256;
    if (wgs > (long long)n_cus * 16) wgs = (long long)n_cus * 16;
    if (wgs < 1) wgs = 1;
    launch_begin();
    hipLaunchKernelGGL(terrain_fill_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, grid, n, seed);
    return launch_end();
}

hipError_t launch_terrain_modify(float *grid, const TerrainShape &sh, const TerrainModifierArgs &m, hipStream_t stream)
{
    if (m.dx <= 0 || m.dy <= 0 || m.dz <= 0) return hipSuccess;
    if ((m.dz + 3) / 4 > 65535 || (m.dy + kYRun - 1) / kYRun > 65535) return hipErrorInvalidValue;
    launch_begin();
    hipLaunchKernelGGL(terrain_modify_kernel, dim3((unsigned)((m.dx + 63) / 64), (unsigned)((m.dz + 3) / 4), (unsigned)((m.dy + kYRun - 1) / kYRun)),
                       dim3(64, 4, 1), 0, stream, grid, sh, m);
    return launch_end();
}

}  // namespace vtmc
