// mc_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the marching-cubes path.
//
// Replaces the reference's three Unity compute kernels (paths relative to
// /root/reference/Unity-Project/Assets/):
//   Shaders/CollectTriNum.compute:41-64   -> classify_blocks_kernel / classify_dense_kernel
//   (single global atomic, VoxelTerrain.cs:394-395 read-back) -> scan_*_kernel (prefix sums)
//   Shaders/SampleNormal.compute:23-34 + Shaders/MarchingCube.compute:101-165 -> emit_kernel (fused)
//
// Design: HBM-bound table-lookup + lerp work, no MFMA.  One wavefront owns one 8x8x8
// block (lane = (x,y) column, loop over z), so every prefix sum is a ballot/mbcnt wave primitive
// and no workgroup barrier sits inside a block.  Lookup tables, the 10^3 density tile, the
// per-cell cases, the triangle-slot map and a 64-triangle staging area live in LDS; triangles
// leave through LDS so the 76-byte records are written as one contiguous, coalesced dword stream
// at offsets fixed by the scan (deterministic canonical order instead of the reference's atomics).
#include "vtmc_internal.h"

namespace vtmc {

// ----------------------------------------------------------------------------------------------
// small wave64 helpers
// ----------------------------------------------------------------------------------------------
typedef unsigned long long u64;

// LDS traffic of ONE wave is processed in issue order, so intra-wave producer/consumer hand-offs
// through LDS only need the compiler kept from reordering -- no s_barrier.
#define VTMC_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

__device__ __forceinline__ unsigned lanes_below(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Exclusive wave prefix sum of a per-lane value in 0..7 from three ballots (no LDS, no shuffles).
__device__ __forceinline__ unsigned wave_prefix3(unsigned n, unsigned &total)
{
    u64 m0 = __builtin_amdgcn_ballot_w64((n & 1u) != 0);
    u64 m1 = __builtin_amdgcn_ballot_w64((n & 2u) != 0);
    u64 m2 = __builtin_amdgcn_ballot_w64((n & 4u) != 0);
    total = (unsigned)__builtin_popcountll(m0) + 2u * (unsigned)__builtin_popcountll(m1) +
            4u * (unsigned)__builtin_popcountll(m2);
    return lanes_below(m0) + 2u * lanes_below(m1) + 4u * lanes_below(m2);
}

__device__ __forceinline__ long long block_origin(const BlockSpace &s, int b)
{
    if (s.list) {
        const int *p = s.list + 3ll * b;
        return 8ll * (p[0] * s.sx + p[1] * s.sy + p[2] * s.sz);
    }
    int v = b / s.bpv;
    int r = b - v * s.bpv;
    int q = r / s.nbx;
    int bx = r - q * s.nbx;
    int bz = q / s.nby;
    int by = q - bz * s.nby;
    return v * s.sv + 8ll * (bx * s.sx + by * s.sy + bz * s.sz);
}

// Gather of one 10x10x10 tile (VoxelTerrain.cs:341-361) straight from the grid into LDS,
// tile[ix + 10*iy + 100*iz].  The lane index walks the axis whose stride is 1.
__device__ __forceinline__ void load_tile(float *tile, const BlockSpace &s, long long org, int lane)
{
    const float *src = s.base + org;
    float v[16];
    int dst[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        int idx = it * 64 + lane;
        idx = idx < 1000 ? idx : 999;
        int a = idx % 10, t = idx / 10;
        int m = t % 10, c = t / 10;
        int ix = s.zfast ? c : a;
        int iz = s.zfast ? a : c;
        v[it] = src[ix * s.sx + m * s.sy + iz * s.sz];
        dst[it] = ix + 10 * m + 100 * iz;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) tile[dst[it]] = v[it];
}

// 4-bit half case of the cell column (x,y) at sample layer z: corners 0,1,2,3 of
// CollectTriNum.compute:27-31 (strict '>' as CollectTriNum.compute:50; NaN => outside).
__device__ __forceinline__ unsigned layer_nibble(const float *tile, int t0, int z)
{
    const float *p = tile + t0 + 100 * z;
    return (unsigned)(p[0] > 0.f) | ((unsigned)(p[1] > 0.f) << 1) | ((unsigned)(p[11] > 0.f) << 2) |
           ((unsigned)(p[10] > 0.f) << 3);
}

// ----------------------------------------------------------------------------------------------
// classify_blocks_kernel: generic per-block classify + count (any strides, block lists, tile
// batches).  One wave per block, persistent waves striding the block list.
//   counts[b]            = triangles of block b          (replaces InterlockedAdd(_TriNum[0]))
//   cases[512b + cell]   = case byte, optional           (replaces _CornerFlags)
// ----------------------------------------------------------------------------------------------
constexpr int kWavesPerWg = 4;

__global__ __launch_bounds__(256) void classify_blocks_kernel(BlockSpace sp, DeviceTables tb,
                                                               uint32_t *__restrict__ counts,
                                                               uint8_t *__restrict__ cases)
{
    __shared__ float s_tile[kWavesPerWg][1000];
    __shared__ unsigned char s_trinum[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    __syncthreads();

    float *tile = s_tile[wave];
    const int t0 = (lane & 7) + 10 * (lane >> 3);
    const int n_waves = gridDim.x * kWavesPerWg;
    for (int b = blockIdx.x * kWavesPerWg + wave; b < sp.n_blocks; b += n_waves) {
        VTMC_WAVE_SYNC();  // previous iteration's reads are done before the tile is overwritten
        load_tile(tile, sp, block_origin(sp, b), lane);
        VTMC_WAVE_SYNC();
        unsigned total = 0;
        unsigned lo = layer_nibble(tile, t0, 0);
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            unsigned hi = layer_nibble(tile, t0, z + 1);
            unsigned cs = lo | (hi << 4);
            lo = hi;
            total += s_trinum[cs];
            if (cases) cases[512ll * b + 64 * z + lane] = (uint8_t)cs;
        }
        // wave sum of per-lane totals (<= 40 each)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
        if (lane == 0) counts[b] = total;
    }
}

// ----------------------------------------------------------------------------------------------
// classify_dense_kernel: the streaming classify + count for dense volumes with stride_x == 1.
// One wave owns a brick of 64 x 8 x 8 cells (8 blocks along x): 81 coalesced 256-byte row loads
// (lane = x) + one strided load for the 65th column.  Signs are kept as bit planes, the case of a
// cell is assembled from 4 row bit-pairs, triangle counts come from the 256-byte LDS table and are
// reduced over 8-lane groups.  Each sample is requested once per brick (9/8 x 9/8 halo re-reads
// are served by L2).  No per-cell output: cases are recomputed by the emit kernel from its LDS tile.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned extract9(u64 e0, u64 e1, int r0)
{
    u64 v;
    if (r0 + 9 <= 64) v = e0 >> r0;
    else if (r0 >= 64) v = e1 >> (r0 - 64);
    else v = (e0 >> r0) | (e1 << (64 - r0));
    return (unsigned)v & 0x1FFu;
}

__global__ __launch_bounds__(256) void classify_dense_kernel(BlockSpace sp, DeviceTables tb,
                                                              uint32_t *__restrict__ counts,
                                                              int nsegx, int n_bricks, int n_wgs)
{
    __shared__ unsigned char s_trinum[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    __syncthreads();

    // XCD-aware bijective remap: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch), give
    // each XCD one contiguous range of bricks so neighbouring bricks' halo planes meet in one L2.
    int wg;
    {
        int q = n_wgs >> 3, r = n_wgs & 7, x = blockIdx.x & 7, j = blockIdx.x >> 3;
        wg = x * q + (x < r ? x : r) + j;
    }
    const int brick = wg * kWavesPerWg + wave;
    if (brick >= n_bricks) return;

    int segx = brick % nsegx;
    int t = brick / nsegx;
    int by = t % sp.nby;
    t /= sp.nby;
    int bz = t % sp.nbz;
    int v = t / sp.nbz;

    const int gx = segx * 64 + lane;                     // cell / sample x of this lane
    const int gxc = gx < sp.nx + 1 ? gx : sp.nx + 1;     // clamp loads inside the volume
    int xe = segx * 64 + 64;                             // the 65th column
    xe = xe < sp.nx + 1 ? xe : sp.nx + 1;
    const float *brick_base = sp.base + v * sp.sv + (8ll * by) * sp.sy + (8ll * bz) * sp.sz;

    // 81 row loads, lane-contiguous
    float val[9][9];
#pragma unroll
    for (int zz = 0; zz < 9; ++zz)
#pragma unroll
        for (int yy = 0; yy < 9; ++yy) val[zz][yy] = brick_base[gxc + yy * sp.sy + zz * sp.sz];
    // 65th column: row r = zz*9 + yy is read by lane r (two passes cover 81 rows)
    float ex0, ex1;
    {
        int r = lane, zz = r / 9, yy = r - 9 * zz;
        ex0 = brick_base[xe + yy * sp.sy + zz * sp.sz];
        r = lane + 64;
        r = r < 81 ? r : 80;
        zz = r / 9;
        yy = r - 9 * zz;
        ex1 = brick_base[xe + yy * sp.sy + zz * sp.sz];
    }

    unsigned A[9], N[9];
#pragma unroll
    for (int zz = 0; zz < 9; ++zz) {
        unsigned a = 0;
#pragma unroll
        for (int yy = 0; yy < 9; ++yy) a |= (unsigned)(val[zz][yy] > 0.f) << yy;
        A[zz] = a;
    }
    const u64 e0 = __builtin_amdgcn_ballot_w64(ex0 > 0.f);
    const u64 e1 = __builtin_amdgcn_ballot_w64(ex1 > 0.f);
    unsigned or_all = 0, and_all = 0x1FFu;
#pragma unroll
    for (int zz = 0; zz < 9; ++zz) {
        unsigned nb = (unsigned)__shfl_down((int)A[zz], 1);
        unsigned ne = extract9(e0, e1, zz * 9);
        N[zz] = lane == 63 ? ne : nb;
        or_all |= A[zz] | N[zz];
        and_all &= A[zz] & N[zz];
    }

    unsigned total = 0;
    const bool uniform = (or_all == 0u) || (and_all == 0x1FFu);
    if (__builtin_amdgcn_ballot_w64(!uniform) != 0) {
        // NIB[zz] nibble yy = corners (0,1,2,3) of the cell column at sample layer zz
        unsigned NIB[9];
#pragma unroll
        for (int zz = 0; zz < 9; ++zz) {
            unsigned w = 0;
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) {
                unsigned a2 = (A[zz] >> yy) & 3u, n2 = (N[zz] >> yy) & 3u;
                unsigned nib = (a2 & 1u) | (n2 << 1) | ((a2 & 2u) << 2);
                w |= nib << (4 * yy);
            }
            NIB[zz] = w;
        }
#pragma unroll
        for (int zz = 0; zz < 8; ++zz) {
            const unsigned lo = NIB[zz], hi = NIB[zz + 1];
            const bool flat = ((lo | hi) == 0u) || ((lo & hi) == 0xFFFFFFFFu);
            if (__builtin_amdgcn_ballot_w64(!flat) == 0) continue;
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) {
                unsigned cs = ((lo >> (4 * yy)) & 15u) | (((hi >> (4 * yy)) & 15u) << 4);
                total += s_trinum[cs];
            }
        }
        if (gx >= sp.nx) total = 0;  // lanes past the last cell of a partial segment
    }
    // 8-lane group sums = per-block counts
    total += __shfl_xor(total, 1);
    total += __shfl_xor(total, 2);
    total += __shfl_xor(total, 4);
    const int bx = segx * 8 + (lane >> 3);
    if ((lane & 7) == 0 && bx < sp.nbx) counts[v * sp.bpv + bx + sp.nbx * (by + sp.nby * bz)] = total;
}

// ----------------------------------------------------------------------------------------------
// scan: exclusive prefix sum of per-block triangle counts + compaction of the non-empty blocks.
// Replaces the single-address InterlockedAdd (CollectTriNum.compute:54), the 4-byte read-back
// (VoxelTerrain.cs:394-395) and the append cursor (MarchingCube.compute:160-162).
//   reduce : per 2048-block tile {triangles, non-empty blocks}
//   spine  : one workgroup scans the tile sums, publishes totals = {T, nActive}
//   apply  : per-block exclusive offsets, active list, per-volume {vertices, triangles}
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

// inclusive scan over a 256-thread workgroup of two values at once; returns totals in tot0/tot1
__device__ __forceinline__ void wg_incl_scan2(uint32_t &a, uint32_t &b, uint32_t &tot0, uint32_t &tot1,
                                              uint32_t (*s_w)[2][4])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_incl_scan(a, lane);
    b = wave_incl_scan(b, lane);
    if (lane == 63) {
        (*s_w)[0][wave] = a;
        (*s_w)[1][wave] = b;
    }
    __syncthreads();
    uint32_t ca = 0, cb = 0, ta = 0, tb2 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t xa = (*s_w)[0][w], xb = (*s_w)[1][w];
        if (w < wave) {
            ca += xa;
            cb += xb;
        }
        ta += xa;
        tb2 += xb;
    }
    a += ca;
    b += cb;
    tot0 = ta;
    tot1 = tb2;
    __syncthreads();
}

__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t *__restrict__ counts, int n,
                                                           uint32_t *__restrict__ partials)
{
    __shared__ uint32_t s_w[2][4];
    const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
    uint32_t sum = 0, act = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int i = base + k;
        uint32_t c = i < n ? counts[i] : 0u;
        sum += c;
        act += c != 0u;
    }
    uint32_t ts, ta;
    wg_incl_scan2(sum, act, ts, ta, &s_w);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = ts;
        partials[2 * blockIdx.x + 1] = ta;
    }
}

__global__ __launch_bounds__(256) void scan_spine_kernel(uint32_t *__restrict__ partials, int n_tiles,
                                                          uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_w[2][4];
    uint32_t carry_s = 0, carry_a = 0;
    for (int start = 0; start < n_tiles; start += 256) {
        int i = start + threadIdx.x;
        uint32_t s = i < n_tiles ? partials[2 * i] : 0u;
        uint32_t a = i < n_tiles ? partials[2 * i + 1] : 0u;
        uint32_t is = s, ia = a, ts, ta;
        wg_incl_scan2(is, ia, ts, ta, &s_w);
        if (i < n_tiles) {
            partials[2 * i] = carry_s + is - s;  // exclusive
            partials[2 * i + 1] = carry_a + ia - a;
        }
        carry_s += ts;
        carry_a += ta;
    }
    if (threadIdx.x == 0) {
        totals[0] = carry_s;  // T
        totals[1] = carry_a;  // number of non-empty blocks
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ counts, int n,
                                                          const uint32_t *__restrict__ partials,
                                                          uint32_t *__restrict__ offsets,
                                                          int32_t *__restrict__ active_list)
{
    __shared__ uint32_t s_w[2][4];
    const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
    uint32_t c[8];
    uint32_t sum = 0, act = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int i = base + k;
        c[k] = i < n ? counts[i] : 0u;
        sum += c[k];
        act += c[k] != 0u;
    }
    uint32_t is = sum, ia = act, ts, ta;
    wg_incl_scan2(is, ia, ts, ta, &s_w);
    uint32_t off = partials[2 * blockIdx.x] + is - sum;
    uint32_t aoff = partials[2 * blockIdx.x + 1] + ia - act;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int i = base + k;
        if (i < n) {
            offsets[i] = off;
            if (c[k] != 0u) active_list[aoff++] = i;
            off += c[k];
            if (i == n - 1) offsets[n] = off;
        }
    }
}

__global__ void volume_counts_kernel(const uint32_t *__restrict__ offsets, int bpv, int n_volumes,
                                     uint32_t *__restrict__ volume_counts)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_volumes) {
        uint32_t t = offsets[(long long)(v + 1) * bpv] - offsets[(long long)v * bpv];
        volume_counts[2 * v] = 3u * t;  // unindexed soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
        volume_counts[2 * v + 1] = t;
    }
}

// ----------------------------------------------------------------------------------------------
// emit_kernel: SampleNormal + MarchingCube fused, one wave per non-empty block.
// ----------------------------------------------------------------------------------------------
constexpr int kSlotCap = 640;      // triangle-slot map entries kept before a flush (2 x 320)
constexpr int kTriDwords = 19;     // 76-byte record

struct EmitLds {
    float tile[1000];
    unsigned short slot[kSlotCap];
    unsigned char cases[512];
    float stage[64 * kTriDwords];
};

// lattice normal of SampleNormal.compute:27-33 at tile index ti (forward differences, normalised)
__device__ __forceinline__ void lattice_normal(const float *tile, int ti, float n[3])
{
    float v = tile[ti];
    float dx = v - tile[ti + 1];
    float dy = v - tile[ti + 10];
    float dz = v - tile[ti + 100];
    float len = __fsqrt_rn(dx * dx + dy * dy + dz * dz);
    n[0] = __fdiv_rn(dx, len);
    n[1] = __fdiv_rn(dy, len);
    n[2] = __fdiv_rn(dz, len);
}

// One mesh vertex on cube edge e of cell (cx,cy,cz): position (MarchingCube.compute:128-133) and
// normal (SampleNormalTrilinear, MarchingCube.compute:69-99).  A vertex sits on a lattice edge, so
// the 8-point trilinear blend collapses to a 2-point lerp along the edge axis with the weight
// taken from the ROUNDED position (c0 = floor(P), c1 = ceil(P), t = P - c0), exactly as the
// reference derives it; the collapsed terms are exact (u + 0*(u-u)).
__device__ __forceinline__ void edge_vertex(const float *tile, int cx, int cy, int cz, unsigned e,
                                            float *pos, float *nrm)
{
    // endpoints of the 12 edges (MarchingCube.compute:40-43), one nibble each
    const u64 EA = 0x321076543210ull, EB = 0x765447650321ull;
    const unsigned a = (unsigned)(EA >> (4 * e)) & 7u, b = (unsigned)(EB >> (4 * e)) & 7u;
    // corner offsets (MarchingCube.compute:46-50) as bit sets over the corner index
    const int ax = cx + ((0x66u >> a) & 1), ay = cy + ((0xCCu >> a) & 1), az = cz + ((0xF0u >> a) & 1);
    const int bx = cx + ((0x66u >> b) & 1), by = cy + ((0xCCu >> b) & 1), bz = cz + ((0xF0u >> b) & 1);
    const float va = tile[ax + 10 * ay + 100 * az];
    const float vb = tile[bx + 10 * by + 100 * bz];
    const float t = __fdiv_rn(-va, vb - va);
    const float px = (float)ax + t * ((float)bx - (float)ax);
    const float py = (float)ay + t * ((float)by - (float)ay);
    const float pz = (float)az + t * ((float)bz - (float)az);
    pos[0] = px;
    pos[1] = py;
    pos[2] = pz;
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int c0 = (int)fx + 10 * (int)fy + 100 * (int)fz;
    const int c1 = (int)ceilf(px) + 10 * (int)ceilf(py) + 100 * (int)ceilf(pz);
    // weight along the edge axis: edges 0,2,4,6 run along x, 1,3,5,7 along y, 8..11 along z
    const float w = e >= 8u ? pz - fz : ((e & 1u) ? py - fy : px - fx);
    float n0[3], n1[3];
    lattice_normal(tile, c0, n0);
    lattice_normal(tile, c1, n1);
    nrm[0] = n0[0] + w * (n1[0] - n0[0]);
    nrm[1] = n0[1] + w * (n1[1] - n0[1]);
    nrm[2] = n0[2] + w * (n1[2] - n0[2]);
}

__device__ __forceinline__ void emit_flush(EmitLds *L, const u64 *s_vert, int pending, size_t tri_base,
                                           int block_id, float *__restrict__ out, int lane)
{
    VTMC_WAVE_SYNC();
    for (int s0 = 0; s0 < pending; s0 += 64) {
        const int s = s0 + lane;
        if (s < pending) {
            const unsigned sc = L->slot[s];
            const int cell = sc & 511u, i = sc >> 9;
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const u64 w = s_vert[L->cases[cell]] >> (12 * i);
            float *rec = L->stage + lane * kTriDwords;
            // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
            edge_vertex(L->tile, cx, cy, cz, (unsigned)w & 15u, rec + 0, rec + 9);
            edge_vertex(L->tile, cx, cy, cz, (unsigned)(w >> 8) & 15u, rec + 3, rec + 12);
            edge_vertex(L->tile, cx, cy, cz, (unsigned)(w >> 4) & 15u, rec + 6, rec + 15);
            rec[18] = __int_as_float(block_id);
        }
        VTMC_WAVE_SYNC();
        const int cnt = pending - s0 < 64 ? pending - s0 : 64;
        const int n_dw = cnt * kTriDwords;
        float *dst = out + (tri_base + (size_t)s0) * kTriDwords;
        for (int d = lane; d < n_dw; d += 64) dst[d] = L->stage[d];
        VTMC_WAVE_SYNC();
    }
}

__global__ __launch_bounds__(256) void emit_kernel(BlockSpace sp, DeviceTables tb,
                                                    const uint32_t *__restrict__ offsets,
                                                    const int32_t *__restrict__ active_list,
                                                    const uint32_t *__restrict__ totals, uint32_t capacity,
                                                    float *__restrict__ out)
{
    __shared__ EmitLds s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ unsigned char s_trinum[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    __syncthreads();

    const uint32_t total_tris = totals[0];
    const int n_active = (int)totals[1];
    if (total_tris > capacity) return;  // host grows the buffer and re-launches (vtmc_api.hip)

    EmitLds *L = &s_lds[wave];
    const int t0 = (lane & 7) + 10 * (lane >> 3);
    const int n_waves = gridDim.x * kWavesPerWg;
    for (int ai = blockIdx.x * kWavesPerWg + wave; ai < n_active; ai += n_waves) {
        const int b = active_list[ai];
        size_t tri_base = offsets[b];
        // the scan's budget for this block; flushes are clamped to it so a classify/emit mismatch
        // could never write outside the block's own slice of the triangle buffer
        int budget = (int)(offsets[b + 1] - offsets[b]);
        VTMC_WAVE_SYNC();
        load_tile(L->tile, sp, block_origin(sp, b), lane);
        VTMC_WAVE_SYNC();

        int pending = 0;
        unsigned lo = layer_nibble(L->tile, t0, 0);
        for (int z = 0; z < 8; ++z) {
            if (pending > kSlotCap - 320) {  // wave-uniform
                const int n_out = pending < budget ? pending : budget;
                emit_flush(L, s_vert, n_out, tri_base, b, out, lane);
                tri_base += n_out;
                budget -= n_out;
                pending = 0;
            }
            unsigned hi = layer_nibble(L->tile, t0, z + 1);
            unsigned cs = lo | (hi << 4);
            lo = hi;
            const int cell = 64 * z + lane;
            L->cases[cell] = (unsigned char)cs;
            unsigned n = s_trinum[cs], layer_total;
            unsigned pre = wave_prefix3(n, layer_total);
            for (unsigned i = 0; i < n; ++i) L->slot[pending + pre + i] = (unsigned short)(cell | (i << 9));
            pending += (int)layer_total;
        }
        if (pending > budget) pending = budget;
        if (pending > 0) emit_flush(L, s_vert, pending, tri_base, b, out, lane);
    }
}

// ----------------------------------------------------------------------------------------------
// launch wrappers
// ----------------------------------------------------------------------------------------------
hipError_t launch_classify_blocks(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                  uint8_t *cases_or_null, int n_cus, hipStream_t stream)
{
    int wgs = (sp.n_blocks + kWavesPerWg - 1) / kWavesPerWg;
    int cap = n_cus * 8;
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(classify_blocks_kernel, dim3(wgs), dim3(256), 0, stream, sp, tb, counts, cases_or_null);
    return hipGetLastError();
}

hipError_t launch_classify_dense(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                 hipStream_t stream)
{
    int nsegx = (sp.nx + 63) / 64;
    long long n_vol = sp.n_blocks / sp.bpv;
    long long n_bricks = n_vol * sp.nbz * sp.nby * nsegx;
    long long n_wgs = (n_bricks + kWavesPerWg - 1) / kWavesPerWg;
    if (n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(classify_dense_kernel, dim3((unsigned)n_wgs), dim3(256), 0, stream, sp, tb, counts, nsegx,
                       (int)n_bricks, (int)n_wgs);
    return hipGetLastError();
}

hipError_t launch_scan(const uint32_t *counts, int n_blocks, uint32_t *offsets, int32_t *active_list,
                       uint32_t *partials, uint32_t *totals, int bpv, int n_volumes,
                       uint32_t *volume_counts, hipStream_t stream)
{
    int n_tiles = (n_blocks + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(n_tiles), dim3(256), 0, stream, counts, n_blocks, partials);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(256), 0, stream, partials, n_tiles, totals);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(n_tiles), dim3(256), 0, stream, counts, n_blocks, partials, offsets,
                       active_list);
    if (volume_counts && n_volumes > 0)
        hipLaunchKernelGGL(volume_counts_kernel, dim3((n_volumes + 255) / 256), dim3(256), 0, stream, offsets, bpv,
                           n_volumes, volume_counts);
    return hipGetLastError();
}

hipError_t launch_emit(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets,
                       const int32_t *active_list, const uint32_t *totals, uint32_t capacity,
                       void *triangles, int n_cus, hipStream_t stream)
{
    int wgs = (sp.n_blocks + kWavesPerWg - 1) / kWavesPerWg;
    int cap = n_cus * 3;  // LDS-limited residency: 3 workgroups (12 waves) per CU
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(emit_kernel, dim3(wgs), dim3(256), 0, stream, sp, tb, offsets, active_list, totals, capacity,
                       (float *)triangles);
    return hipGetLastError();
}

}  // namespace vtmc
