// mc_device.h -- device-side helpers shared by the classify and emit kernels (gfx950, wave64).
#ifndef VTMC_MC_DEVICE_H
#define VTMC_MC_DEVICE_H
#include "vtmc_internal.h"

namespace vtmc {

constexpr int kWavesPerWg = 4;

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
    const int v = (int)s.d_bpv.quot((unsigned)b);
    const int r = b - v * s.bpv;
    const int q = (int)s.d_nbx.quot((unsigned)r);
    const int bx = r - q * s.nbx;
    const int bz = (int)s.d_nby.quot((unsigned)q);
    const int by = q - bz * s.nby;
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

}  // namespace vtmc
#endif
