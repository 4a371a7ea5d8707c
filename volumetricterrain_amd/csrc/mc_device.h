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
// through LDS need no s_barrier and no hardware wait -- only the compiler kept from moving LDS
// accesses across the hand-off.  A wavefront-scope FENCE does that too, but hipcc lowers it to
// s_waitcnt vmcnt(0) as well: every hand-off then also waits for the wave's outstanding global
// stores (the triangle stream) to complete, which exposes the full store latency once per batch.
#define VTMC_WAVE_SYNC()                                        \
    do {                                                        \
        asm volatile("" ::: "memory");                          \
        __builtin_amdgcn_wave_barrier();                        \
        asm volatile("" ::: "memory");                          \
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
        // the dirty list is written before the launch and never during it: read through the constant address space, i.e. by scalar loads
        // (b is wave-uniform) -- a vector load here would put a `vmcnt(0)` of the compiler's into the emit kernels' asynchronous loop
        typedef const __attribute__((address_space(4))) int *const_int_ptr;
        const_int_ptr p = (const_int_ptr)(s.list + 3ll * b);
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

__device__ __forceinline__ unsigned extract9(u64 e0, u64 e1, int r0)
{
    u64 v;
    if (r0 + 9 <= 64) v = e0 >> r0;
    else if (r0 >= 64) v = e1 >> (r0 - 64);
    else v = (e0 >> r0) | (e1 << (64 - r0));
    return (unsigned)v & 0x1FFu;
}

// Streaming classify of one brick of 64 x 8 x 8 cells (8 blocks along x) by one wave, lane = x:
// 81 coalesced 256-byte row loads + one strided load for the 65th column.  Signs are kept as bit
// planes, the case of a cell (CollectTriNum.compute:27-51) is assembled from 4 row bit-pairs,
// triangle counts come from the 256-byte LDS table.  Returns the triangle count of this lane's
// 8 x 8 cell column; the sum over an 8-lane group is a block's count.
// WANT_V: *vcount receives this lane's share of the block's WELDED vertex count = lattice edges with
// a sign change (every such edge carries exactly one mesh vertex): the x-, y- and z-edges starting
// in this lane's 9 x 9 sample column, plus -- for the last lane of an 8-lane group -- the y- and
// z-edges of the block's far x = 8 plane.  Same enumeration as the indexed emit (emit_device.h).
// *rowmask (optional) receives, for this lane's cell column, the layers that hold a cell with triangles:
// bits 0-7 over y, bits 8-15 over z.  OR-ed over a block they tell the emit kernel which rows of the
// block's 10^3 tile it will touch at all.
// FROM_BITS: the samples' signs come from a sign volume (one bit per sample, written by the sampler: density.hip) instead of
// the samples: per row two wave-uniform 64-bit words (scalar loads) hold the 65 bits of the brick's row; `sign_plane` points at
// the plane of the brick's first z layer, `p0` is the bit offset of its first row (x0 + dx * y0), `dxs` the row pitch in bits.
template <bool WANT_V = false, bool FROM_BITS = false>
__device__ __forceinline__ unsigned classify_brick_column(const BlockSpace &sp, const unsigned char *s_trinum,
                                                          const float *brick_base, int gx, int gxc, int xe, int lane,
                                                          int ablate = 0, unsigned *vcount = nullptr, unsigned *rowmask = nullptr,
                                                          const unsigned long long *sign_plane = nullptr, int plane_words = 0, int p0 = 0,
                                                          int dxs = 0)
{
    unsigned A[9], N[9];
    unsigned or_all = 0, and_all = 0x1FFu;
    if constexpr (FROM_BITS) {
        // lane r holds the 65-bit window of row r = zz * 9 + yy (two 16-byte loads per lane cover the 81 rows): bits x0 .. x0 + 63
        // in `win`, bit x0 + 64 in `e`
        auto row_window = [&](int r, unsigned long long &win, bool &e) {
            const int zz = r / 9, yy = r - 9 * zz;
            const int bit0 = p0 + yy * dxs;
            const unsigned long long *q = sign_plane + (long long)zz * plane_words + (bit0 >> 6);
            const unsigned long long w0 = q[0], w1 = q[1];
            const int sh = bit0 & 63;
            win = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            e = ((w1 >> sh) & 1ull) != 0ull;
        };
        unsigned long long win_a, win_b;
        bool e_a, e_b;
        row_window(lane, win_a, e_a);
        row_window(lane + 64 < 81 ? lane + 64 : 80, win_b, e_b);
        // a brick whose 81 x 65 sign bits are all equal holds no triangle and no welded vertex -- four of five bricks end here
        const bool none = win_a == 0ull && win_b == 0ull && !e_a && !e_b, all = win_a == ~0ull && win_b == ~0ull && e_a && e_b;
        if (__builtin_amdgcn_ballot_w64(!none) == 0ull || __builtin_amdgcn_ballot_w64(!all) == 0ull) {
            if (WANT_V) *vcount = 0;
            if (rowmask) *rowmask = 0;
            return 0;
        }
        const u64 e0 = __builtin_amdgcn_ballot_w64(e_a);
        const u64 e1 = __builtin_amdgcn_ballot_w64(e_b);
        const unsigned a_lo = (unsigned)win_a, a_hi = (unsigned)(win_a >> 32), b_lo = (unsigned)win_b, b_hi = (unsigned)(win_b >> 32);
#pragma unroll
        for (int zz = 0; zz < 9; ++zz) {
            unsigned a = 0;
#pragma unroll
            for (int yy = 0; yy < 9; ++yy) {
                const int r = zz * 9 + yy;   // compile-time: the row's window comes out of lane r by v_readlane
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(r < 64 ? a_lo : b_lo), r & 63);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(r < 64 ? a_hi : b_hi), r & 63);
                const unsigned long long win = ((unsigned long long)hi << 32) | lo;
                a |= (unsigned)((win >> lane) & 1ull) << yy;
            }
            A[zz] = a;
        }
#pragma unroll
        for (int zz = 0; zz < 9; ++zz) {
            const unsigned nb = (unsigned)__shfl_down((int)A[zz], 1);
            const unsigned ne = extract9(e0, e1, zz * 9);
            N[zz] = lane == 63 ? ne : nb;
            or_all |= A[zz] | N[zz];
            and_all &= A[zz] & N[zz];
        }
    } else {
    // 81 row loads, lane-contiguous
    float val[9][9];
#pragma unroll
    for (int zz = 0; zz < 9; ++zz)
#pragma unroll
        for (int yy = 0; yy < 9; ++yy) {
            // diagnostics only (ablate & 1): alias the halo rows to their neighbours, i.e. no halo traffic
            const int y2 = (ablate & 1) && yy == 8 ? 7 : yy, z2 = (ablate & 1) && zz == 8 ? 7 : zz;
            const float *src = brick_base + gxc + y2 * sp.sy + z2 * sp.sz;
            val[zz][yy] = (ablate & 2) ? __builtin_nontemporal_load(src) : *src;  // diagnostics: streaming-hint loads
        }
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

#pragma unroll
    for (int zz = 0; zz < 9; ++zz) {
        unsigned a = 0;
#pragma unroll
        for (int yy = 0; yy < 9; ++yy) a |= (unsigned)(val[zz][yy] > 0.f) << yy;
        A[zz] = a;
    }
    const u64 e0 = __builtin_amdgcn_ballot_w64(ex0 > 0.f);
    const u64 e1 = __builtin_amdgcn_ballot_w64(ex1 > 0.f);
#pragma unroll
    for (int zz = 0; zz < 9; ++zz) {
        unsigned nb = (unsigned)__shfl_down((int)A[zz], 1);
        unsigned ne = extract9(e0, e1, zz * 9);
        N[zz] = lane == 63 ? ne : nb;
        or_all |= A[zz] | N[zz];
        and_all &= A[zz] & N[zz];
    }
    }

    unsigned total = 0, rows = 0, yacc = 0, vtotal = 0;
    const bool uniform = (or_all == 0u) || (and_all == 0x1FFu);
    if (__builtin_amdgcn_ballot_w64(!uniform) != 0) {
        // NIB[zz] nibble yy = corners (0,1,2,3) of the cell column at sample layer zz
        // built bit-parallel: the 8 + 1 sign bits of a column are spread to every 4th bit (three shift-or-mask
        // steps), the neighbours' follow by a shift -- 21 VALU per plane instead of 8 per cell
        unsigned NIB[9];
        auto spread4 = [](unsigned x) {   // bit i (i < 8) -> bit 4 i
            x = (x | (x << 12)) & 0x000F000Fu;
            x = (x | (x << 6)) & 0x03030303u;
            return (x | (x << 3)) & 0x11111111u;
        };
#pragma unroll
        for (int zz = 0; zz < 9; ++zz) {
            const unsigned a0 = spread4(A[zz] & 0xFFu), n0 = spread4(N[zz] & 0xFFu);
            const unsigned a1 = (a0 >> 4) | ((A[zz] & 0x100u) << 20), n1 = (n0 >> 4) | ((N[zz] & 0x100u) << 20);   // rows y + 1
            NIB[zz] = a0 | (n0 << 1) | (n1 << 2) | (a1 << 3);   // corners 0 (x,y), 1 (x+1,y), 2 (x+1,y+1), 3 (x,y+1)
        }
#pragma unroll
        for (int zz = 0; zz < 8; ++zz) {
            const unsigned lo = NIB[zz], hi = NIB[zz + 1];
            const bool flat = ((lo | hi) == 0u) || ((lo & hi) == 0xFFFFFFFFu);
            if (__builtin_amdgcn_ballot_w64(!flat) == 0) continue;
            if (WANT_V) {   // lattice edges with a sign change in sample plane zz (x-, y-edges), between planes zz and zz + 1
                            // (z-edges) and, from the last layer, in plane 8: a layer skipped above holds none of them
                const bool far_plane = (lane & 7) == 7;
                vtotal += __builtin_popcount((A[zz] ^ N[zz]) & 0x1FFu) + __builtin_popcount((A[zz] ^ (A[zz] >> 1)) & 0xFFu) +
                          __builtin_popcount((A[zz] ^ A[zz + 1]) & 0x1FFu);
                if (far_plane)
                    vtotal += __builtin_popcount((N[zz] ^ (N[zz] >> 1)) & 0xFFu) + __builtin_popcount((N[zz] ^ N[zz + 1]) & 0x1FFu);
                if (zz == 7) {
                    vtotal += __builtin_popcount((A[8] ^ N[8]) & 0x1FFu) + __builtin_popcount((A[8] ^ (A[8] >> 1)) & 0xFFu);
                    if (far_plane) vtotal += __builtin_popcount((N[8] ^ (N[8] >> 1)) & 0xFFu);
                }
            }
            if (!(ablate & 8)) {   // cells of this layer with a mixed case (= with triangles), one bit per nibble: not all 0, not all 1
                const unsigned any = lo | hi, all = lo & hi;
                const unsigned nz = (any | (any >> 1) | (any >> 2) | (any >> 3)) & 0x11111111u;
                const unsigned full = (all & (all >> 1) & (all >> 2) & (all >> 3)) & 0x11111111u;
                const unsigned act = nz & ~full;
                yacc |= act;
                rows |= act ? (0x100u << zz) : 0u;
            }
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) {
                unsigned cs = ((lo >> (4 * yy)) & 15u) | (((hi >> (4 * yy)) & 15u) << 4);
                total += s_trinum[cs];
            }
        }
        {   // nibble-spaced y bits -> bits 0-7
            unsigned x = yacc;
            x = (x | (x >> 3)) & 0x03030303u;
            x = (x | (x >> 6)) & 0x000F000Fu;
            x = (x | (x >> 12)) & 0xFFu;
            rows |= x;
        }
        if (gx >= sp.nx) total = 0, rows = 0, vtotal = 0;  // lanes past the last cell of a partial segment
    }
    if (WANT_V) *vcount = vtotal;
    if (rowmask) *rowmask = rows;
    return total;
}

}  // namespace vtmc
#endif
