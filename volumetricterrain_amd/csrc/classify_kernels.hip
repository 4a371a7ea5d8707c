// classify_kernels.hip -- classify + count + scan kernels (hand-written gfx950 / CDNA4, wave64).
//
// Replaces (paths relative to /root/reference/Unity-Project/Assets/):
//   Shaders/CollectTriNum.compute:41-64   -> classify_blocks_kernel / classify_dense_kernel
//   the single global atomic (CollectTriNum.compute:54), the 4-byte read-back
//   (Scripts/VoxelTerrain.cs:394-395) and the append cursor (MarchingCube.compute:160-162)
//                                          -> scan_*_kernel (prefix sums + compaction)
// HBM-bound integer/byte work: coalesced row loads, sign bit-planes in registers, the 256-byte
// triangle-count table in LDS, wave ballots / shuffles for every reduction.  No MFMA.
#include "mc_device.h"

namespace vtmc {

// ----------------------------------------------------------------------------------------------
// classify_blocks_kernel: generic per-block classify + count (any strides, block lists, tile
// batches).  One wave per block, persistent waves striding the block list.
//   counts[b]            = triangles of block b          (replaces InterlockedAdd(_TriNum[0]))
//   cases[512b + cell]   = case byte, optional           (replaces _CornerFlags)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void classify_blocks_kernel(BlockSpace sp, DeviceTables tb,
                                                               uint32_t *__restrict__ counts,
                                                               uint8_t *__restrict__ cases,
                                                               uint32_t *__restrict__ vcounts,
                                                               unsigned long long *__restrict__ scan_ctrl, int n_scan_ctrl)
{
    __shared__ float s_tile[kWavesPerWg][1000];
    __shared__ unsigned char s_trinum[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the fused scan that follows on the stream finds its ticket counter and tile status words zeroed
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_scan_ctrl; i += gridDim.x * 256) scan_ctrl[i] = 0ull;
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    __syncthreads();

    float *tile = s_tile[wave];
    const int t0 = (lane & 7) + 10 * (lane >> 3);
    const int n_waves = gridDim.x * kWavesPerWg;
    for (int b = blockIdx.x * kWavesPerWg + wave; b < sp.n_blocks; b += n_waves) {
        VTMC_WAVE_SYNC();  // previous iteration's reads are done before the tile is overwritten
        load_tile(tile, sp, block_origin(sp, b), lane);
        VTMC_WAVE_SYNC();
        unsigned total = 0, zmask = 0;
        bool any = false;
        unsigned lo = layer_nibble(tile, t0, 0);
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            unsigned hi = layer_nibble(tile, t0, z + 1);
            unsigned cs = lo | (hi << 4);
            lo = hi;
            const unsigned n = s_trinum[cs];
            total += n;
            any |= n != 0u;
            zmask |= (__builtin_amdgcn_ballot_w64(n != 0u) != 0 ? 1u : 0u) << z;
            if (cases) cases[512ll * b + 64 * z + lane] = (uint8_t)cs;
        }
        // wave sum of per-lane totals (<= 40 each)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
        {   // layers that hold a cell with triangles: bits 16-23 over y (lane >> 3), 24-31 over z, above the count
            const u64 cols = __builtin_amdgcn_ballot_w64(any);
            unsigned ymask = 0;
#pragma unroll
            for (int y = 0; y < 8; ++y) ymask |= ((cols >> (8 * y)) & 0xFFull) ? (1u << y) : 0u;
            if (lane == 0) counts[b] = total | (ymask << 16) | (zmask << 24);
        }
        if (vcounts) {
            // welded vertices = lattice edges of the 9^3 lattice with a sign change, enumerated as the
            // indexed emit does: point p = x + 9y + 81z, axes x, y, z
            unsigned v = 0;
            for (int p0 = 0; p0 < 729; p0 += 64) {
                const int p = p0 + lane;
                if (p < 729) {
                    const int x = p % 9, y = (p / 9) % 9, z = p / 81;
                    const float *q = tile + x + 10 * y + 100 * z;
                    const bool s0 = q[0] > 0.f;
                    v += (x < 8 && s0 != (q[1] > 0.f)) + (y < 8 && s0 != (q[10] > 0.f)) + (z < 8 && s0 != (q[100] > 0.f));
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) vcounts[b] = v;
        }
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
// FROM_BITS: the signs come from the sampler's sign volume (SignVolume: tuning key "fill_keeps_signs") -- 2 x 8 bytes per row through the
// scalar unit instead of 260 bytes per row through the vector memory path; the kernel is then instruction-bound.
#ifdef VTMC_TIMELINE   // diagnostic build only (tools/classify_timeline.py): when every wave of the streaming classify started and ended, and where
__device__ unsigned long long *g_vtmc_timeline = nullptr;   // [brick][4] = {start, end (s_memrealtime: 100 MHz), HW_ID, XCC_ID}
#endif

template <bool WANT_V, bool FROM_BITS>
__global__ __launch_bounds__(256, WANT_V ? 3 : 1) void classify_dense_kernel(BlockSpace sp, DeviceTables tb,
                                                              uint32_t *__restrict__ counts,
                                                              uint32_t *__restrict__ vcounts,
                                                              int nsegx, int n_bricks, int n_wgs, int ablate_arg,
                                                              unsigned long long *__restrict__ scan_ctrl, int n_scan_ctrl, SignVolume sg, int lane_is_z)
{
    __shared__ unsigned char s_trinum[256];
    const int ablate = VTMC_ABLATE(ablate_arg);
    const int lane = threadIdx.x & 63, wave = FROM_BITS ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : threadIdx.x >> 6;
#ifdef VTMC_TIMELINE
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
    // the fused scan that follows on the stream finds its ticket counter and tile status words zeroed
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_scan_ctrl; i += gridDim.x * 256) scan_ctrl[i] = 0ull;
    {
        // z-fastest grids (the C# float[,,] order): the lane walks z and the plane loop x, i.e. the brick code sees the volume with x and
        // z exchanged and assembles every case with corner bits 1 <-> 4 and 2 <-> 7 exchanged (corner order CollectTriNum.compute:27-37):
        // the count table is read through that permutation, everything else it derives is symmetric
        unsigned cs = threadIdx.x;
        if (lane_is_z) cs = (cs & 0x69u) | ((cs & 0x02u) << 3) | ((cs & 0x10u) >> 3) | ((cs & 0x04u) << 5) | ((cs & 0x80u) >> 5);
        s_trinum[threadIdx.x] = tb.tri_num[cs];
    }
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

    // brick -> (segment along the lane axis, block row y, block index along the plane-loop axis, volume)
    int segx = brick % nsegx;
    int t = brick / nsegx;
    int by = t % sp.nby;
    t /= sp.nby;
    const int n_loop = lane_is_z ? sp.nbx : sp.nbz;
    int bz = t % n_loop;      // z block (x fastest) or x block (z fastest)
    int v = t / n_loop;

    BlockSpace spv = sp;      // the volume as the brick code addresses it: lane axis, y, plane-loop axis
    if (lane_is_z) {
        spv.sz = sp.sx;
        spv.nx = sp.nbz * 8;
    }
    const int gx = segx * 64 + lane;                     // cell / sample index of this lane along the lane axis
    const int gxc = gx < spv.nx + 1 ? gx : spv.nx + 1;   // clamp loads inside the volume
    int xe = segx * 64 + 64;                             // the 65th column
    xe = xe < spv.nx + 1 ? xe : spv.nx + 1;
    const float *brick_base = sp.base + v * sp.sv + (8ll * by) * sp.sy + (8ll * bz) * spv.sz;

    unsigned vc = 0, rows = 0;
    unsigned total;
    if constexpr (FROM_BITS)
        total = classify_brick_column<WANT_V, true>(sp, s_trinum, brick_base, gx, gxc, xe, lane, ablate, &vc, &rows,
                                                    sg.words + ((long long)v * sg.dz + 8 * bz) * sg.plane_words, sg.plane_words,
                                                    segx * 64 + sg.dx * (8 * by), sg.dx);
    else
        total = classify_brick_column<WANT_V, false>(spv, s_trinum, brick_base, gx, gxc, xe, lane, ablate, &vc, &rows);
    if (lane_is_z)   // the row mask's upper byte is "z layers with triangles": here the lane's own layer, not the plane loop's
        rows = (rows & 0xFFu) | ((rows & 0xFFu) ? 0x100u << (lane & 7) : 0u);
    // 8-lane group sums = per-block counts
    total += __shfl_xor(total, 1);
    total += __shfl_xor(total, 2);
    total += __shfl_xor(total, 4);
    const int bl = segx * 8 + (lane >> 3);               // block index along the lane axis
    const int bx = lane_is_z ? bz : bl, bzz = lane_is_z ? bl : bz;
    const int bid = v * sp.bpv + bx + sp.nbx * (by + sp.nby * bzz);
    const bool in_volume = bl < (lane_is_z ? sp.nbz : sp.nbx);
    rows |= (unsigned)__shfl_xor((int)rows, 1);
    rows |= (unsigned)__shfl_xor((int)rows, 2);
    rows |= (unsigned)__shfl_xor((int)rows, 4);
    // a block holds at most 2560 triangles: the count shares its word with the block's row mask
    if ((lane & 7) == 0 && in_volume) counts[bid] = total | (rows << 16);
    if (WANT_V) {
        vc += __shfl_xor(vc, 1);
        vc += __shfl_xor(vc, 2);
        vc += __shfl_xor(vc, 4);
        if ((lane & 7) == 0 && in_volume) vcounts[bid] = vc;
    }
#ifdef VTMC_TIMELINE
    if (g_vtmc_timeline && lane == 0) {
        unsigned long long *t = g_vtmc_timeline + 4ll * brick;
        t[0] = tl_start;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        t[3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// scan: exclusive prefix sum of per-block triangle counts + compaction of the non-empty blocks.
// A count word holds the triangle count in its low 16 bits (kCountMask) and the block's row mask above.
// Replaces the single-address InterlockedAdd (CollectTriNum.compute:54), the 4-byte read-back
// (VoxelTerrain.cs:394-395) and the append cursor (MarchingCube.compute:160-162).
//   one launch (scan_fused_kernel): per-block exclusive offsets, active list, totals = {T, nActive}
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

// ----------------------------------------------------------------------------------------------
// scan_fused_kernel: exclusive scan + compaction in ONE launch (chained scan with decoupled look-back).  A step of the path is then three dispatches -- classify, scan, emit --
// instead of seven plus two copies; what that buys is fixed cost (~35 us per step), which is what
// strong scaling over 8 GPUs is short of (a rank's kernels take 0.3 ms there).
//   * tiles are handed out by a ticket taken when a workgroup STARTS, so every predecessor of a tile
//     is already running and publishes its aggregate before it waits for anything: no deadlock,
//     whatever the residency; every spin is bounded all the same (error word -> VTMC_ERR_DEVICE);
//   * one self-describing 64-bit status word per tile -- state (2 bits: 1 aggregate, 2 inclusive
//     prefix) | non-empty blocks (30) | triangles (32, saturating) -- written and polled with relaxed
//     agent-scope atomics: a granule carries its own flag, nothing else needs ordering;
//   * the words and the ticket counter are zeroed by the classify kernel that precedes the scan on
//     the stream; this kernel in turn zeroes the emit kernel's ticket queue and writes the totals
//     straight into the host's pinned words (no copy node).
// ----------------------------------------------------------------------------------------------
constexpr unsigned long long kScanAggregate = 1ull << 62, kScanInclusive = 2ull << 62;
constexpr int kScanSpinLimit = 1 << 22;

__device__ __forceinline__ unsigned long long scan_pack(unsigned long long state, unsigned long long tri, uint32_t act)
{
    return state | ((unsigned long long)(act & 0x3FFFFFFFu) << 32) | (tri > 0xFFFFFFFFull ? 0xFFFFFFFFull : tri);
}

// DUAL: the welded-vertex counts of the indexed output ride along (second count array, second status word
// per tile, second offsets array): one launch for both scans.
template <bool DUAL>
__global__ __launch_bounds__(256) void scan_fused_kernel(BlockSpace sp, const uint32_t *__restrict__ counts, int n, uint32_t *__restrict__ offsets,
                                                          BlockDesc *__restrict__ active, unsigned long long *__restrict__ ctrl,
                                                          uint32_t *__restrict__ totals, uint32_t *__restrict__ host_totals,
                                                          uint32_t *__restrict__ zero_words, int n_zero,
                                                          const uint32_t *__restrict__ vcounts, uint32_t *__restrict__ voffsets,
                                                          unsigned long long *__restrict__ vstatus, uint32_t *__restrict__ vtotals,
                                                          uint32_t *__restrict__ volume_counts, int tiles_per_volume)
{
    __shared__ uint32_t s_w[2][4];
    __shared__ uint32_t s_v[4];
    __shared__ unsigned s_tile;
    __shared__ unsigned long long s_ex_tri, s_ex_vert;
    __shared__ uint32_t s_ex_act;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *ticket = reinterpret_cast<unsigned *>(ctrl);   // ctrl[0]: ticket counter, ctrl[1]: error word, ctrl[2 + t]: status of tile t
    unsigned long long *status = ctrl + 2;
    if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = (int)s_tile, n_tiles = (int)gridDim.x;
    for (int i = t * 256 + threadIdx.x; i < n_zero; i += n_tiles * 256) zero_words[i] = 0u;   // the emit kernel's ticket queue

    const int base = t * kScanTile + threadIdx.x * 8;
    uint32_t c[8], vc[8];
    uint32_t sum = 0, act = 0, vsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = base + k;
        c[k] = i < n ? counts[i] : 0u;   // triangles | row mask << 16
        sum += c[k] & kCountMask;
        act += (c[k] & kCountMask) != 0u;
        if (DUAL) {
            vc[k] = i < n ? vcounts[i] : 0u;
            vsum += vc[k];
        }
    }
    uint32_t is = sum, ia = act, ts, ta;
    wg_incl_scan2(is, ia, ts, ta, &s_w);   // ts <= 2048 * 2560
    uint32_t iv = vsum, tv = 0;
    if (DUAL) {   // a third inclusive scan over the workgroup (<= 2048 * 1944)
        iv = wave_incl_scan(vsum, lane);
        if (lane == 63) s_v[wave] = iv;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) carry += s_v[w];
            tv += s_v[w];
        }
        iv += carry;
    }

    if (wave == 0) {
        if (lane == 0) {
            __hip_atomic_store(&status[t], scan_pack(t == 0 ? kScanInclusive : kScanAggregate, ts, ta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (DUAL)
                __hip_atomic_store(&vstatus[t], (t == 0 ? kScanInclusive : kScanAggregate) | (unsigned long long)tv, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long ex_tri = 0, ex_vert = 0;
        uint32_t ex_act = 0;
        bool failed = false, done_a = false, done_v = !DUAL;
        for (int j = t - 1; j >= 0 && !(done_a && done_v); j -= 64) {   // windows of 64 predecessors, nearest first (lane 0 = tile j)
            const int idx = j - lane;
            const bool valid = idx >= 0;
            unsigned long long w = 0, wv = kScanInclusive;
            int spins = 0;
            for (;;) {
                if (valid) {
                    w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (DUAL) wv = __hip_atomic_load(&vstatus[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!__builtin_amdgcn_ballot_w64(valid && ((w >> 62) == 0ull || (wv >> 62) == 0ull))) break;
                if (++spins > kScanSpinLimit) {
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!done_a) {
                const u64 incl = __builtin_amdgcn_ballot_w64(valid && (w >> 62) == 2ull);
                const int first = incl ? __builtin_ctzll(incl) : 64;   // nearest tile that already knows its inclusive prefix
                const bool take = valid && lane <= first;
                unsigned long long tri = take ? (w & 0xFFFFFFFFull) : 0ull;
                uint32_t ac = take ? (uint32_t)(w >> 32) & 0x3FFFFFFFu : 0u;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    tri += __shfl_xor(tri, off);
                    ac += __shfl_xor(ac, off);
                }
                ex_tri += tri;   // saturating words sum to >= 2^32 - 1 whenever the true sum does
                ex_act += ac;
                done_a = incl != 0;
            }
            if (DUAL && !done_v) {
                const u64 incl = __builtin_amdgcn_ballot_w64(valid && (wv >> 62) == 2ull);
                const int first = incl ? __builtin_ctzll(incl) : 64;
                unsigned long long vv = (valid && lane <= first) ? (wv & 0x3FFFFFFFFFFFFFFFull) : 0ull;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) vv += __shfl_xor(vv, off);
                ex_vert += vv;
                done_v = incl != 0;
            }
            if (failed) break;
        }
        if (lane == 0) {
            if (t > 0) {
                __hip_atomic_store(&status[t], scan_pack(kScanInclusive, ex_tri + ts, ex_act + ta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (DUAL) __hip_atomic_store(&vstatus[t], kScanInclusive | (ex_vert + tv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (failed) {
                reinterpret_cast<unsigned *>(ctrl + 1)[0] = 1u;
                totals[8] = 1u;
                if (host_totals) host_totals[8] = 1u;
            }
            s_ex_tri = ex_tri;
            s_ex_act = ex_act;
            s_ex_vert = ex_vert;
            // Per-volume {vertices, triangles} as soon as the scan knows them -- a multi-GPU caller's
            // all-gather then runs beside the emit kernel instead of behind it.  Possible when a volume is a whole
            // number of tiles: the tile that ends volume v subtracts the inclusive prefix of the tile that ends v - 1,
            // read from its (self-describing) status word once that has turned inclusive.
            if (volume_counts && tiles_per_volume > 0 && (t + 1) % tiles_per_volume == 0 && !failed) {
                const int v = (t + 1) / tiles_per_volume - 1, tp = t - tiles_per_volume;
                unsigned long long ptri = 0, pvert = 0;
                bool ok = true;
                if (tp >= 0) {
                    unsigned long long w = 0, wv = kScanInclusive;
                    int spins = 0;
                    for (;;) {
                        w = __hip_atomic_load(&status[tp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (DUAL) wv = __hip_atomic_load(&vstatus[tp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((w >> 62) == 2ull && (wv >> 62) == 2ull) break;
                        if (++spins > kScanSpinLimit) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    ptri = w & 0xFFFFFFFFull;
                    pvert = wv & 0x3FFFFFFFFFFFFFFFull;
                }
                if (ok) {
                    const uint32_t tv_ = (uint32_t)(ex_tri + ts - ptri);
                    volume_counts[2 * v + 1] = tv_;
                    volume_counts[2 * v] = DUAL ? (uint32_t)(ex_vert + tv - pvert) : 3u * tv_;   // soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
                } else {
                    reinterpret_cast<unsigned *>(ctrl + 1)[0] = 1u;
                    totals[8] = 1u;
                    if (host_totals) host_totals[8] = 1u;
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long ex_tri = s_ex_tri;
    uint32_t off = (uint32_t)ex_tri + is - sum;   // 32-bit offsets: meaningless once T passes 2^32, nothing is emitted then
    uint32_t aoff = s_ex_act + ia - act;
    uint32_t voff = DUAL ? (uint32_t)s_ex_vert + iv - vsum : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = base + k;
        if (i < n) {
            offsets[i] = off;
            if (active && (c[k] & kCountMask) != 0u) {   // the emit kernel's work item: everything it needs to know about the block in one 32-byte record
                BlockDesc d;
                d.b = (uint32_t)i;
                d.tri_base = off;
                d.cnt_mask = c[k];
                d.vert_base = DUAL ? voff : 0u;
                d.origin = block_origin(sp, i);
                d.vert_cnt = DUAL ? vc[k] : 0u;
                d.pad = 0u;
                active[aoff++] = d;
            }
            off += c[k] & kCountMask;
            if (i == n - 1) offsets[n] = off;
            if (DUAL) {
                voffsets[i] = voff;
                voff += vc[k];
                if (i == n - 1) voffsets[n] = voff;
            }
        }
    }
    if (t == n_tiles - 1 && threadIdx.x == 0) {
        unsigned long long T = ex_tri + ts;
        if (T > 0xFFFFFFFFull) T = 0xFFFFFFFFull;   // saturated somewhere: at least 2^32 - 1
        const unsigned long long V = DUAL ? s_ex_vert + tv : 0ull;
        const uint32_t tot[8] = {(uint32_t)T, s_ex_act + ta, (uint32_t)T, 0u, V > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)V, 0u, (uint32_t)V, (uint32_t)(V >> 32)};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < 4) totals[k] = tot[k];
            else if (DUAL) vtotals[k - 4] = tot[k];
            if (host_totals && (k < 4 || DUAL)) host_totals[k] = tot[k];
        }
    }
}

// ----------------------------------------------------------------------------------------------
// launch wrappers
// ----------------------------------------------------------------------------------------------
hipError_t launch_classify_blocks(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                  uint8_t *cases_or_null, uint32_t *vcounts_or_null, int n_cus, unsigned long long *scan_ctrl,
                                  int n_scan_ctrl, hipStream_t stream)
{
    int wgs = (sp.n_blocks + kWavesPerWg - 1) / kWavesPerWg;
    int cap = n_cus * 8;
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    launch_begin();
    hipLaunchKernelGGL(classify_blocks_kernel, dim3(wgs), dim3(256), 0, stream, sp, tb, counts, cases_or_null, vcounts_or_null, scan_ctrl,
                       n_scan_ctrl);
    return launch_end();
}

hipError_t launch_classify_dense(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                 uint32_t *vcounts_or_null, int ablate, int wgs_per_cu, unsigned long long *scan_ctrl, int n_scan_ctrl,
                                 const SignVolume &signs, hipStream_t stream)
{
    // the lane axis is the one with stride 1: x, or z for the C# float[,,] order
    const int lane_is_z = (sp.sx != 1 && sp.sz == 1) ? 1 : 0;
    const int n_lane_cells = lane_is_z ? sp.nbz * 8 : sp.nx;
    const int nsegx = (n_lane_cells + 63) / 64;
    const long long n_vol = sp.n_blocks / sp.bpv;
    long long n_bricks = n_vol * (lane_is_z ? sp.nbx : sp.nbz) * sp.nby * nsegx;
    long long n_wgs = (n_bricks + kWavesPerWg - 1) / kWavesPerWg;
    if (n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
    // Residency cap: the kernel needs 65 VGPRs and 256 bytes of LDS, so seven workgroups fit a CU -- and the stream runs
    // faster with fewer (each wave already keeps 81 row loads in flight).  Unused dynamic LDS is what caps it.
    const size_t dyn = wgs_per_cu > 0 && wgs_per_cu < 8 ? (size_t)(160 * 1024 / wgs_per_cu - 1024) & ~(size_t)255 : 0;
    const dim3 g((unsigned)n_wgs), b(256);
    launch_begin();
    if (signs.words && !lane_is_z) {   // instruction-bound variant: no residency cap
        if (vcounts_or_null)
            hipLaunchKernelGGL((classify_dense_kernel<true, true>), g, b, 0, stream, sp, tb, counts, vcounts_or_null, nsegx, (int)n_bricks, (int)n_wgs,
                               ablate, scan_ctrl, n_scan_ctrl, signs, 0);
        else
            hipLaunchKernelGGL((classify_dense_kernel<false, true>), g, b, 0, stream, sp, tb, counts, vcounts_or_null, nsegx, (int)n_bricks, (int)n_wgs,
                               ablate, scan_ctrl, n_scan_ctrl, signs, 0);
    } else if (vcounts_or_null)
        hipLaunchKernelGGL((classify_dense_kernel<true, false>), g, b, dyn, stream, sp, tb, counts, vcounts_or_null, nsegx, (int)n_bricks, (int)n_wgs,
                           ablate, scan_ctrl, n_scan_ctrl, signs, lane_is_z);
    else
        hipLaunchKernelGGL((classify_dense_kernel<false, false>), g, b, dyn, stream, sp, tb, counts, vcounts_or_null, nsegx, (int)n_bricks, (int)n_wgs,
                           ablate, scan_ctrl, n_scan_ctrl, signs, lane_is_z);
    return launch_end();
}

hipError_t launch_scan_fused(const BlockSpace &sp, const uint32_t *counts, int n_blocks, uint32_t *offsets, BlockDesc *active, unsigned long long *ctrl,
                             uint32_t *totals, uint32_t *host_totals, uint32_t *zero_words, int n_zero, const uint32_t *vcounts_or_null,
                             uint32_t *voffsets, uint32_t *vtotals, uint32_t *volume_counts_or_null, int bpv, hipStream_t stream)
{
    const int n_tiles = (n_blocks + kScanTile - 1) / kScanTile;
    unsigned long long *vstatus = ctrl + scan_ctrl_words(n_blocks);   // the second half of the control words
    const int tpv = (volume_counts_or_null && bpv > 0 && bpv % kScanTile == 0) ? bpv / kScanTile : 0;
    launch_begin();
    if (vcounts_or_null)
        hipLaunchKernelGGL((scan_fused_kernel<true>), dim3(n_tiles), dim3(256), 0, stream, sp, counts, n_blocks, offsets, active, ctrl, totals,
                           host_totals, zero_words, n_zero, vcounts_or_null, voffsets, vstatus, vtotals, volume_counts_or_null, tpv);
    else
        hipLaunchKernelGGL((scan_fused_kernel<false>), dim3(n_tiles), dim3(256), 0, stream, sp, counts, n_blocks, offsets, active, ctrl, totals,
                           host_totals, zero_words, n_zero, nullptr, nullptr, nullptr, nullptr, volume_counts_or_null, tpv);
    return launch_end();
}

}  // namespace vtmc

#ifdef VTMC_TIMELINE
extern "C" int32_t vtmc_debug_timeline(unsigned long long *d_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(vtmc::g_vtmc_timeline), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -4;
}
#endif
