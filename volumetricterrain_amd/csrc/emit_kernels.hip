// emit_kernels.hip -- fused normals + triangle emit (hand-written gfx950 / CDNA4, wave64).
//
// Replaces Shaders/SampleNormal.compute:23-34 and Shaders/MarchingCube.compute:101-165 of the
// reference (/root/reference/Unity-Project/Assets/): no 18 GB normal lattice is materialised, and
// triangles land at offsets fixed by the scan (canonical order) instead of an atomic append.
#include "emit_device.h"

#include <type_traits>

namespace vtmc {

// ----------------------------------------------------------------------------------------------
// emit_kernel: SampleNormal + MarchingCube fused, one wave per non-empty block, persistent waves.
//   * each XCD (blockIdx % 8, round-robin dispatch -- a speed heuristic only) sweeps contiguous
//     parts of the active list through per-part ticket counters, so blocks that share halo rows /
//     128-byte lines meet in one L2;
//   * the next block's tile is prefetched into registers while the current one is processed
//     (20 loads of 5 rows x 10 samples, scalar slab base + two loop-invariant 32-bit lane offsets);
//   * per-block work: emit_block_from_tile (emit_device.h).
// ----------------------------------------------------------------------------------------------
//   INDEXED: welded vertices + block-local indices (emit_block_indexed) instead of 76-byte records;
//   `out` then is the vertex buffer, voffsets / vcapacity / out_indices its extra operands.
template <bool FAST, bool INDEXED>
__global__ __launch_bounds__(256, 4) void emit_kernel(BlockSpace sp, DeviceTables tb,
                                                    const uint32_t *__restrict__ offsets,
                                                    const int32_t *__restrict__ active_list,
                                                    const uint32_t *__restrict__ totals, uint32_t capacity,
                                                    float *__restrict__ out, int group_log2, int ablate, unsigned *__restrict__ queue, int sub_log2,
                                                    const uint32_t *__restrict__ voffsets, const uint32_t *__restrict__ vtotals,
                                                    uint32_t vcapacity, int *__restrict__ out_indices,
                                                    const uint32_t *__restrict__ rowmasks, uint32_t *__restrict__ volume_counts, int n_volumes)
{
    using Lds = typename std::conditional<INDEXED, EmitLdsIdx, EmitLds2>::type;
    __shared__ Lds s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ unsigned short s_own[INDEXED ? 96 : 1];   // (cube edge, which coordinates are 7) -> owner cell offset | owner-side edge id
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    if (INDEXED && threadIdx.x < 96) s_own[threadIdx.x] = owner_entry(threadIdx.x >> 3, threadIdx.x & 7u);
#ifdef VTMC_DEBUG_POISON_LDS  // diagnostic build: NaN-fill LDS so any read of a never-written word shows up in the output
    for (unsigned i = threadIdx.x; i < sizeof(s_lds) / 4; i += 256) reinterpret_cast<unsigned *>(s_lds)[i] = 0x7FC00000u;
#endif
    __syncthreads();

    // per-volume {vertices, triangles} (the array a multi-GPU caller all-gathers) from the scan's
    // offsets: a few lanes of the first workgroup instead of a dispatch of its own
    if (volume_counts && blockIdx.x == 0) {
        for (int v = threadIdx.x; v < n_volumes; v += 256) {
            const long long lo = (long long)v * sp.bpv, hi = lo + sp.bpv;
            const uint32_t t = offsets[hi] - offsets[lo];
            volume_counts[2 * v] = INDEXED ? voffsets[hi] - voffsets[lo] : 3u * t;   // soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
            volume_counts[2 * v + 1] = t;
        }
    }
    const uint32_t total_tris = totals[0];
    const int n_active = (int)totals[1];
    if (total_tris > capacity) return;  // host grows the buffer and re-launches (vtmc_api.hip)
    if (INDEXED && vtotals[0] > vcapacity) return;

    Lds *L = &s_lds[wave];

    // Tile fetch: an instruction covers 5 rows of 10 samples (lane = sample along the stride-1 axis +
    // 10 * row-in-group; lanes 50-63 idle), 20 instructions cover the 100 rows.  A lane's address is one
    // of two loop-invariant 32-bit offsets (rows 0-4 / 5-9 of a slab) plus a wave-uniform slab offset, its
    // LDS destination one index plus an immediate: 3 address registers instead of the 48 a flat
    // (16 x 64 lanes) enumeration of the 1000 samples needs -- what keeps the kernel at 128 VGPRs.
    const int lq = lane % 10, rq = lane / 10;          // sample along the fast axis, row within the group
    const bool lane_ok = rq < 5;
    const int rqc = lane_ok ? rq : 4;
    const long long s_fast = sp.zfast ? sp.sz : sp.sx, s_slab = sp.zfast ? sp.sx : sp.sz;
    const unsigned off0 = (unsigned)(lq * s_fast + rqc * sp.sy) * 4u, off1 = off0 + (unsigned)(5 * sp.sy) * 4u;
    const unsigned slab_bytes = (unsigned)s_slab * 4u;
    const int lds0 = sp.zfast ? 100 * lq + 10 * rqc : lq + 10 * rqc;   // tile index of (fast = lq, y = rq, slab 0)
    const int lds_slab = sp.zfast ? 1 : 100;
    // Row masks from the classify pass (upper half of the block's count word): a tile row (y, z) is only fetched when a cell with triangles
    // can touch it (its layer or one of the two below, on both axes) -- about 64 % of the rows on the
    // benchmark field.  Rows not fetched keep stale values; pass 1 skips their cells.
    auto load_rows = [&](const char *src, unsigned mask, float (&dst)[20]) {
        const unsigned ym = mask & 0xFFu, zm = mask >> 8;
        const unsigned ny = ym | (ym << 1) | (ym << 2), nz = zm | (zm << 1) | (zm << 2);
        const bool zl = sp.zfast ? ((nz >> lq) & 1u) != 0u : true;   // z-fastest: the lane's own z row
        const bool need0 = lane_ok && zl && ((ny >> rqc) & 1u), need1 = lane_ok && zl && ((ny >> (5 + rqc)) & 1u);
        // the two row groups under ONE exec mask each (a lane's need is the same for all ten slabs), slab pointer
        // advanced by addition: the scalar unit sees two mask set-ups and ten adds per tile, not twenty of each
        if (need0) {
            const char *p = src + off0;
#pragma unroll
            for (int c = 0; c < 10; ++c, p += slab_bytes)
                if ((sp.zfast || ((nz >> c) & 1u)) && (!(ablate & 8) || c < 6)) dst[2 * c] = *reinterpret_cast<const float *>(p);   // x-fastest: a z slab nobody needs is skipped (wave-uniform); ablate 8: diagnostics
        }
        if (need1) {
            const char *p = src + off1;
#pragma unroll
            for (int c = 0; c < 10; ++c, p += slab_bytes)
                if ((sp.zfast || ((nz >> c) & 1u)) && (!(ablate & 8) || c < 6)) dst[2 * c + 1] = *reinterpret_cast<const float *>(p);
        }
    };
    auto store_tile = [&](float *tile, const float (&v)[20]) {
        if (lane_ok) {
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                tile[lds0 + lds_slab * c] = v[2 * c];
                tile[lds0 + lds_slab * c + 50] = v[2 * c + 1];
            }
        }
    };

    // each XCD (blockIdx % 8 under round-robin dispatch; a speed heuristic only) sweeps one
    // contiguous eighth of the active list
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;  // gridDim % 8 == 0
    // dynamic mode splits every XCD's eighth into n_sub sub-ranges, one ticket counter each (on its
    // own 256-byte line: counters sharing a line serialise at ~88 atomics/us chip-wide)
    const int n_sub = queue ? (1 << sub_log2) : 1;
    const int part = xcd * n_sub + (queue ? (j & (n_sub - 1)) : 0), n_part = 8 * n_sub;
    const int ai_begin = (int)((long long)n_active * part / n_part);
    const int ai_end = (int)((long long)n_active * (part + 1) / n_part);
    // the k-th block of this wave: rounds of (waves per XCD) groups, each wave takes 2^group_log2
    // consecutive list entries per round (x-adjacent blocks share 128-byte lines)
    const int u = j * kWavesPerWg + wave, n_u = per_xcd * kWavesPerWg;
    auto entry = [&](int k) {
        const int r = k >> group_log2, g = k & ((1 << group_log2) - 1);
        return ai_begin + (((r * n_u + u) << group_log2) | g);
    };

    // Work distribution.  Static: entry(k).  Dynamic (queue != nullptr): one ticket counter per XCD,
    // so the blocks in flight on an XCD are always the next ones in list order -- spatial neighbours
    // (shared halo rows / 128-byte lines) stay within the few microseconds a line survives in L2.
    // The ticket for block k+2 is requested while block k is processed, its tile one block ahead.
    int k_static = 0;
    unsigned tick_raw = 0;  // lane 0 holds the ticket the last request returned
    auto request = [&]() {
        if (queue) {
            if (lane == 0) tick_raw = __hip_atomic_fetch_add(queue + part * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            tick_raw = (unsigned)(entry(k_static++) - ai_begin);
        }
    };
    auto collect = [&]() { return ai_begin + (int)__builtin_amdgcn_readfirstlane(tick_raw); };

    float pre[20] = {};  // rows a block does not need keep whatever an earlier block left: never used
    int b_next = 0;
    unsigned mask_next = 0xFFFFu;
    request();
    int ai = collect();
    request();
    int ai_next = collect();
    if (ai < ai_end) {
        b_next = active_list[ai];
        if (rowmasks) mask_next = rowmasks[b_next] >> 16;
        load_rows(reinterpret_cast<const char *>(sp.base + block_origin(sp, b_next)), mask_next, pre);
    }
    for (int k = 0; ai < ai_end; ++k) {
        const int b = b_next;
        const unsigned mask = mask_next;
        const size_t tri_base = offsets[b];
        const int budget = (int)(offsets[b + 1] - offsets[b]);  // the scan's count for this block
        VTMC_WAVE_SYNC();
        store_tile(L->tile, pre);
        if (ai_next < ai_end) {  // prefetch the next block's tile; it lands while this one is processed
            b_next = active_list[(ablate & 2) ? ai_begin + (k & 3) : ai_next];
            if (rowmasks) mask_next = rowmasks[b_next] >> 16;
            load_rows(reinterpret_cast<const char *>(sp.base + block_origin(sp, b_next)), mask_next, pre);
        }
        request();  // ticket for the block after next; collected at the bottom of this iteration
        VTMC_WAVE_SYNC();

        if constexpr (INDEXED)
            emit_block_indexed<FAST>(L, s_vert, s_own, tri_base, budget, (size_t)voffsets[b], (int)(voffsets[b + 1] - voffsets[b]), out,
                                     out_indices, lane, ablate, mask);
        else
            emit_block_from_tile<FAST>(L, s_vert, tri_base, budget, b, out, lane, ablate, mask);
        ai = ai_next;
        ai_next = collect();
    }
}

hipError_t launch_emit(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets,
                       const int32_t *active_list, const uint32_t *totals, const uint32_t *counts_or_null, uint32_t capacity,
                       void *triangles, int n_cus, const Tuning &tune, unsigned *queue, uint32_t *volume_counts, int n_volumes,
                       hipStream_t stream)
{
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : 4;  // 4 x 40 KB of LDS, 128 VGPRs
    int wgs = n_cus * per_cu;
    wgs = (wgs + 7) & ~7;  // the XCD sweep needs a multiple of 8
    if (wgs > 8 + tune.emit_spare_wgs) wgs -= tune.emit_spare_wgs & ~7;
    dim3 g(wgs), blk(256);
    float *o = (float *)triangles;
    unsigned *q = tune.emit_dynamic ? queue : nullptr;
    if (tune.emit_fast_math)
        hipLaunchKernelGGL((emit_kernel<true, false>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o, tune.emit_group_log2, tune.emit_ablate, q, tune.emit_sub_log2, nullptr, nullptr, 0u, nullptr, tune.emit_row_masks ? counts_or_null : nullptr, volume_counts, n_volumes);
    else
        hipLaunchKernelGGL((emit_kernel<false, false>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o, tune.emit_group_log2, tune.emit_ablate, q, tune.emit_sub_log2, nullptr, nullptr, 0u, nullptr, tune.emit_row_masks ? counts_or_null : nullptr, volume_counts, n_volumes);
    return hipGetLastError();
}

hipError_t launch_emit_indexed(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets, const uint32_t *voffsets,
                               const int32_t *active_list, const uint32_t *totals, const uint32_t *vtotals, const uint32_t *counts_or_null,
                               uint32_t tri_capacity,
                               uint32_t vert_capacity, void *vertices, void *indices, int n_cus, const Tuning &tune, unsigned *queue,
                               uint32_t *volume_counts, int n_volumes, hipStream_t stream)
{
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : 4;   // 39.9 KB of LDS, <= 128 VGPRs
    int wgs = n_cus * per_cu;
    wgs = (wgs + 7) & ~7;
    if (wgs > 8 + tune.emit_spare_wgs) wgs -= tune.emit_spare_wgs & ~7;
    dim3 g(wgs), blk(256);
    unsigned *q = tune.emit_dynamic ? queue : nullptr;
    if (tune.emit_fast_math)
        hipLaunchKernelGGL((emit_kernel<true, true>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, tri_capacity, (float *)vertices, tune.emit_group_log2, tune.emit_ablate, q, tune.emit_sub_log2, voffsets, vtotals, vert_capacity, (int *)indices, tune.emit_row_masks ? counts_or_null : nullptr, volume_counts, n_volumes);
    else
        hipLaunchKernelGGL((emit_kernel<false, true>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, tri_capacity, (float *)vertices, tune.emit_group_log2, tune.emit_ablate, q, tune.emit_sub_log2, voffsets, vtotals, vert_capacity, (int *)indices, tune.emit_row_masks ? counts_or_null : nullptr, volume_counts, n_volumes);
    return hipGetLastError();
}

}  // namespace vtmc
