// emit_kernels.hip -- fused normals + triangle emit (hand-written gfx950 / CDNA4, wave64).
//
// Replaces Shaders/SampleNormal.compute:23-34 and Shaders/MarchingCube.compute:101-165 of the
// reference (/root/reference/Unity-Project/Assets/): no 18 GB normal lattice is materialised, and
// triangles land at offsets fixed by the scan (canonical order) instead of an atomic append.
#include "emit_device.h"

#include <type_traits>

namespace vtmc {

#ifdef VTMC_EMIT_TIMING
__device__ unsigned long long *g_vtmc_emit_phases = nullptr;   // [8] shader cycles per phase + [8] = blocks, summed over all waves
#endif

// ----------------------------------------------------------------------------------------------
// Loads hipcc does not count.  gfx950 keeps ONE counter (vmcnt) for vector loads, stores and returning
// atomics, retired in issue order.  A wave that prefetches its next tile, emits a data-dependent number
// of stores and then needs the tile gets `s_waitcnt vmcnt(0)` from the compiler -- it waits for every
// store of the block it has just written (round 2: three such waits per block, 43 % of the wave cycles
// parked).  The tile loads and the ticket atomic are therefore issued from inline asm, invisible to the
// compiler's bookkeeping, and retired by wait_vm_at_most(n): n = a LOWER bound of the vector-memory
// instructions issued after them (the body stores of the flush, counted in a scalar register), rounded
// down to a step of the ladder below.  The destination registers travel through the statements as "+v"
// operands, so no consumer is scheduled above the wait; tests/test_isa_audit.py checks in the compiled
// ISA that nothing else touches them between issue and wait.
// ----------------------------------------------------------------------------------------------
// Every statement is UNCONDITIONAL for the compiler -- a wave-uniform skip is a branch inside the string, a lane that needs
// nothing loads a harmless address -- so a destination register has one definition chain and no merge (phi) a copy could be
// inserted for while the data is still in flight.
// the two row-group loads of one tile slab (rows 0-4 / 5-9): skipped as a whole when bit `BIT` of the wave-uniform `live` is clear -- ONE
// test and ONE branch per slab (the kernel is bound by issued instructions: round 4 tested every load on its own)
template <int BIT>
__device__ __forceinline__ void gload_slab_async(float &dst0, float &dst1, unsigned voff0, unsigned voff1, const char *sbase, unsigned live)
{
    // the base may come straight out of a v_readlane (an SGPR reloaded from its spill lane): a VALU-written SGPR needs five wait states
    // before a VMEM instruction reads it, and hipcc pads nothing inside the string (tools/isa_audit.py checks the distance)
    asm volatile("s_bitcmp0_b32 %5, %6\n\ts_cbranch_scc1 .Lskip_%=\n\ts_nop 4\n\tglobal_load_dword %0, %2, %4\n\tglobal_load_dword %1, %3, %4\n.Lskip_%=:"
                 : "+v"(dst0), "+v"(dst1)
                 : "v"(voff0), "v"(voff1), "s"(sbase), "s"(live), "n"(BIT)
                 : "scc", "memory");
}
// lane 0 takes the next ticket of `counter` (nothing happens when counter is null: static distribution)
__device__ __forceinline__ void ticket_async(unsigned &dst, unsigned *counter)
{
    unsigned long long keep;
    asm volatile("s_cmp_eq_u64 %4, 0\n\ts_cbranch_scc1 .Lnone_%=\n\t"
                 "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\ts_nop 4\n\t"   // the counter's address may be fresh from a v_readlane
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_mov_b64 exec, %1\n.Lnone_%=:"
                 : "+v"(dst), "=&s"(keep)
                 : "v"(0u), "v"(1u), "s"(counter)
                 : "scc", "memory");
}
// waits until at most min(n, 48) of the wave's vector-memory instructions are outstanding (n wave-uniform)
#define VTMC_WAIT_LADDER                                                                                   \
    "s_cmp_lt_u32 %[n], 8\n\ts_cbranch_scc1 .Llt8_%=\n\t"                                                  \
    "s_cmp_lt_u32 %[n], 20\n\ts_cbranch_scc1 .Llt20_%=\n\t"                                                \
    "s_cmp_lt_u32 %[n], 32\n\ts_cbranch_scc1 .Llt32_%=\n\t"                                                \
    "s_cmp_lt_u32 %[n], 48\n\ts_cbranch_scc1 .Lw32_%=\n\t"                                                 \
    "s_waitcnt vmcnt(48)\n\ts_branch .Ldone_%=\n"                                                          \
    ".Lw32_%=:\n\ts_waitcnt vmcnt(32)\n\ts_branch .Ldone_%=\n"                                             \
    ".Llt32_%=:\n\ts_cmp_lt_u32 %[n], 24\n\ts_cbranch_scc1 .Lw20_%=\n\t"                                   \
    "s_waitcnt vmcnt(24)\n\ts_branch .Ldone_%=\n"                                                          \
    ".Lw20_%=:\n\ts_waitcnt vmcnt(20)\n\ts_branch .Ldone_%=\n"                                             \
    ".Llt20_%=:\n\ts_cmp_lt_u32 %[n], 12\n\ts_cbranch_scc1 .Llt12_%=\n\t"                                  \
    "s_cmp_lt_u32 %[n], 16\n\ts_cbranch_scc1 .Lw12_%=\n\t"                                                 \
    "s_waitcnt vmcnt(16)\n\ts_branch .Ldone_%=\n"                                                          \
    ".Lw12_%=:\n\ts_waitcnt vmcnt(12)\n\ts_branch .Ldone_%=\n"                                             \
    ".Llt12_%=:\n\ts_cmp_lt_u32 %[n], 10\n\ts_cbranch_scc1 .Lw8_%=\n\t"                                    \
    "s_waitcnt vmcnt(10)\n\ts_branch .Ldone_%=\n"                                                          \
    ".Lw8_%=:\n\ts_waitcnt vmcnt(8)\n\ts_branch .Ldone_%=\n"                                               \
    ".Llt8_%=:\n\ts_cmp_lt_u32 %[n], 4\n\ts_cbranch_scc1 .Llt4_%=\n\t"                                     \
    "s_cmp_lt_u32 %[n], 6\n\ts_cbranch_scc1 .Lw4_%=\n\t"                                                   \
    "s_waitcnt vmcnt(6)\n\ts_branch .Ldone_%=\n"                                                           \
    ".Lw4_%=:\n\ts_waitcnt vmcnt(4)\n\ts_branch .Ldone_%=\n"                                               \
    ".Llt4_%=:\n\ts_cmp_lt_u32 %[n], 2\n\ts_cbranch_scc1 .Llt2_%=\n\t"                                     \
    "s_cmp_lt_u32 %[n], 3\n\ts_cbranch_scc1 .Lw2_%=\n\t"                                                   \
    "s_waitcnt vmcnt(3)\n\ts_branch .Ldone_%=\n"                                                           \
    ".Lw2_%=:\n\ts_waitcnt vmcnt(2)\n\ts_branch .Ldone_%=\n"                                               \
    ".Llt2_%=:\n\ts_cmp_lt_u32 %[n], 1\n\ts_cbranch_scc1 .Lw0_%=\n\t"                                      \
    "s_waitcnt vmcnt(1)\n\ts_branch .Ldone_%=\n"                                                           \
    ".Lw0_%=:\n\ts_waitcnt vmcnt(0)\n"                                                                     \
    ".Ldone_%=:"
// the tile registers and the ticket pass through the wait: whatever reads them comes after it
__device__ __forceinline__ void wait_vm_at_most(int n, float (&t)[20], unsigned &tick)
{
    asm volatile(VTMC_WAIT_LADDER
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]),
                   "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]), "+v"(t[16]), "+v"(t[17]), "+v"(t[18]),
                   "+v"(t[19]), "+v"(tick)
                 : [n] "s"(n)
                 : "scc", "memory");
}

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
//   The next tile and the next ticket travel outside the compiler's vmcnt bookkeeping (gload_async above) and a block's stores are never
//   waited for (round 2's synchronous loop: profiles/r05/experiments/onepass_and_sync_loop.patch).
//   ONCE (soup only): every welded vertex of a block is evaluated one time into LDS and the records are expanded from there
//   (emit_block_once, emit_device.h); 53 KB of LDS per workgroup: three workgroups per CU.
//   (Round 3's three-wave workgroups of the indexed output -- six per CU = 18 waves, measured no faster than 16 -- left the library in
//   round 5: profiles/r05/experiments/README.md.)  The indexed kernel by occupancy: 0.92 / 0.68 / 0.59 ms at
//   8 / 12 / 16 waves).  Workgroups of more than 256 threads get fewer slots than their LDS would allow (measured, tools/_ab/occ2.hip:
//   52 KB x 384 threads: two per CU where the occupancy query says three).
template <bool FAST, bool INDEXED, bool ONCE = false>
__global__ __launch_bounds__(64 * kWavesPerWg, ONCE ? 3 : 4) void emit_kernel(BlockSpace sp, DeviceTables tb,
                                                    const uint32_t *__restrict__ offsets,
                                                    const BlockDesc *__restrict__ active,
                                                    const uint32_t *__restrict__ totals, uint32_t capacity,
                                                    float *__restrict__ out, int ablate_arg, unsigned *__restrict__ queue, int sub_log2,
                                                    const uint32_t *__restrict__ voffsets, const uint32_t *__restrict__ vtotals,
                                                    uint32_t vcapacity, int *__restrict__ out_indices,
                                                    int use_row_masks, uint32_t *__restrict__ volume_counts, int n_volumes)
{
    static_assert(!(ONCE && INDEXED), "ONCE is a form of the soup");
    constexpr int WAVES = kWavesPerWg;
    const int ablate = VTMC_ABLATE(ablate_arg);   // product build: 0, every diagnostic branch below folds away
    using Lds = typename std::conditional<INDEXED, EmitLdsIdx, typename std::conditional<ONCE, EmitLdsOnce, EmitLds2>::type>::type;
    __shared__ Lds s_lds[WAVES];
    __shared__ u64 s_vert[256];
    struct NoTables { unsigned char unused; };
    __shared__ typename std::conditional<INDEXED, IdxTables, typename std::conditional<ONCE, OnceTables, NoTables>::type>::type s_once[1];
    // (cube edge, which coordinates are 7) -> owner cell offset | owner-side edge id: a table for four-wave workgroups; three-wave ones compute
    // it (blocks of more than 255 vertices only) -- their 26 832 bytes of LDS are 21 allocation granules of 1 280 bytes, six workgroups per CU
    __shared__ unsigned short s_own_tab[INDEXED ? 96 : 1];
    const unsigned short *s_own = INDEXED ? s_own_tab : nullptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 256; i += 64 * WAVES) s_vert[i] = tb.vert_packed[i];
    if (INDEXED && threadIdx.x < 96) s_own_tab[threadIdx.x] = owner_entry(threadIdx.x >> 3, threadIdx.x & 7u);
    if constexpr (INDEXED) idx_tables_init(&s_once[0], threadIdx.x);
    else if constexpr (ONCE) once_tables_init(&s_once[0], threadIdx.x);
#ifdef VTMC_DEBUG_POISON_LDS  // diagnostic build: NaN-fill LDS so any read of a never-written word shows up in the output
    for (unsigned i = threadIdx.x; i < sizeof(s_lds) / 4; i += 64 * WAVES) reinterpret_cast<unsigned *>(s_lds)[i] = 0x7FC00000u;
#endif
    __syncthreads();

    // per-volume {vertices, triangles} (the array a multi-GPU caller all-gathers) from the scan's
    // offsets: a few lanes of the first workgroup instead of a dispatch of its own
    if (volume_counts && blockIdx.x == 0) {
        for (int v = threadIdx.x; v < n_volumes; v += 64 * WAVES) {
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
    auto tile_of = [](Lds *l) -> float * {
        if constexpr (INDEXED) return l->t.tile;
        else return l->tile;
    };

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
    // static distribution: the k-th block of this wave, round-robin over the waves of the XCD
    const int u = j * WAVES + wave, n_u = per_xcd * WAVES;
    auto entry = [&](int k) { return ai_begin + k * n_u + u; };

    // Work distribution.  Static: entry(k).  Dynamic (queue != nullptr): one ticket counter per XCD,
    // so the blocks in flight on an XCD are always the next ones in list order -- spatial neighbours
    // (shared halo rows / 128-byte lines) stay within the few microseconds a line survives in L2.
    // The ticket for block k+2 is requested while block k is processed, its tile one block ahead.
    int k_static = 0;
    float pre[20] = {};  // rows a block does not need keep whatever an earlier block left: never used
    {
        // the same two row groups, issued from inline asm: the compiler neither counts nor waits for them.  A lane that needs no
        // row reads the tile's first sample instead (one address for all of them: a broadcast hit), a slab nobody needs is skipped.
        auto load_rows_async = [&](const char *src, unsigned mask, float (&dst)[20]) {
            const unsigned ym = mask & 0xFFu, zm = mask >> 8;
            const unsigned ny = ym | (ym << 1) | (ym << 2), nz = zm | (zm << 1) | (zm << 2);
            const bool zl = sp.zfast ? ((nz >> lq) & 1u) != 0u : true;
            const bool need0 = lane_ok && zl && ((ny >> rqc) & 1u), need1 = lane_ok && zl && ((ny >> (5 + rqc)) & 1u);
            // a lane with nothing to fetch asks for a sample its block NEEDS anyway: the first one of the lowest needed row (ny x nz is an outer
            // product, so that row is needed in every live slab).  Round 3 sent those lanes to the tile's first sample: row (y = 0, z = slab),
            // not needed by four blocks in ten -- one extra 128-byte line per live slab of such a block (tools/_ab/line_fetch_exact.py).
            const unsigned idle = ((sp.zfast ? (unsigned)__builtin_ctz(nz | 0x200u) * (unsigned)s_fast : 0u) + (unsigned)__builtin_ctz(ny | 0x200u) * (unsigned)sp.sy) * 4u;
            const unsigned o0 = need0 ? off0 : idle, o1 = need1 ? off1 : idle;
            const unsigned live = sp.zfast ? 0x3FFu : nz;   // x-fastest: a z slab nobody needs is skipped (wave-uniform)
            const char *p = src;
#define VTMC_ROW(C)                                              \
    gload_slab_async<C>(dst[2 * C], dst[2 * C + 1], o0, o1, p, live);  \
    p += slab_bytes;
            VTMC_ROW(0) VTMC_ROW(1) VTMC_ROW(2) VTMC_ROW(3) VTMC_ROW(4) VTMC_ROW(5) VTMC_ROW(6) VTMC_ROW(7) VTMC_ROW(8) VTMC_ROW(9)
#undef VTMC_ROW
        };
        // Tickets run two blocks ahead of the tile loads and three ahead of the work: at the top of step k the wave knows blocks k and
        // k + 1 (id, row mask, offsets: scalar loads issued a step earlier), holds tile k in flight in `pre` and ticket k + 2 in `tick`.
        struct Blk {
            int b;
            unsigned mask;
            uint32_t tri_base, tri_cnt, vert_base, vert_cnt;
            long long origin;
        };
        auto describe = [&](int entry) {   // ONE 32-byte scalar load (lgkmcnt): the scan left the block's record in list order; nothing here touches vmcnt
            Blk d{-1, 0xFFFFu, 0u, 0u, 0u, 0u, 0ll};
            if (entry < ai_end) {
                typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
                typedef const __attribute__((address_space(4))) u32x8 *desc_ptr;   // written by the scan, never during this launch: a scalar load
                const u32x8 r = *((desc_ptr)active + entry);   // BlockDesc: b, tri_base, cnt_mask, vert_base, origin lo / hi, vert_cnt, pad
                d.b = (int)r[0];
                d.mask = use_row_masks ? r[2] >> 16 : 0xFFFFu;
                d.tri_base = r[1];
                d.tri_cnt = r[2] & kCountMask;
                d.origin = (long long)(((unsigned long long)r[5] << 32) | r[4]);
                if constexpr (INDEXED) {
                    d.vert_base = r[3];
                    d.vert_cnt = r[6];
                }
            }
            return d;
        };
        unsigned tick = 0;        // written by the asm atomic only (lane 0)
        int tick_static = 0;      // static distribution: the entry, kept in a scalar register
        unsigned *const counter = queue ? queue + part * 64 : nullptr;
        auto request_async = [&]() {
            ticket_async(tick, counter);
            if (!queue) tick_static = entry(k_static++);
        };
        auto collect_async = [&]() { return queue ? ai_begin + (int)__builtin_amdgcn_readfirstlane(tick) : tick_static; };
        int vm_issued = 0;
        PhaseClock pc;
        [[maybe_unused]] unsigned long long n_blocks_done = 0;
        request_async();
        wait_vm_at_most(0, pre, tick);
        Blk cur = describe(collect_async());
        request_async();
        wait_vm_at_most(0, pre, tick);
        Blk nxt = describe(collect_async());
        if (cur.b >= 0) load_rows_async(reinterpret_cast<const char *>(sp.base + cur.origin), cur.mask, pre);
        request_async();
        pc.start();
        while (cur.b >= 0) {
            wait_vm_at_most(vm_issued, pre, tick);   // tile `cur` and the ticket of the block after `nxt` have landed; younger stores stay in flight
            vm_issued = 0;
            pc.mark(0);
            ++n_blocks_done;
            VTMC_WAVE_SYNC();
            store_tile(tile_of(L), pre);
            if (ablate & 128) {   // diagnostic builds: the tile's 20 ds_write_b32 a second time (same values: results stay right) -- the difference is what
                VTMC_WAVE_SYNC();  // they cost, i.e. the most a fetch straight into LDS (LDS-DMA) could save
                store_tile(tile_of(L), pre);
            }
            const int far_entry = collect_async();
            if (nxt.b >= 0) load_rows_async(reinterpret_cast<const char *>(sp.base + nxt.origin), nxt.mask, pre);
            request_async();
            const Blk far = describe(far_entry);   // one scalar load, behind the tile loads: only the wave's own LDS work waits for it
            VTMC_WAVE_SYNC();
            pc.mark(1);
            const int budget = (int)cur.tri_cnt;
            if constexpr (INDEXED)
                emit_block_indexed<FAST>(L, s_vert, s_own, &s_once[0], (size_t)cur.tri_base, budget, (size_t)cur.vert_base, (int)cur.vert_cnt, out,
                                         out_indices, lane, ablate, cur.mask, vm_issued, pc);
            else if constexpr (ONCE)
                emit_block_once<FAST>(L, s_vert, &s_once[0], (size_t)cur.tri_base, budget, cur.b, out, lane, ablate, cur.mask, vm_issued, pc);
            else
                emit_block_from_tile<FAST>(corner_view(L), s_vert, (size_t)cur.tri_base, budget, cur.b, out, lane, ablate, cur.mask, vm_issued);
            cur = nxt;
            nxt = far;
            pc.mark(7);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last ticket (nobody reads it) has landed before the wave ends
#ifdef VTMC_EMIT_TIMING
        if (g_vtmc_emit_phases && lane == 0) {
            for (int i = 0; i < 8; ++i) atomicAdd(g_vtmc_emit_phases + i, pc.acc[i]);
            atomicAdd(g_vtmc_emit_phases + 8, n_blocks_done);
        }
#endif
    }
}

hipError_t launch_emit(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets,
                       const BlockDesc *active, const uint32_t *totals, uint32_t capacity,
                       void *triangles, int n_cus, const Tuning &tune, unsigned *queue, uint32_t *volume_counts, int n_volumes,
                       hipStream_t stream)
{
    const bool once = tune.emit_once && tune.emit_fast_math;   // the exact mode stays bit-compatible with the oracle: per-corner evaluation
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : (once ? 3 : 4);  // 4 x 40 KB of LDS, 128 VGPRs; vertex-once: 3 x 53 KB
    int wgs = n_cus * per_cu;
    wgs = (wgs + 7) & ~7;  // the XCD sweep needs a multiple of 8
    if (wgs > 8 + tune.emit_spare_wgs) wgs -= tune.emit_spare_wgs & ~7;
    dim3 g(wgs), blk(256);
    float *o = (float *)triangles;
    unsigned *q = tune.emit_dynamic ? queue : nullptr;
    launch_begin();
#define VTMC_LAUNCH_SOUP(F, O) hipLaunchKernelGGL((emit_kernel<F, false, O>), g, blk, 0, stream, sp, tb, offsets, active, totals, capacity, o, tune.emit_ablate, q, tune.emit_sub_log2, nullptr, nullptr, 0u, nullptr, tune.emit_row_masks, volume_counts, n_volumes)
    if (once) VTMC_LAUNCH_SOUP(true, true);
    else if (tune.emit_fast_math) VTMC_LAUNCH_SOUP(true, false);
    else VTMC_LAUNCH_SOUP(false, false);
#undef VTMC_LAUNCH_SOUP
    return launch_end();
}

hipError_t launch_emit_indexed(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets, const uint32_t *voffsets,
                               const BlockDesc *active, const uint32_t *totals, const uint32_t *vtotals,
                               uint32_t tri_capacity,
                               uint32_t vert_capacity, void *vertices, void *indices, int n_cus, const Tuning &tune, unsigned *queue,
                               uint32_t *volume_counts, int n_volumes, hipStream_t stream)
{
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : 4;   // 35.2 KB of LDS per workgroup, <= 128 VGPRs: four per CU = 16 waves
    int wgs = n_cus * per_cu;
    wgs = (wgs + 7) & ~7;
    if (wgs > 8 + tune.emit_spare_wgs) wgs -= tune.emit_spare_wgs & ~7;
    dim3 g(wgs), blk(256);
    unsigned *q = tune.emit_dynamic ? queue : nullptr;
    launch_begin();
#define VTMC_LAUNCH_IDX(F) hipLaunchKernelGGL((emit_kernel<F, true, false>), g, blk, 0, stream, sp, tb, offsets, active, totals, tri_capacity, (float *)vertices, tune.emit_ablate, q, tune.emit_sub_log2, voffsets, vtotals, vert_capacity, (int *)indices, tune.emit_row_masks, volume_counts, n_volumes)
    if (tune.emit_fast_math) VTMC_LAUNCH_IDX(true);
    else VTMC_LAUNCH_IDX(false);
#undef VTMC_LAUNCH_IDX
    return launch_end();
}

}  // namespace vtmc

#ifdef VTMC_EMIT_TIMING
extern "C" int32_t vtmc_debug_emit_phases(unsigned long long *d_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(vtmc::g_vtmc_emit_phases), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -4;
}
#endif
