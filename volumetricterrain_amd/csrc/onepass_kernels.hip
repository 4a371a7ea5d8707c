// onepass_kernels.hip -- classify + scan + emit of a dense batch in ONE launch (hand-written gfx950 / CDNA4, wave64).
//
// The three-launch path (classify_kernels.hip -> scan -> emit_kernels.hip) reads the samples of every block with triangles twice: the
// classify pass streams the whole volume, the emit pass comes back for the 10^3 tiles (1.8 GB of 128-byte lines on the 1024^3 field,
// long after the first pass has left the caches).  Here a wave classifies a brick (64 x 8 x 8 cells, classify_brick_column), learns where
// its triangles go from a chained scan over the bricks (decoupled look-back, two levels), and emits the brick's blocks right away: the
// counts never travel through memory and a step is one launch.  (The idea that the tile rows would come out of the caches the wave has
// just filled did not survive the counters: see "Measured" below.)
//
//   * work distribution: persistent waves, one brick per ticket.  Eight ticket counters, one per XCD (blockIdx % 8: a speed heuristic
//     only); counter x hands out the bricks of the units u = x, x + 8, ... in ascending order (a unit = `unit_bricks` consecutive bricks:
//     ONE 64-brick group by default.  Larger units would keep the bricks that share halo planes in one L2, but the chained scan needs
//     every XCD at the same frontier of the brick order: 11 ms at 8 groups, 42 ms at a volume); a wave whose own counter has run dry
//     takes from the next one.
//   * order: triangles land in canonical order (block, cell, triangle): brick b's first triangle is the sum of the counts of all
//     bricks before it.  Level 1: a brick publishes its count (bstat) the moment it is classified and adds it to its group's sum (gsum,
//     64 bricks); the wave whose add completes the group publishes the group's aggregate (gstat), looks back over the earlier groups
//     until it meets one with an inclusive prefix and publishes its own.  Level 2: a brick's prefix = inclusive prefix of the previous
//     group + the counts of the bricks before it in its own group (one 256-byte load).
//   * progress: a ticket is taken by a running wave, per counter in ascending order, and a wave publishes its count BEFORE it waits for
//     anything -- so the lowest brick not yet published is either being classified or next on a counter whose earlier bricks are all
//     published, i.e. whose waves are emitting and will come back for it.  Every spin is bounded all the same (error word ->
//     VTMC_ERR_DEVICE).  A ticket is requested when the wave is free to classify its brick at once: a ticket held by a wave that is
//     still emitting keeps every later brick waiting (measured: 42 ms instead of 3.9 when requested one block early).
//   * classify ahead: a wave classifies and publishes the NEXT brick before it emits the current one, so the wait for the current
//     brick's prefix (every lower brick classified + the trips of the group words through memory-side coherence) passes under useful
//     work: 3.9 -> 2.0 ms on the 1024^3 field.
// Measured (profiles/r03/experiments/one_pass_and_reread.txt): 2.03 ms against 1.75 ms for the three launches on the same box, 1.72-1.88 ms
// with the look-back switched off; and 7.22 GB of memory-side reads against 4.50 + 1.82 GB: with all eight (non-coherent) L2s working along
// the same stretch of the volume, halo planes and most tile rows are fetched by an XCD that did not read them first.  The kernel is an
// opt-in (tuning key "one_pass"), bit-identical to the default path.
#include "emit_device.h"

#include <type_traits>

namespace vtmc {

namespace {

constexpr unsigned kOpPublished = 0x80000000u;
constexpr unsigned long long kOpAggregate = 1ull << 62, kOpInclusive = 2ull << 62, kOpValue = (1ull << 62) - 1ull;
constexpr unsigned long long kOpOne = 1ull << 40, kOpSum = kOpOne - 1ull;
constexpr int kOpSpinLimit = 1 << 21;

template <typename T>
__device__ __forceinline__ T ld_agent(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

#ifdef VTMC_ONEPASS_TIMING   // diagnostic build: where a wave's time goes (100 MHz ticks summed over all waves, printed by the counts kernel)
#define VTMC_T(i)                                   \
    do {                                            \
        const long long now_ = wall_clock64();      \
        tacc[i] += now_ - tlast;                    \
        tlast = now_;                               \
    } while (0)
#else
#define VTMC_T(i) do { } while (0)
#endif

template <bool FAST>
__global__ __launch_bounds__(256, 3) void onepass_kernel(BlockSpace sp, DeviceTables tb, OnePassCtrl ctl, uint32_t *__restrict__ offsets,
                                                                    float *__restrict__ out, uint32_t capacity, int nsegx, int n_bricks,
                                                                    int unit_bricks, uint32_t *__restrict__ totals,
                                                                    uint32_t *__restrict__ host_totals, int classify_ablate, int ablate, int depth, int prefetch)
{
    using Lds = typename std::conditional<FAST, EmitLdsOnce, EmitLds2>::type;
    __shared__ Lds s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ OnceTables s_once[1];
    __shared__ unsigned char s_trinum[256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef VTMC_ONEPASS_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#endif
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    once_tables_init(&s_once[0], threadIdx.x);
#ifdef VTMC_DEBUG_POISON_LDS
    for (unsigned i = threadIdx.x; i < sizeof(s_lds) / 4; i += 256) reinterpret_cast<unsigned *>(s_lds)[i] = 0x7FC00000u;
#endif
    __syncthreads();
    Lds *L = &s_lds[wave];
    float *tile;
    if constexpr (FAST) tile = L->c.tile;
    else tile = L->tile;

    // tile fetch of one block: 20 instructions of 5 rows x 10 samples (the emit kernel's shape, emit_kernels.hip)
    const int lq = lane % 10, rq = lane / 10;
    const bool lane_ok = rq < 5;
    const int rqc = lane_ok ? rq : 4;
    const unsigned off0 = (unsigned)(lq + rqc * sp.sy) * 4u, off1 = off0 + (unsigned)(5 * sp.sy) * 4u;
    const unsigned slab_bytes = (unsigned)sp.sz * 4u;
    const int lds0 = lq + 10 * rqc;
    auto load_rows = [&](const char *src, unsigned mask, float (&dst)[20]) {
        const unsigned ym = mask & 0xFFu, zm = mask >> 8;
        const unsigned ny = ym | (ym << 1) | (ym << 2), nz = zm | (zm << 1) | (zm << 2);
        const bool need0 = lane_ok && ((ny >> rqc) & 1u), need1 = lane_ok && ((ny >> (5 + rqc)) & 1u);
        if (need0) {
            const char *p = src + off0;
#pragma unroll
            for (int c = 0; c < 10; ++c, p += slab_bytes)
                if ((nz >> c) & 1u) dst[2 * c] = *reinterpret_cast<const float *>(p);
        }
        if (need1) {
            const char *p = src + off1;
#pragma unroll
            for (int c = 0; c < 10; ++c, p += slab_bytes)
                if ((nz >> c) & 1u) dst[2 * c + 1] = *reinterpret_cast<const float *>(p);
        }
    };
    auto store_tile = [&](const float (&v)[20]) {
        if (lane_ok) {
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                tile[lds0 + 100 * c] = v[2 * c];
                tile[lds0 + 100 * c + 50] = v[2 * c + 1];
            }
        }
    };

    // tickets
    int q = (int)(blockIdx.x & 7u), tries = 0;
    unsigned tick_raw = 0;
    bool requested = false;
    auto request = [&]() {
        if (lane == 0) tick_raw = __hip_atomic_fetch_add(ctl.queue + (q & 7) * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        requested = true;
    };
    auto collect = [&]() -> int {   // the brick of the requested ticket, or -1 when every counter has run dry
        for (;;) {
            if (!requested) request();
            requested = false;
            const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)tick_raw);
            const unsigned long long u = (unsigned long long)(k / (unsigned)unit_bricks) * 8ull + (unsigned)(q & 7);
            const unsigned long long brick = u * (unsigned)unit_bricks + k % (unsigned)unit_bricks;
            if (brick < (unsigned long long)n_bricks) return (int)brick;
            ++q;
            if (++tries >= 8) return -1;
        }
    };
    bool failed = false;
    auto fail = [&]() {
        failed = true;
        if (lane == 0) {
            st_agent(ctl.err, 1u);
            totals[8] = 1u;
            if (host_totals) host_totals[8] = 1u;
        }
    };

    // A brick between its classification and its emission: the block counts stay in the lanes that computed them.
    struct Entry {
        int brick;
        unsigned total, rows, excl;     // per lane: triangles of the lane's block, its row mask, triangles of the brick's earlier blocks
        unsigned brick_total;
        bool have_group_excl;           // this wave completed the brick's group and knows the groups' prefix already
        unsigned long long group_excl;
    };
    struct Where {
        int segx, bid0;
        const float *base;
    };
    auto locate = [&](int brick) {
        Where w;
        w.segx = brick % nsegx;
        int t = brick / nsegx;
        const int by = t % sp.nby;
        t /= sp.nby;
        const int bz = t % sp.nbz;
        const int v = t / sp.nbz;
        w.base = sp.base + v * sp.sv + (8ll * by) * sp.sy + (8ll * bz) * sp.sz;
        w.bid0 = v * sp.bpv + sp.nbx * (by + sp.nby * bz);
        return w;
    };

    // ticket -> counts -> published: nothing in here waits for another brick's EMISSION
    auto classify_publish = [&](int brick) {
        Entry e;
        e.brick = brick;
        const Where at = locate(brick);
        const int gx = at.segx * 64 + lane;
        const int gxc = gx < sp.nx + 1 ? gx : sp.nx + 1;
        int xe = at.segx * 64 + 64;
        xe = xe < sp.nx + 1 ? xe : sp.nx + 1;
        unsigned rows = 0;
        unsigned total = classify_brick_column<false, false>(sp, s_trinum, at.base, gx, gxc, xe, lane, classify_ablate, nullptr, &rows);
        total += __shfl_xor(total, 1);
        total += __shfl_xor(total, 2);
        total += __shfl_xor(total, 4);
        rows |= (unsigned)__shfl_xor((int)rows, 1);
        rows |= (unsigned)__shfl_xor((int)rows, 2);
        rows |= (unsigned)__shfl_xor((int)rows, 4);
        // inclusive sum over the brick's eight blocks (every lane of an 8-lane group holds its block's count)
        unsigned incl = total, o;
        o = (unsigned)__shfl_up((int)incl, 8);
        if (lane >= 8) incl += o;
        o = (unsigned)__shfl_up((int)incl, 16);
        if (lane >= 16) incl += o;
        o = (unsigned)__shfl_up((int)incl, 32);
        if (lane >= 32) incl += o;
        VTMC_T(1);   // rows + counts
        e.total = total;
        e.rows = rows;
        e.excl = incl - total;
        e.brick_total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);

        const int g = brick >> 6;
        const int gsize = n_bricks - (g << 6) < 64 ? n_bricks - (g << 6) : 64;
        unsigned long long old = 0;
        if (lane == 0) {
            st_agent(ctl.bstat + brick, kOpPublished | e.brick_total);
            old = __hip_atomic_fetch_add(ctl.gsum + g, kOpOne | (unsigned long long)e.brick_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(old >> 32)) << 32) |
              (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        e.group_excl = 0;
        e.have_group_excl = g == 0;
        if ((int)(old >> 40) == gsize - 1) {   // this add completed the group: its aggregate, then its inclusive prefix
            const unsigned long long agg = (old & kOpSum) + e.brick_total;
            if (g > 0) {
                if (lane == 0) st_agent(ctl.gstat + g, kOpAggregate | agg);
                for (int j = g - 1; j >= 0; j -= 64) {   // windows of 64 earlier groups, nearest first (lane 0 = group j)
                    const int idx = j - lane;
                    const bool valid = idx >= 0;
                    unsigned long long w = valid ? 0ull : kOpInclusive;
                    int spins = 0;
                    for (;;) {
                        if ((w >> 62) == 0ull) w = ld_agent(ctl.gstat + idx);   // only the lanes still waiting ask again
                        if (!__builtin_amdgcn_ballot_w64((w >> 62) == 0ull)) break;
                        if (++spins > kOpSpinLimit) {
                            fail();
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (failed) break;
                    const u64 inc = __builtin_amdgcn_ballot_w64(valid && (w >> 62) == 2ull);
                    const int first = inc ? __builtin_ctzll(inc) : 64;
                    unsigned long long part = (valid && lane <= first) ? (w & kOpValue) : 0ull;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
                    e.group_excl += part;
                    if (inc) break;
                }
                e.have_group_excl = true;
            }
            if (lane == 0 && !failed) st_agent(ctl.gstat + g, kOpInclusive | (e.group_excl + agg));
        }
        VTMC_T(2);   // publication (+ the group's prefix if this brick completed it)
        return e;
    };

    int vm_unused = 0;
    PhaseClock pc_unused;
    bool more = true;   // tickets left
    // prefix (the groups before + the bricks before it in its group), offsets, emission
    auto finish = [&](const Entry &e) {
        const int brick = e.brick, g = brick >> 6, bi = brick & 63;
        // The next ticket travels beside the END of this brick: beside its last block, or beside the look-back of a brick without
        // triangles.  Never earlier: until the ticket's brick is classified, no brick after it can be emitted.
        if (more && prefetch && e.brick_total == 0u) request();   // opt-in ("one_pass_prefetch"): measured slower
        const Where at = locate(brick);
        // the first block's tile travels beside the look-back wait; every later one beside the emission of the block before it
        u64 act = __builtin_amdgcn_ballot_w64((lane & 7) == 0 && e.total != 0u);
        if (ablate & 64) act = 0;   // diagnostics: no emission
        float pre[20] = {};   // rows a block does not need keep whatever was there: never read
        auto fetch = [&](int l8) {
            const unsigned rmask = (unsigned)__builtin_amdgcn_readlane((int)e.rows, l8);
            load_rows(reinterpret_cast<const char *>(at.base + at.segx * 64 + l8), rmask, pre);   // 8 cells per block: the block's first sample is l8
        };
        if (act) fetch(__builtin_ctzll(act));
        unsigned long long prefix = 0;
        if (!(ablate & 32)) {   // ablate 32 (diagnostics): no look-back -- every brick writes from offset 0
            unsigned w = lane < bi ? 0u : kOpPublished;
            unsigned long long gw = (e.have_group_excl || lane != 0) ? kOpInclusive : 0ull;   // lane 0 asks for the previous group's prefix
            int spins = 0;
            for (;;) {
                if (!(w & kOpPublished)) w = ld_agent(ctl.bstat + (g << 6) + lane);   // only the lanes still waiting ask again
                if ((gw >> 62) != 2ull) gw = ld_agent(ctl.gstat + g - 1);
                if (!__builtin_amdgcn_ballot_w64(!(w & kOpPublished) || (gw >> 62) != 2ull)) break;
                if (++spins > kOpSpinLimit) {
                    fail();
                    return;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            gw = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gw >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)gw);
            unsigned part = lane < bi ? (w & ~kOpPublished) : 0u;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
            prefix = (e.have_group_excl ? e.group_excl : (gw & kOpValue)) + part;
        }
        VTMC_T(3);   // look-back
        if (ablate & 128) prefix = 162ull * (unsigned)brick;   // diagnostics (with 32): no look-back, the stores spread over the real buffer

        const int bl = at.segx * 8 + (lane >> 3);
        if ((lane & 7) == 0 && bl < sp.nbx) offsets[at.bid0 + bl] = (uint32_t)(prefix + e.excl);
        if (brick == n_bricks - 1 && lane == 0) {
            unsigned long long T = prefix + e.brick_total;
            offsets[sp.n_blocks] = (uint32_t)T;
            const uint32_t sat = T > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)T;
            const uint32_t tot[4] = {sat, 0u, (uint32_t)T, (uint32_t)(T >> 32)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                totals[k] = tot[k];
                if (host_totals) host_totals[k] = tot[k];
            }
        }

        // emit the brick's blocks, ascending
        while (act) {
            const int l8 = __builtin_ctzll(act);
            act &= act - 1;
            if (!act && more && prefetch) request();
            const int budget = __builtin_amdgcn_readlane((int)e.total, l8);
            const unsigned long long tri_base = prefix + (unsigned)__builtin_amdgcn_readlane((int)e.excl, l8);
            const unsigned rmask = (unsigned)__builtin_amdgcn_readlane((int)e.rows, l8);
            VTMC_WAVE_SYNC();
            store_tile(pre);
            if (act) fetch(__builtin_ctzll(act));
            VTMC_WAVE_SYNC();
            VTMC_T(4);   // tile
            if (tri_base + (unsigned long long)budget > (unsigned long long)capacity) continue;   // the host grows the buffer and runs the step again
            const int bid = at.bid0 + at.segx * 8 + (l8 >> 3);
            if constexpr (FAST) emit_block_once<true>(L, s_vert, &s_once[0], (size_t)tri_base, budget, bid, out, lane, ablate, rmask, vm_unused, pc_unused);
            else emit_block_from_tile<false>(L, s_vert, (size_t)tri_base, budget, bid, out, lane, ablate, rmask, vm_unused);
            VTMC_T(5);   // emission
        }
    };

    // The wave classifies (and publishes) up to `depth` bricks ahead of the one it emits: the wait for a brick's prefix -- every lower
    // brick classified, the group words through memory-side coherence -- passes while the next brick is classified.
    Entry e0{}, e1{};   // e0: the oldest
    int n = 0;
    for (;;) {
        while (more && n < depth && !failed) {
            const int b = collect();
            if (b < 0) {
                more = false;
                break;
            }
            VTMC_T(0);   // ticket
            const Entry x = classify_publish(b);
            if (n == 0) e0 = x;
            else e1 = x;
            ++n;
        }
        if (n == 0 || failed) break;
        finish(e0);
        VTMC_T(6);   // offsets, bookkeeping
        e0 = e1;
        --n;
    }
#ifdef VTMC_ONEPASS_TIMING
    VTMC_T(7);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(ctl.err) + 2 + i, (unsigned long long)tacc[i]);
#endif
}

// per-volume {vertices, triangles} (the array a multi-GPU caller all-gathers) from the offsets the one-pass kernel left
__global__ void onepass_volume_counts_kernel(const uint32_t *__restrict__ offsets, int bpv, int n_volumes, uint32_t *__restrict__ volume_counts,
                                             const unsigned long long *timing, int n_waves)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef VTMC_ONEPASS_TIMING
    if (v == 0 && timing) {
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += timing[i];
        printf("onepass wave time [us per wave, %% ]: ticket %.1f (%.0f) rows+counts %.1f (%.0f) publish %.1f (%.0f) look-back %.1f (%.0f) tile %.1f (%.0f) emit %.1f (%.0f) rest %.1f (%.0f) tail %.1f (%.0f)\n",
               timing[0] / 100.0 / n_waves, 100.0 * timing[0] / tot, timing[1] / 100.0 / n_waves, 100.0 * timing[1] / tot, timing[2] / 100.0 / n_waves, 100.0 * timing[2] / tot,
               timing[3] / 100.0 / n_waves, 100.0 * timing[3] / tot, timing[4] / 100.0 / n_waves, 100.0 * timing[4] / tot, timing[5] / 100.0 / n_waves, 100.0 * timing[5] / tot,
               timing[6] / 100.0 / n_waves, 100.0 * timing[6] / tot, timing[7] / 100.0 / n_waves, 100.0 * timing[7] / tot);
    }
#endif
    if (v >= n_volumes) return;
    const long long lo = (long long)v * bpv, hi = lo + bpv;
    const uint32_t t = offsets[hi] - offsets[lo];
    volume_counts[2 * v] = 3u * t;   // soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
    volume_counts[2 * v + 1] = t;
}

size_t onepass_ctrl_bytes(const BlockSpace &sp)
{
    const int nsegx = (sp.nx + 63) / 64;
    const long long n_bricks = (long long)(sp.n_blocks / sp.bpv) * sp.nbz * sp.nby * nsegx;
    const long long n_groups = (n_bricks + 63) / 64;
    return 2048 + 256 + (size_t)n_groups * 16 + (size_t)(n_groups * 64) * 4;
}

hipError_t launch_onepass(const BlockSpace &sp, const DeviceTables &tb, void *ctrl, uint32_t *offsets, void *triangles, uint32_t capacity,
                          uint32_t *totals, uint32_t *host_totals, uint32_t *volume_counts, int n_volumes, int n_cus, const Tuning &tune,
                          hipStream_t stream)
{
    const int nsegx = (sp.nx + 63) / 64;
    const long long n_vol = sp.n_blocks / sp.bpv;
    const long long bricks_per_volume = (long long)sp.nbz * sp.nby * nsegx;
    const long long n_bricks = n_vol * bricks_per_volume;
    if (n_bricks > 0x7fffffffll) return hipErrorInvalidValue;
    const long long n_groups = (n_bricks + 63) / 64;
    hipError_t e = hipMemsetAsync(ctrl, 0, onepass_ctrl_bytes(sp), stream);
    if (e != hipSuccess) return e;
    OnePassCtrl c;
    unsigned char *p = (unsigned char *)ctrl;
    c.queue = (unsigned *)p;
    c.err = (unsigned *)(p + 2048);
    c.gsum = (unsigned long long *)(p + 2048 + 256);
    c.gstat = c.gsum + n_groups;
    c.bstat = (unsigned *)(c.gstat + n_groups);
    int unit = tune.one_pass_unit > 0 ? tune.one_pass_unit * 64 : 64;
    const int depth = tune.one_pass_depth == 1 ? 1 : 2;
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : (tune.emit_fast_math ? 3 : 4);
    int wgs = (n_cus * per_cu + 7) & ~7;
    const dim3 g(wgs), blk(256);
    launch_begin();
    if (tune.emit_fast_math)
        hipLaunchKernelGGL((onepass_kernel<true>), g, blk, 0, stream, sp, tb, c, offsets, (float *)triangles, capacity, nsegx, (int)n_bricks, unit, totals,
                           host_totals, tune.classify_ablate, tune.emit_ablate, depth, tune.one_pass_prefetch);
    else
        hipLaunchKernelGGL((onepass_kernel<false>), g, blk, 0, stream, sp, tb, c, offsets, (float *)triangles, capacity, nsegx, (int)n_bricks, unit, totals,
                           host_totals, tune.classify_ablate, tune.emit_ablate, depth, tune.one_pass_prefetch);
    e = launch_end();
    if (e != hipSuccess) return e;
    if (volume_counts && n_volumes > 0) {
        hipLaunchKernelGGL(onepass_volume_counts_kernel, dim3((n_volumes + 255) / 256), dim3(256), 0, stream, offsets, sp.bpv, n_volumes, volume_counts,
                           reinterpret_cast<const unsigned long long *>(c.err) + 2, wgs * 4);
        e = launch_end();
    }
    return e;
}

}  // namespace vtmc
