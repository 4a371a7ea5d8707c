// fused_kernels.hip -- the whole of BatchUpdate's device side in ONE launch for dense x-fastest soup output
// (hand-written gfx950 / CDNA4, wave64): CollectTriNum.compute:23-64 + the prefix sum of
// VoxelTerrain.cs:406-420 + SampleNormal.compute:23-34 + MarchingCube.compute:101-165 of the reference
// (/root/reference/Unity-Project/Assets/).
//
// Why: the staged path reads the field twice from HBM (classify streams it, emit gathers the tiles of
// the non-empty blocks again ~a millisecond later, long after they left the caches).  Here a persistent
// workgroup classifies a UNIT of four bricks (64 x 8 x 8 cells each, 32 blocks), publishes the unit's
// triangle total, and emits the unit it classified ONE ITERATION EARLIER: by then
//   * that unit's exclusive prefix can be summed from its predecessors' status words without spinning
//     (chained scan with decoupled look-back, units handed out by a ticket in start order, so every
//     predecessor is running or done; the lag gives them a whole classify phase to publish), and
//   * its rows are still in L2 / the memory-side cache, so the tile gather costs no HBM traffic.
// Triangles land at the offsets the staged path gives (block order): the outputs are byte-identical.
#include "emit_device.h"

namespace vtmc {

// Control words (64-bit, zeroed before every launch):
//   [32 x]            ticket counter of XCD x (x < 8), one 256-byte line each
//   [256]             workgroups finished      [257] error word
//   [kCtrlHead + g]               SUP[g]: bricks of group g published (bits 56+) | their triangle sum
//   [kCtrlHead + G + g]           P[g]:   bit 63 valid | triangles of groups 0..g
//   [kCtrlHead + 2 G + b]         status[b]: bit 63 published | triangles of brick b
// A group is 64 consecutive bricks.
constexpr int kCtrlHead = 264;
constexpr int kRun = 16;   // bricks an XCD takes in a row (brick order is interleaved over the 8 XCDs run by run)
constexpr unsigned long long kFlag = 1ull << 63, kSupOne = 1ull << 56, kSupSum = kSupOne - 1;
constexpr int kFusedSpinLimit = 1 << 22;

struct FusedGeom {
    int nsegx, n_bricks, n_groups;
    FastDiv d_nsegx, d_nby, d_nbz;
};

// fused_kernel: every WAVE is on its own -- no workgroup barrier inside the loop.  A wave takes a brick (64 x 8 x 8
// cells, 8 blocks) from its XCD's ticket counter, classifies it, publishes its triangle total, and then emits the
// brick it classified ONE ITERATION EARLIER:
//   * the exclusive prefix of that brick = P[group - 1] + the totals of the bricks before it in its own group of 64:
//     one round trip (64 lanes, 64 words), and after a whole classify phase of lag the words are there -- no spinning;
//     the wave that completes a group (the 64th atomic add into SUP[g]) derives P[g] by a decoupled look-back over the
//     group words, also one round trip;
//   * a ticket is taken right before its brick is classified and nothing slow sits between ticket and publication:
//     every later brick waits for that total;
//   * bricks are handed out in brick order per XCD counter, runs of 16 interleaved over the XCDs: whoever holds the
//     smallest unpublished brick is in its classify phase, which never waits -- no deadlock, whatever the residency
//     (the grid is a multiple of 8 workgroups, so every counter has a home workgroup that stays until it is exhausted);
//   * the rows of the previous brick's tiles were read by this very wave some microseconds ago: the gather is served
//     by L2 / the memory-side cache, not by HBM.
template <bool FAST>
__global__ __launch_bounds__(256, 4) void fused_kernel(BlockSpace sp, DeviceTables tb, FusedGeom geo, uint32_t *__restrict__ counts,
                                                     uint32_t *__restrict__ offsets, unsigned long long *__restrict__ ctrl,
                                                     uint32_t *__restrict__ totals, uint32_t *__restrict__ host_totals, uint32_t capacity,
                                                     float *__restrict__ out, int cl_ablate, int ablate,
                                                     uint32_t *__restrict__ volume_counts, int n_volumes)
{
    __shared__ EmitLds2 s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ unsigned s_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    __syncthreads();
    unsigned long long *sup = ctrl + kCtrlHead, *pfx = sup + geo.n_groups, *status = pfx + geo.n_groups;
    const int n_bricks = geo.n_bricks;
    EmitLds2 *L = &s_lds[wave];

    // tile fetch of the emit stage (emit_kernels.hip): 5 rows of 10 samples per instruction, x fastest
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
    auto store_tile = [&](float *tile, const float (&v)[20]) {
        if (lane_ok) {
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                tile[lds0 + 100 * c] = v[2 * c];
                tile[lds0 + 100 * c + 50] = v[2 * c + 1];
            }
        }
    };
    struct BrickPos {
        int segx, row_bid;        // first block id of the brick's block row
        long long row_origin;     // element offset of the block row's (x = 0) corner
    };
    auto brick_pos = [&](int brick) {
        BrickPos p;
        unsigned t = geo.d_nsegx.quot((unsigned)brick);
        p.segx = brick - (int)t * geo.nsegx;
        const unsigned t2 = geo.d_nby.quot(t);
        const int by = (int)(t - t2 * (unsigned)sp.nby);
        const unsigned v = geo.d_nbz.quot(t2);
        const int bz = (int)(t2 - v * (unsigned)sp.nbz);
        p.row_bid = (int)v * sp.bpv + sp.nbx * (by + sp.nby * bz);
        p.row_origin = (long long)v * sp.sv + 8ll * (by * sp.sy + bz * sp.sz);
        return p;
    };
    auto fail = [&]() {
        totals[8] = 1u;
        if (host_totals) host_totals[8] = 1u;
    };

    int xcd = blockIdx.x & 7, tried = 0;
    int prev = -1;
    uint32_t prev_word = 0;   // lanes 8 k: count | row mask of block k of the previous brick
    for (;;) {
        // ---- ticket -> brick (this XCD's counter; an exhausted counter sends the wave on to the next one) ----
        int brick = -1;
        while (tried < 8) {
            unsigned n = 0;
            if (lane == 0) n = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ctrl + 32 * xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
            const long long b = ((long long)(n / kRun) * 8 + xcd) * kRun + (n % kRun);
            if (b < n_bricks) {
                brick = (int)b;
                break;
            }
            xcd = (xcd + 1) & 7;
            ++tried;
        }
        const bool have = brick >= 0;
        uint32_t word = 0;
        if (have) {
            const BrickPos p = brick_pos(brick);
            const int gx = p.segx * 64 + lane;
            const int gxc = gx < sp.nx + 1 ? gx : sp.nx + 1;
            int xe = p.segx * 64 + 64;
            xe = xe < sp.nx + 1 ? xe : sp.nx + 1;
            unsigned rows = 0;
            unsigned total = classify_brick_column<false, true>(sp, reinterpret_cast<const unsigned char *>(s_vert), sp.base + p.row_origin, gx, gxc, xe,
                                                                lane, cl_ablate, nullptr, &rows);
            total += __shfl_xor(total, 1);
            total += __shfl_xor(total, 2);
            total += __shfl_xor(total, 4);
            rows |= (unsigned)__shfl_xor((int)rows, 1);
            rows |= (unsigned)__shfl_xor((int)rows, 2);
            rows |= (unsigned)__shfl_xor((int)rows, 4);
            const int bx = p.segx * 8 + (lane >> 3);
            if (bx < sp.nbx && (lane & 7) == 0) {
                word = total | (rows << 16);
                counts[p.row_bid + bx] = word;
            }
            uint32_t bt = word & kCountMask;   // brick total: the 8 group leaders
            bt += __shfl_xor(bt, 8);
            bt += __shfl_xor(bt, 16);
            bt += __shfl_xor(bt, 32);
            // publish: the brick's own word, then its share of the group's
            const int g = brick >> 6;
            unsigned long long old = 0;
            if (lane == 0) {
                __hip_atomic_store(&status[brick], kFlag | (unsigned long long)bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = __hip_atomic_fetch_add(&sup[g], kSupOne | (unsigned long long)bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            old = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(old >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)old);
            const int need = n_bricks - (g << 6) < 64 ? n_bricks - (g << 6) : 64;
            if ((int)(old >> 56) == need - 1) {   // this wave completed group g: P[g] by a look-back over the group words
                unsigned long long ex = 0;
                bool done = g == 0, bad = false;
                for (int j = g - 1; j >= 0 && !done; j -= 64) {
                    const int idx = j - lane;
                    const bool valid = idx >= 0;
                    unsigned long long pw = kFlag, sw = 0;
                    int spins = 0;
                    for (;;) {
                        if (valid) {
                            pw = __hip_atomic_load(&pfx[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sw = __hip_atomic_load(&sup[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        // usable: P valid, or the group complete (every group before the last holds 64 bricks)
                        if (!__builtin_amdgcn_ballot_w64(valid && !(pw & kFlag) && (sw >> 56) != 64ull)) break;
                        if (++spins > kFusedSpinLimit) {
                            bad = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const u64 anch = __builtin_amdgcn_ballot_w64(valid && (pw & kFlag) != 0ull);
                    const int first = anch ? __builtin_ctzll(anch) : 64;   // nearest group whose inclusive prefix is known
                    unsigned long long val = 0;
                    if (valid && lane < first) val = sw & kSupSum;
                    else if (valid && lane == first) val = pw & ~kFlag;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
                    ex += val;
                    done = anch != 0;
                    if (bad) break;
                }
                if (lane == 0) {
                    const unsigned long long incl = ex + (old & kSupSum) + bt;
                    __hip_atomic_store(&pfx[g], kFlag | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (bad) fail();
                    if (g == geo.n_groups - 1) {   // the grand total: straight into the host's pinned words, as the staged scan does
                        const uint32_t t32 = incl > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)incl;
                        offsets[sp.n_blocks] = t32;
                        totals[0] = t32;
                        totals[2] = (uint32_t)incl;
                        totals[3] = (uint32_t)(incl >> 32);
                        if (host_totals) {
                            host_totals[0] = t32;
                            host_totals[2] = (uint32_t)incl;
                            host_totals[3] = (uint32_t)(incl >> 32);
                        }
                    }
                }
            }
        }

        // ---- the brick classified one iteration ago: prefix, offsets, emit ----
        if (prev >= 0) {
            const BrickPos p = brick_pos(prev);
            const uint32_t cw = prev_word;
            unsigned items = 0;
            {
                const u64 m = __builtin_amdgcn_ballot_w64((cw & kCountMask) != 0u);   // bits 8 k
                // compress bits 0, 8, 16, 24 (low half) and 32..56 (high half) into bits 0..7
                const unsigned lo = (unsigned)m & 0x01010101u, hi = (unsigned)(m >> 32) & 0x01010101u;
                const unsigned clo = (lo | (lo >> 7) | (lo >> 14) | (lo >> 21)) & 0xFu, chi = (hi | (hi >> 7) | (hi >> 14) | (hi >> 21)) & 0xFu;
                items = clo | (chi << 4);
            }
            // the 8 block counts in lanes 0..7, exclusive prefix within the brick
            const uint32_t w8 = (uint32_t)__shfl((int)cw, (lane & 7) * 8);
            const uint32_t c8 = lane < 8 ? w8 & kCountMask : 0u;
            uint32_t incl8 = c8;
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const uint32_t o = __shfl_up(incl8, off);
                if (lane >= off) incl8 += o;
            }
            const uint32_t excl8 = incl8 - c8;
            const uint32_t brick_total = (uint32_t)__builtin_amdgcn_readlane((int)incl8, 7);

            float pre[20] = {};
            auto item = [&](int k, const char *&src, unsigned &mask, uint32_t &rel, int &budget, int &bid) {
                const int bx = p.segx * 8 + k;
                bid = p.row_bid + bx;
                src = reinterpret_cast<const char *>(sp.base + p.row_origin + 8ll * bx);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)w8, k);
                mask = w >> 16;
                budget = (int)(w & kCountMask);
                rel = (uint32_t)__builtin_amdgcn_readlane((int)excl8, k);
            };
            const char *src = nullptr;
            unsigned mask = 0;
            uint32_t rel = 0;
            int budget = 0, bid = 0;
            const bool emitting = items && !(cl_ablate & 16);   // ablate 16: diagnostics, nothing emitted
            if (emitting) {   // the first tile is requested before the prefix round trip
                item(__builtin_ctz(items), src, mask, rel, budget, bid);
                load_rows(src, mask, pre);
            }
            // exclusive prefix: P[group - 1] (lane 63) + the bricks before this one in its group (lanes < r)
            const int g = prev >> 6, r = prev & 63;
            unsigned long long base = 0;
            {
                const bool want = lane < r || (lane == 63 && g > 0);
                const unsigned long long *addr = lane == 63 ? &pfx[g > 0 ? g - 1 : 0] : &status[(g << 6) + (lane < r ? lane : 0)];
                unsigned long long w = kFlag;
                int spins = 0;
                bool bad = false;
                for (;;) {
                    if (want) w = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!__builtin_amdgcn_ballot_w64(want && !(w & kFlag)) || (cl_ablate & 32)) break;   // ablate 32: diagnostics, no waiting
                    if (++spins > kFusedSpinLimit) {
                        bad = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                unsigned long long val = want ? (w & ~kFlag) : 0ull;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
                base = val;
                if (bad && lane == 0) fail();
            }
            if (lane < 8 && p.segx * 8 + lane < sp.nbx) offsets[p.row_bid + p.segx * 8 + lane] = (uint32_t)base + excl8;   // 32-bit: meaningless once T passes 2^32, nothing is emitted then
            const bool fits = base + brick_total <= (unsigned long long)capacity;   // else: the host grows the buffer and launches again
            if (emitting && fits) {
                for (;;) {
                    const unsigned mask_now = mask;
                    const size_t tb_now = (size_t)(base + rel);
                    const int budget_now = budget, bid_now = bid;
                    VTMC_WAVE_SYNC();
                    store_tile(L->tile, pre);
                    items &= items - 1u;
                    if (items) {   // the next block's tile lands while this one is processed
                        item(__builtin_ctz(items), src, mask, rel, budget, bid);
                        load_rows(src, mask, pre);
                    }
                    VTMC_WAVE_SYNC();
                    emit_block_from_tile<FAST>(L, s_vert, tb_now, budget_now, bid_now, out, lane, ablate, mask_now);
                    if (!items) break;
                }
            }
        }
        if (!have) break;
        prev = brick;
        prev_word = word;
    }

    // per-volume {vertices, triangles} by the last workgroup to finish, from the finished offsets
    if (volume_counts) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ctrl + 256), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_last == gridDim.x - 1) {
            __threadfence();
            for (int v = threadIdx.x; v < n_volumes; v += 256) {
                const long long lo = (long long)v * sp.bpv, hi = lo + sp.bpv;
                const uint32_t t = __hip_atomic_load(&offsets[hi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                   __hip_atomic_load(&offsets[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                volume_counts[2 * v] = 3u * t;   // soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
                volume_counts[2 * v + 1] = t;
            }
        }
    }
}

static FusedGeom fused_geom(const BlockSpace &sp)
{
    FusedGeom g;
    g.nsegx = (sp.nx + 63) / 64;
    const long long n_vol = sp.n_blocks / (sp.bpv > 0 ? sp.bpv : 1);
    const long long nb = n_vol * sp.nbz * sp.nby * g.nsegx;
    g.n_bricks = (int)nb;
    g.n_groups = (int)((nb + 63) / 64);
    g.d_nsegx = FastDiv((unsigned)g.nsegx);
    g.d_nby = FastDiv((unsigned)(sp.nby > 0 ? sp.nby : 1));
    g.d_nbz = FastDiv((unsigned)(sp.nbz > 0 ? sp.nbz : 1));
    return g;
}

hipError_t launch_fused(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts, uint32_t *offsets, unsigned long long *ctrl,
                        uint32_t *totals, uint32_t *host_totals, uint32_t capacity, void *triangles, int n_cus, const Tuning &tune,
                        uint32_t *volume_counts, int n_volumes, hipStream_t stream)
{
    const long long n_vol = sp.n_blocks / (sp.bpv > 0 ? sp.bpv : 1);
    if (n_vol * sp.nbz * sp.nby * ((sp.nx + 63) / 64) > 0x3fffffffll) return hipErrorInvalidValue;
    const FusedGeom geo = fused_geom(sp);
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : 4;
    long long wgs = (long long)n_cus * per_cu;
    const long long useful = (geo.n_bricks + kWavesPerWg - 1) / kWavesPerWg;
    if (wgs > useful) wgs = useful;
    wgs = (wgs + 7) & ~7ll;   // every XCD counter needs a home workgroup
    if (wgs < 8) wgs = 8;
    if (tune.emit_fast_math)
        hipLaunchKernelGGL((fused_kernel<true>), dim3((unsigned)wgs), dim3(256), 0, stream, sp, tb, geo, counts, offsets, ctrl, totals, host_totals, capacity,
                           (float *)triangles, tune.classify_ablate, tune.emit_ablate, volume_counts, n_volumes);
    else
        hipLaunchKernelGGL((fused_kernel<false>), dim3((unsigned)wgs), dim3(256), 0, stream, sp, tb, geo, counts, offsets, ctrl, totals, host_totals, capacity,
                           (float *)triangles, tune.classify_ablate, tune.emit_ablate, volume_counts, n_volumes);
    return hipGetLastError();
}

size_t fused_ctrl_words(const BlockSpace &sp)
{
    const FusedGeom g = fused_geom(sp);
    return (size_t)kCtrlHead + 2 * (size_t)g.n_groups + (size_t)g.n_bricks + 8;
}

}  // namespace vtmc
