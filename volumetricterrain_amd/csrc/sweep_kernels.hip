// sweep_kernels.hip -- the single-pass extractor for dense volumes with stride_x == 1
// (hand-written gfx950 / CDNA4, wave64): every sample is streamed from HBM once and every triangle
// is written once, in canonical order, by ONE kernel.
//
// Replaces, in one launch (paths relative to /root/reference/Unity-Project/Assets/):
//   Shaders/CollectTriNum.compute:41-64 (classify + count), its single-address InterlockedAdd
//   (:54) and the 4-byte read-back of Scripts/VoxelTerrain.cs:394-395; Shaders/SampleNormal.compute:23-34;
//   Shaders/MarchingCube.compute:101-165 including its append cursor (:160-162).
//
// Structure.  A workgroup (4 waves) repeatedly draws a ticket = 4 consecutive bricks of 64x8x8
// cells (8 blocks along x each), in the linear order that makes block ids ascend:
//   1. classify: each wave streams its brick (81 row loads, classify_brick_column) and reduces the
//      per-block triangle counts;
//   2. chained scan ("decoupled look-back"): the wave publishes its brick total in an 8-byte status
//      word {flag, value} (one agent-scope store, the data IS the flag), then sums its
//      predecessors' words -- 64 at a time, one per lane -- until it meets an inclusive prefix.
//      Tickets are handed out in brick order, so every predecessor is resident or finished when
//      a wave waits on it: the chain cannot deadlock, whatever the dispatch order or XCD placement;
//      every spin is bounded all the same (a timeout raises ctrl[kCtrlError] and the host fails);
//   3. emit: the ticket's non-empty blocks go to a work list in LDS and are shared out over the four
//      waves (emit_block_from_tile); their 10^3 tiles are re-read through L2 / Infinity Cache, where
//      the classify pass of the same workgroup has just put them.
// Triangles beyond `capacity` are counted but not written; the host grows the buffer and re-runs.
// HBM-bound table lookup + lerp, no MFMA.  Algorithmic bytes per launch: 4*S + 4*B + 76*T.
#include "emit_device.h"

namespace vtmc {

constexpr u64 kFlagAgg = 1ull << 62;   // value = this brick's own triangle count
constexpr u64 kFlagInc = 2ull << 62;   // value = triangles of every brick up to and including this one
constexpr u64 kValueMask = (1ull << 62) - 1;
constexpr unsigned kSpinLimit = 1u << 20;  // >= 50 ms of polling; a healthy wait lasts microseconds

constexpr int kEmitBatch = 24;  // non-empty blocks a workgroup collects (over several tickets) before its waves emit them

struct SweepWork {
    u64 base;        // first triangle of the block in the output
    int block;       // block id
    unsigned count;  // its triangle count
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Exclusive prefix (triangles of all bricks before `brick`) by decoupled look-back; publishes this
// brick's aggregate first and its inclusive prefix last.  Returns false on a spin timeout.
__device__ __forceinline__ bool chained_scan(u64 *__restrict__ status, int brick, uint32_t own, int lane, u64 &excl_out)
{
    u64 excl = 0;
    bool ok = true;
    if (brick > 0) {
        if (lane == 0) __hip_atomic_store(status + brick, kFlagAgg | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int idx = brick - 1 - lane;  // lane i reads predecessor brick-1-i of the current window
        unsigned spins = 0;
        for (;;) {
            const u64 s = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagInc;
            const u64 m_inc = __builtin_amdgcn_ballot_w64((s >> 62) == 2u);
            const u64 m_empty = __builtin_amdgcn_ballot_w64((s >> 62) == 0u);
            // lanes up to and including the nearest inclusive prefix (all 64 if there is none)
            const u64 need = m_inc ? (((m_inc & (0 - m_inc)) << 1) - 1) : ~0ull;
            if (m_empty & need) {
                if (++spins > kSpinLimit) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            const bool mine = (need >> lane) & 1;
            const bool is_inc = (s >> 62) == 2u;
            excl += wave_sum_u32(mine && !is_inc ? (uint32_t)s : 0u);  // aggregates are <= 20480 each
            if (m_inc) {
                const int src = __builtin_ctzll(m_inc);
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)s, src);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(s >> 32), src);
                excl += (((u64)hi << 32) | lo) & kValueMask;
                break;
            }
            idx -= 64;
        }
    }
    // also on a timeout: successors must not wait for this brick (the launch has failed anyway)
    if (lane == 0)
        __hip_atomic_store(status + brick, kFlagInc | ((excl + own) & kValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    excl_out = excl;
    return ok;
}

template <bool FAST>
__global__ __launch_bounds__(256, 3) void sweep_kernel(BlockSpace sp, DeviceTables tb, int nsegx, int n_bricks,
                                                        u64 *__restrict__ status, unsigned *__restrict__ ctrl,
                                                        uint32_t *__restrict__ offsets, u64 capacity,
                                                        float *__restrict__ out, int ablate)
{
    __shared__ EmitLds2 s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ SweepWork s_work[kEmitBatch + 8 * kWavesPerWg];
    __shared__ unsigned char s_trinum[256];
    __shared__ unsigned s_ticket, s_nwork, s_next;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    EmitLds2 *L = &s_lds[wave];
    if (threadIdx.x == 0) {
        s_nwork = 0;
        s_next = 0;
    }

    for (bool more = true; more;) {
        if (threadIdx.x == 0) {
            // a timed-out wait anywhere ends the launch: no further tickets are drawn
            const unsigned failed = __hip_atomic_load(ctrl + kCtrlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // kTicketGroups counters, 256 bytes apart (one word saturates near 88 atomics/us chip-wide):
            // group g = blockIdx % kTicketGroups draws the tickets congruent to g, in ascending order
            const unsigned grp = blockIdx.x % kTicketGroups;
            const unsigned tk = __hip_atomic_fetch_add(ctrl + kCtrlTicket + 64 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = failed || tk > 0x0FFFFFFFu ? 0xFFFFFFFFu : tk * kTicketGroups + grp;
        }
        __syncthreads();
        const unsigned ticket = __builtin_amdgcn_readfirstlane(s_ticket);
        more = (long long)ticket * kWavesPerWg < n_bricks;  // workgroup-uniform
        const int brick = more ? (int)ticket * kWavesPerWg + wave : n_bricks;

        if (brick < n_bricks) {
            if (!(ablate & 256)) __builtin_amdgcn_s_setprio(3);  // the chain waits for the slowest classify: run it ahead of other workgroups' emit phases
            // ---- 1. classify --------------------------------------------------------------
            const int segx = brick % nsegx;
            int t = brick / nsegx;
            const int by = t % sp.nby;
            t /= sp.nby;
            const int bz = t % sp.nbz;
            const int v = t / sp.nbz;
            const int gx = segx * 64 + lane;                  // cell / sample x of this lane
            const int gxc = gx < sp.nx + 1 ? gx : sp.nx + 1;  // clamp loads inside the volume
            int xe = segx * 64 + 64;                          // the 65th column
            xe = xe < sp.nx + 1 ? xe : sp.nx + 1;
            const float *brick_base = sp.base + v * sp.sv + (8ll * by) * sp.sy + (8ll * bz) * sp.sz;
            uint32_t cnt = (ablate & 128) ? 0u : classify_brick_column(sp, s_trinum, brick_base, gx, gxc, xe, lane);
            cnt += __shfl_xor(cnt, 1);
            cnt += __shfl_xor(cnt, 2);
            cnt += __shfl_xor(cnt, 4);  // every lane of an 8-lane group: its block's count
            const int bx = segx * 8 + (lane >> 3);
            const bool head = (lane & 7) == 0 && bx < sp.nbx;
            // exclusive prefix over the brick's 8 blocks and the brick total
            const uint32_t mine = head ? cnt : 0u;
            uint32_t incl = mine;
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            const uint32_t brick_total = __builtin_amdgcn_readlane(incl, 56);

            // ---- 2. chained scan ----------------------------------------------------------
            u64 base = 0;
            const bool ok = (ablate & 64) ? true : chained_scan(status, brick, brick_total, lane, base);
            if (!ok && lane == 0) __hip_atomic_store(ctrl + kCtrlError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 my_base = base + (incl - mine);
            const int block_id = v * sp.bpv + bx + sp.nbx * (by + sp.nby * bz);
            if (head) {
                offsets[block_id] = (uint32_t)my_base;
                if (ok && cnt != 0u && !(ablate & 32)) {
                    const unsigned w = atomicAdd(&s_nwork, 1u);
                    s_work[w].base = my_base;
                    s_work[w].block = block_id;
                    s_work[w].count = cnt;
                }
            }
            if (brick == n_bricks - 1 && lane == 0) {
                const u64 T = base + brick_total;
                offsets[sp.n_blocks] = (uint32_t)T;
                ctrl[kCtrlTotalLo] = (uint32_t)T;
                ctrl[kCtrlTotalHi] = (uint32_t)(T >> 32);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();

        // ---- 3. emit: the collected non-empty blocks, shared out over the waves ----------------
        const unsigned n_work = s_nwork;
        if (more && n_work < (unsigned)kEmitBatch) continue;  // keep streaming (workgroup-uniform)
        for (;;) {
            unsigned w = 0;
            if (lane == 0) w = atomicAdd(&s_next, 1u);
            w = __builtin_amdgcn_readfirstlane(w);
            if (w >= n_work) break;
            const u64 tri_base = s_work[w].base;
            const int b = s_work[w].block;
            const int budget = (int)s_work[w].count;
            if (tri_base + (u64)budget > capacity) continue;  // counted, not written: the host grows the buffer
            // keep the tile's per-lane address arithmetic inside this loop: hoisted out of the ticket
            // loop it would stay live across the classify phase and spill
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            VTMC_WAVE_SYNC();  // the previous block's LDS reads are done before the tile is overwritten
            load_tile(L->tile, sp, block_origin(sp, b), lane_e);
            VTMC_WAVE_SYNC();
            emit_block_from_tile<FAST>(L, s_vert, (size_t)tri_base, budget, b, out, lane_e, ablate);
        }
        __syncthreads();  // every wave is done with the list before it is reset
        if (threadIdx.x == 0) {
            s_nwork = 0;
            s_next = 0;
        }
    }
}

__global__ void sweep_volume_counts_kernel(const uint32_t *__restrict__ offsets, int bpv, int n_volumes,
                                           uint32_t *__restrict__ volume_counts)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_volumes) {
        const uint32_t t = offsets[(long long)(v + 1) * bpv] - offsets[(long long)v * bpv];
        volume_counts[2 * v] = 3u * t;  // unindexed soup: 3 vertices per triangle (VoxelTerrain.cs:456-459)
        volume_counts[2 * v + 1] = t;
    }
}

long long sweep_bricks(const BlockSpace &sp)
{
    const int nsegx = (sp.nx + 63) / 64;
    const long long n_vol = sp.bpv > 0 ? sp.n_blocks / sp.bpv : 0;
    return n_vol * sp.nbz * sp.nby * nsegx;
}

size_t sweep_scratch_bytes(const BlockSpace &sp)
{
    return (size_t)kCtrlWords * sizeof(uint32_t) + (size_t)sweep_bricks(sp) * sizeof(u64);
}

hipError_t launch_sweep(const BlockSpace &sp, const DeviceTables &tb, void *scratch, uint32_t *offsets,
                        unsigned long long capacity, void *triangles, int n_cus, int n_volumes,
                        uint32_t *volume_counts, const Tuning &tune, hipStream_t stream)
{
    const int nsegx = (sp.nx + 63) / 64;
    const long long n_bricks = sweep_bricks(sp);
    if (n_bricks > 0x7fffffffll) return hipErrorInvalidValue;
    // every polled word (ticket, error flag, status words) starts at zero in every launch
    hipError_t e = hipMemsetAsync(scratch, 0, sweep_scratch_bytes(sp), stream);
    if (e != hipSuccess) return e;
    unsigned *ctrl = (unsigned *)scratch;
    static_assert(kCtrlTicket + 64 * kTicketGroups <= kCtrlWords, "ticket counters must fit the control block");
    u64 *status = (u64 *)(ctrl + kCtrlWords);
    const long long tickets = (n_bricks + kWavesPerWg - 1) / kWavesPerWg;
    const int per_cu = tune.sweep_wgs_per_cu > 0 ? tune.sweep_wgs_per_cu : 3;  // LDS-limited residency: 3 x 49 KB
    long long wgs = (long long)n_cus * per_cu;
    if (wgs > tickets) wgs = tickets;
    if (wgs < 1) wgs = 1;
    if (tune.emit_fast_math)
        hipLaunchKernelGGL((sweep_kernel<true>), dim3((unsigned)wgs), dim3(256), 0, stream, sp, tb, nsegx, (int)n_bricks, status,
                           ctrl, offsets, (u64)capacity, (float *)triangles, tune.emit_ablate);
    else
        hipLaunchKernelGGL((sweep_kernel<false>), dim3((unsigned)wgs), dim3(256), 0, stream, sp, tb, nsegx, (int)n_bricks, status,
                           ctrl, offsets, (u64)capacity, (float *)triangles, tune.emit_ablate);
    if (volume_counts && n_volumes > 0)
        hipLaunchKernelGGL(sweep_volume_counts_kernel, dim3((n_volumes + 255) / 256), dim3(256), 0, stream, offsets, sp.bpv,
                           n_volumes, volume_counts);
    return hipGetLastError();
}

}  // namespace vtmc
