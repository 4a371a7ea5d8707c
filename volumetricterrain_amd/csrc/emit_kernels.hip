// emit_kernels.hip -- fused normals + triangle emit (hand-written gfx950 / CDNA4, wave64).
//
// Replaces Shaders/SampleNormal.compute:23-34 and Shaders/MarchingCube.compute:101-165 of the
// reference (/root/reference/Unity-Project/Assets/): no 18 GB normal lattice is materialised, and
// triangles land at offsets fixed by the scan (canonical order) instead of an atomic append.
#include "mc_device.h"

namespace vtmc {

// ----------------------------------------------------------------------------------------------
// emit_kernel: SampleNormal + MarchingCube fused, one wave per non-empty block.
//   ASSIGN 0: waves stride the active list;  1: each XCD (blockIdx % 8, round-robin dispatch -- a
//             speed heuristic only) sweeps one contiguous eighth of the list, so blocks that share
//             halo rows / 128-byte lines meet in one L2.
//   FAST     : v_rcp / v_rsq (1 ulp) instead of correctly rounded divide / sqrt.
//   WIDE     : 16-byte global stores (staging shifted so LDS and global alignment coincide).
// ----------------------------------------------------------------------------------------------
constexpr int kSlotCap = 640;      // triangle-slot map entries kept before a flush (2 x 320)
constexpr int kTriDwords = 19;     // 76-byte record

struct __attribute__((aligned(16))) EmitLds {
    float tile[1000];
    unsigned short slot[kSlotCap];
    unsigned char cases[512];
    float stage[64 * kTriDwords + 4];
};
static_assert(sizeof(EmitLds) % 16 == 0 && offsetof(EmitLds, stage) % 16 == 0, "stage must stay 16-byte aligned");

// lattice normal of SampleNormal.compute:27-33 at tile index ti (forward differences, normalised)
template <bool FAST>
__device__ __forceinline__ void lattice_normal(const float *tile, int ti, float n[3])
{
    float v = tile[ti];
    float dx = v - tile[ti + 1];
    float dy = v - tile[ti + 10];
    float dz = v - tile[ti + 100];
    float len2 = dx * dx + dy * dy + dz * dz;
    if (FAST) {
        float r = __builtin_amdgcn_rsqf(len2);
        n[0] = dx * r;
        n[1] = dy * r;
        n[2] = dz * r;
    } else {
        float len = __builtin_sqrtf(len2);  // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
        n[0] = dx / len;
        n[1] = dy / len;
        n[2] = dz / len;
    }
}

// One mesh vertex on cube edge e of cell (cx,cy,cz): position (MarchingCube.compute:128-133) and
// normal (SampleNormalTrilinear, MarchingCube.compute:69-99).  A vertex sits on a lattice edge, so
// the 8-point trilinear blend collapses to a 2-point lerp along the edge axis with the weight
// taken from the ROUNDED position (c0 = floor(P), c1 = ceil(P), t = P - c0), exactly as the
// reference derives it; the collapsed terms are exact (u + 0*(u-u)).
template <bool FAST>
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
    const float t = FAST ? __builtin_amdgcn_fmed3f(-va * __builtin_amdgcn_rcpf(vb - va), 0.0f, 1.0f) : (-va) / (vb - va);
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
    lattice_normal<FAST>(tile, c0, n0);
    lattice_normal<FAST>(tile, c1, n1);
    nrm[0] = n0[0] + w * (n1[0] - n0[0]);
    nrm[1] = n0[1] + w * (n1[1] - n0[1]);
    nrm[2] = n0[2] + w * (n1[2] - n0[2]);
}

template <bool FAST, bool WIDE>
__device__ __forceinline__ void emit_flush(EmitLds *L, const u64 *s_vert, int pending, size_t tri_base,
                                           int block_id, float *__restrict__ out, int lane)
{
    VTMC_WAVE_SYNC();
    for (int s0 = 0; s0 < pending; s0 += 64) {
        const int s = s0 + lane;
        const size_t d0 = (tri_base + (size_t)s0) * kTriDwords;  // first global dword of this batch
        const int sh = WIDE ? (int)(d0 & 3) : 0;                 // staging shift = global misalignment
        if (s < pending) {
            const unsigned sc = L->slot[s];
            const int cell = sc & 511u, i = sc >> 9;
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const u64 w = s_vert[L->cases[cell]] >> (12 * i);
            float *rec = L->stage + sh + lane * kTriDwords;
            // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
            edge_vertex<FAST>(L->tile, cx, cy, cz, (unsigned)w & 15u, rec + 0, rec + 9);
            edge_vertex<FAST>(L->tile, cx, cy, cz, (unsigned)(w >> 8) & 15u, rec + 3, rec + 12);
            edge_vertex<FAST>(L->tile, cx, cy, cz, (unsigned)(w >> 4) & 15u, rec + 6, rec + 15);
            rec[18] = __int_as_float(block_id);
        }
        VTMC_WAVE_SYNC();
        const int cnt = pending - s0 < 64 ? pending - s0 : 64;
        const int n_dw = cnt * kTriDwords;
        if (WIDE) {
            float *gal = out + (d0 - sh);  // 16-byte aligned
            const int lo = sh, hi = sh + n_dw;
            for (int q4 = 4 * lane; q4 < hi; q4 += 256) {
                if (q4 >= lo && q4 + 4 <= hi) {
                    *reinterpret_cast<float4 *>(gal + q4) = *reinterpret_cast<const float4 *>(L->stage + q4);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (q4 + k >= lo && q4 + k < hi) gal[q4 + k] = L->stage[q4 + k];
                }
            }
        } else {
            float *dst = out + d0;
            for (int d = lane; d < n_dw; d += 64) dst[d] = L->stage[d];
        }
        VTMC_WAVE_SYNC();
    }
}

template <int ASSIGN, bool FAST, bool WIDE>
__global__ __launch_bounds__(256) void emit_v1_kernel(BlockSpace sp, DeviceTables tb,
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
    int ai, ai_end, ai_step;
    if (ASSIGN == 1) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;  // gridDim % 8 == 0
        ai = (int)((long long)n_active * xcd / 8) + j * kWavesPerWg + wave;
        ai_end = (int)((long long)n_active * (xcd + 1) / 8);
        ai_step = per_xcd * kWavesPerWg;
    } else {
        ai = blockIdx.x * kWavesPerWg + wave;
        ai_end = n_active;
        ai_step = gridDim.x * kWavesPerWg;
    }
    for (; ai < ai_end; ai += ai_step) {
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
                emit_flush<FAST, WIDE>(L, s_vert, n_out, tri_base, b, out, lane);
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
        if (pending > 0) emit_flush<FAST, WIDE>(L, s_vert, pending, tri_base, b, out, lane);
    }
}


// ----------------------------------------------------------------------------------------------
// emit_kernel (v2): same contract as emit_v1_kernel, ~2x fewer wave instructions per block.
//   * the next block's tile is prefetched into registers while the current one is processed
//     (16 loads per lane, scalar base + 32-bit lane offsets hoisted out of the block loop);
//   * pass 1 only classifies (8 unrolled layers) and compacts the ACTIVE cells with one ballot per
//     layer; pass 2 runs the triangle-slot prefix sum over 64 active cells at a time instead of
//     once per mostly-empty layer;
//   * a vertex is computed along its edge axis only (one lerp, one floor/ceil), edge geometry comes
//     from a 12-entry LDS table; FAST uses v_rcp/v_rsq (1 ulp, results within ~5e-7 of the exact
//     path; the north-star bar is 1e-5), !FAST is bit-compatible with the CPU oracle.
// ----------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) EmitLds2 {
    float tile[1000];
    unsigned short acell[512];      // active cells of the block, ascending cell id
    unsigned short slot[kSlotCap];  // triangle slot -> cell | (i << 9)
    unsigned char cases[512];
    float stage[64 * kTriDwords + 4];
};
static_assert(sizeof(EmitLds2) % 16 == 0 && offsetof(EmitLds2, stage) % 16 == 0, "stage must stay 16-byte aligned");

// Edge table entry: offA | offB << 7 | axis << 14 | backwards << 16 | oax << 17 | oay << 18 | oaz << 19
// (corner offsets MarchingCube.compute:46-50, edge endpoints MarchingCube.compute:40-43).
__host__ __device__ constexpr unsigned edge_entry(int e)
{
    const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
    const int eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    const int a = ea[e], b = eb[e];
    const int oax = (0x66 >> a) & 1, oay = (0xCC >> a) & 1, oaz = (0xF0 >> a) & 1;
    const int obx = (0x66 >> b) & 1, oby = (0xCC >> b) & 1, obz = (0xF0 >> b) & 1;
    const int axis = oax != obx ? 0 : (oay != oby ? 1 : 2);
    const int back = (obx + oby + obz) < (oax + oay + oaz) ? 1 : 0;
    return (unsigned)(oax + 10 * oay + 100 * oaz) | ((unsigned)(obx + 10 * oby + 100 * obz) << 7) |
           ((unsigned)axis << 14) | ((unsigned)back << 16) | ((unsigned)oax << 17) | ((unsigned)oay << 18) |
           ((unsigned)oaz << 19);
}

template <bool FAST>
__device__ __forceinline__ void vertex_on_edge(const float *tile, const unsigned *s_edge, int cx, int cy, int cz,
                                               int tcell, unsigned e, float *pos, float *nrm)
{
    const unsigned inf = s_edge[e];
    const int ta = tcell + (int)(inf & 127u), tb = tcell + (int)((inf >> 7) & 127u);
    const float va = tile[ta], vb = tile[tb];
    // t lies in [0,1] (the endpoints differ in sign class); the 1-ulp v_rcp can land a hair outside,
    // which would push floor/ceil one lattice point beyond the edge -- clamp it back (v_med3_f32).
    const float t = FAST ? __builtin_amdgcn_fmed3f(-va * __builtin_amdgcn_rcpf(vb - va), 0.0f, 1.0f) : (-va) / (vb - va);
    const unsigned axis = (inf >> 14) & 3u;
    const int iax = cx + (int)((inf >> 17) & 1u), iay = cy + (int)((inf >> 18) & 1u), iaz = cz + (int)((inf >> 19) & 1u);
    const int iak = axis == 0 ? iax : (axis == 1 ? iay : iaz);
    const int sk = axis == 0 ? 1 : (axis == 1 ? 10 : 100);
    // lerp(u, v, t) = u + t*(v-u) with v-u = +-1 exactly on the edge axis and 0 on the others
    const float q = (float)iak + ((inf >> 16) & 1u ? -t : t);
    const float fq = floorf(q);
    const float w = q - fq;  // weight from the ROUNDED position (MarchingCube.compute:71-72)
    const int l0 = ta + ((int)fq - iak) * sk, l1 = ta + ((int)ceilf(q) - iak) * sk;
    pos[0] = axis == 0 ? q : (float)iax;
    pos[1] = axis == 1 ? q : (float)iay;
    pos[2] = axis == 2 ? q : (float)iaz;
    float n0[3], n1[3];
    lattice_normal<FAST>(tile, l0, n0);
    lattice_normal<FAST>(tile, l1, n1);
    nrm[0] = n0[0] + w * (n1[0] - n0[0]);
    nrm[1] = n0[1] + w * (n1[1] - n0[1]);
    nrm[2] = n0[2] + w * (n1[2] - n0[2]);
}

template <bool FAST>
__device__ __forceinline__ void emit_flush2(EmitLds2 *L, const u64 *s_vert, const unsigned *s_edge, int pending,
                                            size_t tri_base, int block_id, float *__restrict__ out, int lane, int ablate)
{
    VTMC_WAVE_SYNC();
    for (int s0 = 0; s0 < pending; s0 += 64) {
        const int s = s0 + lane;
        const size_t d0 = (tri_base + (size_t)s0) * kTriDwords;  // first global dword of this batch
        const int sh = (int)(d0 & 3);                            // staging shift = global misalignment
        if (s < pending && !(ablate & 4)) {
            const unsigned sc = L->slot[s];
            const int cell = sc & 511u, i = sc >> 9;
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const int tcell = cx + 10 * cy + 100 * cz;
            const unsigned w = (unsigned)(s_vert[L->cases[cell]] >> (12 * i));
            float *rec = L->stage + sh + lane * kTriDwords;
            // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
            vertex_on_edge<FAST>(L->tile, s_edge, cx, cy, cz, tcell, w & 15u, rec + 0, rec + 9);
            vertex_on_edge<FAST>(L->tile, s_edge, cx, cy, cz, tcell, (w >> 8) & 15u, rec + 3, rec + 12);
            vertex_on_edge<FAST>(L->tile, s_edge, cx, cy, cz, tcell, (w >> 4) & 15u, rec + 6, rec + 15);
            rec[18] = __int_as_float(block_id);
        }
        VTMC_WAVE_SYNC();
        const int cnt = pending - s0 < 64 ? pending - s0 : 64;
        const int lo = sh, hi = sh + cnt * kTriDwords;
        float *gal = out + (d0 - sh);  // 16-byte aligned
        typedef float v4f __attribute__((ext_vector_type(4)));
        // body: whole 16-byte quads, no per-element predicates (write-once stream: non-temporal)
        const int body_lo = (lo + 3) & ~3, body_hi = hi & ~3;
        if (!(ablate & 1))
        if (ablate & 8) {  // experiment: write-through sc1 stores (line dropped from L2)
            typedef int v4i __attribute__((ext_vector_type(4)));
            const unsigned long long gaddr = (unsigned long long)gal;
            const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));
            float *gu = reinterpret_cast<float *>(((unsigned long long)ghi << 32) | glo);
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(gu, 0, (64 * kTriDwords + 4) * 4, 0x00020000);
            if (ablate & 16) {
                for (int q4 = body_lo + 4 * lane; q4 < body_hi; q4 += 256)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4i *>(L->stage + q4), rsrc, q4 * 4, 0, 17);
            } else {
                for (int q4 = body_lo + 4 * lane; q4 < body_hi; q4 += 256)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4i *>(L->stage + q4), rsrc, q4 * 4, 0, 16);
            }
        } else {
            for (int q4 = body_lo + 4 * lane; q4 < body_hi; q4 += 256)
                __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(L->stage + q4), reinterpret_cast<v4f *>(gal + q4));
        }
        // head (< 4 dwords before the first whole quad) and tail (< 4 after the last): lanes 0-3 / 4-7
        {
            const int k = lane & 3;
            const int idx = lane < 4 ? lo + k : body_hi + k;
            const bool on = lane < 4 ? (idx < body_lo && idx < hi) : (lane < 8 && idx < hi && idx >= body_lo);
            if (on) __builtin_nontemporal_store(L->stage[idx], gal + idx);
        }
        VTMC_WAVE_SYNC();
    }
}

template <bool FAST>
__global__ __launch_bounds__(256) void emit_kernel(BlockSpace sp, DeviceTables tb,
                                                    const uint32_t *__restrict__ offsets,
                                                    const int32_t *__restrict__ active_list,
                                                    const uint32_t *__restrict__ totals, uint32_t capacity,
                                                    float *__restrict__ out, int group_log2, int ablate, unsigned *__restrict__ queue, int sub_log2)
{
    __shared__ EmitLds2 s_lds[kWavesPerWg];
    __shared__ u64 s_vert[256];
    __shared__ unsigned char s_trinum[256];
    __shared__ unsigned s_edge[16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s_vert[threadIdx.x] = tb.vert_packed[threadIdx.x];
    s_trinum[threadIdx.x] = tb.tri_num[threadIdx.x];
    if (threadIdx.x < 12) {
        constexpr unsigned tbl[12] = {edge_entry(0), edge_entry(1), edge_entry(2),  edge_entry(3),
                                      edge_entry(4), edge_entry(5), edge_entry(6),  edge_entry(7),
                                      edge_entry(8), edge_entry(9), edge_entry(10), edge_entry(11)};
        unsigned v = tbl[0];
#pragma unroll
        for (int k = 1; k < 12; ++k) v = threadIdx.x == (unsigned)k ? tbl[k] : v;
        s_edge[threadIdx.x] = v;
    }
#ifdef VTMC_DEBUG_POISON_LDS  // diagnostic build: NaN-fill LDS so any read of a never-written word shows up in the output
    for (unsigned i = threadIdx.x; i < sizeof(s_lds) / 4; i += 256) reinterpret_cast<unsigned *>(s_lds)[i] = 0x7FC00000u;
#endif
    __syncthreads();

    const uint32_t total_tris = totals[0];
    const int n_active = (int)totals[1];
    if (total_tris > capacity) return;  // host grows the buffer and re-launches (vtmc_api.hip)

    EmitLds2 *L = &s_lds[wave];
    const int t0 = (lane & 7) + 10 * (lane >> 3);

    // loop-invariant byte offsets of this lane's 16 tile samples relative to the block origin, and
    // their LDS destinations (lane index walks the stride-1 axis)
    unsigned toff[16];
    int tdst[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        int idx = it * 64 + lane;
        idx = idx < 1000 ? idx : 999;
        const int a = idx % 10, t = idx / 10;
        const int m = t % 10, c = t / 10;
        const int ix = sp.zfast ? c : a, iz = sp.zfast ? a : c;
        toff[it] = (unsigned)(ix * sp.sx + m * sp.sy + iz * sp.sz) * 4u;
        tdst[it] = ix + 10 * m + 100 * iz;
    }

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

    float pre[16];
    int b_next = 0;
    request();
    int ai = collect();
    request();
    int ai_next = collect();
    if (ai < ai_end) {
        b_next = active_list[ai];
        const char *src = reinterpret_cast<const char *>(sp.base + block_origin(sp, b_next));
#pragma unroll
        for (int it = 0; it < 16; ++it) pre[it] = *reinterpret_cast<const float *>(src + toff[it]);
    }
    for (int k = 0; ai < ai_end; ++k) {
        const int b = b_next;
        size_t tri_base = offsets[b];
        // the scan's budget for this block; flushes are clamped to it so a classify/emit mismatch
        // could never write outside the block's own slice of the triangle buffer
        int budget = (int)(offsets[b + 1] - offsets[b]);
        VTMC_WAVE_SYNC();
#pragma unroll
        for (int it = 0; it < 16; ++it) L->tile[tdst[it]] = pre[it];
        if (ai_next < ai_end) {  // prefetch the next block's tile; it lands while this one is processed
            b_next = active_list[(ablate & 2) ? ai_begin + (k & 3) : ai_next];
            const char *src = reinterpret_cast<const char *>(sp.base + block_origin(sp, b_next));
#pragma unroll
            for (int it = 0; it < 16; ++it) pre[it] = *reinterpret_cast<const float *>(src + toff[it]);
        }
        request();  // ticket for the block after next; collected at the bottom of this iteration
        VTMC_WAVE_SYNC();

        // pass 1: cases (CollectTriNum.compute:48-51) + compaction of the cells that hold triangles
        int n_act = 0;
        unsigned lo = layer_nibble(L->tile, t0, 0);
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            const unsigned hi = layer_nibble(L->tile, t0, z + 1);
            const unsigned cs = lo | (hi << 4);
            lo = hi;
            const int cell = 64 * z + lane;
            L->cases[cell] = (unsigned char)cs;
            const bool act = ((cs + 1u) & 0xFFu) > 1u;  // neither 0x00 nor 0xFF
            const u64 m = __builtin_amdgcn_ballot_w64(act);
            if (act) L->acell[n_act + (int)lanes_below(m)] = (unsigned short)cell;
            n_act += __builtin_popcountll(m);
        }
        VTMC_WAVE_SYNC();

        // pass 2: triangle slots, 64 active cells per step
        int pending = 0;
        for (int c0 = 0; c0 < n_act; c0 += 64) {
            if (pending > kSlotCap - 320) {  // wave-uniform
                const int n_out = pending < budget ? pending : budget;
                emit_flush2<FAST>(L, s_vert, s_edge, n_out, tri_base, b, out, lane, ablate);
                tri_base += n_out;
                budget -= n_out;
                pending = 0;
            }
            const int idx = c0 + lane;
            const bool valid = idx < n_act;
            const unsigned cell = valid ? L->acell[idx] : 0u;
            const unsigned n = valid ? s_trinum[L->cases[cell]] : 0u;
            unsigned step_total;
            const unsigned pre_n = wave_prefix3(n, step_total);
            unsigned short *dst = L->slot + pending + pre_n;
#pragma unroll
            for (unsigned i = 0; i < 5; ++i)
                if (i < n) dst[i] = (unsigned short)(cell | (i << 9));
            pending += (int)step_total;
        }
        if (pending > budget) pending = budget;
        if (pending > 0) emit_flush2<FAST>(L, s_vert, s_edge, pending, tri_base, b, out, lane, ablate);
        ai = ai_next;
        ai_next = collect();
    }
}

hipError_t launch_emit(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets,
                       const int32_t *active_list, const uint32_t *totals, uint32_t capacity,
                       void *triangles, int n_cus, const Tuning &tune, unsigned *queue, hipStream_t stream)
{
    int per_cu = tune.emit_wgs_per_cu > 0 ? tune.emit_wgs_per_cu : 3;  // LDS-limited residency: 3 x 48 KB
    int wgs = n_cus * per_cu;
    wgs = (wgs + 7) & ~7;  // the XCD sweep needs a multiple of 8
    dim3 g(wgs), blk(256);
    float *o = (float *)triangles;
    if (tune.emit_version == 1) {
        if (tune.emit_fast_math)
            hipLaunchKernelGGL((emit_v1_kernel<1, true, true>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o);
        else
            hipLaunchKernelGGL((emit_v1_kernel<1, false, true>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o);
    } else {
        if (tune.emit_fast_math)
            hipLaunchKernelGGL((emit_kernel<true>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o, tune.emit_group_log2, tune.emit_ablate, tune.emit_dynamic ? queue : nullptr, tune.emit_sub_log2);
        else
            hipLaunchKernelGGL((emit_kernel<false>), g, blk, 0, stream, sp, tb, offsets, active_list, totals, capacity, o, tune.emit_group_log2, tune.emit_ablate, tune.emit_dynamic ? queue : nullptr, tune.emit_sub_log2);
    }
    return hipGetLastError();
}

}  // namespace vtmc
