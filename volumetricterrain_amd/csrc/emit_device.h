// emit_device.h -- device code of the kernels that emit triangles (emit_kernels.hip: one wave per
// non-empty block after the scan): lattice normals
// (Shaders/SampleNormal.compute:27-33), edge vertices and the trilinear normal fetch
// (Shaders/MarchingCube.compute:69-99, 128-133), triangle records (MarchingCube.compute:143-162).
#ifndef VTMC_EMIT_DEVICE_H
#define VTMC_EMIT_DEVICE_H
#include "mc_device.h"

namespace vtmc {

constexpr int kTriDwords = 19;     // 76-byte record

// ----------------------------------------------------------------------------------------------
// Per-block emit:
//   * pass 1 classifies (8 unrolled layers) and compacts the ACTIVE cells with one ballot per layer;
//   * pass 2 runs the triangle-slot prefix sum over 64 active cells at a time; a slot carries the
//     cell and the triangle's three edge ids, read from the packed table once per CELL;
//   * the flush gives one lane per triangle: all LDS reads of its three vertices are issued before
//     any staging write (three independent chains in flight instead of one), edge geometry comes
//     from a 60-bit constant, a vertex is computed along its edge axis only (one lerp, one
//     floor/ceil); FAST uses v_rcp / v_rsq / fma (results within ~5e-7 of the exact path; the
//     north-star bar is 1e-5), !FAST is bit-compatible with the CPU oracle.
// ----------------------------------------------------------------------------------------------
constexpr int kSlotCap = 384;  // triangle slots kept before a flush (a step adds at most 320)
constexpr int kStageTris = 33; // records staged at a time: half a batch + 1 -- 9.6 KB of LDS per wave, four workgroups per CU

struct __attribute__((aligned(16))) EmitLds2 {
    float tile[1000];
    unsigned slot[kSlotCap];        // triangle slot -> cell | edge triple << 9
    unsigned short acell[512];      // active cells of the block, ascending cell id
    unsigned char cases[512];
    float stage[kStageTris * kTriDwords + 4];
};
static_assert(sizeof(EmitLds2) % 16 == 0 && offsetof(EmitLds2, stage) % 16 == 0, "stage must stay 16-byte aligned");

// Edge geometry, 5 bits per edge: offset of endpoint a (x | y << 1 | z << 2) | axis << 3
// (corner offsets MarchingCube.compute:46-50, edge endpoints MarchingCube.compute:40-43).
__host__ __device__ constexpr u64 edge_geom_word()
{
    const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
    const int eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    u64 w = 0;
    for (int e = 0; e < 12; ++e) {
        const int a = ea[e], b = eb[e];
        const int oax = (0x66 >> a) & 1, oay = (0xCC >> a) & 1, oaz = (0xF0 >> a) & 1;
        const int obx = (0x66 >> b) & 1, oby = (0xCC >> b) & 1;
        const int axis = oax != obx ? 0 : (oay != oby ? 1 : 2);
        w |= (u64)(oax | (oay << 1) | (oaz << 2) | (axis << 3)) << (5 * e);
    }
    return w;
}
constexpr u64 kEdgeGeom = edge_geom_word();

// Where a vertex on cube edge e of cell (cx,cy,cz) lives: tile indices of the edge's endpoints a, b
// (in the reference's order), the edge axis and endpoint a's integer coordinates.
struct EdgeSite {
    int ta, tb, sk, iak;
    unsigned axis, back;
    int ia[3];
};

__device__ __forceinline__ EdgeSite edge_site(int cx, int cy, int cz, int tcell, unsigned e)
{
    EdgeSite s;
    const unsigned g = (unsigned)(kEdgeGeom >> (5u * e)) & 31u;
    const unsigned oax = g & 1u, oay = (g >> 1) & 1u, oaz = (g >> 2) & 1u;
    s.axis = g >> 3;
    s.sk = s.axis == 0 ? 1 : (s.axis == 1 ? 10 : 100);
    s.back = s.axis == 0 ? oax : (s.axis == 1 ? oay : oaz);  // a sits on the far end: the edge runs towards -axis
    s.ta = tcell + (int)oax + 10 * (int)oay + 100 * (int)oaz;
    s.tb = s.back ? s.ta - s.sk : s.ta + s.sk;
    s.ia[0] = cx + (int)oax;
    s.ia[1] = cy + (int)oay;
    s.ia[2] = cz + (int)oaz;
    s.iak = s.axis == 0 ? s.ia[0] : (s.axis == 1 ? s.ia[1] : s.ia[2]);
    return s;
}

// gradient of SampleNormal.compute:27-30 at tile index ti, not yet normalised
__device__ __forceinline__ void lattice_gradient(const float *tile, int ti, float d[3])
{
    const float v = tile[ti];
    d[0] = v - tile[ti + 1];
    d[1] = v - tile[ti + 10];
    d[2] = v - tile[ti + 100];
}

template <bool FAST>
__device__ __forceinline__ void normalise(float d[3])
{
    if (FAST) {
        const float len2 = __builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], d[0] * d[0]));
        const float r = __builtin_amdgcn_rsqf(len2);
        d[0] *= r;
        d[1] *= r;
        d[2] *= r;
    } else {
        const float len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float len = __builtin_sqrtf(len2);  // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
        d[0] /= len;
        d[1] /= len;
        d[2] /= len;
    }
}

// Streams the staged dwords [lo, hi) (stream coordinates: stage[i] goes to gal[i], gal 16-byte aligned)
// with 16-byte non-temporal stores; the < 4 dwords before the first whole quad and after the last go
// out as single dwords from lanes 0-3 / 4-7.
__device__ __forceinline__ void stream_out_range(const float *stage, float *__restrict__ gal, int lo, int hi, int lane, int ablate)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int body_lo = (lo + 3) & ~3, body_hi = hi & ~3;
    if (!(ablate & 1)) {
        for (int q4 = body_lo + 4 * lane; q4 < body_hi; q4 += 256) {
            const v4f v = *reinterpret_cast<const v4f *>(stage + q4);
            v4f *p = reinterpret_cast<v4f *>(gal + q4);
            if (ablate & 16) __builtin_nontemporal_store(v, p);   // diagnostics: streaming hint (4 % slower at four workgroups per CU)
            else *p = v;
        }
    }
    const int k = lane & 3;
    const int idx = lane < 4 ? lo + k : body_hi + k;
    const bool on = lane < 4 ? (idx < body_lo && idx < hi) : (lane < 8 && idx < hi && idx >= body_lo);
    if (on) __builtin_nontemporal_store(stage[idx], gal + idx);
}

// One lane per triangle.  Vertex = position along the edge (MarchingCube.compute:128-133: t =
// -cube[a] / (cube[b] - cube[a]), lerp with v-u = +-1 on the edge axis and 0 on the others) and the
// trilinear normal fetch of MarchingCube.compute:69-99, which on a lattice edge is a 2-point lerp
// whose weight comes from the ROUNDED position (c0 = floor(P), c1 = ceil(P), t = P - c0).
template <bool FAST>
__device__ __forceinline__ void emit_flush2(EmitLds2 *L, int pending, size_t tri_base, int block_id,
                                            float *__restrict__ out, int lane, int ablate)
{
    VTMC_WAVE_SYNC();
    for (int s0 = 0; s0 < pending; s0 += 64) {
        const int s = s0 + lane;
        const size_t d0 = (tri_base + (size_t)s0) * kTriDwords;  // first global dword of this batch
        const int sh = (int)(d0 & 3);                            // staging shift = global misalignment
        float rec[18];
#pragma unroll
        for (int c = 0; c < 18; ++c) rec[c] = 0.f;
        if (s < pending && !(ablate & 4)) {
            const unsigned sc = L->slot[s];
            const int cell = sc & 511u;
            const unsigned trip = sc >> 9;
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const int tcell = cx + 10 * cy + 100 * cz;
            // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
            const unsigned e[3] = {trip & 15u, (trip >> 8) & 15u, (trip >> 4) & 15u};
            const float *tile = L->tile;
            EdgeSite es[3];
            float va[3], vb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                es[k] = edge_site(cx, cy, cz, tcell, e[k]);
                va[k] = tile[es[k].ta];
                vb[k] = tile[es[k].tb];
            }
            float q[3], w[3], g0[3][3], g1[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // t lies in [0,1] (the endpoints differ in sign class); the 1-ulp v_rcp can land a hair
                // outside, which would push floor/ceil one lattice point beyond the edge -- clamp it back
                const float t = FAST ? __builtin_amdgcn_fmed3f(-va[k] * __builtin_amdgcn_rcpf(vb[k] - va[k]), 0.0f, 1.0f)
                                     : (-va[k]) / (vb[k] - va[k]);
                q[k] = (float)es[k].iak + (es[k].back ? -t : t);
                const float fq = floorf(q[k]);
                w[k] = q[k] - fq;
                const int l0 = es[k].ta + ((int)fq - es[k].iak) * es[k].sk;
                const int l1 = es[k].ta + ((int)ceilf(q[k]) - es[k].iak) * es[k].sk;
                lattice_gradient(tile, l0, g0[k]);
                lattice_gradient(tile, l1, g1[k]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                normalise<FAST>(g0[k]);
                normalise<FAST>(g1[k]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    rec[3 * k + c] = es[k].axis == (unsigned)c ? q[k] : (float)es[k].ia[c];
                    rec[9 + 3 * k + c] = FAST ? __builtin_fmaf(w[k], g1[k][c] - g0[k][c], g0[k][c])
                                              : g0[k][c] + w[k] * (g1[k][c] - g0[k][c]);
                }
            }
        }
        // The 64 records of the batch leave through a staging area of 33, in two rounds.  In stream
        // coordinates (dwords from the 16-byte aligned address below the batch) record r sits at
        // sh + 19 r; round 0 stages records 0..32 and stores every whole quad below 608 + (sh ? 4 : 0),
        // round 1 stages records 32..63 (record 32 again, so the quad that straddles the two halves left
        // complete in round 0) and continues from that quad boundary: no partial stores in mid-batch.
        const int cnt = pending - s0 < 64 ? pending - s0 : 64;
        float *gal = out + (d0 - sh);  // 16-byte aligned
        const int split = 32 * kTriDwords + (sh ? 4 : 0);   // multiple of 4
        const int end = sh + cnt * kTriDwords;
        for (int h = 0; h == 0 || cnt > 32 * h; ++h) {   // wave-uniform, at most two rounds
            VTMC_WAVE_SYNC();
            const int r = lane - 32 * h;   // record slot in the staging area
            if (r >= 0 && r < kStageTris && s < pending && !(ablate & 4)) {
                float *dstrec = L->stage + sh + r * kTriDwords;
#pragma unroll
                for (int c = 0; c < 18; ++c) dstrec[c] = rec[c];
                dstrec[18] = __int_as_float(block_id);
            }
            VTMC_WAVE_SYNC();
            // stream coordinates of this round, and the same relative to the staging area (which starts at 608 h)
            const int lo = h == 0 ? sh : split, hi = (h == 0 && cnt > 32) ? split : end;
            const int base = 32 * kTriDwords * h;
            stream_out_range(L->stage - base, gal, lo, hi, lane, ablate);
        }
        VTMC_WAVE_SYNC();
    }
}

// Passes 1 + 2 + flush of one block whose 10^3 tile is already in L->tile: cases
// (CollectTriNum.compute:48-51), compaction of the cells that hold triangles, triangle slots, then
// one lane per triangle.  `budget` = the scan's triangle count of the block; flushes are clamped to
// it so a classify/emit mismatch could never write outside the block's own slice of the buffer.
// `rowmask` (bits 0-7: y layers, 8-15: z layers that hold a cell with triangles; 0xFFFF = unknown):
// only tile rows next to such layers need to be valid, cells outside them are skipped.
template <bool FAST>
__device__ __forceinline__ void emit_block_from_tile(EmitLds2 *L, const u64 *s_vert, size_t tri_base, int budget,
                                                     int block_id, float *__restrict__ out, int lane, int ablate,
                                                     unsigned rowmask = 0xFFFFu)
{
    const int t0 = (lane & 7) + 10 * (lane >> 3);
    const bool y_live = (rowmask >> (lane >> 3)) & 1u;
    // pass 1
    int n_act = 0;
    unsigned lo = layer_nibble(L->tile, t0, 0);
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        const unsigned hi = layer_nibble(L->tile, t0, z + 1);
        const unsigned cs = lo | (hi << 4);
        lo = hi;
        const int cell = 64 * z + lane;
        L->cases[cell] = (unsigned char)cs;
        const bool act = ((cs + 1u) & 0xFFu) > 1u && y_live && ((rowmask >> (8 + z)) & 1u);  // neither 0x00 nor 0xFF, in a live layer
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
            emit_flush2<FAST>(L, n_out, tri_base, block_id, out, lane, ablate);
            tri_base += n_out;
            budget -= n_out;
            pending = 0;
        }
        const int idx = c0 + lane;
        const bool valid = idx < n_act;
        const unsigned cell = valid ? L->acell[idx] : 0u;
        const u64 vw = valid ? s_vert[L->cases[cell]] : 0ull;  // fifteen 4-bit edge ids + the count in the top nibble
        const unsigned n = (unsigned)(vw >> 60);
        unsigned step_total;
        const unsigned pre_n = wave_prefix3(n, step_total);
        unsigned *dst = L->slot + pending + pre_n;
#pragma unroll
        for (unsigned i = 0; i < 5; ++i)
            if (i < n) dst[i] = cell | (((unsigned)(vw >> (12 * i)) & 0xFFFu) << 9);
        pending += (int)step_total;
    }
    if (pending > budget) pending = budget;
    if (pending > 0) emit_flush2<FAST>(L, pending, tri_base, block_id, out, lane, ablate);
}


// ----------------------------------------------------------------------------------------------
// Indexed (welded) output of one block -- new in the build (the reference welds later, on the CPU,
// with Mesh.Optimize(), VoxelTerrain.cs:460).  A mesh vertex lives on a lattice edge with a sign
// change; all cells around that edge share it:
//   vertices : one 24-byte record {position, normal} per such edge of the block's 9^3 lattice,
//              ordered by lattice point p = x + 9y + 81z, then axis x, y, z;
//   indices  : three block-local int32 per triangle, canonical triangle order (as the soup).
// The vertex is evaluated from the edge's LOW endpoint (the orientation the reference uses for cube
// edges 0, 1, 4, 5, 8..11); where the reference walks an edge backwards (edges 2, 3, 6, 7) its
// t' = 1 - t differs from this one by rounding only (<= ~1e-6 in cell units, bar 1e-5).
// ----------------------------------------------------------------------------------------------
constexpr int kVertDwords = 6;  // 24-byte vertex record
constexpr int kVlistCap = 448;  // queued vertices before a flush (a 64-point step adds at most 192)

struct __attribute__((aligned(16))) EmitLdsIdx {
    float tile[1000];
    unsigned slot[kSlotCap];        // triangle slot -> cell | edge triple << 9
    unsigned short acell[512];      // active cells of the block, ascending cell id
    unsigned char cases[512];
    unsigned short vmap[736];       // lattice point -> first vertex id | axis flags << 12
    unsigned short vlist[kVlistCap];  // vertices waiting to be evaluated: point << 2 | axis
    float stage[64 * kVertDwords + 4];
};
static_assert(sizeof(EmitLdsIdx) % 16 == 0 && offsetof(EmitLdsIdx, stage) % 16 == 0, "stage must stay 16-byte aligned");

// streams `n_dw` staged dwords (L->stage + sh ...) to out + d0 with 16-byte stores; sh = d0 & 3
__device__ __forceinline__ void stream_out_staged(const float *stage, float *__restrict__ out, size_t d0, int sh, int n_dw, int lane)
{
    const int lo = sh, hi = sh + n_dw;
    float *gal = out + (d0 - sh);  // 16-byte aligned
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int body_lo = (lo + 3) & ~3, body_hi = hi & ~3;
    for (int q4 = body_lo + 4 * lane; q4 < body_hi; q4 += 256)
        __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(stage + q4), reinterpret_cast<v4f *>(gal + q4));
    const int k = lane & 3;
    const int idx = lane < 4 ? lo + k : body_hi + k;
    const bool on = lane < 4 ? (idx < body_lo && idx < hi) : (lane < 8 && idx < hi && idx >= body_lo);
    if (on) __builtin_nontemporal_store(stage[idx], gal + idx);
}

template <bool FAST>
__device__ __forceinline__ void emit_block_indexed(EmitLdsIdx *L, const u64 *s_vert, size_t tri_base, int tri_budget,
                                                   size_t vert_base, int vert_budget, float *__restrict__ out_vertices,
                                                   int *__restrict__ out_indices, int lane, int ablate = 0)
{
    const float *tile = L->tile;
    // pass 1: cases + compaction of the active cells (as the soup path)
    const int t0 = (lane & 7) + 10 * (lane >> 3);
    int n_act = 0;
    unsigned lo = layer_nibble(tile, t0, 0);
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        const unsigned hi = layer_nibble(tile, t0, z + 1);
        const unsigned cs = lo | (hi << 4);
        lo = hi;
        const int cell = 64 * z + lane;
        L->cases[cell] = (unsigned char)cs;
        const bool act = ((cs + 1u) & 0xFFu) > 1u;
        const u64 m = __builtin_amdgcn_ballot_w64(act);
        if (act) L->acell[n_act + (int)lanes_below(m)] = (unsigned short)cell;
        n_act += __builtin_popcountll(m);
    }

    // vertices: number the sign-change edges 64 lattice points at a time; their (point, axis) entries
    // queue up in vlist and are evaluated 64 per step, one lane per vertex
    int vrun = 0;    // vertices numbered so far
    int vdone = 0;   // vertices already written out
    auto flush_vertices = [&]() {
        VTMC_WAVE_SYNC();
        int n_v = (ablate & 16) ? 0 : vrun - vdone;
        if (vdone + n_v > vert_budget) n_v = vert_budget - vdone > 0 ? vert_budget - vdone : 0;  // never outside the block's slice
        for (int s0 = 0; s0 < n_v; s0 += 64) {
            const int s = s0 + lane;
            const size_t d0 = (vert_base + (size_t)(vdone + s0)) * kVertDwords;
            const int sh = (int)(d0 & 3);
            if (s < n_v) {
                const unsigned ent = L->vlist[s];
                const unsigned axis = ent & 3u;
                const int pp = (int)(ent >> 2);
                const int c[3] = {pp % 9, (pp / 9) % 9, pp / 81};
                const int sk = axis == 0 ? 1 : (axis == 1 ? 10 : 100);
                const int tl = c[0] + 10 * c[1] + 100 * c[2];
                const float va = tile[tl], vb = tile[tl + sk];
                const float t = FAST ? __builtin_amdgcn_fmed3f(-va * __builtin_amdgcn_rcpf(vb - va), 0.0f, 1.0f) : (-va) / (vb - va);
                const int ck = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
                const float q = (float)ck + t;
                const float fq = floorf(q);
                const float w = q - fq;
                float g0[3], g1[3];
                lattice_gradient(tile, tl + ((int)fq - ck) * sk, g0);
                lattice_gradient(tile, tl + ((int)ceilf(q) - ck) * sk, g1);
                normalise<FAST>(g0);
                normalise<FAST>(g1);
                float *rec = L->stage + sh + lane * kVertDwords;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    rec[a] = axis == (unsigned)a ? q : (float)c[a];
                    rec[3 + a] = FAST ? __builtin_fmaf(w, g1[a] - g0[a], g0[a]) : g0[a] + w * (g1[a] - g0[a]);
                }
            }
            VTMC_WAVE_SYNC();
            const int cnt = n_v - s0 < 64 ? n_v - s0 : 64;
            stream_out_staged(L->stage, out_vertices, d0, sh, cnt * kVertDwords, lane);
            VTMC_WAVE_SYNC();
        }
        vdone = vrun;
    };
    for (int p0 = 0; p0 < ((ablate & 64) ? 0 : 729); p0 += 64) {
        if (vrun - vdone > kVlistCap - 192) flush_vertices();  // wave-uniform: the next step may add 192
        const int p = p0 + lane;
        unsigned flags = 0;
        if (p < 729) {
            const int x = p % 9, y = (p / 9) % 9, z = p / 81;
            const float *q = tile + x + 10 * y + 100 * z;
            const bool s0 = q[0] > 0.f;
            flags = (unsigned)(x < 8 && s0 != (q[1] > 0.f)) | ((unsigned)(y < 8 && s0 != (q[10] > 0.f)) << 1) |
                    ((unsigned)(z < 8 && s0 != (q[100] > 0.f)) << 2);
        }
        unsigned step_total;
        const unsigned pre = wave_prefix3((unsigned)__builtin_popcount(flags), step_total);
        if (p < 729) L->vmap[p] = (unsigned short)((unsigned)(vrun + (int)pre) | (flags << 12));
        unsigned k = (unsigned)(vrun - vdone) + pre;
#pragma unroll
        for (unsigned a = 0; a < 3; ++a)
            if (flags & (1u << a)) L->vlist[k++] = (unsigned short)(((unsigned)p << 2) | a);
        vrun += (int)step_total;
    }
    flush_vertices();

    // pass 2 + index flush: triangle slots, 64 active cells per step
    auto flush = [&](int pending, size_t base) {
        VTMC_WAVE_SYNC();
        if (ablate & 32) pending = 0;
        for (int s0 = 0; s0 < pending; s0 += 64) {
            const int s = s0 + lane;
            const size_t d0 = (base + (size_t)s0) * 3;
            const int sh = (int)(d0 & 3);
            if (s < pending) {
                const unsigned sc = L->slot[s];
                const int cell = sc & 511u;
                const unsigned trip = sc >> 9;
                const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
                const unsigned e[3] = {trip & 15u, (trip >> 8) & 15u, (trip >> 4) & 15u};  // winding swap, MarchingCube.compute:147-157
                int *rec = reinterpret_cast<int *>(L->stage) + sh + lane * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned g = (unsigned)(kEdgeGeom >> (5u * e[k])) & 31u;
                    const unsigned axis = g >> 3;
                    int lx = cx + (int)(g & 1u), ly = cy + (int)((g >> 1) & 1u), lz = cz + (int)((g >> 2) & 1u);
                    // endpoint a on the far end: the lattice edge starts one step back along the axis
                    lx -= axis == 0 ? (int)(g & 1u) : 0;
                    ly -= axis == 1 ? (int)((g >> 1) & 1u) : 0;
                    lz -= axis == 2 ? (int)((g >> 2) & 1u) : 0;
                    const unsigned vm = L->vmap[lx + 9 * ly + 81 * lz];
                    rec[k] = (int)((vm & 0xFFFu) + (unsigned)__builtin_popcount((vm >> 12) & ((1u << axis) - 1u)));
                }
            }
            VTMC_WAVE_SYNC();
            const int cnt = pending - s0 < 64 ? pending - s0 : 64;
            stream_out_staged(L->stage, reinterpret_cast<float *>(out_indices), d0, sh, cnt * 3, lane);
            VTMC_WAVE_SYNC();
        }
    };
    int pending = 0;
    for (int c0 = 0; c0 < n_act; c0 += 64) {
        if (pending > kSlotCap - 320) {  // wave-uniform
            const int n_out = pending < tri_budget ? pending : tri_budget;
            flush(n_out, tri_base);
            tri_base += n_out;
            tri_budget -= n_out;
            pending = 0;
        }
        const int idx = c0 + lane;
        const bool valid = idx < n_act;
        const unsigned cell = valid ? L->acell[idx] : 0u;
        const u64 vw = valid ? s_vert[L->cases[cell]] : 0ull;
        const unsigned n = (unsigned)(vw >> 60);
        unsigned step_total;
        const unsigned pre_n = wave_prefix3(n, step_total);
        unsigned *dst = L->slot + pending + pre_n;
#pragma unroll
        for (unsigned i = 0; i < 5; ++i)
            if (i < n) dst[i] = cell | (((unsigned)(vw >> (12 * i)) & 0xFFFu) << 9);
        pending += (int)step_total;
    }
    if (pending > tri_budget) pending = tri_budget;
    if (pending > 0) flush(pending, tri_base);
}

}  // namespace vtmc
#endif
