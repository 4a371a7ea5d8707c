// emit_device.h -- device code of the kernels that emit triangles (emit_kernels.hip: one wave per
// non-empty block after the scan): lattice normals
// (Shaders/SampleNormal.compute:27-33), edge vertices and the trilinear normal fetch
// (Shaders/MarchingCube.compute:69-99, 128-133), triangle records (MarchingCube.compute:143-162).
#ifndef VTMC_EMIT_DEVICE_H
#define VTMC_EMIT_DEVICE_H
#include "mc_device.h"

namespace vtmc {

constexpr int kTriDwords = 19;     // 76-byte record

// Diagnostic build -DVTMC_EMIT_TIMING (tools/emit_phases.py): shader cycles a wave spends in each phase of a block, summed over all waves.
// In the product build the clock is an empty object and every mark compiles to nothing.
#ifdef VTMC_EMIT_TIMING
struct PhaseClock {
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i)
    {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[i] += now - last;
        last = now;
    }
};
#else
struct PhaseClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
};
#endif

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
constexpr int kOnceStageDwords = 64 * kTriDwords + 4;   // vertex-once kernel: a whole batch of 64 records + the alignment shift

struct __attribute__((aligned(16))) EmitLds2 {
    float tile[1000];
    unsigned slot[kSlotCap];        // triangle slot -> cell | edge triple << 9
    unsigned short acell[512];      // active cells of the block, ascending cell id
    unsigned char cases[512];       // cases[i]: the case of cell acell[i] (pass 1 holds it in registers anyway)
    float stage[kStageTris * kTriDwords + 4];
};
static_assert(sizeof(EmitLds2) % 16 == 0 && offsetof(EmitLds2, stage) % 16 == 0, "stage must stay 16-byte aligned");

// What the per-corner path works on: its own LDS block (EmitLds2) or the corresponding areas of the vertex-once kernel's, whose
// blocks with too many vertices / triangles take this path (EmitLdsOnce::corner_view).
struct CornerView {
    float *tile;
    unsigned *slot;
    int slot_cap;            // >= 320 + 1: a step of 64 cells adds at most 320 slots
    unsigned short *acell;
    unsigned char *cases;
    float *stage;            // kStageTris records + 4 dwords, 16-byte aligned, NOT over the tile (the flush reads the tile)
};
__device__ __forceinline__ CornerView corner_view(EmitLds2 *L) { return CornerView{L->tile, L->slot, kSlotCap, L->acell, L->cases, L->stage}; }

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
// `vm_issued` grows by the number of body store INSTRUCTIONS this call issues for certain (pass p is issued exactly when lane 0 has a
// quad, i.e. body_lo + 256 p < body_hi): a lower bound of the wave's vector-memory instructions, which the counted wait of the
// asynchronous tile prefetch needs (emit_kernels.hip, wait_vm_at_most); the two edge stores are left out of it.
template <int MAX_PASSES = 3>   // 256 dwords per pass: a bound known at compile time keeps the loop's bookkeeping out of the scalar unit
__device__ __forceinline__ void stream_out_range(const float *stage, float *__restrict__ gal, int lo, int hi, int lane, int ablate, int &vm_issued,
                                                 int stage_last_quad)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int body_lo = (lo + 3) & ~3, body_hi = hi & ~3;
    // every LDS read of the call is issued before the first store (a read inside its store's conditional is a round trip of its own: three
    // `ds_read_b128 -> s_waitcnt -> global_store` chains per round in round 4's ISA); a lane beyond the range reads the staging area's last quad
    v4f v[MAX_PASSES];
#pragma unroll
    for (int pass = 0; pass < MAX_PASSES; ++pass) {
        const int q4 = body_lo + 4 * lane + 256 * pass;
        v[pass] = *reinterpret_cast<const v4f *>(stage + (q4 < stage_last_quad ? q4 : stage_last_quad));
    }
    const int k = lane & 3;
    const int idx = lane < 4 ? lo + k : body_hi + k;
    const bool on = lane < 4 ? (idx < body_lo && idx < hi) : (lane < 8 && idx < hi && idx >= body_lo);
    const float edge = stage[on ? idx : lo];
    if (!(ablate & 1)) {
        const int n_pass = body_hi > body_lo ? (body_hi - body_lo + 255) >> 8 : 0;
        vm_issued += n_pass < MAX_PASSES ? n_pass : MAX_PASSES;
#pragma unroll
        for (int pass = 0; pass < MAX_PASSES; ++pass) {
            const int q4 = body_lo + 4 * lane + 256 * pass;
            if (q4 < body_hi) {
                v4f *p = reinterpret_cast<v4f *>(gal + q4);
                if (ablate & 16) __builtin_nontemporal_store(v[pass], p);   // diagnostics: streaming hint (no faster at four workgroups per CU)
                else *p = v[pass];
            }
        }
    }
    if (on) __builtin_nontemporal_store(edge, gal + idx);
}

// The 64 records of a batch (lane r holds record r in `rec`; `on`: the lane has one) leave through a staging area
// of 33, in two rounds.  In stream coordinates (dwords from the 16-byte aligned address below the batch) record r sits
// at sh + 19 r; round 0 stages records 0..32 and stores every whole quad below 608 + (sh ? 4 : 0), round 1 stages
// records 32..63 (record 32 again, so the quad that straddles the two halves left complete in round 0) and continues
// from that quad boundary: no partial stores in mid-batch.  d0 = first global dword of the batch.
__device__ __forceinline__ void stream_batch76(float *stage, const float (&rec)[18], bool on, int cnt, size_t d0, int block_id,
                                               float *__restrict__ out, int lane, int ablate, int &vm_issued)
{
    if (ablate & 64) d0 &= (size_t)0x7FFFFFu;   // diagnostics: every record lands in the buffer's first 32 MB (stores issued as shipped, nothing reaches the HBM)
    const int sh = (int)(d0 & 3);   // staging shift = global misalignment
    float *gal = out + (d0 - sh);   // 16-byte aligned
    const int split = 32 * kTriDwords + (sh ? 4 : 0);   // multiple of 4
    const int end = sh + cnt * kTriDwords;
    for (int h = 0; h == 0 || cnt > 32 * h; ++h) {   // wave-uniform, at most two rounds
        VTMC_WAVE_SYNC();
        const int r = lane - 32 * h;   // record slot in the staging area
        if (r >= 0 && r < kStageTris && on) {
            float *dstrec = stage + sh + r * kTriDwords;
#pragma unroll
            for (int c = 0; c < 18; ++c) dstrec[c] = rec[c];
            dstrec[18] = __int_as_float(block_id);
        }
        VTMC_WAVE_SYNC();
        // stream coordinates of this round, and the same relative to the staging area (which starts at 608 h)
        const int lo = h == 0 ? sh : split, hi = (h == 0 && cnt > 32) ? split : end;
        const int base = 32 * kTriDwords * h;
        stream_out_range<3>(stage - base, gal, lo, hi, lane, ablate, vm_issued, base + (kStageTris * kTriDwords & ~3));
    }
    VTMC_WAVE_SYNC();
}

// The 64 records of a batch through a staging area that holds all of them: one round (stage, one wave sync, five unrolled passes of
// 16-byte stores) instead of the two of stream_batch76.
__device__ __forceinline__ void stream_batch76_full(float *stage, const float (&rec)[18], bool on, int cnt, size_t d0, int block_id,
                                                    float *__restrict__ out, int lane, int ablate, int &vm_issued)
{
    if (ablate & 64) d0 &= (size_t)0x7FFFFFu;   // diagnostics: every record lands in the buffer's first 32 MB
    const int sh = (int)(d0 & 3);   // staging shift = global misalignment
    float *gal = out + (d0 - sh);   // 16-byte aligned
    VTMC_WAVE_SYNC();
    if (on) {
        float *dstrec = stage + sh + lane * kTriDwords;
#pragma unroll
        for (int c = 0; c < 18; ++c) dstrec[c] = rec[c];
        dstrec[18] = __int_as_float(block_id);
    }
    VTMC_WAVE_SYNC();
    stream_out_range<5>(stage, gal, sh, sh + cnt * kTriDwords, lane, ablate, vm_issued, (kOnceStageDwords - 4) & ~3);
    VTMC_WAVE_SYNC();
}

// One lane per triangle.  Vertex = position along the edge (MarchingCube.compute:128-133: t =
// -cube[a] / (cube[b] - cube[a]), lerp with v-u = +-1 on the edge axis and 0 on the others) and the
// trilinear normal fetch of MarchingCube.compute:69-99, which on a lattice edge is a 2-point lerp
// whose weight comes from the ROUNDED position (c0 = floor(P), c1 = ceil(P), t = P - c0).
template <bool FAST>
__device__ __forceinline__ void emit_flush2(const CornerView &L, int pending, size_t tri_base, int block_id,
                                            float *__restrict__ out, int lane, int ablate, int &vm_issued)
{
    VTMC_WAVE_SYNC();
    for (int s0 = 0; s0 < pending; s0 += 64) {
        const int s = s0 + lane;
        const size_t d0 = (tri_base + (size_t)s0) * kTriDwords;  // first global dword of this batch
        float rec[18];
#pragma unroll
        for (int c = 0; c < 18; ++c) rec[c] = 0.f;
        if (s < pending && !(ablate & 4)) {
            const unsigned sc = L.slot[s];
            const int cell = sc & 511u;
            const unsigned trip = sc >> 9;
            const int cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
            const int tcell = cx + 10 * cy + 100 * cz;
            // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
            const unsigned e[3] = {trip & 15u, (trip >> 8) & 15u, (trip >> 4) & 15u};
            const float *tile = L.tile;
            EdgeSite es[3];
            float va[3], vb[3], ga[3][3], gb[3][3];
            // One round of LDS reads per vertex: the samples at both endpoints of the edge and their forward neighbours.  t lies in [0, 1], so
            // floor(P) and ceil(P) of MarchingCube.compute:71-72 are the edge's endpoints a or b themselves: the two lattice gradients the
            // trilinear fetch blends (SampleNormal.compute:27-30) are known before t is -- which of them is c0 / c1 is a select afterwards.
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                es[k] = edge_site(cx, cy, cz, tcell, e[k]);
                va[k] = tile[es[k].ta];
                vb[k] = tile[es[k].tb];
                ga[k][0] = va[k] - tile[es[k].ta + 1];
                ga[k][1] = va[k] - tile[es[k].ta + 10];
                ga[k][2] = va[k] - tile[es[k].ta + 100];
                gb[k][0] = vb[k] - tile[es[k].tb + 1];
                gb[k][1] = vb[k] - tile[es[k].tb + 10];
                gb[k][2] = vb[k] - tile[es[k].tb + 100];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // t lies in [0,1] (the endpoints differ in sign class); the 1-ulp v_rcp can land a hair
                // outside, which would push floor/ceil one lattice point beyond the edge -- clamp it back
                const float t = FAST ? __builtin_amdgcn_fmed3f(-va[k] * __builtin_amdgcn_rcpf(vb[k] - va[k]), 0.0f, 1.0f)
                                     : (-va[k]) / (vb[k] - va[k]);
                const float q = (float)es[k].iak + (es[k].back ? -t : t);
                const float fq = floorf(q);
                const float w = q - fq;
                const bool c0_at_b = ((int)fq - es[k].iak) != 0, c1_at_b = ((int)ceilf(q) - es[k].iak) != 0;   // offset 0: endpoint a; +-1: endpoint b
                normalise<FAST>(ga[k]);
                normalise<FAST>(gb[k]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g0 = c0_at_b ? gb[k][c] : ga[k][c], g1 = c1_at_b ? gb[k][c] : ga[k][c];
                    rec[3 * k + c] = es[k].axis == (unsigned)c ? q : (float)es[k].ia[c];
                    rec[9 + 3 * k + c] = FAST ? __builtin_fmaf(w, g1 - g0, g0) : g0 + w * (g1 - g0);
                }
            }
        }
        const int cnt = pending - s0 < 64 ? pending - s0 : 64;
        stream_batch76(L.stage, rec, s < pending && !(ablate & 4), cnt, d0, block_id, out, lane, ablate, vm_issued);
    }
}

// case of cell (cx, cy, cz) from the LDS tile (CollectTriNum.compute:27-51)
__device__ __forceinline__ unsigned cell_case(const float *tile, unsigned cell)
{
    const int t2 = (int)(cell & 7u) + 10 * (int)((cell >> 3) & 7u), cz = (int)(cell >> 6);
    return layer_nibble(tile, t2, cz) | (layer_nibble(tile, t2, cz + 1) << 4);
}

// The four sign bits of a sample layer under a cell column as a nibble, corners 0, 1, 2, 3 of CollectTriNum.compute:27-31 (strict '>' as
// CollectTriNum.compute:50; NaN => outside).  The emit kernels are bound by ISSUED instructions (a SIMD retires one per four cycles whatever
// its kind: profiles/r05/emit_issue_bound.txt), and what hipcc makes of `(a > 0) | (b > 0) << 1 | ...` is four v_cmp + four v_cndmask + two
// v_or + up to four s_nop (gfx950 wants two wait states between a vector compare and a vector read of its mask).  Here: four compares in a
// row (each mask is two instructions old when it is read), then the bits are shifted in by add-with-carry -- n = 2 n + bit -- eight
// instructions, no wait states.
__device__ __forceinline__ unsigned sign_nibble(float c0, float c1, float c2, float c3)
{
    unsigned n;
    unsigned long long m0, m1, m2, m3, dump;
    asm("v_cmp_lt_f32 %[m3], 0, %[c3]\n\t"
        "v_cmp_lt_f32 %[m2], 0, %[c2]\n\t"
        "v_cmp_lt_f32 %[m1], 0, %[c1]\n\t"
        "v_cmp_lt_f32 %[m0], 0, %[c0]\n\t"
        "v_addc_co_u32 %[n], %[dump], 0, 0, %[m3]\n\t"
        "v_addc_co_u32 %[n], %[dump], %[n], %[n], %[m2]\n\t"
        "v_addc_co_u32 %[n], %[dump], %[n], %[n], %[m1]\n\t"
        "v_addc_co_u32 %[n], %[dump], %[n], %[n], %[m0]"
        : [n] "=&v"(n), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [dump] "=&s"(dump)
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3));
    return n;
}

__device__ __forceinline__ int compact_active_cells(const float *tile, unsigned short *acell, unsigned char *acase, int lane, unsigned rowmask)
{
    const float *p0 = tile + (lane & 7) + 10 * (lane >> 3);
    float sm[9][4];
#pragma unroll
    for (int z = 0; z < 9; ++z) {
        const float *p = p0 + 100 * z;
        sm[z][0] = p[0];
        sm[z][1] = p[1];
        sm[z][2] = p[11];
        sm[z][3] = p[10];
    }
    unsigned nib[9];   // corners 0, 1, 2, 3 of CollectTriNum.compute:27-31 at sample layer z
#pragma unroll
    for (int z = 0; z < 9; ++z)
        nib[z] = sign_nibble(sm[z][0], sm[z][1], sm[z][2], sm[z][3]);
    const u64 y_live = __builtin_amdgcn_ballot_w64(((rowmask >> (lane >> 3)) & 1u) != 0u);   // lanes of a live y layer
    int n_act = 0;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        if (!((rowmask >> (8 + z)) & 1u)) continue;   // wave-uniform
        const unsigned cs = nib[z] | (nib[z + 1] << 4);
        // triangles <=> the case is neither 0x00 nor 0xFF <=> (cs - 1) mod 256 < 254: one compare
        const u64 m = __builtin_amdgcn_ballot_w64(((cs - 1u) & 0xFFu) < 254u) & y_live;
        if (__builtin_amdgcn_inverse_ballot_w64(m)) {
            const int i = n_act + (int)lanes_below(m);
            acell[i] = (unsigned short)(64 * z + lane);
            acase[i] = (unsigned char)cs;
        }
        n_act += __builtin_popcountll(m);
    }
    return n_act;
}

// Passes 1 + 2 + flush of one block whose 10^3 tile is already in L->tile: cases
// (CollectTriNum.compute:48-51), compaction of the cells that hold triangles, triangle slots, then
// one lane per triangle.  `budget` = the scan's triangle count of the block; flushes are clamped to
// it so a classify/emit mismatch could never write outside the block's own slice of the buffer.
// `rowmask` (bits 0-7: y layers, 8-15: z layers that hold a cell with triangles; 0xFFFF = unknown):
// only tile rows next to such layers need to be valid, cells outside them are skipped.
template <bool FAST>
__device__ __forceinline__ void emit_block_from_tile(const CornerView &L, const u64 *s_vert, size_t tri_base, int budget,
                                                     int block_id, float *__restrict__ out, int lane, int ablate,
                                                     unsigned rowmask, int &vm_issued)
{
    const int n_act = compact_active_cells(L.tile, L.acell, L.cases, lane, rowmask);   // pass 1
    VTMC_WAVE_SYNC();

    // pass 2: triangle slots, 64 active cells per step
    int pending = 0;
    for (int c0 = 0; c0 < n_act; c0 += 64) {
        if (pending > L.slot_cap - 320) {  // wave-uniform
            const int n_out = pending < budget ? pending : budget;
            emit_flush2<FAST>(L, n_out, tri_base, block_id, out, lane, ablate, vm_issued);
            tri_base += n_out;
            budget -= n_out;
            pending = 0;
        }
        const int idx = c0 + lane;
        const bool valid = idx < n_act;
        const unsigned cell = valid ? L.acell[idx] : 0u;
        const u64 vw = valid ? s_vert[L.cases[idx]] : 0ull;  // fifteen 4-bit edge ids + the count in the top nibble
        const unsigned n = (unsigned)(vw >> 60);
        unsigned step_total;
        const unsigned pre_n = wave_prefix3(n, step_total);
        unsigned *dst = L.slot + pending + pre_n;
#pragma unroll
        for (unsigned i = 0; i < 5; ++i)
            if (i < n) dst[i] = cell | (((unsigned)(vw >> (12 * i)) & 0xFFFu) << 9);
        pending += (int)step_total;
    }
    if (pending > budget) pending = budget;
    if (pending > 0) emit_flush2<FAST>(L, pending, tri_base, block_id, out, lane, ablate, vm_issued);
}


// ----------------------------------------------------------------------------------------------
// Soup, vertex-once ("ONCE"): the 76-byte records of the reference, but every mesh vertex of the block is
// evaluated ONE time.  A triangle lane of emit_flush2 evaluates its three corners itself -- 3T / V = 4.7
// evaluations per welded vertex, 64 % of the kernel's vector instructions (profiles/r03a) -- here
//   N : the active cells number the block's vertices (a vertex = a lattice edge with a sign change; its
//       owner cell is the one whose corner 0 is the edge's low point, or the boundary cell next to a far
//       face: owned_edges below) and leave, per vertex, a descriptor (low lattice point | axis) in `vlist`
//       and, per lattice edge, the vertex id in `vtab` (one byte: axis * 729 + x + 9 y + 81 z);
//   V : one lane per vertex: position and normal (the arithmetic of MarchingCube.compute:128-133 and
//       :69-99, from the edge's low endpoint as the indexed output does), 24 bytes into `verts`;
//   T : one lane per triangle: three table look-ups (cube edge -> lattice edge -> vertex id), three
//       24-byte LDS reads, the record staged over the (now dead) tile and streamed out as before.
// The vertex of a cube edge the reference walks from its HIGH end (edges 2, 3, 6, 7 of
// MarchingCube.compute:40-43) differs from the reference's by rounding only (<= ~1e-6 in cell units against
// the 1e-5 bar; tests/test_gpu_parity.py); emit_fast_math = 0 keeps the bit-compatible kernel.
// A block with more than kVertCap vertices or more than kSlotCap triangles takes the per-corner path.
// ----------------------------------------------------------------------------------------------
// cube edges with a sign change from the case (the reference's cornerToEdgeTable, VoxelTerrain.cs:489-507, as three nibble
// operations: edges 0-3 / 4-7 are the low / high corner ring xor its rotation, edges 8-11 the two rings xor each other)
__device__ __forceinline__ unsigned case_edge_mask(unsigned cs)
{
    const unsigned n = cs & 15u, m = cs >> 4;
    const unsigned rn = (n ^ ((n >> 1) | (n << 3))) & 15u, rm = (m ^ ((m >> 1) | (m << 3))) & 15u;
    return rn | (rm << 4) | ((n ^ m) << 8);
}
constexpr int kVertCap = 176;          // vertices a block may hold in LDS (1024^3 perlin3d: at most 172)
constexpr int kOnceSlotCap = 352;      // triangles a block may hold in LDS (1024^3 perlin3d: at most 295)

// LDS of one wave of the vertex-once kernel.  Life times: tile (fetch .. V) | slot, vtab (N .. T) | acell, cases (pass 1 .. N) | vlist
// (N .. V) | verts (V .. T) | stage (T).  The stage holds a WHOLE batch of 64 records (round 4 staged 33 at a time, two rounds per
// batch: twice the fixed cost of a round) and lies over what is dead in T: the tile, the vertex list behind it and a pad.
struct __attribute__((aligned(16))) EmitLdsOnce {
    unsigned slot[kOnceSlotCap];                 // triangle slot -> cell | edge triple << 9
    float tile[1000];
    unsigned short vlist[kVertCap + 8];          // vertex id -> low lattice point | axis << 12; entry kVertCap: dump of an overflowing block
    unsigned char stage_tail[kOnceStageDwords * 4 - 4000 - (kVertCap + 8) * 2];
    union {
        struct {
            unsigned short acell[512];           // active cells of the block, ascending cell id
            unsigned char cases[512];            // cases[i]: the case of cell acell[i]
            float corner_stage[kStageTris * kTriDwords + 4];   // the per-corner path's staging area (blocks that overflow the caps)
        } p;
        float verts[kVertCap * 6];               // {position, normal} per vertex
    } u;
    unsigned char vtab[2192];                    // lattice edge (axis * 729 + x + 9 y + 81 z) -> vertex id
    __device__ __forceinline__ float *stage() { return tile; }
    __device__ __forceinline__ CornerView corner_view() { return CornerView{tile, slot, kOnceSlotCap, u.p.acell, u.p.cases, u.p.corner_stage}; }
};
static_assert(offsetof(EmitLdsOnce, tile) % 16 == 0 && offsetof(EmitLdsOnce, u) % 16 == 0 && sizeof(EmitLdsOnce) % 16 == 0, "16-byte aligned areas");
static_assert(offsetof(EmitLdsOnce, u) - offsetof(EmitLdsOnce, tile) >= kOnceStageDwords * 4, "the stage ends before the vertex records");
static_assert(sizeof(EmitLdsOnce) <= 12784 - 64, "three workgroups per CU: 42 LDS granules of 1280 bytes less the shared tables, a quarter each");

// per-workgroup constant tables of the ONCE path
struct OnceTables {
    unsigned short emask[256];   // case -> cube edges with a sign change (VoxelTerrain.cs:489-507)
    unsigned edge[12];           // cube edge -> low point's offset (x | y << 4 | z << 8) | axis << 12 | lattice-edge offset << 16
    unsigned short ownx[8];      // which coordinates of a cell are 7 -> the far-face edges it owns
};

__host__ __device__ constexpr unsigned once_edge_entry(unsigned e)
{
    const unsigned g = (unsigned)(kEdgeGeom >> (5u * e)) & 31u;
    const unsigned axis = g >> 3, low = g & 7u & ~(1u << axis);
    const unsigned lx = low & 1u, ly = (low >> 1) & 1u, lz = low >> 2;
    return lx | (ly << 4) | (lz << 8) | (axis << 12) | ((axis * 729u + lx + 9u * ly + 81u * lz) << 16);
}
// the cube edges a cell can own, in the order the vertex-once numbering takes them: the three at its corner 0 (x, y, z lattice edges
// starting there), then the nine far-face edges of a boundary cell (once_ownx_entry).  Every entry is a compile-time constant in the
// unrolled numbering rounds: no table look-up, no LDS round trip
constexpr unsigned kOwnedEdgeOrder[12] = {0, 3, 8, 1, 2, 4, 5, 6, 7, 9, 10, 11};

__host__ __device__ constexpr unsigned short once_ownx_entry(unsigned b7)
{
    const unsigned X = b7 & 1, Y = (b7 >> 1) & 1, Z = (b7 >> 2) & 1;
    return (unsigned short)((X * 0x202u) | (Y * 0x804u) | (Z * 0x090u) | ((X & Y) * 0x400u) | ((X & Z) * 0x020u) | ((Y & Z) * 0x040u));
}
__device__ __forceinline__ void once_tables_init(OnceTables *t, int tid)   // 256 threads
{
    t->emask[tid] = (unsigned short)case_edge_mask((unsigned)tid);
    if (tid < 12) t->edge[tid] = once_edge_entry((unsigned)tid);
    if (tid < 8) t->ownx[tid] = once_ownx_entry((unsigned)tid);
}
// the indexed output keeps only the two small tables (its per-wave LDS decides how many waves fit a CU: the 512-byte case -> edge-mask table
// is three nibble operations per active cell instead)
struct IdxTables {
    unsigned edge[12];
    unsigned short ownx[8];
};
__device__ __forceinline__ void idx_tables_init(IdxTables *t, int tid)   // >= 12 threads
{
    if (tid < 12) t->edge[tid] = once_edge_entry((unsigned)tid);
    if (tid < 8) t->ownx[tid] = once_ownx_entry((unsigned)tid);
}

// One mesh vertex from its descriptor (low lattice point x | y << 4 | z << 8, axis << 12): position along the edge
// (MarchingCube.compute:128-133, from the edge's low endpoint) and the trilinear normal fetch of MarchingCube.compute:69-99, which on a
// lattice edge is a 2-point lerp whose weight comes from the ROUNDED position.  rec = {position, normal}.
template <bool FAST>
__device__ __forceinline__ void eval_vertex(const float *tile, unsigned d, float (&rec)[6])
{
    const int c[3] = {(int)(d & 15u), (int)((d >> 4) & 15u), (int)((d >> 8) & 15u)};
    const unsigned axis = d >> 12;
    const int sk = axis == 0 ? 1 : (axis == 1 ? 10 : 100);
    const int tl = c[0] + 10 * c[1] + 100 * c[2], th = tl + sk;
    // ONE round of LDS reads: the samples at both ends of the edge and their forward neighbours.  t lies in [0, 1], so floor(P) and ceil(P)
    // (MarchingCube.compute:71-72) are the edge's own endpoints: both lattice gradients (SampleNormal.compute:27-30) are known before t is,
    // which of them is c0 / c1 is a select afterwards (round 4 computed t first and fetched the gradients behind it: two round trips)
    const float va = tile[tl], vb = tile[th];
    float glo[3] = {va - tile[tl + 1], va - tile[tl + 10], va - tile[tl + 100]};
    float ghi[3] = {vb - tile[th + 1], vb - tile[th + 10], vb - tile[th + 100]};
    // t lies in [0,1] (the endpoints differ in sign class); the 1-ulp v_rcp can land a hair outside: clamp it back
    const float t = FAST ? __builtin_amdgcn_fmed3f(-va * __builtin_amdgcn_rcpf(vb - va), 0.0f, 1.0f) : (-va) / (vb - va);
    const float ckf = (float)(axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]));
    const float q = ckf + t;
    const float fq = floorf(q);
    const float w = q - fq;   // the weight comes from the ROUNDED position (MarchingCube.compute:71-72)
    const bool c0_hi = (int)(fq - ckf) != 0, c1_hi = (int)(ceilf(q) - ckf) != 0;
    normalise<FAST>(glo);
    normalise<FAST>(ghi);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float g0 = c0_hi ? ghi[a] : glo[a], g1 = c1_hi ? ghi[a] : glo[a];
        rec[a] = axis == (unsigned)a ? q : (float)c[a];
        rec[3 + a] = FAST ? __builtin_fmaf(w, g1 - g0, g0) : g0 + w * (g1 - g0);
    }
}

// one triangle of the vertex-once expansion: slot -> three table look-ups (cube edge -> lattice edge -> vertex id) -> three 24-byte records
__device__ __forceinline__ void once_gather_record(const unsigned *slot, const unsigned char *vtab, const float *verts, const OnceTables *tb, int s,
                                                   float (&rec)[18])
{
    const unsigned sc = slot[s];
    const unsigned cell = sc & 511u, trip = sc >> 9;
    const unsigned cell9 = (cell & 7u) + 9u * ((cell >> 3) & 7u) + 81u * (cell >> 6);
    // table entries (3i, 3i+2, 3i+1): the winding swap of MarchingCube.compute:147-157
    const unsigned e[3] = {trip & 15u, (trip >> 8) & 15u, (trip >> 4) & 15u};
    unsigned vid[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vid[k] = vtab[cell9 + (tb->edge[e[k]] >> 16)];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *v = verts + vid[k] * 6u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rec[3 * k + a] = v[a];
            rec[9 + 3 * k + a] = v[3 + a];
        }
    }
}

template <bool FAST>
__device__ __forceinline__ void emit_block_once(EmitLdsOnce *L, const u64 *s_vert, const OnceTables *tb, size_t tri_base, int budget,
                                                int block_id, float *__restrict__ out, int lane, int ablate, unsigned rowmask, int &vm_issued, PhaseClock &pc)
{
    if (budget > kOnceSlotCap) {   // wave-uniform: more triangles than the slot buffer holds -- the per-corner path with its own flushes
        emit_block_from_tile<FAST>(L->corner_view(), s_vert, tri_base, budget, block_id, out, lane, ablate, rowmask, vm_issued);
        return;
    }
    const float *tile = L->tile;
    const int n_act = compact_active_cells(tile, L->u.p.acell, L->u.p.cases, lane, rowmask);   // pass 1
    VTMC_WAVE_SYNC();
    pc.mark(2);
    unsigned short *vlist = L->vlist;
    unsigned char *vtab = L->vtab;

    // N + pass 2, 64 active cells per step: vertex numbering and triangle slots.  One LDS round for the cell and its case, one for the
    // three tables; everything behind them runs on registers and compile-time constants.
    int n_vert = 0, pending = 0;
    for (int c0 = 0; c0 < n_act; c0 += 64) {
        const int idx = c0 + lane;
        const bool valid = idx < n_act;
        const int ic = valid ? idx : n_act - 1;
        const unsigned cell = L->u.p.acell[ic];
        const unsigned cs = valid ? (unsigned)L->u.p.cases[ic] : 0u;   // case 0: no edges, no triangles
        const unsigned cx = cell & 7u, cy = (cell >> 3) & 7u, cz = cell >> 6;
        const unsigned b7 = (unsigned)(cx == 7u) | ((unsigned)(cy == 7u) << 1) | ((unsigned)(cz == 7u) << 2);
        const u64 vw = s_vert[cs];
        const unsigned owned = tb->emask[cs] & (0x109u | tb->ownx[b7]);   // the three edges at corner 0, and the far-face edges of a boundary cell
        const unsigned desc = cx | (cy << 4) | (cz << 8);
        const unsigned cell9 = cx + 9u * cy + 81u * cz;
#pragma unroll
        for (int r = 0; r < 12; ++r) {   // one ballot per cube edge a cell can own; a round nobody needs costs a compare and a branch
            const unsigned ed = once_edge_entry(kOwnedEdgeOrder[r]);   // folds to an immediate
            const u64 m = __builtin_amdgcn_ballot_w64((owned & (1u << kOwnedEdgeOrder[r])) != 0u);
            if (m == 0ull) continue;   // wave-uniform
            if (__builtin_amdgcn_inverse_ballot_w64(m)) {   // the ballot IS the exec mask: no second compare
                const int id = n_vert + (int)lanes_below(m);
                vlist[id < kVertCap ? id : kVertCap] = (unsigned short)(desc + (ed & 0xFFFFu));   // entry kVertCap: the dump of a block that overflows (it leaves for the per-corner path below)
                vtab[cell9 + (ed >> 16)] = (unsigned char)id;
            }
            n_vert += __builtin_popcountll(m);
        }
        // triangle slots (as emit_block_from_tile)
        const unsigned n = (unsigned)(vw >> 60);
        unsigned step_total;
        const unsigned pre_n = wave_prefix3(n, step_total);
        if (pending + (int)step_total > kOnceSlotCap) break;   // wave-uniform; cannot happen while the scan's count (budget <= kOnceSlotCap) describes this tile: never outside the slot buffer
        unsigned *dst = L->slot + pending + pre_n;
#pragma unroll
        for (unsigned i = 0; i < 5; ++i)
            if (i < n) dst[i] = cell | (((unsigned)(vw >> (12 * i)) & 0xFFFu) << 9);
        pending += (int)step_total;
    }
    if (pending > budget) pending = budget;   // never outside the block's slice of the buffer
    if (n_vert > kVertCap) {   // wave-uniform: the slots are the per-corner path's own
        if (pending > 0) emit_flush2<FAST>(L->corner_view(), pending, tri_base, block_id, out, lane, ablate, vm_issued);
        return;
    }
    VTMC_WAVE_SYNC();
    pc.mark(3);

    // V: one lane per vertex, from the edge's low endpoint.  Branch-free up to the store: a lane past the end evaluates the last vertex again
    float *verts = L->u.verts;
    for (int s0 = 0; s0 < n_vert && !(ablate & 4); s0 += 64) {
        const int s = s0 + lane;
        float r6[6];
        eval_vertex<FAST>(tile, vlist[s < n_vert ? s : n_vert - 1], r6);
        if (s < n_vert) {
            float *rec = verts + s * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) rec[a] = r6[a];
        }
    }
    VTMC_WAVE_SYNC();
    pc.mark(4);

    // T: one lane per triangle; the staging area (a whole batch) lies over the tile and the vertex list, which nobody reads any more.  Two batches of 64 at a time: the
    // look-up chain of the second (slot -> edge table -> vertex id -> vertex records: four LDS round trips) runs beside the first one's
    float *stage = L->stage();
    for (int s0 = 0; s0 < pending; s0 += 128) {
        const int sa = s0 + lane, sb = sa + 64;
        float ra[18], rb[18];
        const bool two = s0 + 64 < pending;   // wave-uniform
        if (two) {   // both chains in one basic block: the scheduler issues their loads side by side
            once_gather_record(L->slot, vtab, verts, tb, sa, ra);
            once_gather_record(L->slot, vtab, verts, tb, sb < pending ? sb : pending - 1, rb);
        } else {
            once_gather_record(L->slot, vtab, verts, tb, sa < pending ? sa : pending - 1, ra);
        }
        pc.mark(5);
        const int na = pending - s0 < 64 ? pending - s0 : 64;
        stream_batch76_full(stage, ra, sa < pending, na, (tri_base + (size_t)s0) * kTriDwords, block_id, out, lane, ablate, vm_issued);
        if (two) {
            const int nb = pending - s0 - 64 < 64 ? pending - s0 - 64 : 64;
            stream_batch76_full(stage, rb, sb < pending, nb, (tri_base + (size_t)s0 + 64) * kTriDwords, block_id, out, lane, ablate, vm_issued);
        }
        pc.mark(6);
    }
}

// ----------------------------------------------------------------------------------------------
// Indexed (welded) output of one block -- new in the build (the reference welds later, on the CPU,
// with Mesh.Optimize(), VoxelTerrain.cs:460).  A mesh vertex lives on a lattice edge with a sign
// change; all cells around that edge share it:
//   vertices : one 24-byte record {position, normal} per such edge of the block's 9^3 lattice.
//              Every such edge is a cube edge of exactly one OWNER cell -- the cell whose corner 0 is
//              the edge's low point (cube edges 0, 3, 8 of MarchingCube.compute:40-43), or, on the
//              block's x = 8 / y = 8 / z = 8 faces, the boundary cell next to it (its edges 1, 9 / 2, 11 /
//              4, 7 / 10, 5, 6) -- and vertices are ordered by owner cell x + 8y + 64z, then cube edge id.
//              That order falls out of the ACTIVE cells' cases: owned = edge mask(case) & ownership
//              mask(cell position), so numbering is one 4-ballot prefix sum per 64 active cells
//              (round 1 walked the 729 lattice points: twelve steps with four LDS reads each);
//   indices  : three block-local int32 per triangle, canonical triangle order (as the soup).
// The vertex is evaluated from the edge's LOW endpoint (the orientation the reference uses for cube
// edges 0, 1, 4, 5, 8..11); where the reference walks an edge backwards (edges 2, 3, 6, 7) its
// t' = 1 - t differs from this one by rounding only (<= ~1e-6 in cell units, bar 1e-5).
// ----------------------------------------------------------------------------------------------
constexpr int kVertDwords = 6;   // 24-byte vertex record
constexpr int kVlistCap = 256;   // vertex descriptors per evaluation window
constexpr int kIdxFastVerts = 255;   // blocks with at most this many vertices number them in one byte per lattice edge (vtab)

// owner-side id of a lattice edge: [axis * 4 + (o_u + 2 o_v)] -> cube edge, (u, v) = the two other axes in order
constexpr u64 kOwnerEdge = 0x0ull | (2ull << 4) | (4ull << 8) | (6ull << 12) |            // x: (oy, oz)
                           (3ull << 16) | (1ull << 20) | (7ull << 24) | (5ull << 28) |    // y: (ox, oz)
                           (8ull << 32) | (9ull << 36) | (11ull << 40) | (10ull << 44);   // z: (ox, oy)

struct __attribute__((aligned(16))) EmitLdsIdx {
    union {
        float tile[1000];
        unsigned slot[kSlotCap];          // pass 2: triangle slot -> cell | edge triple << 9.  Over the tile: nothing reads a sample after the vertex
                                          // phase (pass 2 takes the cases from `cases`), and the next tile is stored at the top of the next block
    } t;
    unsigned short vlist[kVlistCap + 8];  // vertex phase: vertex id - window -> low lattice point (x | y << 4 | z << 8) | axis << 12; entry kVlistCap: the
                                          // dump of lanes that own nothing in a numbering round (branch-free rounds, round 6)
    unsigned short acell[512];      // active cells of the block, ascending cell id
    union {
        unsigned char vtab[2192];   // <= 255 vertices: lattice edge (axis * 729 + x + 9 y + 81 z) -> vertex id
        unsigned cellmap[512];      // more: owner cell -> first vertex id | owned-edge mask << 16 (active cells only)
    } m;
    unsigned char cases[512];       // cases[i]: the case of cell acell[i] (pass 1 -> numbering -> pass 2)
};   // 8256 bytes: with the shared tables a workgroup of four waves takes 35.3 KB -- four per CU, 16 waves
static_assert(sizeof(EmitLdsIdx) % 16 == 0, "keeps the waves' blocks 16-byte aligned");


// s_own[e * 8 + b7] for cube edge e of a cell whose coordinates equal 7 where b7 has a bit set: low byte = offset
// of the OWNER cell's id from the cell's, high byte = the edge's id as the owner sees it (kOwnerEdge)
__device__ __forceinline__ unsigned short owner_entry(unsigned e, unsigned b7)
{
    const unsigned g = (unsigned)(kEdgeGeom >> (5u * e)) & 31u;
    const unsigned axis = g >> 3, lowoff = g & 7u & ~(1u << axis);   // the low point's offsets on the two other axes
    const unsigned step = lowoff & ~b7, keep = lowoff & b7;         // into the neighbour cell, unless that leaves the block
    const unsigned delta = (step & 1u) + 8u * ((step >> 1) & 1u) + 64u * (step >> 2);
    const unsigned ou = axis == 0 ? (keep >> 1) & 1u : (keep & 1u), ov = axis == 2 ? (keep >> 1) & 1u : (keep >> 2) & 1u;
    const unsigned eo = (unsigned)(kOwnerEdge >> (4u * (axis * 4u + ou + 2u * ov))) & 15u;
    return (unsigned short)(delta | (eo << 8));
}

// exclusive wave prefix sum of a per-lane value in 0..15 from four ballots
__device__ __forceinline__ unsigned wave_prefix4(unsigned n, unsigned &total)
{
    u64 m0 = __builtin_amdgcn_ballot_w64((n & 1u) != 0);
    u64 m1 = __builtin_amdgcn_ballot_w64((n & 2u) != 0);
    u64 m2 = __builtin_amdgcn_ballot_w64((n & 4u) != 0);
    u64 m3 = __builtin_amdgcn_ballot_w64((n & 8u) != 0);
    total = (unsigned)__builtin_popcountll(m0) + 2u * (unsigned)__builtin_popcountll(m1) + 4u * (unsigned)__builtin_popcountll(m2) +
            8u * (unsigned)__builtin_popcountll(m3);
    return lanes_below(m0) + 2u * lanes_below(m1) + 4u * lanes_below(m2) + 8u * lanes_below(m3);
}

// Indexed output of one block.  Numbering (N), vertex evaluation (V), then triangle slots and indices (pass 2 + T):
//   N : the active cells number the block's vertices in the canonical order (owner cell, then cube edge id): per cell the edges it
//       owns with a sign change, one 4-ballot prefix sum per 64 cells, and per vertex a descriptor (low lattice point | axis) in
//       `vlist`.  Blocks of at most 255 vertices (all but ~0.3 % on the benchmark fields) also leave the id of every lattice edge in
//       the byte table `vtab`, so that a triangle lane finds a vertex with two table reads; larger blocks keep round 2's owner-cell map.
//   V : one lane per vertex (eval_vertex), two 12-byte stores per lane straight from registers.
//   T : one lane per triangle, one 12-byte store per lane: 64 lanes write 768 contiguous bytes, no staging.
template <bool FAST>
__device__ __forceinline__ void emit_block_indexed(EmitLdsIdx *L, const u64 *s_vert, const unsigned short *s_own, const IdxTables *tb,
                                                   size_t tri_base, int tri_budget, size_t vert_base, int vert_budget,
                                                   float *__restrict__ out_vertices, int *__restrict__ out_indices, int lane, int ablate,
                                                   unsigned rowmask, int &vm_issued, PhaseClock &pc)
{
    typedef float v3u __attribute__((ext_vector_type(3), aligned(4)));
    typedef int i3u __attribute__((ext_vector_type(3), aligned(4)));
    const float *tile = L->t.tile;
    // pass 1: compaction of the active cells (as the soup path, row masks included)
    const int n_act = compact_active_cells(tile, L->acell, L->cases, lane, rowmask);
    VTMC_WAVE_SYNC();
    pc.mark(2);
    const bool big = vert_budget > kIdxFastVerts;   // wave-uniform: the scan's vertex count of this block decides the numbering's form

    // N: vertex numbering over the active cells, 64 per step.  Descriptors of the ids in [window, window + kVlistCap) are queued.
    // The kernel is bound by ISSUED instructions and a scalar one costs what a vector one does (profiles/r05/emit_issue_bound.txt;
    // profiles/r06/sq_counters_indexed.txt: 412 scalar + 114 branch instructions per block beside 643 vector ones), so (round 6) the two forms
    // of the numbering are two loops -- no wave-uniform `big` test inside the rounds -- and the three rounds every step runs (the edges at a
    // cell's corner 0) are branch-free: a lane that owns nothing writes the lists' dump entries instead of sitting out an exec region.
    auto number_small = [&]() {   // at most kIdxFastVerts vertices: one window, every id has its byte in vtab
        int vrun = 0;
        for (int c0 = 0; c0 < n_act; c0 += 64) {
            const int idx = c0 + lane;
            const bool valid = idx < n_act;
            const int ic = valid ? idx : n_act - 1;
            const unsigned cell = L->acell[ic];
            const unsigned cs = valid ? (unsigned)L->cases[ic] : 0u;   // case 0: no edges
            const unsigned cx = cell & 7u, cy = (cell >> 3) & 7u, cz = cell >> 6;
            const unsigned b7 = (unsigned)(cx == 7u) | ((unsigned)(cy == 7u) << 1) | ((unsigned)(cz == 7u) << 2);
            const unsigned owned = case_edge_mask(cs) & (0x109u | tb->ownx[b7]);
            const unsigned desc = cx | (cy << 4) | (cz << 8);
            const unsigned cell9 = cx + 9u * cy + 81u * cz;
            unsigned step_total;
            const unsigned id0 = (unsigned)vrun + wave_prefix4((unsigned)__builtin_popcount(owned), step_total);
#pragma unroll
            for (int r = 0; r < 3; ++r) {   // cube edges 0, 3, 8: no branch, no exec region
                const unsigned e = kOwnedEdgeOrder[r], ed = once_edge_entry(kOwnedEdgeOrder[r]);
                const bool has = ((owned >> e) & 1u) != 0u;
                const unsigned id = id0 + (unsigned)__builtin_popcount(owned & ((1u << e) - 1u));
                L->vlist[has ? (id & 255u) : (unsigned)kVlistCap] = (unsigned short)(desc + (ed & 0xFFFFu));   // & 255: a count mismatch could never write outside the list
                L->m.vtab[has ? cell9 + (ed >> 16) : 2191u] = (unsigned char)id;                                 // 2187 lattice edges: bytes 2187-2191 are spare
            }
            if (__builtin_amdgcn_ballot_w64((owned & 0xEF6u) != 0u) != 0ull) {   // wave-uniform: some cell of the step owns an edge of a far face
#pragma unroll
                for (int r = 3; r < 12; ++r) {
                    const unsigned e = kOwnedEdgeOrder[r], ed = once_edge_entry(kOwnedEdgeOrder[r]);
                    const bool has = ((owned >> e) & 1u) != 0u;
                    if (__builtin_amdgcn_ballot_w64(has) == 0ull) continue;   // wave-uniform
                    if (has) {
                        const unsigned id = id0 + (unsigned)__builtin_popcount(owned & ((1u << e) - 1u));
                        L->vlist[id & 255u] = (unsigned short)(desc + (ed & 0xFFFFu));
                        L->m.vtab[cell9 + (ed >> 16)] = (unsigned char)id;
                    }
                }
            }
            vrun += (int)step_total;
        }
        return vrun;
    };
    auto number_big = [&](int window, bool first) {   // more: the owner-cell map of round 2, descriptors window by window
        int vrun = 0;
        for (int c0 = 0; c0 < n_act; c0 += 64) {
            const int idx = c0 + lane;
            const bool valid = idx < n_act;
            const unsigned cell = valid ? L->acell[idx] : 0u;
            const unsigned cs = valid ? (unsigned)L->cases[idx] : 0u;   // pass 1 left the case beside the cell
            const unsigned cx = cell & 7u, cy = (cell >> 3) & 7u, cz = cell >> 6;
            const unsigned b7 = (unsigned)(cx == 7u) | ((unsigned)(cy == 7u) << 1) | ((unsigned)(cz == 7u) << 2);
            const unsigned owned = valid ? (case_edge_mask(cs) & (0x109u | tb->ownx[b7])) : 0u;
            const unsigned desc = cx | (cy << 4) | (cz << 8);
            unsigned step_total;
            const int id0 = vrun + (int)wave_prefix4((unsigned)__builtin_popcount(owned), step_total);
            if (first && valid) L->m.cellmap[cell] = (unsigned)id0 | (owned << 16);
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const unsigned e = kOwnedEdgeOrder[r], ed = once_edge_entry(kOwnedEdgeOrder[r]);
                const bool has = ((owned >> e) & 1u) != 0u;
                if (r >= 3 && __builtin_amdgcn_ballot_w64(has) == 0ull) continue;   // wave-uniform
                if (has) {
                    const int q = id0 + __builtin_popcount(owned & ((1u << e) - 1u)) - window;
                    if (q >= 0 && q < kVlistCap) L->vlist[q] = (unsigned short)(desc + (ed & 0xFFFFu));
                }
            }
            vrun += (int)step_total;
        }
        return vrun;
    };
    int n_vert = big ? number_big(0, true) : number_small();
    if (n_vert > vert_budget) n_vert = vert_budget;  // never outside the block's slice of the vertex buffer
    pc.mark(3);

    // V: one lane per vertex, from the edge's low endpoint; position and normal leave as two 12-byte stores per lane
    for (int window = 0; window < n_vert; window += kVlistCap) {
        if (window > 0) number_big(window, false);
        VTMC_WAVE_SYNC();
        const int n_w = n_vert - window < kVlistCap ? n_vert - window : kVlistCap;
        for (int s0 = 0; s0 < n_w && !(ablate & 16); s0 += 64) {
            const int s = s0 + lane;
            if (s < n_w) {
                float rec[kVertDwords];
                eval_vertex<FAST>(tile, L->vlist[s], rec);
                float *p = out_vertices + (vert_base + (size_t)(window + s)) * kVertDwords;
                if (!(ablate & 1)) {
                    *reinterpret_cast<v3u *>(p) = v3u{rec[0], rec[1], rec[2]};
                    *reinterpret_cast<v3u *>(p + 3) = v3u{rec[3], rec[4], rec[5]};
                }
            }
            vm_issued += (ablate & 1) ? 0 : 2;   // issued for certain: lane 0 of the batch has a vertex
        }
        VTMC_WAVE_SYNC();
    }
    pc.mark(4);

    // pass 2 + T: triangle slots, 64 active cells per step; a triangle lane finds each of its three vertices in the lattice-edge table
    // (or, in a block of more than 255 vertices, through its edge's owner cell) and writes its index triple from registers
    auto flush = [&](int pending, size_t base) {
        VTMC_WAVE_SYNC();
        if (ablate & 32) pending = 0;
        for (int s0 = 0; s0 < pending; s0 += 64) {
            const int s = s0 + lane;
            if (s < pending) {
                const unsigned sc = L->t.slot[s];
                const unsigned cell = sc & 511u, trip = sc >> 9;
                const unsigned e[3] = {trip & 15u, (trip >> 8) & 15u, (trip >> 4) & 15u};  // winding swap, MarchingCube.compute:147-157
                int id[3];
                if (!big) {
                    const unsigned cell9 = (cell & 7u) + 9u * ((cell >> 3) & 7u) + 81u * (cell >> 6);
#pragma unroll
                    for (int k = 0; k < 3; ++k) id[k] = (int)L->m.vtab[cell9 + (tb->edge[e[k]] >> 16)];
                } else {
                    const unsigned b7 = (unsigned)((cell & 7u) == 7u) | ((unsigned)(((cell >> 3) & 7u) == 7u) << 1) | ((unsigned)((cell >> 6) == 7u) << 2);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const unsigned ow = s_own ? s_own[e[k] * 8u + b7] : owner_entry(e[k], b7);   // owner cell offset | owner-side edge id << 8 (no table: computed -- blocks of > 255 vertices are rare)
                        const unsigned vm = L->m.cellmap[cell + (ow & 0xFFu)];
                        id[k] = (int)((vm & 0xFFFFu) + (unsigned)__builtin_popcount((vm >> 16) & ((1u << (ow >> 8)) - 1u)));
                    }
                }
                if (!(ablate & 1)) *reinterpret_cast<i3u *>(out_indices + (base + (size_t)s) * 3) = i3u{id[0], id[1], id[2]};
            }
            vm_issued += (ablate & 1) ? 0 : 1;
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
        const u64 vw = valid ? s_vert[(unsigned)L->cases[idx]] : 0ull;   // the tile is gone: the slots lie over it
        const unsigned n = (unsigned)(vw >> 60);
        unsigned step_total;
        const unsigned pre_n = wave_prefix3(n, step_total);
        const unsigned d0 = (unsigned)pending + pre_n;
#pragma unroll
        for (unsigned i = 0; i < 5; ++i)   // no exec region per slot: a triangle the cell does not have lands in the dump word behind the slots
            L->t.slot[i < n ? d0 + i : (unsigned)kSlotCap] = cell | (((unsigned)(vw >> (12 * i)) & 0xFFFu) << 9);
        pending += (int)step_total;
    }
    if (pending > tri_budget) pending = tri_budget;
    pc.mark(5);
    if (pending > 0) flush(pending, tri_base);
    pc.mark(6);
}

}  // namespace vtmc
#endif
