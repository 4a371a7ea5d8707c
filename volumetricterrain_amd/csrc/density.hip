// density.hip -- synthetic density samplers on the GPU (perlin3d / fbm8).
//
// The reference fills its density grid on the CPU, one virtual QueryDensity call per sample
// (VoxelTerrain.cs:284-305), and holds no noise field of its own (its only noise modifier wraps the
// un-vendored LibNoise, TerrainModifier.cs:158-196).  These kernels are the build's "density-field
// sampler" stage for the benchmark grids: Ken Perlin's 2002 improved noise with a 256-entry
// permutation from a SplitMix64-driven Fisher-Yates shuffle, summed over octaves, minus a vertical
// ramp.  VALU + small LDS tables; writes are lane-contiguous along the stride-1 axis.
#include "vtmc_internal.h"
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"   // the sign words' v_writelane names m0 as clobbered: a reserved register, which this file's kernels never use otherwise

namespace vtmc {

static uint64_t splitmix64(uint64_t &state)
{
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void density_permutation(uint64_t seed, unsigned char perm[256])
{
    for (int i = 0; i < 256; ++i) perm[i] = (unsigned char)i;
    uint64_t s = seed;
    for (int i = 255; i >= 1; --i) {
        int j = (int)(splitmix64(s) % (uint64_t)(i + 1));
        unsigned char t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
}

__device__ __forceinline__ float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
__device__ __forceinline__ float mixf(float t, float a, float b) { return a + t * (b - a); }
__device__ __forceinline__ float gradf(int hash, float x, float y, float z)
{
    int h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

__device__ __forceinline__ float noise3(const unsigned char *p, float x, float y, float z)
{
    float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
    x -= fx;
    y -= fy;
    z -= fz;
    float u = fade(x), v = fade(y), w = fade(z);
#define VTMC_P(i) ((int)p[(i) & 255])
    int A = VTMC_P(X) + Y, AA = VTMC_P(A) + Z, AB = VTMC_P(A + 1) + Z;
    int B = VTMC_P(X + 1) + Y, BA = VTMC_P(B) + Z, BB = VTMC_P(B + 1) + Z;
    float r = mixf(w,
                   mixf(v, mixf(u, gradf(VTMC_P(AA), x, y, z), gradf(VTMC_P(BA), x - 1, y, z)),
                        mixf(u, gradf(VTMC_P(AB), x, y - 1, z), gradf(VTMC_P(BB), x - 1, y - 1, z))),
                   mixf(v, mixf(u, gradf(VTMC_P(AA + 1), x, y, z - 1), gradf(VTMC_P(BA + 1), x - 1, y, z - 1)),
                        mixf(u, gradf(VTMC_P(AB + 1), x, y - 1, z - 1), gradf(VTMC_P(BB + 1), x - 1, y - 1, z - 1))));
#undef VTMC_P
    return r;
}

// Per-sample form (more than 8 octaves): one workgroup = 256 consecutive samples along the fast axis; blockIdx.x enumerates
// (segment, y, slow-axis index, volume) -- flattened because grid.y/z stop at 65535
__global__ __launch_bounds__(256) void density_generic_kernel(DensityLaunch dl, const unsigned char *__restrict__ perm,
                                                       const int *__restrict__ origins, float *__restrict__ out,
                                                       int fast_is_z, int nseg)
{
    __shared__ unsigned char s_perm[256];
    s_perm[threadIdx.x] = perm[threadIdx.x];
    __syncthreads();
    const int dfast = fast_is_z ? dl.dz : dl.dx;
    const int dslow = fast_is_z ? dl.dx : dl.dz;
    unsigned r = blockIdx.x;
    const int seg = r % nseg;
    r /= nseg;
    const int j = r % dl.dy;  // y
    r /= dl.dy;
    const int c = r % dslow;  // index along the slow axis
    const int vol = r / dslow;
    const int a = seg * 256 + threadIdx.x;  // index along the fast axis
    if (a >= dfast) return;
    const int i = fast_is_z ? c : a, k = fast_is_z ? a : c;
    const float px = (float)(origins[3 * vol] + i), py = (float)(origins[3 * vol + 1] + j),
                pz = (float)(origins[3 * vol + 2] + k);
    float x = px * dl.frequency, y = py * dl.frequency, z = pz * dl.frequency;
    float amp = 1.0f, sum = 0.0f;
    for (int o = 0; o < dl.octaves; ++o) {
        sum = sum + amp * noise3(s_perm, x, y, z);
        x *= dl.lacunarity;
        y *= dl.lacunarity;
        z *= dl.lacunarity;
        amp *= dl.gain;
    }
    out[vol * dl.sv + i * dl.sx + j * dl.sy + k * dl.sz] = sum - (py - dl.ramp_center) * dl.ramp_scale;
}


// ----------------------------------------------------------------------------------------------
// density_column_kernel -- the wave64 sampler (octaves <= 8).
//
// A lane owns one column of the volume and walks it along z; the lanes of a workgroup are 256
// consecutive points of the (x, y) plane, x first.  For compact x-fastest volumes that plane is one
// contiguous slab of memory and every step of a workgroup writes 1 KB of it; for the C# z-fastest
// layout a column is contiguous instead and the values leave transposed through LDS (ZT).  The walk
// axis never depends on the layout, so a sample is one function of its position: the same bits in
// every layout.  What makes the sampler cheap is what is UNIFORM or CONSTANT along the walk:
//   * everything that depends on the walk coordinate alone (lattice cell, fraction, fade weight,
//     which octaves enter a new cell) is the same for every lane of every wave at that step: a small
//     kernel writes it once per (volume, step) into a 96-byte row; a workgroup stages the rows of its
//     segment in LDS once and every step reads its row back with four broadcast ds_read_b128 (+ two on a
//     step that rebuilds).  Scalar loads of the rows were tried twice: round 2 without a prefetch (one
//     scalar-cache miss per step, ~900 cycles, every wave of the workgroup at once), round 5 with the next
//     row's s_load_dwordx16 a step ahead and the fmas reading SGPR pairs: 23 % slower than the LDS rows;
//   * inside one lattice cell the eight gradient dot products are linear in the walk fraction t and
//     the interpolation weights of the two lane axes are constant, so the two lane-interpolated faces
//     of the cell collapse to S_j(t) = alpha_j + beta_j * t (j = low / high face) and a sample is
//         noise = mix(fade(t), alpha_0 + beta_0 * t, alpha_1 + beta_1 * (t - 1)) = a0 + b0 t + fade(t) (c + d t)
//     -- three fmas per octave instead of 8 hashes + 8 gradients + 7 lerps, and PACKED: octaves 2p | 2p + 1
//     in the halves of one v_pk_fma_f32.  The kernel is bound by the vector ALU's issue slots (round 4: 61
//     vector instructions per 64 samples x 4 cycles = its duration; now 43: profiles/r05/sampler_valu_bound.txt);
//   * the (alpha, beta) pairs are rebuilt only when the walk enters a new cell (wave-uniform: a bit of
//     the row's mask word); stepping into the next cell re-uses the high face as the new low face, so
//     one face = 4 LDS lookups of gradient vectors (the x-y part of the hash chain is constant along the walk), and
//     its lerps over the two lane axes run on (x-y dot product, gz) PAIRS: the dot product is written into the
//     spare third component of the gradient it came from (table layout gx, gy, 0, gz).
// Same noise definition as the per-sample kernel above and as oracle/density_ref.c; the lerp order
// (lane axes first, walk axis last) and the fma contractions move results by a few 1e-7 (bar of the
// twin test: 2e-6).
// ----------------------------------------------------------------------------------------------
constexpr int kColSeg = 160;    // steps one workgroup walks before the next segment starts with a two-face rebuild (20 KB of rows in LDS)
constexpr int kRowUsed = 24;    // dwords of a row the walk reads
constexpr int kRowDwords = 24;  // [0..7] t, [8..15] fade(t), 16 ramp (y walk), 17 masks "next cell" | "rebuild both faces" << 8, 18 free, 19/20 cell & 255 of octaves 0-3 / 4-7

// one thread per (volume, step): the row of everything that depends on the walk coordinate alone
__global__ __launch_bounds__(256) void density_row_kernel(DensityLaunch dl, const int *__restrict__ origins, int axis, int n_steps,
                                                           int seg_len, float *__restrict__ rows)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dl.n_volumes * n_steps) return;
    const int vol = idx / n_steps, j = idx - vol * n_steps;
    const float pw = (float)(origins[3 * vol + axis] + j);
    float y = pw * dl.frequency, yp = (pw - 1.0f) * dl.frequency;
    float *row = rows + (long long)idx * kRowDwords;
    unsigned m1 = 0, m2 = 0, yc[2] = {0, 0};
    for (int o = 0; o < 8; ++o) {
        const float fy = floorf(y);
        const int cell = (int)fy, prev = (int)floorf(yp);
        const float t = y - fy;
        row[o] = t;
        row[8 + o] = fade(t);
        if (o < dl.octaves) {
            if (j % seg_len == 0 || (cell != prev && cell != prev + 1)) m2 |= 1u << o;   // first step of a walk, or a jump
            else if (cell == prev + 1) m1 |= 1u << o;
        }
        yc[o >> 2] |= ((unsigned)cell & 255u) << (8 * (o & 3));
        y *= dl.lacunarity;
        yp *= dl.lacunarity;
    }
    row[16] = axis == 1 ? (pw - dl.ramp_center) * dl.ramp_scale : 0.f;
    row[17] = __uint_as_float(m1 | (m2 << 8));   // one word: the walk asks "does this step rebuild anything" with one test
    row[18] = 0.f;
    row[19] = __uint_as_float(yc[0]);
    row[20] = __uint_as_float(yc[1]);
    for (int q = 21; q < kRowDwords; ++q) row[q] = 0.f;
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32: two fmas, one issue slot

struct ColOct {
    float a1, b1;          // the high face  S_1 = a1 + b1 * (t - 1), scaled by the octave's amplitude; the low face S_0 = a0 + b0 * t and
                           // S_1 - S_0 = c + d * t (c = (a1 - b1) - a0, d = b1 - b0, derived whenever a face changes) live in the kernel's
                           // PAIRED arrays (octaves 2p, 2p + 1 in one 64-bit register pair: the walk's fmas are packed).  Each face is a pure
                           // function of (column, lattice row, octave) -- a sample never depends on where its walk started
    float ra, rb;          // fractions along the two lane axes (x, y)
    unsigned key;          // 16 * P(P(X+i)+Y+j) for (i,j) = 00 (low half), 01 (high half)
    unsigned key2;         // the same for (i,j) = 10, 11 -- byte offsets into s_g512 before the lattice row is added
};

// The walk is along z, lane plane (x, y), whatever the memory layout: a sample is one function of its position (round 2 also had a walk
// along y for z-fastest and padded volumes: other axes interpolated first, last-bit differences, 26 of 5.4 M triangles on a 512^3 volume).
// ZT: the volume is z-fastest in memory (the C# float[,,] order): a lane's column is contiguous, the lane plane is not.  The values of
// kTrSteps steps go through LDS and leave as 16-byte stores along z (two lanes per column) instead of 64 four-byte stores to 64 lines a step.
constexpr int kTrPitch = 258;   // 256 columns + 2: the transposed read is bank-conflict free
constexpr int tr_steps(int noct) { return noct <= 4 ? 16 : 8; }   // steps per flush: 64-byte runs where the LDS budget allows (three workgroups per CU either way)
template <int NOCT, bool ZT = false>
__global__ __launch_bounds__(256, ZT ? 3 : 4) void density_column_kernel(DensityLaunch dl, const unsigned char *__restrict__ perm,
                                                              const int *__restrict__ origins, const float *__restrict__ rows,
                                                              float *__restrict__ out, int fast_is_z, int n_plane_wgs, int n_seg,
                                                              int seg_len, unsigned long long *__restrict__ signs)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ unsigned short s_p2[256];  // P(i) | P(i+1) << 8
    __shared__ v4f s_grad[16];            // gradient of hash h as (gx, gy, 0, gz), components in {-1, 0, 1}: a corner's x-y dot product lands beside gz (one register pair)
    __shared__ v4f s_g512[512];      // gradient of hash P(i & 255), i = key byte + lattice row <= 511 -- one lookup per corner, no wrap
    __shared__ __attribute__((aligned(16))) float s_rows[kColSeg][kRowUsed];   // this segment's rows
    __shared__ float2 s_uv[NOCT][256];    // fade weights of the two lane axes, per octave and lane: constant along the walk, read back at a face rebuild
    constexpr int kTrSteps = tr_steps(NOCT);
    constexpr int kLanesPerCol = kTrSteps / 4;   // lanes that share a column's run of kTrSteps values (float4 each)
    __shared__ float s_tr[ZT ? kTrSteps : 1][ZT ? kTrPitch : 1];   // ZT: the last kTrSteps values of every column
    __shared__ long long s_col[ZT ? 256 : 1];                      // ZT: element offset of column c's first sample of this segment (-1: past the plane)

    const int tid = threadIdx.x;
    unsigned r = blockIdx.x;
    const int pw = r % n_plane_wgs;
    r /= n_plane_wgs;
    const int seg = r % n_seg;
    const int vol = r / n_seg;
    const int n_steps = dl.dz;
    const int w_begin = seg * seg_len;
    const int w_count = min(seg_len, n_steps - w_begin);
    const int ox = origins[3 * vol], oy = origins[3 * vol + 1];   // the z origin enters through the rows (density_row_kernel)

    {   // tables
        const unsigned p0 = perm[tid], p1 = perm[(tid + 1) & 255];
        s_p2[tid] = (unsigned short)(p0 | (p1 << 8));
        if (tid < 16) {
            const int h = tid;   // gradf(): u = h<8 ? x : y;  v = h<4 ? y : (h==12||h==14 ? x : z);  (h&1 ? -u : u) + (h&2 ? -v : v)
            const float su = (h & 1) ? -1.f : 1.f, sv = (h & 2) ? -1.f : 1.f;
            const bool vx = h == 12 || h == 14;
            const float g[3] = {(h < 8 ? su : 0.f) + (vx ? sv : 0.f), (h >= 8 ? su : 0.f) + (h < 4 ? sv : 0.f), (h >= 4 && !vx) ? sv : 0.f};
            s_grad[h] = v4f{g[0], g[1], 0.f, g[2]};
        }
        {
            __syncthreads();   // s_grad
            s_g512[tid] = s_grad[p0 & 15u];
            s_g512[tid + 256] = s_grad[p0 & 15u];
        }
        // the segment's rows: everything that depends on the walk coordinate alone, staged once per workgroup
        const float *src = rows + ((long long)vol * n_steps + w_begin) * kRowDwords;
        for (int e = tid; e < w_count * kRowUsed; e += 256) {
            const int rr = e / kRowUsed, cc = e - rr * kRowUsed;
            s_rows[rr][cc] = src[rr * kRowDwords + cc];
        }
    }
    __syncthreads();

    // this lane's column: (i, j, k) of its first sample
    int i, j, k;
    bool live;
    {
        const long long q = (long long)pw * 256 + tid;
        live = q < (long long)dl.dx * dl.dy;
        const long long qc = live ? q : (long long)dl.dx * dl.dy - 1;   // past the plane's end: the plane's last column again (this workgroup's own)
        i = (int)(qc % dl.dx);
        j = (int)(qc / dl.dx);
        k = w_begin;
    }
    // lane axes A, B: (x, z) for the y walk, (x, y) for the z walk
    float pa = (float)(ox + i) * dl.frequency, pb = (float)(oy + j) * dl.frequency;
    const float lane_ramp = ((float)(oy + j) - dl.ramp_center) * dl.ramp_scale;

    ColOct st[NOCT];
    constexpr int NP = (NOCT + 1) / 2;   // octave pairs; the odd octave count's missing half stays all-zero
    v2f a0p[NP], b0p[NP], cp[NP], dp[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) a0p[q] = b0p[q] = cp[q] = dp[q] = v2f{0.f, 0.f};
    float amp[NOCT];
    {
        float am = 1.0f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            const float fa = floorf(pa), fb = floorf(pb);
            const unsigned A = (unsigned)(int)fa & 255u, B = (unsigned)(int)fb & 255u;
            st[o].ra = pa - fa;
            st[o].rb = pb - fb;
            const unsigned px = s_p2[A];   // P(X) | P(X+1) << 8
            {   // the x-y part of the hash chain is constant along the z walk
                const unsigned q0 = s_p2[((px & 255u) + B) & 255u], q1 = s_p2[((px >> 8) + B) & 255u];
                st[o].key = ((q0 & 255u) << 4) | ((q0 >> 8) << 20);
                st[o].key2 = ((q1 & 255u) << 4) | ((q1 >> 8) << 20);
            }
            st[o].a1 = st[o].b1 = 0.f;
            s_uv[o][tid] = make_float2(fade(st[o].ra), fade(st[o].rb));   // only this lane ever reads its entries back
            amp[o] = am;   // wave-uniform: the compiler keeps these in SGPRs
            pa *= dl.lacunarity;
            pb *= dl.lacunarity;
            am *= dl.gain;
        }
    }

    // one face of the cell: the four corners over the lane axes at lattice row Wp of the walk axis,
    // interpolated along A then B.
    // Everything a face derives from the column's constants alone (ra - 1, rb - 1, the halves of the two keys) would be hoisted out of the walk for
    // all octaves at once (~50 VGPRs): those six instructions are volatile asm, which stays where it is and reads the state in place (round 4
    // passed COPIES of the state through an empty asm statement: four v_mov a face).
    typedef const __attribute__((address_space(3))) v4f *lds_v4f;
    const unsigned g512_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)s_g512;
    auto face = [&](const ColOct &s, int o, unsigned Wp, float am, float &alpha, float &beta) {
        const float ra = s.ra, rb = s.rb;
        float a1, b1;
        asm volatile("v_add_f32 %0, -1.0, %2\n\tv_add_f32 %1, -1.0, %3" : "=v"(a1), "=v"(b1) : "v"(ra), "v"(rb));
        const unsigned gw = g512_lds + (Wp << 4);   // wave-uniform: table base + 16 * lattice row
        unsigned ad00, ad01, ad10, ad11;   // g[a-corner][b-corner]: key = 00 | 01 << 16, key2 = 10 | 11 << 16 (byte offsets of the hashes' gradients)
        asm volatile("v_add_u32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_add_u32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                     "v_add_u32_sdwa %2, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_add_u32_sdwa %3, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                     : "=&v"(ad00), "=&v"(ad01), "=&v"(ad10), "=&v"(ad11)
                     : "s"(gw), "v"(s.key), "v"(s.key2));
        const v4f g00 = *(lds_v4f)(uintptr_t)ad00, g01 = *(lds_v4f)(uintptr_t)ad01, g10 = *(lds_v4f)(uintptr_t)ad10, g11 = *(lds_v4f)(uintptr_t)ad11;
        const float2 uv = s_uv[o][tid];
        const v2f u2 = {uv.x, uv.x}, v2 = {uv.y, uv.y};
        const float a0 = ra, b0 = rb;
        // a corner = (its gradient's x-y dot product, its gradient's component along the walk): the lerps along A then B run on both at once
        const v2f p00 = {__builtin_fmaf(g00.y, b0, g00.x * a0), g00.w}, p10 = {__builtin_fmaf(g10.y, b0, g10.x * a1), g10.w};
        const v2f p01 = {__builtin_fmaf(g01.y, b1, g01.x * a0), g01.w}, p11 = {__builtin_fmaf(g11.y, b1, g11.x * a1), g11.w};
        const v2f pu0 = fma2(u2, p10 - p00, p00), pu1 = fma2(u2, p11 - p01, p01);
        const v2f pv = fma2(v2, pu1 - pu0, pu0);
        const v2f ab = pv * v2f{am, am};   // one packed multiply for both
        alpha = ab.x;
        beta = ab.y;
    };

    float *dst = out + vol * dl.sv + (long long)i * dl.sx + (long long)j * dl.sy + (long long)k * dl.sz;
    const long long dst_step = dl.sz;
    if constexpr (ZT) {
        s_col[tid] = live ? vol * dl.sv + (long long)i * dl.sx + (long long)j * dl.sy + (long long)k * dl.sz : -1ll;
        __syncthreads();
    }
    // ZT: steps [first, first + n) of this segment sit in s_tr[0..n): thread t writes quad t % kLanesPerCol of column t / kLanesPerCol + (256 / kLanesPerCol) * pass
    auto flush_transposed = [&](int first, int n) {
        typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < kLanesPerCol; ++pass) {
            const int c = tid / kLanesPerCol + (256 / kLanesPerCol) * pass, q4 = 4 * (tid % kLanesPerCol);
            const long long base = s_col[c];
            if (base >= 0 && q4 < n && !(VTMC_ABLATE(dl.ablate) & 1)) {
                float *p = out + base + first + q4;   // sz == 1
                if (q4 + 4 <= n) {
                    *reinterpret_cast<v4u *>(p) = v4u{s_tr[q4][c], s_tr[q4 + 1][c], s_tr[q4 + 2][c], s_tr[q4 + 3][c]};
                } else {
                    for (int r = 0; q4 + r < n; ++r) p[r] = s_tr[q4 + r][c];
                }
            }
        }
        __syncthreads();
    };
    // sign volume: word ((vol * dz + k) * 4 n_plane_wgs + 4 pw + wave), bit = lane: the sign of plane point 256 pw + tid at step k
    unsigned long long *sign_dst = signs ? signs + ((long long)vol * dl.dz + w_begin) * (4ll * n_plane_wgs) + 4 * pw + (tid >> 6) : nullptr;
    v2f base_sum = {0.f, 0.f};   // sums of the low faces' constants a0 over the even | odd octaves, re-added in octave order whenever one of them changes
    // one step of the walk from its row in LDS: rebuilds (when `rebuild`: some octave enters a new lattice cell here -- the same for every lane)
    // and the sample
    auto step_value = [&](int jj, bool rebuild) __attribute__((always_inline)) {
        const v4f *rp = reinterpret_cast<const v4f *>(s_rows[jj]);
        if (rebuild) {
            const v4f ma = rp[4];
            const unsigned m12 = __builtin_amdgcn_readfirstlane(__float_as_uint(ma.y)), m1 = m12 & 255u, m2 = m12 >> 8;
            const unsigned wc[2] = {(unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(ma.w)),
                                    (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(s_rows[jj][20]))};
#pragma unroll
            for (int o = 0; o < NOCT; ++o) {
                if ((m1 | m2) & (1u << o)) {
                    const unsigned W = (wc[o >> 2] >> (8 * (o & 3))) & 255u;
                    float lo_a = st[o].a1, lo_b = st[o].b1;   // the high face of the cell just left is the low face of this one, bit for bit
                    if (m2 & (1u << o)) face(st[o], o, W, amp[o], lo_a, lo_b);
                    face(st[o], o, W + 1u, amp[o], st[o].a1, st[o].b1);
                    a0p[o >> 1][o & 1] = lo_a;
                    b0p[o >> 1][o & 1] = lo_b;
                    cp[o >> 1][o & 1] = (st[o].a1 - st[o].b1) - lo_a;   // S_1(t) = a1 + b1 (t - 1)
                    dp[o >> 1][o & 1] = st[o].b1 - lo_b;
                }
            }
            base_sum = a0p[0];
#pragma unroll
            for (int q = 1; q < NP; ++q) base_sum += a0p[q];
            base_sum.x -= lane_ramp;   // the ramp rides in the accumulator's start: nothing to subtract per step
        }
        // the step's row: broadcast reads (every lane the same address), AFTER the rebuild: sixteen registers a face does not have to work around
        const v4f ta = rp[0], tb = rp[1], va = rp[2], vb = rp[3];
        const v2f t2[4] = {ta.xy, ta.zw, tb.xy, tb.zw}, f2[4] = {va.xy, va.zw, vb.xy, vb.zw};
        // noise_o = S_0 + fade(t) (S_1 - S_0) = a0 + b0 t + fade(t) (c + d t): three fmas per octave on top of the constant part, PACKED: octaves
        // 2p | 2p + 1 in the halves of one v_pk_fma_f32 (the sampler is bound by the vector ALU's issue slots: 61 a step in round 4, 24 of them
        // these fmas); even and odd octaves accumulate apart and meet at the end
        v2f acc = base_sum;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            acc = fma2(b0p[q], t2[q], acc);
            acc = fma2(f2[q], fma2(dp[q], t2[q], cp[q]), acc);
        }
        return acc.x + acc.y;   // (two chains over pairs 0, 1 | 2, 3 were tried for the shorter dependence: 1 % slower, the extra add costs more)
    };
    if constexpr (ZT) {
        for (int jj = 0; jj < w_count; ++jj) {
            const unsigned mm = __builtin_amdgcn_readfirstlane(__float_as_uint(s_rows[jj][17]));
            s_tr[jj % kTrSteps][tid] = step_value(jj, mm != 0u);
            if (jj % kTrSteps == kTrSteps - 1 || jj == w_count - 1) flush_transposed(jj - jj % kTrSteps, jj % kTrSteps + 1);   // workgroup-uniform
        }
    } else {
        // Round 4's step spent ~38 of its ~62 instructions on things that are not its 24 fmas (most of them scalar -- which, it turned out, issue
        // beside the vector ones here: removing them bought 3.5 %, packing the fmas 11 %).  Per 64 steps now: ONE ballot says which steps rebuild anything (a bit
        // test per step instead of two readfirstlanes + or + compare, and the mask row is read only where it is needed); no exec mask around
        // the stores at all (a lane past the plane's end walks the plane's LAST column: the same value to the same address as that column's
        // own lane); a step's sign ballot goes into lane (step % 64) of a register pair (two v_writelane) and the chunk's 64 words leave in
        // ONE store instead of a compare + exec dance per step.
        const long long sign_pitch = 4ll * n_plane_wgs;
        const int lane = tid & 63;
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
        auto walk = [&](auto with_signs) {
            for (int c0 = 0; c0 < w_count; c0 += 64) {
                const int n = __builtin_amdgcn_readfirstlane(w_count - c0 < 64 ? w_count - c0 : 64);   // workgroup-uniform: keeps the step loop's bound scalar
                const int jl = c0 + lane < w_count ? c0 + lane : w_count - 1;
                unsigned long long reb = __builtin_amdgcn_ballot_w64(__float_as_uint(s_rows[jl][17]) != 0u && lane < n);
                unsigned slo = 0u, shi = 0u;   // lane j: the sign ballot of step c0 + j
                for (int j = 0; j < n; ++j) {
                    const bool rebuild = ((unsigned)reb & 1u) != 0u;
                    reb >>= 1;
                    const float value = step_value(c0 + j, rebuild);
                    if (!(VTMC_ABLATE(dl.ablate) & 1) || value == 1e30f) *dst = value;   // ablate 1: diagnostics, no stores
                    dst += dst_step;
                    if constexpr (decltype(with_signs)::value) {
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(value > 0.f);
                        // two SGPRs in one VOP3 break gfx9's constant-bus rule: the lane select goes through m0 (which nothing else in this kernel uses)
                        asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                                     : "+v"(slo), "+v"(shi)
                                     : "s"((unsigned)m), "s"((unsigned)(m >> 32)), "s"(j)
                                     : "m0");
                    }
                }
                // the sign volume (z walk only): word ((vol * dz + k) * 4 n_plane_wgs + 4 pw + wave), bit = lane: lane j writes step c0 + j's
                // word; the bits of lanes past the plane's end are 0
                if constexpr (decltype(with_signs)::value)
                    if (lane < n) sign_dst[(long long)(c0 + lane) * sign_pitch] = (((unsigned long long)shi << 32) | slo) & live_mask;
            }
        };
        if (signs) walk(std::true_type{}); else walk(std::false_type{});
    }
}

// d_signs (optional): density_sign_words(dl) 64-bit words, written when the walk is along z (density_writes_signs(dl))
bool density_writes_signs(const DensityLaunch &dl)
{
    return dl.octaves <= 8 && dl.sx == 1 && dl.sy == dl.dx && dl.sz >= (long long)dl.dx * dl.dy;
}
int density_sign_plane_words(int dx, int dy) { return 4 * (int)(((long long)dx * dy + 255) / 256); }
size_t density_sign_words(const DensityLaunch &dl) { return (size_t)dl.n_volumes * dl.dz * density_sign_plane_words(dl.dx, dl.dy) + 2; }

hipError_t launch_density(const DensityLaunch &dl, const unsigned char *d_perm, const int *d_origins,
                          float *d_rows, float *d_out, unsigned long long *d_signs, hipStream_t stream)
{
    const int fast_is_z = (dl.sz == 1 && dl.sx != 1) ? 1 : 0;
    const int dfast = fast_is_z ? dl.dz : dl.dx;
    const int dslow = fast_is_z ? dl.dx : dl.dz;
    if (dl.octaves <= 8) {
        // ALWAYS along z, lanes over the (x, y) plane, whatever the memory layout: a sample is then one function of its position, the same
        // bits for an x-fastest chunk, a padded volume and the C# z-fastest order (round 2 walked y for the latter two: the faces were
        // interpolated over other axes first and the fields differed in the last bit -- 26 triangles of 5.4 M on a 512^3 volume).  The
        // lane plane is contiguous memory for compact x-fastest volumes (every step of a workgroup writes 1 KB); for the z-fastest order a
        // lane's column is contiguous instead and the stores of a step are 64 partial lines the L2 merges over the next steps.
        const int n_steps = dl.dz;
        const long long plane = (long long)dl.dx * dl.dy;
        const long long n_plane_wgs = (plane + 255) / 256;
        const int n_seg = (n_steps + kColSeg - 1) / kColSeg;
        const int seg_len = (n_steps + n_seg - 1) / n_seg;  // equal segments: every one pays the same two-face start-up
        const long long n_wgs = n_plane_wgs * n_seg * dl.n_volumes;
        const long long n_rows = (long long)dl.n_volumes * n_steps;
        if (n_wgs <= 0 || n_wgs > 0x7fffffffll || n_rows > 0x7fffffffll) return hipErrorInvalidValue;
        launch_begin();
        hipLaunchKernelGGL(density_row_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, dl, d_origins, 2,
                           n_steps, seg_len, d_rows);
        // residency cap (tuning key "density_wgs_per_cu"): unused dynamic LDS leaves wave slots, registers and LDS to the kernels of
        // another stream -- the extract stages of the previous batch are HBM-bound, this kernel is ALU-bound
        const bool zt = dl.sz == 1 && dl.sx != 1;   // z-fastest: transposed stores
        const size_t lds_static = 22272 + 2048 * (size_t)dl.octaves + (zt ? sizeof(float) * tr_steps(dl.octaves) * kTrPitch + 2048 : 0);
        const size_t lds_want = dl.wgs_per_cu > 0 && dl.wgs_per_cu < 4 ? (size_t)(160 * 1024 / dl.wgs_per_cu - 1024) : 0;
        const size_t dyn = lds_want > lds_static ? (lds_want - lds_static) & ~(size_t)255 : 0;
#define VTMC_COL(N)                                                                                                                   \
    case N:                                                                                                                           \
        if (zt)                                                                                                                       \
            hipLaunchKernelGGL((density_column_kernel<N, true>), dim3((unsigned)n_wgs), dim3(256), dyn, stream, dl, d_perm, d_origins, d_rows, \
                               d_out, fast_is_z, (int)n_plane_wgs, n_seg, seg_len, nullptr);                                         \
        else                                                                                                                          \
            hipLaunchKernelGGL((density_column_kernel<N, false>), dim3((unsigned)n_wgs), dim3(256), dyn, stream, dl, d_perm, d_origins, d_rows, \
                               d_out, fast_is_z, (int)n_plane_wgs, n_seg, seg_len, d_signs);                                        \
        break;
        switch (dl.octaves) {   // the octave count is a template parameter: the eight octaves of a sample are straight-line code
            VTMC_COL(1) VTMC_COL(2) VTMC_COL(3) VTMC_COL(4) VTMC_COL(5) VTMC_COL(6) VTMC_COL(7) VTMC_COL(8)
        }
#undef VTMC_COL
        return launch_end();
    }
    const int nseg = (dfast + 255) / 256;
    long long n_wgs = (long long)dl.n_volumes * dslow * dl.dy * nseg;
    if (n_wgs <= 0 || n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
    launch_begin();
    hipLaunchKernelGGL(density_generic_kernel, dim3((unsigned)n_wgs), dim3(256), 0, stream, dl, d_perm, d_origins, d_out,
                       fast_is_z, nseg);
    return launch_end();
}

size_t density_rows_bytes(int n_volumes, int dy, int dz) { return (size_t)n_volumes * (size_t)(dy > dz ? dy : dz) * kRowDwords * sizeof(float); }

}  // namespace vtmc
