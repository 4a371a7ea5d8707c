// density.hip -- synthetic density samplers on the GPU (perlin3d / fbm8).
//
// The reference fills its density grid on the CPU, one virtual QueryDensity call per sample
// (VoxelTerrain.cs:284-305), and holds no noise field of its own (its only noise modifier wraps the
// un-vendored LibNoise, TerrainModifier.cs:158-196).  These kernels are the build's "density-field
// sampler" stage for the benchmark grids: Ken Perlin's 2002 improved noise with a 256-entry
// permutation from a SplitMix64-driven Fisher-Yates shuffle, summed over octaves, minus a vertical
// ramp.  VALU + small LDS tables; writes are lane-contiguous along the stride-1 axis.
#include "vtmc_internal.h"

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
// A lane owns one (fast, slow) column of the volume and walks it along y; the 64 lanes of a wave
// are 64 consecutive points along the stride-1 axis, so every store is one coalesced 256-byte row.
// What makes it cheap is what is UNIFORM or CONSTANT along that walk:
//   * everything that depends on y alone (lattice cell, fraction, fade weight, the ramp) is the same
//     for all lanes of the workgroup: it is computed once per (y, octave) into an LDS table and read
//     back with one broadcast ds_read_b128;
//   * inside one lattice cell the eight gradient dot products are linear in the y fraction and the
//     x / z interpolation weights are constant, so the two x-z-interpolated faces of the cell collapse
//     to S_j(t) = alpha_j + beta_j * t (j = low / high y face) and a sample is
//         noise = mix(fade(t), alpha_0 + beta_0 * t, alpha_1 + beta_1 * (t - 1))
//     -- 5 VALU instructions per octave instead of 8 hashes + 8 gradients + 7 lerps;
//   * the (alpha, beta) pairs are rebuilt only when the walk enters a new cell (a wave-uniform branch:
//     all lanes share y); stepping into the next cell re-uses the high face as the new low face, so one
//     face = 2 + 2 + 4 LDS lookups (permutation pairs, packed gradient offsets, gradient vectors).
// Same noise definition as the per-sample kernel below and as oracle/density_ref.c; the x-z-y lerp
// order and the fma contractions move results by a few 1e-7 (bar of the twin test: 2e-6).
// VALU roofline: ~5 lane-ops per sample-octave + ~45 per face rebuild, see DESIGN.md.
// ----------------------------------------------------------------------------------------------
constexpr int kColSeg = 160;  // y samples per workgroup segment (table: kColSeg x (NOCT + 1) x 16 bytes)

struct ColOct {
    float a0, b0, a1, b1;  // faces j = 0, 1:  S_j(t) = a_j + b_j * t, already scaled by the octave's amplitude
    float xr, zr, u, w;    // fractions and fade weights along the two lane axes
    unsigned pxz;          // P(X) | P(X+1) << 8 | Z << 16
};

template <int NOCT>
__global__ __launch_bounds__(256) void density_column_kernel(DensityLaunch dl, const unsigned char *__restrict__ perm,
                                                              const int *__restrict__ origins, float *__restrict__ out,
                                                              int fast_is_z, int n_plane_wgs, int n_yseg, int seg_len)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ unsigned short s_p2[256];  // P(i) | P(i+1) << 8
    __shared__ unsigned s_h2[256];        // byte offsets into s_grad of hash P(i) (low half) and P(i+1) (high half)
    __shared__ v4f s_grad[16];            // gradient of hash h as (gx, gy, gz, 0), components in {-1, 0, 1}
    __shared__ float s_amp[8];
    __shared__ v4f s_y[kColSeg][NOCT + 1];  // per (y, octave): {t, t - 1, fade(t), cell & 255 | flag << 8}; last: {ramp, 0, 0, 0}

    const int tid = threadIdx.x;
    unsigned r = blockIdx.x;
    const int pw = r % n_plane_wgs;
    r /= n_plane_wgs;
    const int ys = r % n_yseg;
    const int vol = r / n_yseg;
    const int y_begin = ys * seg_len;
    const int y_count = min(seg_len, dl.dy - y_begin);
    const int ox = origins[3 * vol], oy = origins[3 * vol + 1], oz = origins[3 * vol + 2];

    {   // tables
        const unsigned p0 = perm[tid], p1 = perm[(tid + 1) & 255];
        s_p2[tid] = (unsigned short)(p0 | (p1 << 8));
        s_h2[tid] = ((p0 & 15u) << 4) | (((p1 & 15u) << 4) << 16);
        if (tid < 16) {
            const int h = tid;   // gradf(): u = h<8 ? x : y;  v = h<4 ? y : (h==12||h==14 ? x : z);  (h&1 ? -u : u) + (h&2 ? -v : v)
            const float su = (h & 1) ? -1.f : 1.f, sv = (h & 2) ? -1.f : 1.f;
            float g[3] = {0.f, 0.f, 0.f};
            g[h < 8 ? 0 : 1] += su;
            g[h < 4 ? 1 : ((h == 12 || h == 14) ? 0 : 2)] += sv;
            s_grad[h] = v4f{g[0], g[1], g[2], 0.f};
        }
        if (tid == 0) {
            float amp = 1.0f;
            for (int o = 0; o < 8; ++o) {
                s_amp[o] = amp;
                amp *= dl.gain;
            }
        }
        // y table: one entry per (y, octave); the previous sample's cell decides the flag
        for (int e = tid; e < y_count * (NOCT + 1); e += 256) {
            const int jj = e / (NOCT + 1), o = e - jj * (NOCT + 1);
            const float py = (float)(oy + y_begin + jj);
            if (o == NOCT) {
                s_y[jj][NOCT] = v4f{(py - dl.ramp_center) * dl.ramp_scale, 0.f, 0.f, 0.f};
                continue;
            }
            float y = py * dl.frequency, yp = (py - 1.0f) * dl.frequency;
            for (int k = 0; k < o; ++k) {
                y *= dl.lacunarity;
                yp *= dl.lacunarity;
            }
            const float fy = floorf(y);
            const int cell = (int)fy, prev = (int)floorf(yp);
            const float t = y - fy;
            const unsigned flag = jj == 0 ? 2u : (cell == prev ? 0u : (cell == prev + 1 ? 1u : 2u));
            s_y[jj][o] = v4f{t, t - 1.0f, fade(t), __uint_as_float(((unsigned)cell & 255u) | (flag << 8))};
        }
    }
    __syncthreads();

    // this lane's column
    const int dfast = fast_is_z ? dl.dz : dl.dx;
    const int dslow = fast_is_z ? dl.dx : dl.dz;
    const long long q = (long long)pw * 256 + tid;
    const bool live = q < (long long)dfast * dslow;
    const long long qc = live ? q : 0;
    const int a = (int)(qc % dfast), c = (int)(qc / dfast);
    const int i = fast_is_z ? c : a, k = fast_is_z ? a : c;
    float x = (float)(ox + i) * dl.frequency, z = (float)(oz + k) * dl.frequency;

    ColOct st[NOCT];
#pragma unroll
    for (int o = 0; o < NOCT; ++o) {
        const float fx = floorf(x), fz = floorf(z);
        const unsigned X = (unsigned)(int)fx & 255u, Z = (unsigned)(int)fz & 255u;
        st[o].xr = x - fx;
        st[o].zr = z - fz;
        st[o].u = fade(st[o].xr);
        st[o].w = fade(st[o].zr);
        st[o].pxz = (unsigned)s_p2[X] | (Z << 16);
        st[o].a0 = st[o].b0 = st[o].a1 = st[o].b1 = 0.f;
        x *= dl.lacunarity;
        z *= dl.lacunarity;
    }

    // one face of the cell: the four corners (i, k) at lattice row Yp, interpolated in x then z
    auto face = [&](const ColOct &s, unsigned Yp, float amp, float &alpha, float &beta) {
        const unsigned Z = s.pxz >> 16;
        const unsigned pa = s_p2[((s.pxz & 255u) + Yp) & 255u] & 255u;         // P(P(X) + Yp)
        const unsigned pb = s_p2[(((s.pxz >> 8) & 255u) + Yp) & 255u] & 255u;  // P(P(X+1) + Yp)
        const unsigned ha = s_h2[(pa + Z) & 255u], hb = s_h2[(pb + Z) & 255u];
        const char *gb = reinterpret_cast<const char *>(s_grad);
        const v4f g00 = *reinterpret_cast<const v4f *>(gb + (ha & 0xFFFFu)), g01 = *reinterpret_cast<const v4f *>(gb + (ha >> 16));
        const v4f g10 = *reinterpret_cast<const v4f *>(gb + (hb & 0xFFFFu)), g11 = *reinterpret_cast<const v4f *>(gb + (hb >> 16));
        const float x0 = s.xr, x1 = s.xr - 1.0f, z0 = s.zr, z1 = s.zr - 1.0f;
        const float c00 = __builtin_fmaf(g00.z, z0, g00.x * x0), c10 = __builtin_fmaf(g10.z, z0, g10.x * x1);
        const float c01 = __builtin_fmaf(g01.z, z1, g01.x * x0), c11 = __builtin_fmaf(g11.z, z1, g11.x * x1);
        const float cu0 = __builtin_fmaf(s.u, c10 - c00, c00), cu1 = __builtin_fmaf(s.u, c11 - c01, c01);
        const float bu0 = __builtin_fmaf(s.u, g10.y - g00.y, g00.y), bu1 = __builtin_fmaf(s.u, g11.y - g01.y, g01.y);
        alpha = amp * __builtin_fmaf(s.w, cu1 - cu0, cu0);
        beta = amp * __builtin_fmaf(s.w, bu1 - bu0, bu0);
    };

    float *dst = out + vol * dl.sv + (long long)i * dl.sx + (long long)y_begin * dl.sy + (long long)k * dl.sz;
    for (int jj = 0; jj < y_count; ++jj) {
        float sum = 0.0f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (o < dl.octaves) {
                const v4f e = s_y[jj][o];
                const unsigned bits = __builtin_amdgcn_readfirstlane(__float_as_uint(e.w));
                if (bits >> 8) {  // the walk entered a new lattice cell (same y for every lane: wave-uniform)
                    const unsigned Y = bits & 255u;
                    const float amp = s_amp[o];
                    if ((bits >> 8) == 1u) {
                        st[o].a0 = st[o].a1;
                        st[o].b0 = st[o].b1;
                    } else {
                        face(st[o], Y, amp, st[o].a0, st[o].b0);
                    }
                    face(st[o], Y + 1u, amp, st[o].a1, st[o].b1);
                }
                const float s0 = __builtin_fmaf(st[o].b0, e.x, st[o].a0), s1 = __builtin_fmaf(st[o].b1, e.y, st[o].a1);
                sum += __builtin_fmaf(e.z, s1 - s0, s0);
            }
        }
        const float ramp = s_y[jj][NOCT].x;
        if (live) *dst = sum - ramp;
        dst += dl.sy;
    }
}

hipError_t launch_density(const DensityLaunch &dl, const unsigned char *d_perm, const int *d_origins,
                          float *d_out, hipStream_t stream)
{
    const int fast_is_z = (dl.sz == 1 && dl.sx != 1) ? 1 : 0;
    const int dfast = fast_is_z ? dl.dz : dl.dx;
    const int dslow = fast_is_z ? dl.dx : dl.dz;
    if (dl.octaves <= 8) {
        const long long plane = (long long)dfast * dslow;
        const long long n_plane_wgs = (plane + 255) / 256;
        const int n_yseg = (dl.dy + kColSeg - 1) / kColSeg;
        const int seg_len = (dl.dy + n_yseg - 1) / n_yseg;  // equal segments: every one pays the same two-face start-up
        const long long n_wgs = n_plane_wgs * n_yseg * dl.n_volumes;
        if (n_wgs <= 0 || n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
        if (dl.octaves == 1)
            hipLaunchKernelGGL((density_column_kernel<1>), dim3((unsigned)n_wgs), dim3(256), 0, stream, dl, d_perm, d_origins, d_out,
                               fast_is_z, (int)n_plane_wgs, n_yseg, seg_len);
        else
            hipLaunchKernelGGL((density_column_kernel<8>), dim3((unsigned)n_wgs), dim3(256), 0, stream, dl, d_perm, d_origins, d_out,
                               fast_is_z, (int)n_plane_wgs, n_yseg, seg_len);
        return hipGetLastError();
    }
    const int nseg = (dfast + 255) / 256;
    long long n_wgs = (long long)dl.n_volumes * dslow * dl.dy * nseg;
    if (n_wgs <= 0 || n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(density_generic_kernel, dim3((unsigned)n_wgs), dim3(256), 0, stream, dl, d_perm, d_origins, d_out,
                       fast_is_z, nseg);
    return hipGetLastError();
}

}  // namespace vtmc
