// density.hip -- synthetic density samplers on the GPU (perlin3d / fbm8).
//
// The reference fills its density grid on the CPU, one virtual QueryDensity call per sample
// (VoxelTerrain.cs:284-305), and holds no noise field of its own (its only noise modifier wraps the
// un-vendored LibNoise, TerrainModifier.cs:158-196).  These kernels are the build's "density-field
// sampler" stage for the benchmark grids: Ken Perlin's 2002 improved noise with a 256-entry
// permutation from a SplitMix64-driven Fisher-Yates shuffle, summed over octaves, minus a vertical
// ramp.  Pure VALU + a 512-byte LDS table; writes are lane-contiguous along the stride-1 axis.
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

// one workgroup = 256 consecutive samples along the fast axis; blockIdx.x enumerates
// (segment, y, slow-axis index, volume) -- flattened because grid.y/z stop at 65535
__global__ __launch_bounds__(256) void density_kernel(DensityLaunch dl, const unsigned char *__restrict__ perm,
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

hipError_t launch_density(const DensityLaunch &dl, const unsigned char *d_perm, const int *d_origins,
                          float *d_out, hipStream_t stream)
{
    const int fast_is_z = (dl.sz == 1 && dl.sx != 1) ? 1 : 0;
    const int dfast = fast_is_z ? dl.dz : dl.dx;
    const int dslow = fast_is_z ? dl.dx : dl.dz;
    const int nseg = (dfast + 255) / 256;
    long long n_wgs = (long long)dl.n_volumes * dslow * dl.dy * nseg;
    if (n_wgs <= 0 || n_wgs > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(density_kernel, dim3((unsigned)n_wgs), dim3(256), 0, stream, dl, d_perm, d_origins, d_out,
                       fast_is_z, nseg);
    return hipGetLastError();
}

}  // namespace vtmc
