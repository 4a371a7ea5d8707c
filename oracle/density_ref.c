/*
 * density_ref.c -- CPU twin of the synthetic density samplers (perlin3d, fbm8).
 *
 * TEST INFRASTRUCTURE ONLY.  There is no Perlin / fBm density in the reference;
 * these fields are the build's own synthetic inputs, defined in SURVEY.md 8d / DESIGN.md:
 *   - Ken Perlin's 2002 "improved noise", 256-entry permutation from a Fisher-Yates shuffle driven
 *     by SplitMix64(seed);
 *   - density(i,j,k) = sum_{o<octaves} gain^o * noise(p * f * lacunarity^o) - (p.y - ramp_center) * ramp_scale
 *     with p = (origin + (i,j,k)) as float, all arithmetic FP32.
 * The GPU sampler (volumetricterrain_amd/csrc/density.hip) implements the same definition; parity
 * tests never rely on CPU/GPU noise equality -- both extractors are always fed the same array.
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint64_t seed;
    float frequency;   /* f: perlin3d 8/N, fbm8 4/N */
    int32_t octaves;   /* 1 or 8 */
    float lacunarity;  /* 2 */
    float gain;        /* 0.5 */
    float ramp_scale;  /* fbm8: 2/N, perlin3d: 0 */
    float ramp_center; /* fbm8: N/2 */
} vto_density_params;

static uint64_t splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void vto_density_permutation(uint64_t seed, uint8_t perm[256])
{
    for (int i = 0; i < 256; ++i) perm[i] = (uint8_t)i;
    uint64_t s = seed;
    for (int i = 255; i >= 1; --i) {
        int j = (int)(splitmix64(&s) % (uint64_t)(i + 1));
        uint8_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
}

static inline float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
static inline float mix(float t, float a, float b) { return a + t * (b - a); }
static inline float grad(int hash, float x, float y, float z)
{
    int h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

static float noise3(const uint8_t *p, float x, float y, float z)
{
    float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
    x -= fx;
    y -= fy;
    z -= fz;
    float u = fade(x), v = fade(y), w = fade(z);
#define P(i) ((int)p[(i) & 255])
    int A = P(X) + Y, AA = P(A) + Z, AB = P(A + 1) + Z;
    int B = P(X + 1) + Y, BA = P(B) + Z, BB = P(B + 1) + Z;
    float r = mix(w,
                  mix(v, mix(u, grad(P(AA), x, y, z), grad(P(BA), x - 1, y, z)),
                      mix(u, grad(P(AB), x, y - 1, z), grad(P(BB), x - 1, y - 1, z))),
                  mix(v, mix(u, grad(P(AA + 1), x, y, z - 1), grad(P(BA + 1), x - 1, y, z - 1)),
                      mix(u, grad(P(AB + 1), x, y - 1, z - 1), grad(P(BB + 1), x - 1, y - 1, z - 1))));
#undef P
    return r;
}

/* Fill a (dx,dy,dz)-sample volume whose sample (0,0,0) sits at global sample index (ox,oy,oz);
 * out[i*sx + j*sy + k*sz]. */
void vto_density_fill_threads(const vto_density_params *prm, int32_t ox, int32_t oy, int32_t oz,
                              int32_t dx, int32_t dy, int32_t dz, int64_t sx, int64_t sy, int64_t sz, float *out, int32_t n_threads);

void vto_density_fill(const vto_density_params *prm, int32_t ox, int32_t oy, int32_t oz,
                      int32_t dx, int32_t dy, int32_t dz, int64_t sx, int64_t sy, int64_t sz, float *out)
{
    vto_density_fill_threads(prm, ox, oy, oz, dx, dy, dz, sx, sy, sz, out, 0);
}

/* n_threads: the team of the fill (0: the OpenMP default) -- bench.py's CPU leg of the streaming config states its thread count. */
void vto_density_fill_threads(const vto_density_params *prm, int32_t ox, int32_t oy, int32_t oz,
                              int32_t dx, int32_t dy, int32_t dz, int64_t sx, int64_t sy, int64_t sz, float *out, int32_t n_threads)
{
    uint8_t perm[256];
    vto_density_permutation(prm->seed, perm);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads) if (n_threads > 1)
    for (int32_t k = 0; k < dz; ++k)
        for (int32_t j = 0; j < dy; ++j)
            for (int32_t i = 0; i < dx; ++i) {
                float px = (float)(ox + i), py = (float)(oy + j), pz = (float)(oz + k);
                float x = px * prm->frequency, y = py * prm->frequency, z = pz * prm->frequency;
                float amp = 1.0f, sum = 0.0f;
                for (int o = 0; o < prm->octaves; ++o) {
                    sum = sum + amp * noise3(perm, x, y, z);
                    x *= prm->lacunarity;
                    y *= prm->lacunarity;
                    z *= prm->lacunarity;
                    amp *= prm->gain;
                }
                out[i * sx + j * sy + k * sz] = sum - (py - prm->ramp_center) * prm->ramp_scale;
            }
}
