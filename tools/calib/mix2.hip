// mix2.hip -- what does the memory system deliver for a read : write mix?  (round 3; VERDICT r02 task 2)
//
// Round 2 quoted 4.4-4.9 TB/s for a "3 reads : 7 writes" stream from ONE untuned kernel (calib.hip mix_kernel: 4096 workgroups,
// one float4 per lane per step, plain stores) and called that the emit kernel's ceiling.  This program sweeps the shapes a tuned
// streaming kernel can take before any such number is quoted again:
//   grid      : persistent, k workgroups per CU (k = 1, 2, 4, 8) of 256 threads, each striding the whole buffer
//   in flight : U float4 loads issued per lane before their uses (U = 1, 2, 4, 8)
//   stores    : plain | nontemporal
//   mixes     : copy (1 : 1), read only, write only, r : w = 3 : 7 (the soup emit), 2 : 5, 1 : 3 (closer to the emit kernel after
//               row masks), and the emit kernel's own shape: 40-byte rows gathered + 76-byte records (19 dwords per lane) written
// Every variant is timed over `reps` launches with hipEvents; bytes are the algorithmic ones (reads + writes).
// Output: one JSON object per line.
//
// `mix2 <GiB> box` (round 6): the four numbers bench.py puts next to its kernels' rates -- what THIS box's memory delivers for a plain
// read stream, a plain write stream, a plain copy and a 4 : 7 read : write mix (the emit kernel moves 1.85 GB in and 3.26 GB out: 36 : 64),
// each the best of a few launch shapes, median of 5 launches; about 60 ms of kernels, one JSON line.
//
//   hipcc -O3 --offload-arch=gfx950 -o mix2 mix2.hip && ./mix2 [GiB of buffer, default 4]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// Units of (R + W) float4 per lane-step: the lane reads R float4 from `src` and writes W float4 to `dst`, both streams perfectly
// coalesced (consecutive lanes, consecutive float4).  U units are in flight per lane (loads first, then the stores).
template <int R, int W, int U, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n_units)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long u0 = (long long)blockIdx.x * 256 + threadIdx.x; u0 < n_units; u0 += stride * U) {
        v4f acc[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + j * stride;
            acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (u < n_units) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[j] += src[(long long)r * n_units + u];   // R read streams
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + j * stride;
            if (W == 0 && acc[j].x == 1.2345e30f) dst[0] = acc[j];   // read only: keeps the loads alive, never taken
            if (u < n_units) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    v4f v = acc[j];
                    v.x += (float)w;
                    v4f *p = dst + (long long)w * n_units + u;   // W write streams
                    if (NT) __builtin_nontemporal_store(v, p);
                    else *p = v;
                }
            }
        }
    }
}

// The same R : W mix with ONE read front and ONE write front per workgroup: unit u of a workgroup's step reads the R consecutive 4-KiB pieces
// (u R + r) of `src` and writes the W consecutive pieces (u W + w) of `dst` -- two address streams instead of mix_kernel's R + W.
template <int R, int W, int U>
__global__ __launch_bounds__(256) void mix_contig_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n_units)
{
    for (long long u0 = (long long)blockIdx.x * U; u0 < n_units; u0 += (long long)gridDim.x * U) {
        v4f acc[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (u0 + j < n_units) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[j] += src[((u0 + j) * R + r) * 256 + threadIdx.x];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (u0 + j < n_units) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    v4f v = acc[j];
                    v.x += (float)w;
                    dst[((u0 + j) * W + w) * 256 + threadIdx.x] = v;
                }
            }
        }
    }
}

// The emit kernel's own traffic shape without its arithmetic: a wave gathers `rows` 40-byte rows (10 lanes each, 5 rows per load
// instruction, rows at a 520-byte pitch like a 130-sample grid line) and writes `tris` 76-byte records as one contiguous stream of
// float4 (the staged form).  rows / tris per wave-step = 64 / 128: a tile after row masks and an average block's triangles.
__global__ __launch_bounds__(256) void emit_shape_kernel(const float *__restrict__ src, long long src_floats, v4f *__restrict__ dst,
                                                         long long n_steps)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * 256) >> 6;
    const int lq = lane % 10, rq = lane / 10;
    for (long long s = wave; s < n_steps; s += n_waves) {
        // 64 rows = 13 load instructions of 5 rows (50 lanes)
        const long long base = (s * 8192) % (src_floats - 70000);
        float v[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) v[i] = rq < 5 ? src[base + (long long)(5 * i + rq) * 130 + lq] : 0.f;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 13; ++i) a += v[i];
        // 128 records x 19 dwords = 608 float4 = 9.5 float4 per lane
        v4f *o = dst + s * 608;
        const v4f val = {a, a + 1.f, a + 2.f, a + 3.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) o[i * 64 + lane] = val;
        if (lane < 32) o[576 + lane] = val;
    }
}


// Write-only, in the emit kernel's own granularity: every wave writes whole chunks of `chunk_f4` float4 (a block's records: 608 for
// the average block) with back-to-back 1-KiB store instructions.  MODE 0: chunk = it * n_waves + wave (what a static round-robin over
// the active list does); MODE 1: the four waves of a workgroup write ONE chunk together (wave w the w-th KiB of every 4 KiB);
// MODE 2: like 0 but a dependent ~`spin`-instruction delay between two store instructions of a wave (stores trickle out, as they do
// between the staging rounds of a flush).
template <int MODE>
__global__ __launch_bounds__(256) void chunk_write_kernel(v4f *__restrict__ dst, long long n_chunks, int chunk_f4, int spin)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const v4f val = {1.f, 2.f, 3.f, (float)lane};
    if (MODE == 1) {
        for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
            v4f *o = dst + c * chunk_f4;
            for (int i = w * 64 + lane; i < chunk_f4; i += 256) o[i] = val;
        }
    } else {
        const long long wave = (long long)blockIdx.x * 4 + w, n_waves = (long long)gridDim.x * 4;
        float a = (float)lane;
        // MODE 3: the chunks in a scattered order (c -> c * odd mod 2^k over the largest power of two below n_chunks): every wave's next
        // chunk is far from its last and from its neighbours'.  MODE 4: `spin` sweep windows (the emit kernel's: 16), each an equal part of
        // the buffer, swept front to back by the waves w % windows == its index.
        long long pow2 = 1;
        while (pow2 * 2 <= n_chunks) pow2 *= 2;
        for (long long c0 = wave; c0 < (MODE == 3 ? pow2 : n_chunks); c0 += n_waves) {
            long long c = c0;
            if (MODE == 3) c = (c0 * 2654435761ll) & (pow2 - 1);
            if (MODE == 4) {
                const long long win = c0 % spin, k = c0 / spin, per = n_chunks / spin;
                if (k >= per) continue;
                c = win * per + k;
            }
            v4f *o = dst + c * chunk_f4;
            for (int i = lane; i < chunk_f4; i += 64) {
                if (MODE == 2) {
                    for (int k = 0; k < spin; ++k) a = __builtin_fmaf(a, 1.0000001f, 0.5f);
                    o[i] = (v4f){a, 2.f, 3.f, 4.f};
                } else {
                    o[i] = val;
                }
            }
        }
    }
}

template <int MODE>
static void run_chunks(const char *name, int wgs_per_cu, int n_cus, v4f *dst, long long total_f4, int chunk_f4, int spin, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long n_chunks = total_f4 / chunk_f4;
    std::vector<float> ms(reps);
    for (int i = 0; i < reps + 1; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((chunk_write_kernel<MODE>), dim3(wgs_per_cu * n_cus), dim3(256), 0, 0, dst, n_chunks, chunk_f4, spin);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    long long written = n_chunks;
    if (MODE == 3) {   // the scattered order covers the largest power of two of chunks
        written = 1;
        while (written * 2 <= n_chunks) written *= 2;
    }
    if (MODE == 4) written = (n_chunks / spin) * spin;
    const double bytes = (double)written * chunk_f4 * 16.0;
    printf("{\"kernel\": \"%s\", \"mode\": %d, \"chunk_bytes\": %d, \"spin\": %d, \"wgs_per_cu\": %d, \"GB\": %.3f, \"ms_med\": %.4f, \"TBps_med\": %.3f, "
           "\"TBps_best\": %.3f}\n", name, MODE, chunk_f4 * 16, spin, wgs_per_cu, bytes / 1e9, ms[reps / 2], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
    fflush(stdout);
}


// Read-only stream in the classify kernel's load shape: ONE dword per lane and instruction (a wave-instruction moves 256 bytes), U loads
// in flight per lane, rows `pitch` floats apart (130: a chunk's row pitch; 64: dense).  Is 4 bytes per lane the classify kernel's limit?
template <int U>
__global__ __launch_bounds__(256) void read_dword_kernel(const float *__restrict__ src, long long n_rows, int pitch, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * 256) >> 6;
    float acc = 0.f;
    for (long long r0 = wave * U; r0 < n_rows; r0 += n_waves * U) {
        float v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = r0 + j < n_rows ? src[(r0 + j) * pitch + lane] : 0.f;
#pragma unroll
        for (int j = 0; j < U; ++j) acc += v[j];
    }
    if (acc == 1.2345e30f) sink[0] = acc;
}

template <int U>
static void run_dword(int wgs_per_cu, int n_cus, const float *src, long long n_floats, int pitch, float *sink, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long n_rows = n_floats / pitch - 1;
    std::vector<float> ms(reps);
    for (int i = 0; i < reps + 1; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((read_dword_kernel<U>), dim3(wgs_per_cu * n_cus), dim3(256), 0, 0, src, n_rows, pitch, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)n_rows * 256.0, touched = (double)n_rows * pitch * 4.0;
    printf("{\"kernel\": \"read_dword\", \"in_flight\": %d, \"wgs_per_cu\": %d, \"pitch_floats\": %d, \"GB_loaded\": %.3f, \"ms_med\": %.4f, \"TBps_loaded\": %.3f, "
           "\"TBps_of_lines_touched\": %.3f}\n", U, wgs_per_cu, pitch, bytes / 1e9, ms[reps / 2], bytes / ms[reps / 2] / 1e9, touched / ms[reps / 2] / 1e9);
    fflush(stdout);
}

template <int R, int W, int U, bool NT>
static void run(const char *name, int wgs_per_cu, int n_cus, const v4f *src, v4f *dst, long long n_units, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = wgs_per_cu * n_cus;
    hipLaunchKernelGGL((mix_kernel<R, W, U, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<R, W, U, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)n_units * 16.0 * (R + W);
    printf("{\"kernel\": \"%s\", \"reads\": %d, \"writes\": %d, \"in_flight\": %d, \"nt\": %d, \"wgs_per_cu\": %d, \"GB\": %.3f, "
           "\"ms_med\": %.4f, \"ms_min\": %.4f, \"TBps_med\": %.3f, \"TBps_best\": %.3f}\n",
           name, R, W, U, (int)NT, wgs_per_cu, bytes / 1e9, ms[reps / 2], ms[0], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
    fflush(stdout);
}

// The plainest copy there is -- what a "float4 copy" micro-benchmark usually means: one float4 (or UNROLL of them, a grid-stride apart) per
// thread, the grid as large as the buffer, no persistence.  VERDICT r03 item 7b: the guide quotes 6.29 TB/s for a float4 copy, the
// persistent sweep above finds 5.45.
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_plain_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    v4f v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) v[j] = i + j * stride < n ? src[i + j * stride] : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
        if (i + j * stride < n) dst[i + j * stride] = v[j];
}

// the same plainness for one direction only
__global__ __launch_bounds__(256) void read_plain_kernel(const v4f *__restrict__ src, v4f *__restrict__ sink, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const v4f v = src[i];
        if (v.x == 1.2345e30f) sink[0] = v;   // never taken: keeps the load alive
    }
}
__global__ __launch_bounds__(256) void write_plain_kernel(v4f *__restrict__ dst, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (v4f){1.f, 2.f, 3.f, (float)threadIdx.x};
}

// How much does the NUMBER OF CONCURRENT FRONTS cost a read stream?  Workgroup b reads the 4-KiB piece (b % F) * (n_wgs / F) + b / F: F = 1 is the
// plain kernel (one front sweeping the buffer), F = 8 what "every XCD streams its own contiguous eighth" does, larger F what a brick order
// does that keeps ~100 sample planes open per XCD (classify_dense_kernel).  PIECE float4 per thread (1: 4 KiB per workgroup, the plain kernel).
__global__ __launch_bounds__(256) void read_fronts_kernel(const v4f *__restrict__ src, v4f *__restrict__ sink, long long n_wgs, int fronts)
{
    const long long b = blockIdx.x;
    const long long per = n_wgs / fronts;
    const long long piece = (b % fronts) * per + b / fronts;
    if (b / fronts < per) {
        const v4f v = src[piece * 256 + threadIdx.x];
        if (v.x == 1.2345e30f) sink[0] = v;
    }
}

// Time-division of the two directions (round 4 experiment): the mixes above lose 10-25 % against the weighted read-only / write-only rates.
// Is that the memory's read <-> write turn-around?  Here every wave issues its loads only inside the "read window" of a chip-wide clock
// (s_memrealtime, 100 MHz: the first `read_ticks` of every `period_ticks`) and its stores only outside it -- no communication, every wave
// reads the same clock.  R : W as mix_kernel; U units per lane and window pair.
template <int R, int W, int U>
__global__ __launch_bounds__(256) void phased_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, long long n_units, unsigned period_ticks,
                                                      unsigned read_ticks)
{
    const long long stride = (long long)gridDim.x * 256;
    auto wait_for = [&](bool want_read) {
        if (period_ticks == 0) return;
        for (;;) {
            const unsigned ph = (unsigned)(__builtin_amdgcn_s_memrealtime() % period_ticks);
            if ((ph < read_ticks) == want_read) break;
            __builtin_amdgcn_s_sleep(2);
        }
    };
    for (long long u0 = (long long)blockIdx.x * 256 + threadIdx.x; u0 - threadIdx.x - (long long)blockIdx.x * 256 < n_units; u0 += stride * U) {
        v4f acc[U];
        wait_for(true);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + j * stride;
            acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (u < n_units) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[j] += src[(long long)r * n_units + u];
            }
        }
        // the loads have to be back before the window test means anything: touch them
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < U; ++j) t += acc[j].x;
        if (t == 1.2345e30f) dst[0] = acc[0];
        wait_for(false);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + j * stride;
            if (u < n_units) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    v4f v = acc[j];
                    v.x += (float)w;
                    dst[(long long)w * n_units + u] = v;
                }
            }
        }
    }
}

template <int R, int W, int U>
static void run_phased(int wgs_per_cu, int n_cus, const v4f *src, v4f *dst, long long n_units, unsigned period, unsigned read_ticks, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms(reps);
    for (int i = 0; i < reps + 1; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((phased_kernel<R, W, U>), dim3(wgs_per_cu * n_cus), dim3(256), 0, 0, src, dst, n_units, period, read_ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)n_units * 16.0 * (R + W);
    printf("{\"kernel\": \"phased\", \"reads\": %d, \"writes\": %d, \"in_flight\": %d, \"wgs_per_cu\": %d, \"period_us\": %.1f, \"read_window_us\": %.1f, "
           "\"GB\": %.3f, \"ms_med\": %.4f, \"TBps_med\": %.3f, \"TBps_best\": %.3f}\n",
           R, W, U, wgs_per_cu, period / 100.0, read_ticks / 100.0, bytes / 1e9, ms[reps / 2], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    const long long total_f4 = (long long)(gib * (1ll << 30)) / 16;
    v4f *buf;
    if (getenv("MIX2_CONTIGUOUS"))   // round 6, profiles/r06/placement_probe.txt: physically contiguous memory instead of hipMalloc's fragments
        CK(hipExtMallocWithFlags((void **)&buf, total_f4 * 16, hipDeviceMallocContiguous));
    else
        CK(hipMalloc(&buf, total_f4 * 16));
    CK(hipMemset(buf, 0, total_f4 * 16));
    const int reps = 7;

    if (argc > 2 && !strcmp(argv[2], "box")) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int r5 = 5;
        auto median_ms = [&](auto launch) {
            std::vector<float> ms(r5);
            for (int i = 0; i < r5 + 1; ++i) {
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
            }
            std::sort(ms.begin(), ms.end());
            return (double)ms[r5 / 2];
        };
        const long long n = total_f4;
        const unsigned g1 = (unsigned)((n + 255) / 256);
        // plain read / write over the whole buffer, plain copy half -> half (one float4 per thread; and the persistent U = 4 shapes)
        double rd = (double)n * 16.0 / median_ms([&] { hipLaunchKernelGGL(read_plain_kernel, dim3(g1), dim3(256), 0, 0, buf, buf, n); }) / 1e9;
        rd = std::max(rd, (double)n * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<1, 0, 4, false>), dim3(4 * n_cus), dim3(256), 0, 0, buf, buf, n); }) / 1e9);
        double wr = (double)n * 16.0 / median_ms([&] { hipLaunchKernelGGL(write_plain_kernel, dim3(g1), dim3(256), 0, 0, buf, n); }) / 1e9;
        wr = std::max(wr, (double)n * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<0, 1, 4, false>), dim3(4 * n_cus), dim3(256), 0, 0, buf, buf, n); }) / 1e9);
        const long long h = n / 2;
        double cp = 2.0 * h * 16.0 / median_ms([&] { hipLaunchKernelGGL((copy_plain_kernel<1>), dim3((unsigned)((h + 255) / 256)), dim3(256), 0, 0, buf, buf + h, h); }) / 1e9;
        cp = std::max(cp, 2.0 * h * 16.0 / median_ms([&] { hipLaunchKernelGGL((copy_plain_kernel<4>), dim3((unsigned)((h + 1023) / 1024)), dim3(256), 0, 0, buf, buf + h, h); }) / 1e9);
        const long long u = n / 11;   // 4 read streams + 7 write streams of u float4 each
        double mx = 0.0;
        mx = std::max(mx, 11.0 * u * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<4, 7, 2, false>), dim3(4 * n_cus), dim3(256), 0, 0, buf, buf + 4 * u, u); }) / 1e9);
        mx = std::max(mx, 11.0 * u * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<4, 7, 4, false>), dim3(4 * n_cus), dim3(256), 0, 0, buf, buf + 4 * u, u); }) / 1e9);
        mx = std::max(mx, 11.0 * u * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<4, 7, 4, false>), dim3(2 * n_cus), dim3(256), 0, 0, buf, buf + 4 * u, u); }) / 1e9);
        mx = std::max(mx, 11.0 * u * 16.0 / median_ms([&] { hipLaunchKernelGGL((mix_kernel<4, 7, 8, false>), dim3(2 * n_cus), dim3(256), 0, 0, buf, buf + 4 * u, u); }) / 1e9);
        {   // one read front + one write front per workgroup: units of 4 + 7 pieces of 4 KiB
            const long long nu = n / (11 * 256);
            mx = std::max(mx, 11.0 * nu * 4096.0 / median_ms([&] { hipLaunchKernelGGL((mix_contig_kernel<4, 7, 1>), dim3(8 * n_cus), dim3(256), 0, 0, buf, buf + 4 * nu * 256, nu); }) / 1e9);
            mx = std::max(mx, 11.0 * nu * 4096.0 / median_ms([&] { hipLaunchKernelGGL((mix_contig_kernel<4, 7, 2>), dim3(4 * n_cus), dim3(256), 0, 0, buf, buf + 4 * nu * 256, nu); }) / 1e9);
            mx = std::max(mx, 11.0 * nu * 4096.0 / median_ms([&] { hipLaunchKernelGGL((mix_contig_kernel<4, 7, 1>), dim3((unsigned)nu), dim3(256), 0, 0, buf, buf + 4 * nu * 256, nu); }) / 1e9);
        }
        // the emit kernel's own access shape (40-byte rows gathered, 76-byte records streamed), without its arithmetic
        const long long dst_f4 = n * 3 / 4, src_floats = (n - dst_f4) * 4, n_steps = dst_f4 / 608;
        const float *src = reinterpret_cast<const float *>(buf + dst_f4);
        double es = 0.0;
        for (int per_cu : {4, 8})
            es = std::max(es, (double)n_steps * (608.0 * 16 + 64 * 40) / median_ms([&] { hipLaunchKernelGGL(emit_shape_kernel, dim3(per_cu * n_cus), dim3(256), 0, 0, src, src_floats, buf, n_steps); }) / 1e9);
        printf("{\"kernel\": \"box\", \"GiB\": %.2f, \"read_TBps\": %.3f, \"write_TBps\": %.3f, \"copy_TBps\": %.3f, \"mix_4r7w_TBps\": %.3f, \"emit_shape_TBps\": %.3f, "
               "\"cus\": %d, \"device\": \"%s\"}\n", gib, rd, wr, cp, mx, es, n_cus, prop.gcnArchName);
        fflush(stdout);
        CK(hipFree(buf));
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "dword")) {
        const float *src = reinterpret_cast<const float *>(buf);
        float *sink = reinterpret_cast<float *>(buf);
        const long long n_floats = total_f4 * 4;
        for (int pitch : {64, 65, 130}) {
            for (int per_cu : {8, 4, 3, 2}) {
                run_dword<1>(per_cu, n_cus, src, n_floats, pitch, sink, reps);
                run_dword<9>(per_cu, n_cus, src, n_floats, pitch, sink, reps);
                run_dword<27>(per_cu, n_cus, src, n_floats, pitch, sink, reps);
                run_dword<81>(per_cu, n_cus, src, n_floats, pitch, sink, reps);
            }
        }
        CK(hipFree(buf));
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "copy")) {
        // plain copies over buffer sizes: half the buffer is the source, half the destination
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (double g : {0.125, 0.5, 2.0, gib}) {
            const long long n = (long long)(g * (1ll << 30)) / 32;   // float4 per half
            if (2 * n > total_f4) continue;
            const v4f *src = buf;
            v4f *dst = buf + n;
            for (int variant = 0; variant < 6; ++variant) {
                std::vector<float> ms(reps);
                for (int i = 0; i < reps + 1; ++i) {
                    CK(hipEventRecord(e0));
                    if (variant == 0) hipLaunchKernelGGL((copy_plain_kernel<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, dst, n);
                    else if (variant == 1) hipLaunchKernelGGL((copy_plain_kernel<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, src, dst, n);
                    else if (variant == 2) hipLaunchKernelGGL((copy_plain_kernel<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, src, dst, n);
                    else if (variant == 3) CK(hipMemcpyAsync(dst, src, n * 16, hipMemcpyDeviceToDevice, 0));
                    else if (variant == 4) hipLaunchKernelGGL(read_plain_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, 0, src, dst, 2 * n);
                    else hipLaunchKernelGGL(write_plain_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, 0, buf, 2 * n);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
                }
                std::sort(ms.begin(), ms.end());
                const double bytes = 2.0 * n * 16.0;
                const char *names[6] = {"copy_plain_1", "copy_plain_4", "copy_plain_8", "hipMemcpyDtoD", "read_plain_1", "write_plain_1"};
                printf("{\"kernel\": \"%s\", \"GiB_total\": %.3f, \"GB_moved\": %.3f, \"ms_med\": %.4f, \"TBps_med\": %.3f, \"TBps_best\": %.3f}\n", names[variant], g,
                       bytes / 1e9, ms[reps / 2], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
                fflush(stdout);
            }
        }
        CK(hipFree(buf));
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "fronts")) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const long long n_wgs = total_f4 / 256;
        for (int fronts : {1, 8, 16, 64, 128, 512, 1024, 4096, 32768}) {
            std::vector<float> ms(reps);
            for (int i = 0; i < reps + 1; ++i) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(read_fronts_kernel, dim3((unsigned)n_wgs), dim3(256), 0, 0, buf, buf, n_wgs, fronts);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
            }
            std::sort(ms.begin(), ms.end());
            const double bytes = (double)(n_wgs / fronts) * fronts * 4096.0;
            printf("{\"kernel\": \"read_fronts\", \"fronts\": %d, \"GB\": %.3f, \"ms_med\": %.4f, \"TBps_med\": %.3f, \"TBps_best\": %.3f}\n", fronts, bytes / 1e9,
                   ms[reps / 2], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
            fflush(stdout);
        }
        CK(hipFree(buf));
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "phase")) {
        // 3 : 7 (the soup emit's direction mix) and 1 : 1, free-running (period 0) against time-divided; windows in proportion to the bytes
        {
            const long long n_units = total_f4 / 10;
            const v4f *src = buf;
            v4f *dst = buf + 3 * n_units;
            for (int per_cu : {4, 8}) {
                run_phased<3, 7, 8>(per_cu, n_cus, src, dst, n_units, 0, 0, reps);
                for (unsigned period : {2000u, 4000u, 8000u, 16000u}) run_phased<3, 7, 8>(per_cu, n_cus, src, dst, n_units, period, period * 3 / 10, reps);
            }
        }
        {
            const long long n_units = total_f4 / 2;
            const v4f *src = buf;
            v4f *dst = buf + n_units;
            for (int per_cu : {4, 8}) {
                run_phased<1, 1, 8>(per_cu, n_cus, src, dst, n_units, 0, 0, reps);
                for (unsigned period : {1000u, 2000u, 4000u, 8000u}) run_phased<1, 1, 8>(per_cu, n_cus, src, dst, n_units, period, period / 2, reps);
            }
        }
        CK(hipFree(buf));
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "chunks")) {
        const long long f4 = (3ll << 30) / 16;   // 3 GiB written per launch (the emit kernel writes 3.23 GB)
        for (int per_cu : {4, 3, 2}) {
            for (int chunk : {152, 304, 608, 1216, 2432, 9728}) run_chunks<0>("chunk_write", per_cu, n_cus, buf, f4, chunk, 0, reps);
            for (int chunk : {608, 2432}) run_chunks<1>("chunk_write_wg", per_cu, n_cus, buf, f4, chunk, 0, reps);
            for (int spin : {100, 400, 1600}) run_chunks<2>("chunk_write_trickle", per_cu, n_cus, buf, f4, 608, spin, reps);
            run_chunks<3>("chunk_write_scattered", per_cu, n_cus, buf, f4, 608, 0, reps);
            for (int windows : {8, 16, 128}) run_chunks<4>("chunk_write_windows", per_cu, n_cus, buf, f4, 608, windows, reps);
        }
        CK(hipFree(buf));
        return 0;
    }
#define SWEEP(R, W, NAME)                                                                     \
    {                                                                                         \
        const long long n_units = total_f4 / (R + W);                                         \
        const v4f *src = buf;                                                                 \
        v4f *dst = buf + (long long)R * n_units;                                              \
        run<R, W, 1, false>(NAME, 8, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 2, false>(NAME, 4, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 4, false>(NAME, 4, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 4, false>(NAME, 2, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 8, false>(NAME, 2, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 8, false>(NAME, 1, n_cus, src, dst, n_units, reps);                         \
        run<R, W, 4, true>(NAME, 4, n_cus, src, dst, n_units, reps);                          \
        run<R, W, 8, true>(NAME, 2, n_cus, src, dst, n_units, reps);                          \
    }
    SWEEP(1, 1, "copy")
    SWEEP(1, 0, "read")
    SWEEP(0, 1, "write")
    SWEEP(3, 7, "mix3:7")
    SWEEP(2, 5, "mix2:5")
    SWEEP(1, 3, "mix1:3")
    SWEEP(1, 4, "mix1:4")
    {   // the emit kernel's own shape
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const long long dst_f4 = total_f4 * 3 / 4, src_floats = (total_f4 - dst_f4) * 4;
        const long long n_steps = dst_f4 / 608;
        const float *src = reinterpret_cast<const float *>(buf + dst_f4);
        for (int per_cu : {2, 4, 8}) {
            std::vector<float> ms(reps);
            for (int i = 0; i < reps + 1; ++i) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(emit_shape_kernel, dim3(per_cu * n_cus), dim3(256), 0, 0, src, src_floats, buf, n_steps);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                if (i) CK(hipEventElapsedTime(&ms[i - 1], e0, e1));
            }
            std::sort(ms.begin(), ms.end());
            const double bytes = (double)n_steps * (608.0 * 16 + 64 * 40);
            printf("{\"kernel\": \"emit_shape\", \"wgs_per_cu\": %d, \"GB\": %.3f, \"ms_med\": %.4f, \"TBps_med\": %.3f, \"TBps_best\": %.3f}\n", per_cu,
                   bytes / 1e9, ms[reps / 2], bytes / ms[reps / 2] / 1e9, bytes / ms[0] / 1e9);
        }
    }
    CK(hipFree(buf));
    return 0;
}
