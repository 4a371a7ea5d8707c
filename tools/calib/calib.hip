// calib.hip -- two calibration kernels for the numbers bench.py and DESIGN.md quote (run on the GPU box):
//   rows : gathers 40-byte rows (10 floats, the shape of the emit kernel's tile rows: 5 rows per
//          wave-instruction, rows 520 bytes apart, slabs 67 600 bytes apart) out of a buffer larger than
//          the 256 MiB Infinity Cache, every row exactly once -- a KNOWN byte count for the FETCH_SIZE
//          counter on this access width (MI355X_MICROARCH.md: "other access widths are uncalibrated");
//   mix  : a perfectly coalesced stream that reads 3 and writes 7 of every 10 float4 (the emit kernel's
//          read : write mix, 1.45 GB : 3.38 GB) -- what the memory system gives that mix at best.
// usage: calib rows|mix [MiB]      prints one JSON line; under rocprofv3 --pmc FETCH_SIZE the first gives
//                                  the counter / known-bytes factor.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// one wave per "block" of 10 x 10 x 10 samples at (8 bx, 8 by, 8 bz) of a volume of dim^3 floats:
// 20 instructions x 5 rows x 10 lanes, exactly the emit kernel's tile fetch (emit_kernels.hip)
__global__ __launch_bounds__(256) void rows_kernel(const float *__restrict__ vol, int dim, int nb, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_blocks = nb * nb * nb;
    const int lq = lane % 10, rq = lane / 10;
    float acc = 0.f;
    for (int b = wave; b < n_blocks; b += gridDim.x * 4) {
        const int bx = b % nb, by = (b / nb) % nb, bz = b / (nb * nb);
        // blocks 10 apart: rows of neighbouring blocks never share a sample, every fetched row is new
        const float *org = vol + (size_t)(10 * bx) + (size_t)dim * (10 * by) + (size_t)dim * dim * (10 * bz);
        if (rq < 5) {
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                acc += org[lq + (size_t)dim * rq + (size_t)dim * dim * c];
                acc += org[lq + (size_t)dim * (rq + 5) + (size_t)dim * dim * c];
            }
        }
    }
    if (acc == 1.2345e30f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void mix_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n_iter)
{
    const size_t T = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t j = 0; j < n_iter; ++j) {
        float4 a = in[(j * 3 + 0) * T + t], b = in[(j * 3 + 1) * T + t], c = in[(j * 3 + 2) * T + t];
        float4 s = make_float4(a.x + b.x, a.y + b.y, c.z, c.w);
        out[(j * 7 + 0) * T + t] = a;
        out[(j * 7 + 1) * T + t] = b;
        out[(j * 7 + 2) * T + t] = c;
        out[(j * 7 + 3) * T + t] = s;
        out[(j * 7 + 4) * T + t] = a;
        out[(j * 7 + 5) * T + t] = b;
        out[(j * 7 + 6) * T + t] = c;
    }
}

int main(int argc, char **argv)
{
    const char *what = argc > 1 ? argv[1] : "rows";
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (!strcmp(what, "rows")) {
        const int dim = 1040, nb = 104;   // 1040^3 floats = 4.5 GB, 104^3 disjoint 10^3 tiles
        float *vol, *sink;
        CK(hipMalloc(&vol, sizeof(float) * (size_t)dim * dim * dim));
        CK(hipMalloc(&sink, 64));
        CK(hipMemset(vol, 0, sizeof(float) * (size_t)dim * dim * dim));
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rows_kernel, dim3(256 * 8), dim3(256), 0, 0, vol, dim, nb, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double rows = 100.0 * nb * nb * nb, bytes = rows * 40.0;
        printf("{\"kernel\": \"rows_kernel\", \"rows\": %.0f, \"known_bytes\": %.0f, \"ms\": %.4f, \"GBps_of_row_bytes\": %.1f, "
               "\"note\": \"every 40-byte row read exactly once; compare with FETCH_SIZE (KiB) of the same dispatch\"}\n",
               rows, bytes, best, bytes / best / 1e6);
    } else {
        const size_t T = 256 * 256 * 16, n_iter = 12;   // 1M threads: 3 x 16 MB read, 7 x 16 MB written per iteration
        float4 *in, *out;
        CK(hipMalloc(&in, sizeof(float4) * T * 3 * n_iter));
        CK(hipMalloc(&out, sizeof(float4) * T * 7 * n_iter));
        CK(hipMemset(in, 0, sizeof(float4) * T * 3 * n_iter));
        float best = 1e30f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mix_kernel, dim3(T / 256), dim3(256), 0, 0, in, out, n_iter);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double rd = 16.0 * T * 3 * n_iter, wr = 16.0 * T * 7 * n_iter;
        printf("{\"kernel\": \"mix_kernel\", \"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GBps_total\": %.1f}\n", rd, wr,
               best, (rd + wr) / best / 1e6);
    }
    return 0;
}
