#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f va = {a, a * 1.01f}, vb = {b, b * 0.99f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 scalar fmas
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else {           // 4 packed fmas = the same 8 flops-pairs
            p0 = __builtin_elementwise_fma(p0, va, vb); p1 = __builtin_elementwise_fma(p1, va, vb);
            p2 = __builtin_elementwise_fma(p2, va, vb); p3 = __builtin_elementwise_fma(p3, va, vb);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = MODE == 0 ? x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 : p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, wgs = 4096;
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
        else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fmas = (double)wgs * 256 * iters * 8;
        printf("mode %d (%s): %.3f ms  %.2f T fma-lanes/s\n", mode, mode ? "4 x v_pk_fma_f32" : "8 x v_fma_f32", ms, fmas / (ms * 1e-3) / 1e12);
    }
    return 0;
}
