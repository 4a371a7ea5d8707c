// do 3 workgroups of 384 threads with 52 256 B of static LDS really run together on every CU?  each workgroup counts itself in, waits, counts out
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int *running, int *peak, float *o)
{
    __shared__ float s[BYTES / 4];
    s[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = atomicAdd(running, 1) + 1;
        atomicMax(peak, r);
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(8);   // 100 MHz clock: 200 us
        atomicSub(running, 1);
    }
    __syncthreads();
    if (s[(threadIdx.x * 7) % (BYTES / 4)] == -1.f) o[0] = 1.f;
}
template <int BYTES, int THREADS>
void probe(int per_cu)
{
    int *d, h[2] = {0, 0};
    float *o;
    hipMalloc(&d, 8);
    hipMalloc(&o, 4);
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<BYTES, THREADS>), dim3(256 * per_cu), dim3(THREADS), 0, 0, d, d + 1, o);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("LDS %6d B, %4d threads, grid %4d: peak concurrent workgroups %d (%.2f per CU)\n", BYTES, THREADS, 256 * per_cu, h[1], h[1] / 256.0);
    hipFree(d);
    hipFree(o);
}
int main()
{
    probe<35776, 256>(4);
    probe<53504, 256>(3);
    probe<52256, 384>(3);
    probe<40960, 384>(4);
    probe<32768, 384>(5);
    probe<52256, 320>(3);
    probe<26832, 192>(6);
    probe<26832, 192>(7);
    probe<18272, 128>(9);
    probe<68736, 512>(2);
    probe<80096, 640>(2);
    return 0;
}
