// occupancy probe: how many workgroups of a given size and static LDS does the runtime place on a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *o)
{
    __shared__ float s[BYTES / 4];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    o[threadIdx.x] = s[(threadIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES, int THREADS>
void probe()
{
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES, THREADS>, THREADS, 0);
    printf("LDS %6d B, %4d threads: %d workgroups per CU = %d waves\n", BYTES, THREADS, n, n * THREADS / 64);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxThreadsPerMultiProcessor %d\n", p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, p.maxThreadsPerMultiProcessor);
    probe<35776, 256>();
    probe<39872, 256>();
    probe<53504, 256>();
    probe<52256, 384>();
    probe<52256, 256>();
    probe<49152, 384>();
    probe<40960, 384>();
    probe<32768, 384>();
    probe<65536, 512>();
    probe<80096, 640>();
    probe<65536, 640>();
    return 0;
}
