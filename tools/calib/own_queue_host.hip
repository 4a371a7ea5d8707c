// A plain C++ / HIP host (ROCm's own runtime, no PyTorch) that drives two contexts the way bench.py and streaming.ChunkStream do: each
// context's sampler and extract queued on the context's OWN-QUEUE stream (vtmc_context_stream(ctx, 1, &s): hipExtStreamCreateWithCUMask inside
// the library), two steps in flight, results compared with the same work on the ordinary stream -- and then the process must EXIT.
// Round 5 saw a C++ host whose context used such a stream for everything, pinned staging included, hang in the runtime's tear-down
// (profiles/r05/stream_overlap.txt); the library has kept its own pinned copies off that stream since.  This program is the check that the
// documented pattern -- kernels on the own-queue stream, the library's staging on its ordinary one -- runs and exits under this runtime.
//   hipcc --offload-arch=gfx950 -I include -o own_queue_host tools/calib/own_queue_host.hip -L volumetricterrain_amd -lvtmc -Wl,-rpath,$PWD/volumetricterrain_amd
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vtmc.h"

#define CHECK(x)                                                                       \
    do {                                                                               \
        const int32_t rc_ = (x);                                                       \
        if (rc_ != VTMC_OK) {                                                          \
            std::fprintf(stderr, "%s -> %d (%s)\n", #x, (int)rc_, vtmc_last_error(nullptr)); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

int main()
{
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);
    std::printf("HIP runtime %d\n", ver);
    const int c = 128, dim = c + 2, n_vol = 8, rounds = 6;
    const size_t vol = (size_t)dim * dim * dim;
    vtmc_ctx *ctx[2] = {nullptr, nullptr};
    void *own[2] = {nullptr, nullptr};
    float *d_buf[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
        CHECK(vtmc_create(0, &ctx[i]));
        CHECK(vtmc_context_stream(ctx[i], 1, &own[i]));
        CHECK(vtmc_set_tuning(ctx[i], "fill_keeps_signs", 1));
        if (hipMalloc(&d_buf[i], n_vol * vol * sizeof(float)) != hipSuccess) return 3;
    }
    vtmc_density_params prm;
    prm.seed = 1337;
    prm.frequency = 4.0f / 2048.0f;
    prm.octaves = 8;
    prm.lacunarity = 2.0f;
    prm.gain = 0.5f;
    prm.ramp_scale = 2.0f / 2048.0f;
    prm.ramp_center = 1024.0f;
    std::vector<int32_t> origins((size_t)3 * n_vol);
    long long total[2] = {0, 0};   // [0]: ordinary stream, [1]: own-queue streams, two steps in flight
    for (int mode = 0; mode < 2; ++mode) {
        int64_t T = 0;
        bool pending[2] = {false, false};
        for (int k = 0; k < rounds; ++k) {
            const int s = k & 1;
            void *st = mode ? own[s] : nullptr;
            if (pending[s]) {   // the buffer's previous extract: its T, before the buffer is refilled
                CHECK(vtmc_extract_finish(ctx[s], &T));
                total[mode] += T;
                pending[s] = false;
            }
            for (int v = 0; v < n_vol; ++v) {
                origins[3 * v] = 128 * ((k * n_vol + v) % 16);
                origins[3 * v + 1] = 896 + 128 * (((k * n_vol + v) / 16) % 2);
                origins[3 * v + 2] = 128 * ((k * n_vol + v) / 32);
            }
            CHECK(vtmc_density_fill_device_async(ctx[s], &prm, origins.data(), n_vol, dim, dim, dim, 1, dim, (int64_t)dim * dim, (int64_t)vol, d_buf[s], st));
            vtmc_volume_batch b;
            b.d_samples = d_buf[s];
            b.nx = b.ny = b.nz = c;
            b.stride_x = 1;
            b.stride_y = dim;
            b.stride_z = (int64_t)dim * dim;
            b.n_volumes = n_vol;
            b.volume_stride = (int64_t)vol;
            CHECK(vtmc_extract_volumes_device_async(ctx[s], &b, st, 0));
            pending[s] = true;
        }
        for (int s = 0; s < 2; ++s)
            if (pending[s]) {
                CHECK(vtmc_extract_finish(ctx[s], &T));
                total[mode] += T;
            }
        std::printf("%s: %lld triangles over %d batches\n", mode ? "own-queue streams, two in flight" : "ordinary stream", total[mode], rounds);
    }
    // a read-back through the library (its ordinary stream) of what the last own-queue step left
    const vtmc_triangle *d_tris = nullptr;
    const uint32_t *d_offs = nullptr;
    const uint32_t *d_vc = nullptr;
    CHECK(vtmc_device_results(ctx[1], &d_tris, &d_offs, &d_vc));
    std::vector<uint32_t> vc((size_t)2 * n_vol);
    CHECK(vtmc_copy_to_host(ctx[1], d_vc, vc.data(), (int64_t)(vc.size() * sizeof(uint32_t)), nullptr));
    unsigned long long last = 0;
    for (int v = 0; v < n_vol; ++v) last += vc[(size_t)2 * v + 1];
    std::printf("last batch of context 1: %llu triangles by its per-volume counts\n", last);
    for (int i = 0; i < 2; ++i) {
        CHECK(vtmc_destroy(ctx[i]));
        (void)hipFree(d_buf[i]);
    }
    if (total[0] != total[1] || total[0] <= 0 || last == 0) {
        std::fprintf(stderr, "MISMATCH %lld / %lld\n", total[0], total[1]);
        return 1;
    }
    std::printf("OWN-QUEUE-HOST-OK\n");
    return 0;   // ... and the runtime's tear-down must let the process go
}
