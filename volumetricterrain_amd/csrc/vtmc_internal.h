// vtmc_internal.h -- shared between the HIP kernels and the C-ABI host layer (not installed).
#ifndef VTMC_INTERNAL_H
#define VTMC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vtmc {

// Division by a launch-constant via multiply-high + shifts (Granlund-Montgomery, exact for every
// 32-bit numerator): block ids are decomposed per block on the device, a hardware-less integer
// divide would cost ~40 VALU instructions each.
struct FastDiv {
    unsigned mul = 1, sh1 = 0, sh2 = 0, div = 1;
    FastDiv() = default;
    explicit FastDiv(unsigned d) : div(d)
    {
        unsigned l = 0;
        while ((1ull << l) < d) ++l;
        mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        sh1 = l < 1 ? l : 1;
        sh2 = l > 0 ? l - 1 : 0;
    }
#ifdef __HIPCC__
    __device__ __forceinline__ unsigned quot(unsigned n) const
    {
        unsigned t = __umulhi(mul, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
#endif
};

// How kernels find the 10x10x10 sample tile of block b.
//  dense: b = v*bpv + bx + nbx*(by + nby*bz)  -> origin = v*sv + 8*(bx*sx + by*sy + bz*sz)
//  list : (bx,by,bz) = list[3b..3b+2] (the dirty list, VoxelTerrain.cs:321), single volume
struct BlockSpace {
    const float *base;
    long long sx, sy, sz, sv;  // element strides
    int nbx, nby, nbz;         // blocks per volume
    int bpv;                   // nbx*nby*nbz
    int n_blocks;              // total blocks submitted
    const int *list;           // device, n_blocks x 3, or nullptr
    int zfast;                 // sz == 1: tile loads walk z fastest (C# float[,,] layout)
    int nx;                    // cells along x per volume (dense classify)
    FastDiv d_bpv, d_nbx, d_nby;
};

// What the emit kernel needs to know about a non-empty block, written by the scan in list order: ONE 32-byte scalar load per block instead
// of a chain of four (list entry -> row mask, two offsets) plus the block-id arithmetic.
struct __attribute__((aligned(32))) BlockDesc {
    uint32_t b;          // block id (the record's `block` field, VoxelTerrain.cs:35)
    uint32_t tri_base;   // first triangle of the block in the output
    uint32_t cnt_mask;   // triangles (<= 2560) | row mask << 16 (counts[b] as the classify pass left it)
    uint32_t vert_base;  // indexed output: first welded vertex
    long long origin;    // element offset of the block's 10^3 tile from BlockSpace::base
    uint32_t vert_cnt;   // indexed output: welded vertices
    uint32_t pad;
};
static_assert(sizeof(BlockDesc) == 32, "one s_load_dwordx8");

struct DeviceTables {
    const unsigned long long *vert_packed;  // 256 x u64 (mc_tables_packed.h)
    const unsigned char *tri_num;           // 256 x u8
};

// The sign volume a z-walk density fill can leave for the classify stage of the same, unmodified buffer (density.hip, "fill_keeps_signs")
struct SignVolume {
    const unsigned long long *words = nullptr;   // [(volume * dz + z) * plane_words + (x + dx * y) / 64], bit (x + dx * y) % 64: sample > 0
    int plane_words = 0, dx = 0, dz = 0;
};

// The *_ablate keys switch parts of a kernel off (the output is then INVALID): they exist in diagnostic builds only
// (-DVTMC_DIAGNOSTICS: tools/build_diagnostics.py, loaded through VTMC_LIB); the product library compiles every such branch
// out of its kernels and vtmc_set_tuning answers VTMC_ERR_INVALID_ARG for the keys.
#ifdef VTMC_DIAGNOSTICS
#define VTMC_ABLATE(v) (v)
#else
#define VTMC_ABLATE(v) 0
#endif

// A/B knobs (vtmc_set_tuning); defaults are the shipped configuration.  Every key the product library accepts is compared with the oracle
// by tests/test_tuning_matrix.py.
struct Tuning {
    int emit_fast_math = 1;   // 1: v_rcp/v_rsq (<= ~5e-7 from exact); 0: correctly rounded, bit-compatible with the oracle
    int emit_wgs_per_cu = 0;   // 0: the kernel's own residency (4 for the soup, 3 for the indexed output)
    int emit_sub_log2 = 1;    // dynamic mode: 2^s ticket counters per XCD
    int emit_dynamic = 1;     // per-XCD ticket counters instead of a static round-robin over the active list
    int emit_ablate = 0;      // diagnostic builds only: 1 skip stores, 4 skip vertex math, 16 non-temporal record stores, 64 all records into the buffer's first 32 MB (output invalid)
    int classify_ablate = 0;  // diagnostic builds only: 1 no halo rows (output invalid)
    int classify_wgs_per_cu = 3;   // residency cap of the streaming classify kernel (0: none = 7 workgroups per CU; 3 measured best, A/B in profiles/r02c)
    int emit_row_masks = 1;   // emit loads only the tile rows next to cells with triangles (masks from classify)
    int density_ablate = 0;   // diagnostic builds only: 1 the sampler skips its stores (output invalid)
    int fill_keeps_signs = 0;   // 1: a z-walk density fill also leaves the samples' sign bits; an extract of the same, UNMODIFIED buffer by this
                                // context then classifies from them (1/32 of the bytes) -- the streaming driver's setting
    int density_wgs_per_cu = 0;   // residency cap of the column sampler (0: four workgroups per CU); 3 leaves room for a concurrent extract
    int stage_events = 1;     // 1: events between the three kernels (vtmc_last_stage_ms per stage); 0: only around the whole step
    int gather_beside = 0;    // 1: the all-gather of a queued extract runs on a second stream beside the emit kernel (opt-in: never run with a world > 1); 0: behind it, on the caller's stream
    int emit_once = 1;        // 1 (soup, fast math): every welded vertex of a block is evaluated once into LDS, records expanded from there; 0: per triangle corner
    int place_outputs = 0;    // round 6: > 1 (up to 16) = when the output buffers have just been (re)allocated, that many candidate allocations are timed with the emit stage of the
                              // extract at hand and the fastest kept (profiles/r06/placement_probe.txt: the emit kernel's time is a property of the pair
                              // input field / output allocation, 0.86-1.00 ms for one kernel); 0 / 1: take what hipMalloc gives
    int emit_spare_wgs = 0;   // workgroups the emit launch leaves free (one per XCD: room for the collective's kernel beside it)
};

// scan scratch layout
constexpr int kQueueWords = 8 * 16 * 64;  // up to 16 ticket counters per XCD, 256 bytes apart
constexpr uint32_t kCountMask = 0xFFFFu;  // counts[b]: triangles (<= 2560) | row mask << 16 (y layers 16-23, z layers 24-31)
constexpr int kScanTile = 2048;  // block counts per scan workgroup (256 threads x 8)

// The runtime keeps ONE sticky "last error" word per thread: a failure nobody looked at stays there and is handed to whoever asks next.
// Two rules keep a stale failure from being blamed on a launch (round 3's red suite was exactly that):
//  * a launch wrapper clears the word before its launch and returns only what the launch left (launch_begin / launch_end);
//  * a best-effort call whose failure is deliberately ignored (teardown, an optional pinned allocation) goes through quiet(), which
//    consumes the sticky word together with the status.
inline void launch_begin() { (void)hipGetLastError(); }
inline hipError_t launch_end() { return hipGetLastError(); }
inline void quiet(hipError_t e)
{
    if (e != hipSuccess) (void)hipGetLastError();
}

// Launch wrappers.  All asynchronous on `stream`; each returns the status of its own launch (launch_begin ... launch_end).
// scan_ctrl / n_scan_ctrl: 64-bit words the kernel zeroes for the fused scan that follows (may be null / 0)
hipError_t launch_classify_blocks(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                  uint8_t *cases_or_null, uint32_t *vcounts_or_null, int n_cus, unsigned long long *scan_ctrl,
                                  int n_scan_ctrl, hipStream_t stream);
hipError_t launch_classify_dense(const BlockSpace &sp, const DeviceTables &tb, uint32_t *counts,
                                 uint32_t *vcounts_or_null, int ablate, int wgs_per_cu, unsigned long long *scan_ctrl, int n_scan_ctrl,
                                 const SignVolume &signs, hipStream_t stream);
// one-launch scan: ctrl = 2 + n_tiles words zeroed beforehand (ticket, error, tile status); totals[0..3] = {T saturating, nActive,
// T, 0}, totals[8] = 1 on a look-back time-out, mirrored into host_totals (device-visible pinned memory) when not null;
// zero_words / n_zero: 32-bit words to clear on the way (the emit kernel's ticket queue)
// vcounts_or_null != null: the welded-vertex counts are scanned in the same launch (voffsets, vtotals[0..3] = {V saturating, 0, V lo, V hi},
// host_totals[4..7]); ctrl then holds 2 * scan_ctrl_words(n_blocks) zeroed words
hipError_t launch_scan_fused(const BlockSpace &sp, const uint32_t *counts, int n_blocks, uint32_t *offsets, BlockDesc *active, unsigned long long *ctrl,
                             uint32_t *totals, uint32_t *host_totals, uint32_t *zero_words, int n_zero, const uint32_t *vcounts_or_null,
                             uint32_t *voffsets, uint32_t *vtotals, uint32_t *volume_counts_or_null, int bpv, hipStream_t stream);
// the scan itself leaves the per-volume counts when a volume is a whole number of scan tiles (else the emit kernel's prologue does)
inline bool scan_writes_volume_counts(int bpv) { return bpv > 0 && bpv % 2048 == 0; }
inline size_t scan_ctrl_words(int n_blocks) { return 2 + (size_t)((n_blocks + 2047) / 2048); }
hipError_t launch_emit(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets,
                       const BlockDesc *active, const uint32_t *totals, uint32_t capacity,
                       void *triangles, int n_cus, const Tuning &tune, unsigned *queue, uint32_t *volume_counts, int n_volumes,
                       hipStream_t stream);   // volume_counts != null: the first workgroup also derives the per-volume counts from the offsets

hipError_t launch_emit_indexed(const BlockSpace &sp, const DeviceTables &tb, const uint32_t *offsets, const uint32_t *voffsets,
                               const BlockDesc *active, const uint32_t *totals, const uint32_t *vtotals,
                               uint32_t tri_capacity,
                               uint32_t vert_capacity, void *vertices, void *indices, int n_cus, const Tuning &tune, unsigned *queue,
                               uint32_t *volume_counts, int n_volumes, hipStream_t stream);

// terrain.hip: device-resident density grid with the reference's CSG write semantics.
struct TerrainShape {
    int dim_x, dim_y, dim_z;  // samples per axis = cells + 2 (VoxelTerrain.cs:145); x fastest in memory
    float scale;              // _voxelScale
    float origin[3];          // TerrainOrigin
    uint64_t seed;
};
struct TerrainModifierArgs {
    int kind, add_or_erode;  // vtmc_modifier.kind / .add_or_erode
    float p[8];              // vtmc_modifier.p
    int lx, ly, lz;          // first sample of the clamped AABB (VoxelTerrain.cs:273-276)
    int dx, dy, dz;          // samples per axis of the AABB, inclusive ends (VoxelTerrain.cs:284-286)
    uint32_t event;          // 1-based index of this modifier application since Init (hash input)
    const float *data;       // heightmap (device), row-major [dims0][dims1]
    int dims0, dims1;
};
hipError_t launch_terrain_fill(float *grid, long long n, uint64_t seed, int n_cus, hipStream_t stream);
hipError_t launch_terrain_modify(float *grid, const TerrainShape &sh, const TerrainModifierArgs &m, hipStream_t stream);

// density.hip
struct DensityLaunch {
    float frequency, lacunarity, gain, ramp_scale, ramp_center;
    int octaves;
    int dx, dy, dz;
    long long sx, sy, sz, sv;
    int n_volumes;
    int ablate;  // diagnostics only (vtmc_set_tuning "density_ablate"): 1 skip the stores (output invalid)
    int wgs_per_cu;  // residency cap of the column sampler (0: its own four workgroups per CU)
};
// d_rows: density_rows_bytes(n_volumes, dy, dz) bytes of scratch (the per-(volume, step) rows of the column sampler)
hipError_t launch_density(const DensityLaunch &dl, const unsigned char *d_perm, const int *d_origins,
                          float *d_rows, float *d_out, unsigned long long *d_signs, hipStream_t stream);
// the sign volume a z-walk fill can leave for the classify stage of the SAME, unmodified buffer (tuning key "fill_keeps_signs")
bool density_writes_signs(const DensityLaunch &dl);
int density_sign_plane_words(int dx, int dy);
size_t density_sign_words(const DensityLaunch &dl);

size_t density_rows_bytes(int n_volumes, int dy, int dz);
void density_permutation(uint64_t seed, unsigned char perm[256]);

}  // namespace vtmc
#endif
