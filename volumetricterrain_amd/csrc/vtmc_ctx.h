// vtmc_ctx.h -- the context object behind include/vtmc.h and the small host helpers every
// translation unit of the C-ABI layer shares (vtmc_api.hip, chunk_io.hip, comm.hip).  Not installed.
#ifndef VTMC_CTX_H
#define VTMC_CTX_H
#include "../../include/vtmc.h"
#include "vtmc_internal.h"

#include <string>
#include <vector>

struct VtmcDevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// An extract that has been queued on a stream and not yet completed by extract_finish().
struct VtmcPending {
    bool active = false;    // queued, extract_finish() not yet called
    bool launched = false;  // false: the empty-batch early exit (nothing to wait for but the memsets)
    vtmc::BlockSpace sp{};
    int n_volumes = 0;
    bool indexed = false;
    hipStream_t stream = nullptr;
    size_t tcap = 0, vcap = 0;  // capacities the last emit launch was given
    bool scan_event = false;    // ev[2] was recorded behind this extract's scan
    bool counts_early = false;  // the per-volume counts were final when ev[2] (end of the scan) was recorded
};

struct vtmc_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;         // the context's own stream (what `stream` = NULL means in the ABI)
    bool stream_own_queue = false;        // test switch VTMC_TEST_MAIN_STREAM_OWN_QUEUE=1: `stream` itself sits on a hardware queue of its own
    hipStream_t queue_stream = nullptr;   // vtmc_context_stream(own_queue = 1): a stream on a hardware queue of its own, made on request
    vtmc::DeviceTables tables{nullptr, nullptr};
    VtmcDevBuf d_vert, d_trinum;
    VtmcDevBuf counts, offsets, active, partials, totals, volcounts, cases, tris, input, list, perm, origins, yrows;
    VtmcDevBuf vcounts, voffsets, vtotals, verts, indices;  // indexed output
    int output_mode = VTMC_OUTPUT_SOUP;
    bool last_indexed = false;
    int64_t last_verts = 0;
    uint32_t *h_totals_dev = nullptr;  // the same pinned words as the device sees them (the fused scan writes its totals there)
    uint32_t *h_totals = nullptr;  // pinned: the scan's totals ({T sat, nActive, T lo, T hi}, then the vertex scan's), 64 words
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0..3] stage timing, [4] staging copies
    VtmcDevBuf signs;               // the sign volume of the last z-walk fill (tuning key "fill_keeps_signs")
    struct {
        bool valid = false;
        const float *d_out = nullptr;
        int dx = 0, dy = 0, dz = 0, n_volumes = 0;
        long long sv = 0;
    } sign_of;                      // which buffer / layout `signs` describes
    float *h_stage = nullptr;       // pinned staging of host-gathered tiles (vtmc_extract_grid with a dirty list)
    size_t h_stage_bytes = 0;
    int h_stage_small_calls = 0;    // consecutive calls that needed far less than an over-sized staging buffer holds (the trim waits for kStageTrimAfter of them)
    int32_t *h_origins = nullptr;   // pinned staging of the sampler's chunk origins
    size_t h_origins_bytes = 0;
    hipEvent_t ev_origins = nullptr;   // behind the upload from h_origins
    bool origins_upload_pending = false;
    float stage_ms[4] = {0, 0, 0, 0};
    bool place_pending = false;     // the output buffers were (re)allocated since the last placement trial (tuning key place_outputs)
    float place_ms[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // emit stage per candidate of the last trial ([0]: the buffer that was there), place_n of them
    int place_n = 0, place_kept = 0;
    hipEvent_t ev_fill[2] = {nullptr, nullptr};  // around the last density kernel (vtmc_last_fill_ms)
    bool fill_timed = false;
    VtmcPending pending;
    // last result
    bool has_result = false;
    vtmc::BlockSpace last_space{};
    int last_blocks = 0;
    int last_volumes = 0;
    int64_t last_tris = 0;
    vtmc::Tuning tune;
    // device-resident terrain (vtmc_terrain_*)
    VtmcDevBuf terrain, heightmap;
    vtmc::TerrainShape tshape{};
    bool has_terrain = false;
    uint32_t terrain_events = 0;
    std::vector<int32_t> dirty;  // (bx,by,bz) of the last vtmc_terrain_update, ordered by block id
    bool dirty_is_all = false;   // ... or every block (the list is then materialised on demand only)
    uint64_t perm_seed = 0;
    bool perm_valid = false;
    // chunk_io.hip: file image being assembled / last image read
    VtmcDevBuf chunk_image;
    // comm.hip: RCCL communicator (opaque ncclComm_t) + the padded send buffer of the counts all-gather
    bool comm_borrowed = false;   // comm belongs to another context (vtmc_comm_share): never destroyed through this one
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    VtmcDevBuf comm_send;
    hipStream_t comm_stream = nullptr;   // the all-gather runs here, beside the emit kernel, when the counts leave the scan
    hipEvent_t ev_gather = nullptr;
    hipEvent_t ev_last_gather = nullptr;   // behind the last all-gather this context queued, on the stream it went to
    bool gather_recorded = false;
    hipEvent_t ev_comm_chain = nullptr;    // owner of a communicator: behind the LAST collective anybody issued through it (the chain of comm.hip)
    hipStream_t comm_chain_stream = nullptr;
    bool comm_chain_recorded = false;
    vtmc_ctx *comm_owner = nullptr;              // borrowed: whose communicator this is
    std::vector<vtmc_ctx *> comm_borrowers;      // owned: the contexts that borrowed it (detached when the owner lets go)
    std::string err;
};

namespace vtmc {
// sets the context's (or, with ctx == nullptr, the thread's create-) error text and returns `code`
int fail(vtmc_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int ensure(vtmc_ctx *ctx, VtmcDevBuf &b, size_t bytes);  // grow-only device buffer
void release(VtmcDevBuf &b);
const char *create_error_text();
void comm_release(vtmc_ctx *ctx);  // comm.hip: called by vtmc_destroy
// vtmc_api.hip: a context's streams come from (and return to) a per-device pool and are never destroyed -- see StreamPool there
hipError_t take_stream(int device, bool own_queue, int n_cus, hipStream_t *out);
void park_stream(int device, bool own_queue, hipStream_t s);
int release_parked_streams();
}  // namespace vtmc

#define VTMC_HIP(ctx, expr)                                                                                \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return vtmc::fail(ctx, VTMC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                                         \
    } while (0)

#endif
