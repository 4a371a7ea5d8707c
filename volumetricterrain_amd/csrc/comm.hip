// comm.hip -- the one collective of the path, behind the C ABI: an RCCL all-gather of the per-chunk
// {vertex count, triangle count} pairs over xGMI.  New in the build: the reference is
// single-process, single-GPU and has no collective call site; the call belongs where BatchUpdate
// hands its results to the host (Unity-Project/Assets/Scripts/VoxelTerrain.cs:426-446), which a
// multi-GPU host does once per rank.
//
// librccl is bound at run time (dlopen of the SONAME, so a process that already maps an RCCL -- e.g.
// the one bundled with PyTorch -- shares it and its HIP runtime): a single-GPU host never loads it,
// and a host without RCCL gets VTMC_ERR_DEVICE from vtmc_comm_init_rank instead of a load failure.
// No density or mesh data ever crosses GPUs; the message is 8 bytes per chunk.
#include "vtmc_ctx.h"

#include <dlfcn.h>

// The five entry points and four types of RCCL this file uses, declared here so that the library builds on a host without the RCCL
// headers (a single-GPU host never loads librccl at all).  Values as in rccl.h / nccl.h (stable ABI): ncclSuccess = 0, ncclUint32 = 3,
// a unique id of 128 bytes.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint32 = 3;
static constexpr int NCCL_UNIQUE_ID_BYTES = 128;

#include <algorithm>
#include <cstring>
#include <mutex>

using namespace vtmc;

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    int version = 0;
    std::string error;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi &rccl()
{
    std::call_once(g_rccl_once, [] {
        RcclApi &a = g_rccl;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) {
            const char *e = dlerror();
            a.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        a.GetVersion = (decltype(a.GetVersion))dlsym(a.handle, "ncclGetVersion");
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString || !a.GetVersion) {
            a.error = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetVersion";
            a.handle = nullptr;
            return;
        }
        // the declarations above are the NCCL 2.x ABI (ncclUint32 = 3, a 128-byte id passed by value): anything else is refused, not guessed at.
        // Version code: major * 10000 + minor * 100 + patch from 2.9 on, major * 1000 + ... before.
        if (a.GetVersion(&a.version) != ncclSuccess || a.version < 2000 || (a.version >= 10000 && a.version / 10000 != 2)) {
            a.error = "librccl.so.1 reports version code " + std::to_string(a.version) + ": this library is written against the NCCL 2.x ABI";
            a.handle = nullptr;
        }
    });
    return g_rccl;
}

#define VTMC_NCCL(ctx, api, expr)                                                                                \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return fail(ctx, VTMC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, (api).GetErrorString(r_), __FILE__, \
                        __LINE__);                                                                               \
    } while (0)

}  // namespace

namespace vtmc {
namespace {
// every collective a context has queued through its communicator has finished: its own streams, a queued extract's stream, and -- through
// the event recorded behind the last all-gather -- whatever stream the caller handed in (bench.py's side stream)
void drain_collectives(vtmc_ctx *ctx)
{
    if (ctx->gather_recorded && ctx->ev_last_gather) quiet(hipEventSynchronize(ctx->ev_last_gather));
    ctx->gather_recorded = false;
    if (ctx->comm_chain_recorded && ctx->ev_comm_chain) quiet(hipEventSynchronize(ctx->ev_comm_chain));   // an owner: the end of its communicator's chain
    ctx->comm_chain_recorded = false;
    if (ctx->comm_stream) quiet(hipStreamSynchronize(ctx->comm_stream));
    if (ctx->pending.active && ctx->pending.stream) quiet(hipStreamSynchronize(ctx->pending.stream));
    if (ctx->stream) quiet(hipStreamSynchronize(ctx->stream));
}
}  // namespace

// Contexts that share a communicator (vtmc_comm_share) are driven by ONE host thread -- their collectives must be in one program order
// anyway -- so the owner's borrower list needs no lock.
void comm_release(vtmc_ctx *ctx)
{
    if (!ctx || !ctx->comm) return;
    quiet(hipSetDevice(ctx->device));
    drain_collectives(ctx);   // no collective of this communicator may still be queued when it is destroyed
    if (ctx->comm_borrowed) {   // a borrower leaves: the owner forgets it, the communicator stays
        if (vtmc_ctx *o = ctx->comm_owner) {
            auto &v = o->comm_borrowers;
            v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
        }
    } else {
        // the owner goes first (e.g. the garbage collector's order): every borrower is drained and detached -- its next
        // vtmc_allgather_volume_counts answers VTMC_ERR_NO_RESULT instead of using a destroyed communicator
        for (vtmc_ctx *b : ctx->comm_borrowers) {
            drain_collectives(b);
            b->comm = nullptr;
            b->comm_borrowed = false;
            b->comm_owner = nullptr;
            b->comm_world = 1;
            b->comm_rank = 0;
        }
        ctx->comm_borrowers.clear();
        const RcclApi &a = rccl();
        if (a.handle) (void)a.CommDestroy((ncclComm_t)ctx->comm);
    }
    ctx->comm_borrowed = false;
    ctx->comm_owner = nullptr;
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
}
}  // namespace vtmc

extern "C" {

int32_t vtmc_comm_unique_id(uint8_t id[VTMC_COMM_ID_BYTES])
{
    if (!id) return fail(nullptr, VTMC_ERR_INVALID_ARG, "id is null");
    static_assert(VTMC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "vtmc.h and rccl.h disagree on the unique-id size");
    const RcclApi &a = rccl();
    if (!a.handle) return fail(nullptr, VTMC_ERR_DEVICE, "%s", a.error.c_str());
    ncclUniqueId u;
    VTMC_NCCL(nullptr, a, a.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return VTMC_OK;
}

int32_t vtmc_comm_init_rank(vtmc_ctx *ctx, const uint8_t id[VTMC_COMM_ID_BYTES], int32_t rank, int32_t world_size)
{
    if (!ctx) return VTMC_ERR_INVALID_ARG;
    if (!id) return fail(ctx, VTMC_ERR_INVALID_ARG, "id is null");
    if (world_size <= 0 || rank < 0 || rank >= world_size) return fail(ctx, VTMC_ERR_INVALID_ARG, "bad rank %d / world %d", rank, world_size);
    const RcclApi &a = rccl();
    if (!a.handle) return fail(ctx, VTMC_ERR_DEVICE, "%s", a.error.c_str());
    VTMC_HIP(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    VTMC_NCCL(ctx, a, a.CommInitRank(&c, world_size, u, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_world = world_size;
    return VTMC_OK;
}

int32_t vtmc_comm_share(vtmc_ctx *ctx, vtmc_ctx *owner)
{
    if (!ctx || !owner) return VTMC_ERR_INVALID_ARG;
    if (ctx == owner) return fail(ctx, VTMC_ERR_INVALID_ARG, "a context cannot borrow its own communicator");
    if (!owner->comm || owner->comm_borrowed) return fail(ctx, VTMC_ERR_NO_RESULT, "the owner holds no communicator of its own (vtmc_comm_init_rank)");
    if (owner->device != ctx->device) return fail(ctx, VTMC_ERR_INVALID_ARG, "the two contexts are on different devices (%d, %d)", ctx->device, owner->device);
    VTMC_HIP(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ctx->comm = owner->comm;
    ctx->comm_borrowed = true;
    ctx->comm_owner = owner;
    owner->comm_borrowers.push_back(ctx);
    ctx->comm_rank = owner->comm_rank;
    ctx->comm_world = owner->comm_world;
    return VTMC_OK;
}

int32_t vtmc_comm_destroy(vtmc_ctx *ctx)
{
    if (!ctx) return VTMC_ERR_INVALID_ARG;
    if (ctx->stream) VTMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) VTMC_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    comm_release(ctx);
    return VTMC_OK;
}

int32_t vtmc_allgather_volume_counts(vtmc_ctx *ctx, uint32_t *d_all_counts, int32_t volumes_per_rank, void *stream)
{
    if (!ctx) return VTMC_ERR_INVALID_ARG;
    if (!ctx->comm) return fail(ctx, VTMC_ERR_NO_RESULT, "allgather_volume_counts before vtmc_comm_init_rank");
    // the counts are final once the scan has run: valid for a finished extract and for a queued one
    // (vtmc_extract_volumes_device_async), which is how a rank keeps its host out of the step
    if (!ctx->has_result && !ctx->pending.active) return fail(ctx, VTMC_ERR_NO_RESULT, "allgather_volume_counts before any extract");
    const int n_vol = ctx->pending.active ? ctx->pending.n_volumes : ctx->last_volumes;
    const int n_blk = ctx->pending.active ? ctx->pending.sp.n_blocks : ctx->last_blocks;
    if (!d_all_counts) return fail(ctx, VTMC_ERR_INVALID_ARG, "d_all_counts is null");
    if (volumes_per_rank < n_vol || volumes_per_rank <= 0)
        return fail(ctx, VTMC_ERR_CAPACITY, "volumes_per_rank %d < %d volumes of the last extract", volumes_per_rank, n_vol);
    const RcclApi &a = rccl();
    VTMC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // When the scan of a queued extract has already left the counts (whole scan tiles per volume), the
    // collective goes to the context's second stream behind the scan's event and runs BESIDE the emit
    // kernel, which was launched a workgroup per XCD short for it; `stream` then only waits for its end.
    const bool beside = ctx->pending.active && ctx->pending.launched && ctx->pending.counts_early && ctx->pending.scan_event && ctx->tune.gather_beside;
    hipStream_t gs = st;
    // not beside: the counts are read on `st`.  When that is not the stream the queued extract runs on, order the read behind the
    // extract's emit launch (whose first workgroup may be the one that writes the per-volume counts)
    if (!beside && ctx->pending.active && ctx->pending.launched && st != ctx->pending.stream) VTMC_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[3], 0));
    if (beside) {
        if (!ctx->comm_stream) VTMC_HIP(ctx, take_stream(ctx->device, false, ctx->n_cus, &ctx->comm_stream));
        if (!ctx->ev_gather) VTMC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_gather, hipEventDisableTiming));
        gs = ctx->comm_stream;
        VTMC_HIP(ctx, hipStreamWaitEvent(gs, ctx->ev[2], 0));
    }
    // ranks may own different numbers of chunks (c % N): every rank sends volumes_per_rank pairs, zero-padded
    const size_t words = 2 * (size_t)volumes_per_rank;
    const size_t own = 2 * (size_t)(n_blk > 0 ? n_vol : 0);
    const void *send = ctx->volcounts.p;
    if (own < words) {
        if (ctx->comm_send.bytes < words * sizeof(uint32_t)) {   // about to reallocate
            VTMC_HIP(ctx, hipStreamSynchronize(st));
            if (ctx->comm_stream) VTMC_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
        }
        if (int rc = ensure(ctx, ctx->comm_send, words * sizeof(uint32_t))) return rc;
        VTMC_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->comm_send.p + own, 0, (words - own) * sizeof(uint32_t), gs));
        if (own > 0)
            VTMC_HIP(ctx, hipMemcpyAsync(ctx->comm_send.p, ctx->volcounts.p, own * sizeof(uint32_t), hipMemcpyDeviceToDevice, gs));
        send = ctx->comm_send.p;
    }
    // ONE ORDER PER COMMUNICATOR, WHATEVER THE STREAMS.  The collectives of a communicator -- its owner's and every borrower's -- form a chain of
    // events kept by the owner: a collective that goes to another stream than the one before it waits (on the device, no host wait) for the
    // previous one's event first.  Two contexts that take turns on a stream each (bench.py --streams 2) therefore hand RCCL its collectives
    // exactly as a single stream would -- one after the other, the same order on every rank -- and the last event of the chain stands for all
    // of them when the communicator is destroyed.
    vtmc_ctx *chain = ctx->comm_borrowed && ctx->comm_owner ? ctx->comm_owner : ctx;
    if (!chain->ev_comm_chain) VTMC_HIP(ctx, hipEventCreateWithFlags(&chain->ev_comm_chain, hipEventDisableTiming));
    if (chain->comm_chain_recorded && chain->comm_chain_stream != gs) VTMC_HIP(ctx, hipStreamWaitEvent(gs, chain->ev_comm_chain, 0));
    VTMC_NCCL(ctx, a, a.AllGather(send, d_all_counts, words, ncclUint32, (ncclComm_t)ctx->comm, gs));
    VTMC_HIP(ctx, hipEventRecord(chain->ev_comm_chain, gs));
    chain->comm_chain_recorded = true;
    chain->comm_chain_stream = gs;
    // behind the collective, on whatever stream it went to: comm_release waits for this before the communicator is destroyed
    if (!ctx->ev_last_gather) VTMC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_last_gather, hipEventDisableTiming));
    VTMC_HIP(ctx, hipEventRecord(ctx->ev_last_gather, gs));
    ctx->gather_recorded = true;
    if (beside) {
        VTMC_HIP(ctx, hipEventRecord(ctx->ev_gather, gs));
        VTMC_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_gather, 0));
    }
    return VTMC_OK;
}

int32_t vtmc_copy_to_host(vtmc_ctx *ctx, const void *d_src, void *dst, int64_t bytes, void *stream)
{
    if (!ctx) return VTMC_ERR_INVALID_ARG;
    if (bytes < 0 || (bytes > 0 && (!d_src || !dst))) return fail(ctx, VTMC_ERR_INVALID_ARG, "null pointer or negative size");
    if (bytes == 0) return VTMC_OK;
    VTMC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    VTMC_HIP(ctx, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, st));
    VTMC_HIP(ctx, hipStreamSynchronize(st));
    return VTMC_OK;
}

}  // extern "C"
