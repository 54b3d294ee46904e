// The one collective of the path: an RCCL all-gather of the per-rank spectrum shards
// (include/pyrad_hip.h, "multi-GPU").  One process per GPU; the unique id travels out of
// band (the Python host broadcasts it over its launcher's rendezvous).
#include "../../include/pyrad_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include <mutex>

// private views of the objects defined in lbl_api.hip
extern "C" int lbl_ctx_stream(lbl_ctx* ctx, void** stream);
extern "C" int lbl_buffer_devptr(lbl_buffer* buf, void** devptr);
extern "C" int lbl_buffer_size(const lbl_buffer* buf, int64_t* n);
extern "C" int lbl_comm_fence_dev(lbl_comm* comm, int slot);
namespace lbl {
int comm_fail(lbl_ctx* ctx, int code, const char* msg);
int ctx_device(lbl_ctx* ctx);
bool ctx_capturing(lbl_ctx* ctx);
lbl_ctx* buffer_ctx(lbl_buffer* buf);
void* comm_prof_begin(lbl_ctx* ctx);
void comm_prof_end(lbl_ctx* ctx, void* start);
}

// no C++ exception crosses the C boundary (see lbl_api.hip)
#define LBL_GUARD_END(ctx_expr)                                                                                   \
    catch (const std::bad_alloc&) { return lbl::comm_fail((ctx_expr), LBL_ERR_OOM, "host allocation failed"); }   \
    catch (const std::exception& e) { return lbl::comm_fail((ctx_expr), LBL_ERR_STATE, e.what()); }               \
    catch (...) { return lbl::comm_fail((ctx_expr), LBL_ERR_STATE, "unknown C++ exception"); }

// Every collective of a communicator is issued on ONE stream of its own (cstream), in the same
// order on every rank.  The context stream and cstream are ordered with events only, so the
// all-gather of step k can run while the kernels of step k+1 compute (lbl_allgather_overlap_dev
// + lbl_comm_fence_dev); lbl_allgather_dev is the same path with the fence applied at once.
// The stream a collective is ordered against is the stream of the context that OWNS its buffers
// (any context of the communicator's device): several contexts - independent steps in flight on
// streams of their own - share the one communicator and its one, rank-consistent, order of collectives.
constexpr int kSlots = 8;        // 0..6 for overlapped collectives, 7 for the in-stream form
struct lbl_comm {
    lbl_ctx* ctx;
    ncclComm_t comm;
    int world, rank;
    hipStream_t cstream;
    hipEvent_t ready;      // context stream -> cstream: inputs of the collective are complete
    hipEvent_t done[kSlots];    // cstream -> context stream: the collective issued with this slot has finished
    bool pending[kSlots];       // a collective was issued with this slot since its last fence
    lbl_ctx* owner[kSlots];     // context whose stream waits at the slot's fence (the owner of the slot's buffers)
};

static_assert(sizeof(ncclUniqueId) <= LBL_UNIQUE_ID_BYTES, "unique id does not fit");

// Live communicators, so that freeing a buffer or a context can wait for collectives that still use them:
// an overlapped all-gather runs on the communicator's own stream, which a context's stream knows nothing
// about until the slot's fence.
static std::mutex g_comms_mutex;
static std::vector<lbl_comm*> g_comms;

namespace lbl {
// Before memory of `ctx` is released: collectives issued on its buffers and not fenced yet must have finished.
void comm_quiesce(lbl_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_comms_mutex);
    for (lbl_comm* cm : g_comms)
        for (int i = 0; i < kSlots; ++i)
            if (cm->pending[i] && cm->owner[i] == ctx) { (void)hipStreamSynchronize(cm->cstream); break; }
}
// `ctx` is going away: its slots are finished and no fence may touch its stream any more.
void comm_forget(lbl_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_comms_mutex);
    for (lbl_comm* cm : g_comms) {
        bool synced = false;
        for (int i = 0; i < kSlots; ++i) {
            if (cm->owner[i] != ctx || cm->ctx == ctx) continue;      // (a communicator dies with its own context's objects)
            if (cm->pending[i] && !synced) { (void)hipStreamSynchronize(cm->cstream); synced = true; }
            cm->pending[i] = false;
            cm->owner[i] = cm->ctx;
        }
    }
}
}  // namespace lbl

extern "C" int lbl_comm_unique_id(char id[LBL_UNIQUE_ID_BYTES]) try {
    if (!id) return lbl::comm_fail(nullptr, LBL_ERR_BAD_ARG, "id is NULL");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return lbl::comm_fail(nullptr, LBL_ERR_RCCL, ncclGetErrorString(r));
    memset(id, 0, LBL_UNIQUE_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return LBL_OK;
} LBL_GUARD_END(nullptr)

extern "C" int lbl_comm_create(lbl_ctx* ctx, const char id[LBL_UNIQUE_ID_BYTES], int world_size, int rank,
                               lbl_comm** out) try {
    if (!ctx || !id || !out) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "bad world_size / rank");
    if (hipSetDevice(lbl::ctx_device(ctx)) != hipSuccess) return lbl::comm_fail(ctx, LBL_ERR_HIP, "hipSetDevice failed");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c;
    ncclResult_t r = ncclCommInitRank(&c, world_size, u, rank);
    if (r != ncclSuccess) return lbl::comm_fail(ctx, LBL_ERR_RCCL, ncclGetErrorString(r));
    lbl_comm* cm = new (std::nothrow) lbl_comm();
    if (!cm) { ncclCommDestroy(c); return lbl::comm_fail(ctx, LBL_ERR_OOM, "host allocation failed"); }
    cm->ctx = ctx; cm->comm = c; cm->world = world_size; cm->rank = rank;
    cm->cstream = nullptr; cm->ready = nullptr;
    bool ok = hipStreamCreateWithFlags(&cm->cstream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&cm->ready, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kSlots; ++i) {
        cm->done[i] = nullptr; cm->pending[i] = false; cm->owner[i] = ctx;
        ok = ok && hipEventCreateWithFlags(&cm->done[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        ncclCommDestroy(c);
        delete cm;
        return lbl::comm_fail(ctx, LBL_ERR_HIP, "communicator stream/event creation failed");
    }
    {
        std::lock_guard<std::mutex> lock(g_comms_mutex);
        g_comms.push_back(cm);
    }
    *out = cm;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_comm_destroy(lbl_comm* comm) try {
    if (!comm) return LBL_OK;
    {
        std::lock_guard<std::mutex> lock(g_comms_mutex);
        for (size_t i = 0; i < g_comms.size(); ++i)
            if (g_comms[i] == comm) { g_comms.erase(g_comms.begin() + (long)i); break; }
    }
    void* s = nullptr;
    lbl_ctx_stream(comm->ctx, &s);
    (void)hipStreamSynchronize((hipStream_t)s);
    (void)hipStreamSynchronize(comm->cstream);
    ncclCommDestroy(comm->comm);
    (void)hipEventDestroy(comm->ready);
    for (int i = 0; i < kSlots; ++i) if (comm->done[i]) (void)hipEventDestroy(comm->done[i]);
    (void)hipStreamDestroy(comm->cstream);
    delete comm;
    return LBL_OK;
} LBL_GUARD_END(comm ? comm->ctx : nullptr)

static int allgather_impl(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count, lbl_buffer* recv,
                          int slot, bool fence_now) {
    if (!comm || !send || !recv) return lbl::comm_fail(comm ? comm->ctx : nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    lbl_ctx* ctx = lbl::buffer_ctx(send);            // the context whose stream produced the data
    if (lbl::buffer_ctx(recv) != ctx) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "send and recv belong to different contexts");
    if (lbl::ctx_device(ctx) != lbl::ctx_device(comm->ctx))
        return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "buffers live on another device than the communicator");
    if (slot < 0 || slot >= kSlots) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "slot must be 0..6");
    if (comm->pending[slot] && comm->owner[slot] != ctx)
        return lbl::comm_fail(ctx, LBL_ERR_STATE, "slot still carries an unfenced collective of another context");
    if (lbl::ctx_capturing(ctx))
        return lbl::comm_fail(ctx, LBL_ERR_STATE, "the all-gather stays outside a captured graph: end the capture first");
    int64_t ns = 0, nr = 0;
    lbl_buffer_size(send, &ns);
    lbl_buffer_size(recv, &nr);
    if (count < 0 || send_offset < 0 || send_offset + count > ns) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "send range out of bounds");
    if ((int64_t)comm->world * count > nr) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "recv shorter than world_size*count");
    if (count == 0) return LBL_OK;
    void *ps = nullptr, *pr = nullptr, *s = nullptr;
    lbl_buffer_devptr(send, &ps);
    lbl_buffer_devptr(recv, &pr);
    lbl_ctx_stream(ctx, &s);
    hipStream_t main_stream = (hipStream_t)s;
    if (hipSetDevice(lbl::ctx_device(ctx)) != hipSuccess) return lbl::comm_fail(ctx, LBL_ERR_HIP, "hipSetDevice failed");
    // the collective starts when everything enqueued so far on the context stream is complete
    if (hipEventRecord(comm->ready, main_stream) != hipSuccess ||
        hipStreamWaitEvent(comm->cstream, comm->ready, 0) != hipSuccess)
        return lbl::comm_fail(ctx, LBL_ERR_HIP, "stream ordering (ready) failed");
    ncclResult_t r = ncclAllGather((const double*)ps + send_offset, pr, (size_t)count, ncclDouble, comm->comm, comm->cstream);
    if (r != ncclSuccess) return lbl::comm_fail(ctx, LBL_ERR_RCCL, ncclGetErrorString(r));
    if (hipEventRecord(comm->done[slot], comm->cstream) != hipSuccess) return lbl::comm_fail(ctx, LBL_ERR_HIP, "hipEventRecord(done) failed");
    comm->pending[slot] = true;
    comm->owner[slot] = ctx;
    if (fence_now) return lbl_comm_fence_dev(comm, slot);
    return LBL_OK;
}

extern "C" int lbl_comm_fence_dev(lbl_comm* comm, int slot) try {
    if (!comm) return lbl::comm_fail(nullptr, LBL_ERR_BAD_ARG, "comm is NULL");
    if (slot < -1 || slot >= kSlots) return lbl::comm_fail(comm->ctx, LBL_ERR_BAD_ARG, "slot must be -1 (all) or 0..6");
    for (int i = 0; i < kSlots; ++i) {
        if ((slot >= 0 && i != slot) || !comm->pending[i]) continue;
        void* s = nullptr;
        lbl_ctx_stream(comm->owner[i], &s);
        // Usually the collective of two steps ago has long finished: ask the host side first and put a wait
        // (a barrier packet, a few microseconds of the context stream) into the stream only if it has not
        if (hipEventQuery(comm->done[i]) != hipSuccess) {
            (void)hipGetLastError();                   // hipErrorNotReady is not an error here
            if (hipStreamWaitEvent((hipStream_t)s, comm->done[i], 0) != hipSuccess)
                return lbl::comm_fail(comm->owner[i], LBL_ERR_HIP, "stream ordering (done) failed");
        }
        comm->pending[i] = false;
    }
    return LBL_OK;
} LBL_GUARD_END(comm ? comm->ctx : nullptr)

extern "C" int lbl_allgather_dev(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count, lbl_buffer* recv) try {
    lbl_ctx* pctx = (comm && send) ? lbl::buffer_ctx(send) : nullptr;
    void* ev = pctx ? lbl::comm_prof_begin(pctx) : nullptr;
    int rc = allgather_impl(comm, send, send_offset, count, recv, kSlots - 1, true);
    if (pctx) lbl::comm_prof_end(pctx, ev);
    return rc;
} LBL_GUARD_END(comm ? comm->ctx : nullptr)

extern "C" int lbl_allgather_overlap_dev(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count,
                                         lbl_buffer* recv, int slot) try {
    if (slot == kSlots - 1) return lbl::comm_fail(comm ? comm->ctx : nullptr, LBL_ERR_BAD_ARG, "slot must be 0..6");
    return allgather_impl(comm, send, send_offset, count, recv, slot, false);
} LBL_GUARD_END(comm ? comm->ctx : nullptr)

// Fewer, larger collectives: the shards of several consecutive steps are staged side by side in one batch
// buffer and leave in ONE all-gather.  This is the staging copy, device to device, on the stream of the
// context that owns dst (so it follows the kernels that produced src when both belong to that context).
extern "C" int lbl_gather_stage_dev(lbl_buffer* dst, int64_t dst_offset, lbl_buffer* src, int64_t src_offset, int64_t n) try {
    if (!dst || !src) return lbl::comm_fail(nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    lbl_ctx* ctx = lbl::buffer_ctx(dst);
    int64_t nd = 0, ns = 0;
    lbl_buffer_size(dst, &nd);
    lbl_buffer_size(src, &ns);
    if (n < 0 || dst_offset < 0 || src_offset < 0 || dst_offset + n > nd || src_offset + n > ns)
        return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "copy range out of bounds");
    if (lbl::ctx_device(lbl::buffer_ctx(src)) != lbl::ctx_device(ctx))
        return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "buffers live on different devices");
    if (n == 0) return LBL_OK;
    void *pd = nullptr, *ps = nullptr, *s = nullptr;
    lbl_buffer_devptr(dst, &pd);
    lbl_buffer_devptr(src, &ps);
    lbl_ctx_stream(ctx, &s);
    if (hipSetDevice(lbl::ctx_device(ctx)) != hipSuccess) return lbl::comm_fail(ctx, LBL_ERR_HIP, "hipSetDevice failed");
    if (hipMemcpyAsync((double*)pd + dst_offset, (const double*)ps + src_offset, (size_t)n * sizeof(double),
                       hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess)
        return lbl::comm_fail(ctx, LBL_ERR_HIP, "device-to-device copy failed");
    return LBL_OK;
} LBL_GUARD_END(dst ? lbl::buffer_ctx(dst) : nullptr)
