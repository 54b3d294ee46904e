// The one collective of the path: an RCCL all-gather of the per-rank spectrum shards
// (include/pyrad_hip.h, "multi-GPU").  One process per GPU; the unique id travels out of
// band (the Python host broadcasts it over its launcher's rendezvous).
#include "../../include/pyrad_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

// private views of the objects defined in lbl_api.hip
extern "C" int lbl_ctx_stream(lbl_ctx* ctx, void** stream);
extern "C" int lbl_buffer_devptr(lbl_buffer* buf, void** devptr);
extern "C" int lbl_buffer_size(const lbl_buffer* buf, int64_t* n);
namespace lbl {
int comm_fail(lbl_ctx* ctx, int code, const char* msg);
int ctx_device(lbl_ctx* ctx);
void* comm_prof_begin(lbl_ctx* ctx);
void comm_prof_end(lbl_ctx* ctx, void* start);
}

struct lbl_comm {
    lbl_ctx* ctx;
    ncclComm_t comm;
    int world, rank;
};

static_assert(sizeof(ncclUniqueId) <= LBL_UNIQUE_ID_BYTES, "unique id does not fit");

extern "C" int lbl_comm_unique_id(char id[LBL_UNIQUE_ID_BYTES]) {
    if (!id) return lbl::comm_fail(nullptr, LBL_ERR_BAD_ARG, "id is NULL");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return lbl::comm_fail(nullptr, LBL_ERR_RCCL, ncclGetErrorString(r));
    memset(id, 0, LBL_UNIQUE_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return LBL_OK;
}

extern "C" int lbl_comm_create(lbl_ctx* ctx, const char id[LBL_UNIQUE_ID_BYTES], int world_size, int rank,
                               lbl_comm** out) {
    if (!ctx || !id || !out) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return lbl::comm_fail(ctx, LBL_ERR_BAD_ARG, "bad world_size / rank");
    if (hipSetDevice(lbl::ctx_device(ctx)) != hipSuccess) return lbl::comm_fail(ctx, LBL_ERR_HIP, "hipSetDevice failed");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c;
    ncclResult_t r = ncclCommInitRank(&c, world_size, u, rank);
    if (r != ncclSuccess) return lbl::comm_fail(ctx, LBL_ERR_RCCL, ncclGetErrorString(r));
    lbl_comm* cm = new (std::nothrow) lbl_comm{ctx, c, world_size, rank};
    if (!cm) { ncclCommDestroy(c); return lbl::comm_fail(ctx, LBL_ERR_OOM, "host allocation failed"); }
    *out = cm;
    return LBL_OK;
}

extern "C" int lbl_comm_destroy(lbl_comm* comm) {
    if (!comm) return LBL_OK;
    void* s = nullptr;
    lbl_ctx_stream(comm->ctx, &s);
    (void)hipStreamSynchronize((hipStream_t)s);
    ncclCommDestroy(comm->comm);
    delete comm;
    return LBL_OK;
}

extern "C" int lbl_allgather_dev(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count, lbl_buffer* recv) {
    if (!comm || !send || !recv) return lbl::comm_fail(comm ? comm->ctx : nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    lbl_ctx* ctx = comm->ctx;
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
    void* ev = lbl::comm_prof_begin(ctx);
    ncclResult_t r = ncclAllGather((const double*)ps + send_offset, pr, (size_t)count, ncclDouble, comm->comm, (hipStream_t)s);
    lbl::comm_prof_end(ctx, ev);
    if (r != ncclSuccess) return lbl::comm_fail(ctx, LBL_ERR_RCCL, ncclGetErrorString(r));
    return LBL_OK;
}
