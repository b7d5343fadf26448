// rccl.cpp -- the one collective on the path: the merge of per-GPU partial group tables.
//
// Reference: CombineResults / Result.Combine / BasicHist.Combine (aggregate.go:414-467,
// query_spec.go:138-193, hist_basic.go:259-279) fold per-block results on one host; the
// reference's only multi-node mechanism ships gob files to `sybil aggregate`
// (node_aggregator.go:147-177).  Here every rank holds an identically laid out integer
// table, so the merge is one SUM all-reduce (counts, sums, buckets) plus one MAX
// all-reduce (extrema; minima are stored negated) over RCCL / xGMI.
#include <rccl/rccl.h>
#include <string.h>

#include "engine.h"

using namespace sybl;

static int nccl_fail(ncclResult_t r, const char *what) {
    return fail(SYBL_E_NODEVICE, "RCCL error: %s in %s", ncclGetErrorString(r), what);
}
#define SYBL_NCCL(expr)                                   \
    do {                                                  \
        ncclResult_t r__ = (expr);                        \
        if (r__ != ncclSuccess) return nccl_fail(r__, #expr); \
    } while (0)

extern "C" {

int sybl_comm_unique_id(void *id128) {
    if (!id128) return fail(SYBL_E_INVAL, "id buffer is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    SYBL_NCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return SYBL_OK;
}

int sybl_comm_init(sybl_ctx *ctx, const void *id128, int32_t nranks, int32_t rank) {
    if (!ctx || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(SYBL_E_INVAL, "sybl_comm_init: bad argument");
    if (ctx->comm) return fail(SYBL_E_STATE, "communicator already initialised");
    SYBL_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    SYBL_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return SYBL_OK;
}

int sybl_comm_free(sybl_ctx *ctx) {
    if (!ctx || !ctx->comm) return SYBL_OK;
    hipSetDevice(ctx->device);
    ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_nranks = 1;
    ctx->comm_rank = 0;
    return SYBL_OK;
}

int sybl_query_allreduce(sybl_query *q) {
    if (!q) return fail(SYBL_E_INVAL, "query is NULL");
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_allreduce before sybl_query_scan");
    Ctx *ctx = q->ctx;
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator: call sybl_comm_init first");
    SYBL_HIP(hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    SYBL_NCCL(ncclGroupStart());
    SYBL_NCCL(ncclAllReduce(q->d_sum, q->d_sum, (size_t)q->n_sum_words, ncclInt64, ncclSum, comm, ctx->stream));
    SYBL_NCCL(ncclAllReduce(q->d_max, q->d_max, (size_t)q->n_max_words, ncclInt64, ncclMax, comm, ctx->stream));
    SYBL_NCCL(ncclGroupEnd());
    return SYBL_OK;
}

}  // extern "C"
