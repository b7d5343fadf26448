// rccl.cpp -- the one collective on the path: the merge of per-GPU partial group tables.
//
// Reference: CombineResults / Result.Combine / BasicHist.Combine (aggregate.go:414-467,
// query_spec.go:138-193, hist_basic.go:259-279) fold per-block results on one host; the
// reference's only multi-node mechanism ships gob files to `sybil aggregate`
// (node_aggregator.go:147-177).  Here every rank holds an identically laid out integer
// table, so the merge is one SUM all-reduce (counts, sums, buckets) plus one MAX
// all-reduce (extrema; minima are stored negated) over RCCL / xGMI.
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

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

namespace sybl {

int comm_allgather_inplace(Ctx *ctx, int64_t *buf, size_t words_per_rank) {
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator");
    SYBL_NCCL(ncclAllGather(buf + (size_t)ctx->comm_rank * words_per_rank, buf, words_per_rank, ncclInt64, (ncclComm_t)ctx->comm, ctx->stream));
    return SYBL_OK;
}

int comm_allreduce_sum(Ctx *ctx, int64_t *buf, size_t words) {
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator");
    SYBL_NCCL(ncclAllReduce(buf, buf, words, ncclInt64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return SYBL_OK;
}

}  // namespace sybl

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
    const ScanPlan &P = q->plan;
    const bool has_max = P.n_max_fields > 0;  // (cfg 3: no extremum is tracked -- ONE collective per step)
    const int64_t hist_words = (int64_t)P.n_cells * P.hist_stride, small_words = q->n_sum_words - (hist_words + (P.hist_stride > 0 ? kMaxScatterRanks * P.hist_stride : 0));
    // Big bucket tables (config 4: 65 536 cells x 1002 buckets = 525 MB per rank): all-reducing them moves ~2x the table
    // over every GPU's links although rank 0 only needs the percentiles, the bucket moments, the Cumulative buckets
    // and the printed rows' arrays.  Instead the bucket arrays are reduce-SCATTERED over equal slices of cells (each
    // rank receives 1/R of the table), every rank summarises its slice (k_hist_summary / k_hist_total in
    // query_snapshot) and the summaries are all-gathered: 52 MB of percentiles instead of 525 MB of buckets.
    // snapshot and finalize then are collective calls.
    const bool scatter = P.hist_stride > 0 && query_wants_hist_summary(q) && q->limit > 0 && ctx->comm_nranks <= kMaxScatterRanks &&
                         (ctx->comm_nranks > 1 || getenv("SYBL_FORCE_SCATTER"));
    SYBL_NCCL(ncclGroupStart());
    if (!scatter) {
        SYBL_NCCL(ncclAllReduce(q->d_sum, q->d_sum, (size_t)(small_words + hist_words), ncclInt64, ncclSum, comm, ctx->stream));
    } else {
        const int64_t R = ctx->comm_nranks, per = (P.n_cells + R - 1) / R, count = per * P.hist_stride;
        int64_t *H = q->d_sum + P.hist_off;
        SYBL_NCCL(ncclAllReduce(q->d_sum, q->d_sum, (size_t)small_words, ncclInt64, ncclSum, comm, ctx->stream));
        // (cells past n_cells in the last slice are the zeroed padding of the SUM section)
        SYBL_NCCL(ncclReduceScatter(H, H + (int64_t)ctx->comm_rank * count, (size_t)count, ncclInt64, ncclSum, comm, ctx->stream));
        q->rs_active = true;
        q->rs_cells_per = per;
        q->rs_cell0 = std::min<int64_t>(P.n_cells, (int64_t)ctx->comm_rank * per);
        q->rs_cell1 = std::min<int64_t>(P.n_cells, q->rs_cell0 + per);
    }
    if (has_max) SYBL_NCCL(ncclAllReduce(q->d_max, q->d_max, (size_t)q->n_max_words, ncclInt64, ncclMax, comm, ctx->stream));
    SYBL_NCCL(ncclGroupEnd());
    return SYBL_OK;
}

}  // extern "C"
