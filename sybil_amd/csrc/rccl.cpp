// rccl.cpp -- the one collective on the path: the merge of per-GPU partial group tables.
//
// Reference: CombineResults / Result.Combine / BasicHist.Combine (aggregate.go:414-467,
// query_spec.go:138-193, hist_basic.go:259-279) fold per-block results on one host; the
// reference's only multi-node mechanism ships gob files to `sybil aggregate`
// (node_aggregator.go:147-177).  Here every rank holds an identically laid out integer
// table, so the merge is one SUM all-reduce (counts, sums, buckets) plus one MAX
// all-reduce (extrema; minima are stored negated) over RCCL / xGMI.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "engine.h"
#include "rccl_lazy.h"

using namespace sybl;

namespace sybl {
const RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        void *lib = nullptr;
        auto find = [&](const char *name) -> void * {
            if (void *p = dlsym(RTLD_DEFAULT, name)) return p;  // already in the process (the host's, torch's, a preloaded stand-in)
            if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            return lib ? dlsym(lib, name) : nullptr;
        };
#define SYBL_BIND(field, sym) api.field = (decltype(api.field))find(sym)
        SYBL_BIND(GetUniqueId, "ncclGetUniqueId");
        SYBL_BIND(CommInitRank, "ncclCommInitRank");
        SYBL_BIND(CommDestroy, "ncclCommDestroy");
        SYBL_BIND(GetErrorString, "ncclGetErrorString");
        SYBL_BIND(AllReduce, "ncclAllReduce");
        SYBL_BIND(AllGather, "ncclAllGather");
        SYBL_BIND(ReduceScatter, "ncclReduceScatter");
        SYBL_BIND(GroupStart, "ncclGroupStart");
        SYBL_BIND(GroupEnd, "ncclGroupEnd");
#undef SYBL_BIND
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GetErrorString && api.AllReduce && api.AllGather && api.ReduceScatter &&
                 api.GroupStart && api.GroupEnd;
        if (!api.ok) api.why = "librccl.so.1 is neither in this process nor on the library path";
    });
    return api;
}
}  // namespace sybl

static int nccl_fail(ncclResult_t r, const char *what) {
    return fail(SYBL_E_NODEVICE, "RCCL error: %s in %s", ncclGetErrorString(r), what);
}
#define SYBL_NCCL(expr)                                   \
    do {                                                  \
        ncclResult_t r__ = (expr);                        \
        if (r__ != ncclSuccess) return nccl_fail(r__, #expr); \
    } while (0)

// ... between ncclGroupStart and ncclGroupEnd: a failing call closes the group before it returns
#define SYBL_NCCL_G(expr)                                 \
    do {                                                  \
        ncclResult_t r__ = (expr);                        \
        if (r__ != ncclSuccess) {                         \
            (void)ncclGroupEnd();                         \
            return nccl_fail(r__, #expr);                 \
        }                                                 \
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

// true on every rank iff `mine` is true on every rank (one int64 MIN all-reduce, waited for)
int comm_all_agree(Ctx *ctx, bool mine, bool *all) {
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator");
    DevOwner own;
    SYBL_HIP(hipMalloc(&own.p, 8));
    int64_t v = mine ? 1 : 0;
    SYBL_HIP(hipMemcpyAsync(own.p, &v, 8, hipMemcpyHostToDevice, ctx->stream));
    ncclResult_t nr = ncclAllReduce(own.p, own.p, 1, ncclInt64, ncclMin, (ncclComm_t)ctx->comm, ctx->stream);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(&v, own.p, 8, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllReduce(agree)");
    if (e != hipSuccess) return hip_fail(e, "agree");
    *all = v != 0;
    return SYBL_OK;
}

int comm_allreduce_u32_sum(Ctx *ctx, uint32_t *buf, size_t n) {
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator");
    SYBL_NCCL(ncclAllReduce(buf, buf, n, ncclUint32, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return SYBL_OK;
}

// Hash group-by across ranks: which keys exist differs from rank to rank, so the ranks first agree on the sorted union of
// their key lists (all-gather of the counts, all-gather of the padded lists, sort + unique on every GPU), re-lay their
// dense arrays out over it, and then merge with the usual SUM (+ MAX) all-reduce.
static int query_hash_allreduce(Query *q) {
    Ctx *ctx = q->ctx;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    const int R = ctx->comm_nranks, me = ctx->comm_rank;
    int rc = query_hash_compact(q);
    if (rc) return rc;
    int64_t *d_counts = nullptr;
    SYBL_HIP(hipMalloc((void **)&d_counts, (size_t)R * 8));
    std::vector<int64_t> counts((size_t)R, 0);
    counts[(size_t)me] = q->hash_live;
    SYBL_HIP(hipMemcpyAsync(d_counts + me, &counts[(size_t)me], 8, hipMemcpyHostToDevice, st));
    ncclResult_t nr = ncclAllGather(d_counts + me, d_counts, 1, ncclInt64, comm, st);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(counts.data(), d_counts, (size_t)R * 8, hipMemcpyDeviceToHost, st) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_counts);
    if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllGather(key counts)");
    if (e != hipSuccess) return hip_fail(e, "key counts");
    int64_t per = 0;
    for (int r = 0; r < R; r++) per = std::max(per, counts[(size_t)r]);
    if (per > 0) {
        uint64_t *d_lists = nullptr, *d_union = nullptr;
        int64_t n_union = 0;
        DevOwner own_lists;  // (freed on every exit)
        SYBL_HIP(hipMalloc(&own_lists.p, (size_t)(per * R) * 8));
        d_lists = (uint64_t *)own_lists.p;
        e = hipMemsetAsync(d_lists, 0xFF, (size_t)(per * R) * 8, st);  // padding = kHashEmpty, sorts last
        if (e == hipSuccess && q->hash_live > 0)
            e = hipMemcpyAsync(d_lists + (int64_t)me * per, q->d_dense_keys, (size_t)q->hash_live * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) nr = ncclAllGather(d_lists + (int64_t)me * per, d_lists, (size_t)per, ncclUint64, comm, st);
        if (e == hipSuccess && nr == ncclSuccess) rc = hash_union_of_lists(q, d_lists, per * R, &d_union, &n_union);
        if (e == hipSuccess && nr == ncclSuccess && !rc) rc = query_hash_install_union_device(q, d_union, n_union);
        (void)hipStreamSynchronize(st);
        if (d_union) (void)hipFree(d_union);
        if (e != hipSuccess) return hip_fail(e, "hash key exchange");
        if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllGather(keys)");
        if (rc) return rc;
    }
    const int64_t n = q->hash_live;
    // count distinct: every rank fills a sketch per key of the UNION (its own rows), then the register-wise maximum
    if (q->n_distinct) {
        q->distinct_pending = true;  // (the key set just changed under whatever an earlier pass made)
        if ((rc = query_hash_distinct(q))) return rc;
    }
    SYBL_NCCL(ncclGroupStart());
    if (q->n_distinct && n > 0) SYBL_NCCL_G(ncclAllReduce(q->d_hll, q->d_hll, (size_t)q->hll_bytes, ncclUint8, ncclMax, comm, st));
    SYBL_NCCL_G(ncclAllReduce(q->d_dense_sum, q->d_dense_sum, (size_t)hash_dense_sum_words(q, n), ncclInt64, ncclSum, comm, st));
    if (q->plan.n_max_fields > 0 && n > 0)
        SYBL_NCCL_G(ncclAllReduce(q->d_dense_max, q->d_dense_max, (size_t)hash_dense_max_words(q, n), ncclInt64, ncclMax, comm, st));
    SYBL_NCCL(ncclGroupEnd());
    return SYBL_OK;
}

// Every rank's outlier log into every rank's log (after the header, with the log's cursor, was summed): the counts
// are all-gathered, the logs all-gathered padded to the longest, and the pieces closed up in rank order.  More records
// than the log holds: the result stays marked partial (the printers then refuse rows with outliers, as on one GPU).
static int gather_outlier_logs(Query *q, int64_t local) {
    Ctx *ctx = q->ctx;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    const int R = ctx->comm_nranks, me = ctx->comm_rank;
    local = std::min<int64_t>(local, q->out_cap);  // (records beyond the capacity were only counted)
    DevOwner own_counts, own_all;  // (freed on every exit)
    SYBL_HIP(hipMalloc(&own_counts.p, (size_t)R * 8));
    int64_t *d_counts = (int64_t *)own_counts.p;
    std::vector<int64_t> counts((size_t)R, 0);
    SYBL_HIP(hipMemcpyAsync(d_counts + me, &local, 8, hipMemcpyHostToDevice, st));
    ncclResult_t nr = ncclAllGather(d_counts + me, d_counts, 1, ncclInt64, comm, st);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(counts.data(), d_counts, (size_t)R * 8, hipMemcpyDeviceToHost, st) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllGather(outlier counts)");
    if (e != hipSuccess) return hip_fail(e, "outlier counts");
    int64_t per = 0, total = 0;
    for (int r = 0; r < R; r++) {
        per = std::max(per, counts[(size_t)r]);
        total += counts[(size_t)r];
    }
    if (total > q->out_cap) return SYBL_OK;  // stays partial
    if (per > 0) {
        const size_t words = (size_t)per * kOutLogWords;
        SYBL_HIP(hipMalloc(&own_all.p, words * (size_t)R * 8));
        int64_t *d_all = (int64_t *)own_all.p;
        e = hipMemcpyAsync(d_all + (size_t)me * words, q->d_out_log, (size_t)local * kOutLogWords * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) nr = ncclAllGather(d_all + (size_t)me * words, d_all, words, ncclInt64, comm, st);
        int64_t at = 0;
        for (int r = 0; r < R && e == hipSuccess && nr == ncclSuccess; r++) {
            e = hipMemcpyAsync(q->d_out_log + at * kOutLogWords, d_all + (size_t)r * words, (size_t)counts[(size_t)r] * kOutLogWords * 8,
                               hipMemcpyDeviceToDevice, st);
            at += counts[(size_t)r];
        }
        (void)hipStreamSynchronize(st);
        if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllGather(outlier logs)");
        if (e != hipSuccess) return hip_fail(e, "outlier logs");
    }
    q->out_log_partial = false;
    return SYBL_OK;
}

}  // namespace sybl

extern "C" {

int sybl_comm_unique_id(void *id128) {
    if (!id128) return fail(SYBL_E_INVAL, "id buffer is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!rccl().ok) return fail(SYBL_E_NODEVICE, "no RCCL: %s", rccl().why);
    ncclUniqueId id;
    SYBL_NCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return SYBL_OK;
}

int sybl_comm_init(sybl_ctx *ctx, const void *id128, int32_t nranks, int32_t rank) {
    SYBL_API_GUARD(ctx);
    if (!ctx || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(SYBL_E_INVAL, "sybl_comm_init: bad argument");
    if (ctx->comm) return fail(SYBL_E_STATE, "communicator already initialised");
    if (!rccl().ok) return fail(SYBL_E_NODEVICE, "no RCCL: %s", rccl().why);
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
    SYBL_API_GUARD(ctx);
    if (!ctx || !ctx->comm) return SYBL_OK;
    hipSetDevice(ctx->device);
    ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_nranks = 1;
    ctx->comm_rank = 0;
    return SYBL_OK;
}

int sybl_query_allreduce(sybl_query *q) {
    SYBL_API_GUARD(q);
    if (!q) return fail(SYBL_E_INVAL, "query is NULL");
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_allreduce before sybl_query_scan");
    Ctx *ctx = q->ctx;
    if (!ctx->comm) return fail(SYBL_E_STATE, "no communicator: call sybl_comm_init first");
    SYBL_HIP(hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    {
        int lrc = query_check_layout(q);
        if (lrc) return lrc;
    }
    if (ctx->comm_nranks > 1) q->out_log_partial = true;  // (until gather_outlier_logs has brought every rank's in)
    // Outlier values (plan.h: outlier log): every rank logged its own; the merged result needs all of them
    // (hist_basic.go:132-142,221-257: printed as buckets of their own).  The local count is read before the header is
    // summed -- a host round trip, but only queries whose column bounds allow an outlier at all keep a log.
    int64_t out_local = -1;
    if (q->d_out_log && (ctx->comm_nranks > 1 || env("SYBL_FORCE_SCATTER"))) {
        SYBL_HIP(hipMemcpyAsync(&out_local, q->d_sum + kHdrOutLog, 8, hipMemcpyDeviceToHost, ctx->stream));
        SYBL_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (q->hash_mode) {
        int rc = query_hash_allreduce(q);
        if (!rc && out_local >= 0) rc = gather_outlier_logs(q, out_local);
        return rc;
    }
    const ScanPlan &P = q->plan;
    const bool has_max = P.n_max_fields > 0;  // (cfg 3: no extremum is tracked -- ONE collective per step)
    const int64_t hist_words = (int64_t)P.n_cells * P.hist_stride, small_words = q->n_sum_words - (hist_words + (P.hist_stride > 0 ? kMaxScatterRanks * P.hist_stride : 0));
    // Big bucket tables (config 4: 65 536 cells x 1002 buckets = 525 MB per rank): all-reducing them moves ~2x the table
    // over every GPU's links although rank 0 only needs the percentiles, the bucket moments, the Cumulative buckets
    // and the printed rows' arrays.  Instead the bucket arrays are reduce-SCATTERED over equal slices of cells (each
    // rank receives 1/R of the table), every rank summarises its slice (k_hist_summary / k_hist_total in
    // query_snapshot) and the summaries are all-gathered: 52 MB of percentiles instead of 525 MB of buckets.
    // snapshot and finalize then are collective calls.
    const bool big_limited = P.hist_stride > 0 && query_wants_hist_summary(q) && q->limit > 0 && (ctx->comm_nranks > 1 || env("SYBL_FORCE_SCATTER"));
    // A printer (sybl_query_desc.printed_only) looks at `limit` rows and Cumulative (aggregate.go:469-525, printer.go:291-308):
    // the bucket table then does not travel at all.  The cell fields are all-reduced here -- every rank derives the same
    // sort order from them --, Cumulative's buckets are summed locally and all-reduced by the snapshot (8 KB), the printed
    // rows' arrays gathered locally and all-reduced by the finalize (config 4, -limit 100: 0.8 MB; the reduce-scatter
    // below moves 263 MB per rank and step, and the summaries 52 MB more).
    const bool top_merge = big_limited && q->printed_only;
    const bool scatter = big_limited && !top_merge && ctx->comm_nranks <= kMaxScatterRanks;
    // int32 slices: decided once per query, from a bound every rank computes alike -- the largest shard's rows (one
    // blocking MAX all-reduce at the query's first collective) times the ranks; a bucket of the merged table cannot
    // exceed that.  Weighted queries keep int64 (a bucket holds a sum of weights).
    if (scatter && q->rs_int32 < 0) {
        DevOwner own_rows;
        int64_t rows = q->stats.rows_scanned;
        SYBL_HIP(hipMalloc(&own_rows.p, 8));
        int64_t *d_rows = (int64_t *)own_rows.p;
        SYBL_HIP(hipMemcpyAsync(d_rows, &rows, 8, hipMemcpyHostToDevice, ctx->stream));
        ncclResult_t nr = ncclAllReduce(d_rows, d_rows, 1, ncclInt64, ncclMax, comm, ctx->stream);
        hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(&rows, d_rows, 8, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (nr != ncclSuccess) return nccl_fail(nr, "ncclAllReduce(shard rows)");
        if (e != hipSuccess) return hip_fail(e, "shard rows");
        q->rs_int32 = (!q->weighted && !env("SYBL_NO_SCATTER32") && rows * (int64_t)ctx->comm_nranks < ((int64_t)1 << 31)) ? 1 : 0;
    }
    // (everything that can fail on its own -- the int32 staging buffer, k_pack32 -- runs before the group is opened: an
    // early return between ncclGroupStart and ncclGroupEnd would leave the group open)
    if (scatter && q->rs_int32 == 1) {
        const int64_t R = ctx->comm_nranks, per = (P.n_cells + R - 1) / R, count = per * P.hist_stride;
        if (!q->d_h32) SYBL_HIP(hipMalloc((void **)&q->d_h32, (size_t)(count * (R + 1)) * 4));
        hipError_t e = launch_pack32(q->d_sum + P.hist_off, q->d_h32, count * R, ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "k_pack32");
    }
    // a pushed-down scan (pushdown.hip) keeps Cumulative's maxima in header words: they merge by MAX, not with the header's SUM --
    // set aside before the group, all-reduced in it, put back behind it
    const bool pd_max = q->pushdown && q->pushdown_ran;
    if (pd_max) {
        if (!q->d_pd_max) SYBL_HIP(hipMalloc((void **)&q->d_pd_max, (size_t)kMaxAggs * 8));
        SYBL_HIP(hipMemcpyAsync(q->d_pd_max, q->d_sum + kHdrPdMax, (size_t)kMaxAggs * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    SYBL_NCCL(ncclGroupStart());
    if (pd_max) SYBL_NCCL_G(ncclAllReduce(q->d_pd_max, q->d_pd_max, (size_t)kMaxAggs, ncclInt64, ncclMax, comm, ctx->stream));
    if (top_merge) {
        SYBL_NCCL_G(ncclAllReduce(q->d_sum, q->d_sum, (size_t)small_words, ncclInt64, ncclSum, comm, ctx->stream));
        q->top_merge = true;
    } else if (!scatter) {
        SYBL_NCCL_G(ncclAllReduce(q->d_sum, q->d_sum, (size_t)(small_words + hist_words), ncclInt64, ncclSum, comm, ctx->stream));
    } else {
        const int64_t R = ctx->comm_nranks, per = (P.n_cells + R - 1) / R, count = per * P.hist_stride;
        int64_t *H = q->d_sum + P.hist_off;
        SYBL_NCCL_G(ncclAllReduce(q->d_sum, q->d_sum, (size_t)small_words, ncclInt64, ncclSum, comm, ctx->stream));
        // (cells past n_cells in the last slice are the zeroed padding of the SUM section)
        if (q->rs_int32 == 1) {
            SYBL_NCCL_G(ncclReduceScatter(q->d_h32, q->d_h32 + count * R, (size_t)count, ncclInt32, ncclSum, comm, ctx->stream));
        } else {
            SYBL_NCCL_G(ncclReduceScatter(H, H + (int64_t)ctx->comm_rank * count, (size_t)count, ncclInt64, ncclSum, comm, ctx->stream));
        }
        q->rs_active = true;
        q->rs_cells_per = per;
        q->rs_cell0 = std::min<int64_t>(P.n_cells, (int64_t)ctx->comm_rank * per);
        q->rs_cell1 = std::min<int64_t>(P.n_cells, q->rs_cell0 + per);
    }
    if (has_max) SYBL_NCCL_G(ncclAllReduce(q->d_max, q->d_max, (size_t)q->n_max_words, ncclInt64, ncclMax, comm, ctx->stream));
    // count distinct: Result.Combine merges the sketches register by register (query_spec.go:180-188)
    if (q->n_distinct) SYBL_NCCL_G(ncclAllReduce(q->d_hll, q->d_hll, (size_t)q->hll_bytes, ncclUint8, ncclMax, comm, ctx->stream));
    SYBL_NCCL(ncclGroupEnd());
    if (pd_max) SYBL_HIP(hipMemcpyAsync(q->d_sum + kHdrPdMax, q->d_pd_max, (size_t)kMaxAggs * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (scatter && q->rs_int32 == 1) {
        const int64_t R = ctx->comm_nranks, per = (P.n_cells + R - 1) / R, count = per * P.hist_stride;
        hipError_t e = launch_unpack32(q->d_h32 + count * R, q->d_sum + P.hist_off + (int64_t)ctx->comm_rank * count, count, ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "k_unpack32");
    }
    if (out_local >= 0) {
        int rc = gather_outlier_logs(q, out_local);
        if (rc) return rc;
    }
    return SYBL_OK;
}

}  // extern "C"
