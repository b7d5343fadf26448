// engine.cpp -- context, error reporting, the scan driver and the query half of the C ABI of
// include/sybilgpu.h (the planner lives in planner.cpp, tables in table.cpp, finalize in result.cpp).
//
// Reference mapping (src/lib/ of logv/sybil):
//   Table / Column            <- table.go, table_column.go, record_slab.go (AoS row slabs become
//                                dense per-column arrays in HBM)
//   sybl_table_append_block   <- the output of LoadBlockFromDir (table_block_io.go:225-310)
//   block skipping            <- ShouldLoadBlockFromDir (table_block_io.go:110-182)
//   sybl_query_prepare        <- BuildFilters/Grouping/Aggregation (filter.go:59, query_spec.go:214-219)
//   sybl_query_scan           <- the block loop of LoadAndQueryRecords (table_query.go:96-231)
// There is no CPU fallback in this file: without a HIP device every call fails.
#include "engine.h"
#include "hll.h"
#include "re2lite.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>
#include <regex>

namespace sybl {

extern "C" char **environ;
const char *env(const char *name) {
    static std::unordered_map<std::string, std::string> snap;
    static bool live = false;
    static std::once_flag once;
    std::call_once(once, [] {
        for (char **e = environ; e && *e; e++) {
            if (strncmp(*e, "SYBL_", 5) != 0) continue;
            const char *eq = strchr(*e, '=');
            if (eq) snap.emplace(std::string(*e, (size_t)(eq - *e)), std::string(eq + 1));
        }
        live = snap.count("SYBL_ENV_LIVE") != 0;
    });
    if (live) return getenv(name);
    auto it = snap.find(name);
    return it == snap.end() ? nullptr : it->second.c_str();
}


static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return e == hipErrorOutOfMemory ? SYBL_E_NOMEM : SYBL_E_NODEVICE;
}

int load_sync_all(Ctx *ctx) {
    if (ctx->load_flush) {  // (launches the load still holds back: loader.cpp)
        int rc = ctx->load_flush();
        if (rc) return rc;
    }
    if (!ctx->load_multi) return SYBL_OK;
    for (int i = 0; i < ctx->n_load_streams; i++) SYBL_HIP(hipStreamSynchronize(ctx->load_streams[i]));
    return SYBL_OK;
}

Column *Table::find(const char *n) const {
    if (!n) return nullptr;
    auto it = col_ix.find(n);
    return it == col_ix.end() ? nullptr : cols[it->second].get();
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Side streams get a priority of their own.  HIP multiplexes the streams of one priority onto a few hardware queues
// (in order within a queue), so a normal-priority side stream can land on the queue the caller's stream feeds and sit
// behind a whole scan; the queues of another priority level are separate.  toward > 0: lowest priority, < 0: highest.
hipError_t create_side_stream(hipStream_t *out, int toward) {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, toward < 0 ? greatest : least);
}

static void free_query(Query *q) {
    if (!q) return;
    if (!q->table_gone) {
        query_finish_lazy_results(q);  // (their rows are built from this query's metadata and its table's dictionaries: now, or never)
        if (q->t) {
            auto &v = q->t->queries;
            v.erase(std::remove(v.begin(), v.end(), q), v.end());
        }
    }
    q->eff_weight.reset();  // (shared with the weight column's cache: freed with its last holder)
    if (q->d_plan) hipFree(q->d_plan);
    if (q->d_preplan) hipFree(q->d_preplan);
    if (q->d_prebits) hipFree(q->d_prebits);
    if (q->d_segs) hipFree(q->d_segs);
    if (q->d_wg_seg_begin) hipFree(q->d_wg_seg_begin);
    if (q->d_wg_cell_base) hipFree(q->d_wg_cell_base);
    if (q->d_recs) hipFree(q->d_recs);
    if (q->d_h32) hipFree(q->d_h32);
    if (q->d_cursor) hipFree(q->d_cursor);
    for (void *p : q->d_idmasks) hipFree(p);
    if (q->own_partials) {
        if (q->d_sum) hipFree(q->d_sum);
        if (q->d_max) hipFree(q->d_max);
    }
    if (q->d_ws_sum) hipFree(q->d_ws_sum);
    if (q->d_ws_max) hipFree(q->d_ws_max);
    if (q->h_max) hipHostFree(q->h_max);
    for (auto &e : q->ev)
        if (e) hipEventDestroy(e);
    if (q->snap_on_aux && q->ev_snap) (void)hipEventSynchronize(q->ev_snap);  // (a copy may still be reading the tables)
    if (q->ev_snap) hipEventDestroy(q->ev_snap);
    if (q->ev_ready) hipEventDestroy(q->ev_ready);
    if (q->d_pct) hipFree(q->d_pct);
    if (q->d_mom) hipFree(q->d_mom);
    if (q->d_total) hipFree(q->d_total);
    if (q->h_top) hipHostFree(q->h_top);
    q->h_pct_buf.reset();
    if (q->h_mom) hipHostFree(q->h_mom);
    if (q->h_total) hipHostFree(q->h_total);
    if (q->d_top) hipFree(q->d_top);
    if (q->d_top_cells) hipFree(q->d_top_cells);
    if (q->d_out_log) hipFree(q->d_out_log);
    if (q->d_out_stage) hipFree(q->d_out_stage);
    if (q->d_multi) hipFree(q->d_multi);
    if (q->d_dplan) hipFree(q->d_dplan);
    if (q->d_pd) hipFree(q->d_pd);
    if (q->d_pd_max) hipFree(q->d_pd_max);
    if (q->d_hll) hipFree(q->d_hll);
    if (q->d_hll_idhash) hipFree(q->d_hll_idhash);
    if (q->d_hll_chars) hipFree(q->d_hll_chars);
    if (q->d_hll_stroff) hipFree(q->d_hll_stroff);
    for (auto &kv : q->replaced) {
        if (kv.second->d_keys) hipFree(kv.second->d_keys);
        if (kv.second->d_ranks) hipFree(kv.second->d_ranks);
    }
    query_hash_free(q);
    delete q;
}
static int ensure_partials(Query *q) {
    if (q->d_sum && q->d_max) return SYBL_OK;
    SYBL_HIP(hipMalloc((void **)&q->d_sum, (size_t)q->n_sum_words * 8));
    SYBL_HIP(hipMalloc((void **)&q->d_max, (size_t)q->n_max_words * 8));
    SYBL_HIP(hipMemset(q->d_sum, 0, (size_t)q->n_sum_words * 8));
    SYBL_HIP(hipMemset(q->d_max, 0, (size_t)q->n_max_words * 8));
    q->own_partials = true;
    q->plan_dirty = true;
    return SYBL_OK;
}

// Count distinct: the sketches start empty and are filled by a pass of their own behind the scan (distinct.hip).
static int scan_distinct(Query *q, bool ran, hipStream_t st) {
    if (!q->n_distinct) return SYBL_OK;
    if (q->hash_mode) {
        // which Results exist is only known once the table has been compacted (and, across ranks, the key sets united): the
        // pass runs then, over the same rows (query_hash_distinct, from the snapshot / the all-reduce)
        q->distinct_pending = ran;
        q->hll_bytes = 0;
        return SYBL_OK;
    }
    SYBL_HIP(hipMemsetAsync(q->d_hll, 0, (size_t)q->hll_bytes, st));
    if (!ran) return SYBL_OK;
    hipError_t e = launch_scan_distinct(q->d_dplan, q->dplan.n_slots, q->n_wg, st);
    if (e != hipSuccess) return hip_fail(e, "k_scan_distinct");
    return SYBL_OK;
}

// Count distinct over a hashed group-by (aggregate.go:205-243 with the map of aggregate.go:186-200): a sketch per key of the
// dense, sorted key list -- sized here, because only now is the list final --, filled by k_scan_distinct over the rows the scan
// saw, each row finding its key's place by binary search.
int query_hash_distinct(Query *q) {
    if (!q->hash_mode || !q->n_distinct) return SYBL_OK;
    hipStream_t st = q->ctx->stream;
    const int64_t n = q->hash_live;
    const int64_t bytes = std::max<int64_t>(n, 1) * (int64_t)kHllRegs;
    if (bytes > ((int64_t)8 << 30))
        return fail(SYBL_E_INVAL, "count distinct: %lld groups x 16 KB of sketch exceed 8 GiB", (long long)n);
    if (!q->distinct_pending && q->hll_bytes == bytes) return SYBL_OK;  // (the sketches of this key set are there)
    if (q->d_hll) SYBL_HIP(hipFree(q->d_hll));
    q->d_hll = nullptr;
    SYBL_HIP(hipMalloc((void **)&q->d_hll, (size_t)bytes));
    SYBL_HIP(hipMemsetAsync(q->d_hll, 0, (size_t)bytes, st));
    q->hll_bytes = bytes;
    if (n > 0 && q->scanned) {
        q->dplan.hll = q->d_hll;
        q->dplan.hll_keys = q->d_dense_keys;
        q->dplan.hll_nkeys = n;
        SYBL_HIP(hipMemcpyAsync(q->d_dplan, &q->dplan, sizeof(ScanPlan), hipMemcpyHostToDevice, st));
        SYBL_HIP(hipStreamSynchronize(st));  // (the plan is read from pageable memory)
        hipError_t e = launch_scan_distinct(q->d_dplan, q->dplan.n_slots, q->n_wg, st);
        if (e != hipSuccess) return hip_fail(e, "k_scan_distinct");
    }
    q->distinct_pending = false;
    return SYBL_OK;
}

// Outlier log (plan.h): the stripes' cursors start from zero; behind the scan kernels the stripes are closed up into the
// dense log and the record count lands in the header.
static int out_log_begin(Query *q, hipStream_t st) {
    if (q->d_out_stage) SYBL_HIP(hipMemsetAsync(q->d_out_stage, 0, (size_t)kOutStripes * kOutCursorWords * 8, st));
    return SYBL_OK;
}
static int out_log_end(Query *q, bool ran, hipStream_t st) {
    if (!q->d_out_stage || !ran) return SYBL_OK;
    hipError_t e = launch_outlog_gather(q->d_out_stage, q->out_cap, q->d_out_log, q->d_sum, st);
    if (e != hipSuccess) return hip_fail(e, "k_outlog_gather");
    return SYBL_OK;
}

static int scan(Query *q) {
    PhaseTrace trace("scan");
    int rc = ensure_partials(q);
    if (rc) return rc;
    trace.mark("partials");
    q->rs_active = false;
    q->top_merge = false;
    q->out_log_partial = false;
    hipStream_t st = q->ctx->stream;
    ScanPlan &P = q->plan;
    if (q->snap_on_aux && q->ev_snap) {
        // the previous snapshot of this query is (or was) copied out by the auxiliary stream: nothing below may touch the
        // tables before it is done (a wait on the GPU's side, not the host's)
        SYBL_HIP(hipStreamWaitEvent(st, q->ev_snap, 0));
        q->snap_on_aux = false;
    }
    if (q->hash_mode && (rc = query_hash_reset(q))) return rc;  // (allocates the key table on first use: before the plan is copied)
    if (q->plan_dirty) {
        P.sum_out = q->d_sum;
        P.max_out = q->d_max;
        // synchronous copy: the plan object may be rewritten by the host right after
        SYBL_HIP(hipMemcpy(q->d_plan, &P, sizeof(ScanPlan), hipMemcpyHostToDevice));
        q->plan_dirty = false;
    }
    const bool ran = !q->never_matches && !q->segs.empty();
    // several ranks: the pushed-down scan is only taken when EVERY rank planned it (a rank without rows plans no partitioned
    // histograms at all) -- asked once per prepared query, a collective call of every rank's first scan
    if (q->pd_static && q->pd_agreed < 0) {
        bool all = false;
        if ((rc = comm_all_agree(q->ctx, q->pushdown, &all))) return rc;
        q->pd_agreed = all ? 1 : 0;
        if (!all) q->pushdown = false;
    }
    const bool pd_ranks = q->pushdown && q->ctx->comm && q->ctx->comm_nranks > 1;  // (its count all-reduce: every rank, rows or not)
    hipError_t e = hipSuccess;
    if ((rc = out_log_begin(q, st))) return rc;
    q->pushdown_ran = false;
    if (q->pushdown && (ran || pd_ranks)) {
        q->pushdown_ran = true;
        // -limit pushed into the scan (pushdown.hip): group counts from the key column, the printed cells chosen on the
        // device, then ONE pass over key + value that fills Cumulative and the printed groups only
        PushdownPlan &D = q->dplan_pd;
        const ScanPlan &PP = q->plan;
        if ((rc = query_total_buffers(q))) return rc;
        // header, every cell field (Count is rewritten by the fold; the sums start from zero), Cumulative's buckets, the carries
        SYBL_HIP(hipMemsetAsync(q->d_sum, 0, ((size_t)kHeaderWords + (size_t)PP.n_sum_fields * (size_t)PP.n_cells) * 8, st));
        SYBL_HIP(hipMemsetAsync(q->d_total, 0, (size_t)PP.hist_stride * 8, st));
        SYBL_HIP(hipMemsetAsync(D.carry, 0, (size_t)PP.n_cells * 4, st));
        SYBL_HIP(hipMemsetAsync(D.n_top, 0, 4, st));
        e = launch_fill64(q->d_max, q->n_max_words, INT64_MIN, st);
        if (e == hipSuccess) e = launch_fill64(q->d_sum + kHdrPdMax, kMaxAggs, INT64_MIN, st);
        if (e != hipSuccess) return hip_fail(e, "k_fill64");
        SYBL_HIP(hipEventRecord(q->ev[0], st));
        D.fp.segs = q->d_segs;
        D.fp.wg_seg_begin = q->d_wg_seg_begin;
        D.sum_out = q->d_sum;
        D.max_out = q->d_max;
        D.total = q->d_total;
        e = launch_pushdown_count(D, st);
        if (e != hipSuccess) return hip_fail(e, "k_pd_count");
        // across ranks: the groups' counts over ALL shards decide which cells are printed (the cells' Count fields stay this
        // rank's own: the merge sums them later)
        if (pd_ranks && (rc = comm_allreduce_u32_sum(q->ctx, D.cnt, (size_t)PP.n_cells))) return rc;
        e = launch_pushdown_scan(D, st);
        if (e != hipSuccess) return hip_fail(e, "k_pd_*");
        SYBL_HIP(hipEventRecord(q->ev[1], st));
        SYBL_HIP(hipEventRecord(q->ev[2], st));
        q->scanned = true;
        q->snapshot_pending = false;
        return SYBL_OK;
    }
    if (q->part_hist && ran) {
        // k_part_hist overwrites every cell field, bucket and extremum: only the header and the
        // partition cursors start from zero
        // (every pass of a three / four aggregation query computes its own split: a later pass has fewer partitions and
        // may share them between workgroups when the first does not)
        bool any_split = q->pplan.split > 1;
        for (auto &pp : q->part_more) any_split = any_split || pp.H.split > 1;
        if (any_split) {
            // shared partitions accumulate with atomics into a zeroed table
            SYBL_HIP(hipMemsetAsync(q->d_sum, 0, (size_t)q->n_sum_words * 8, st));
            e = launch_fill64(q->d_max, q->n_max_words, INT64_MIN, st);
            if (e != hipSuccess) return hip_fail(e, "k_fill64");
        } else {
            // (outlier fields are accumulated with atomics: with them the cell fields start from zero as well)
            bool outliers = false;
            for (auto &ai : q->aggs) outliers = outliers || ai.d.f_out >= 0;
            const size_t words = (size_t)kHeaderWords + (outliers ? (size_t)P.n_sum_fields * (size_t)P.n_cells : 0);
            SYBL_HIP(hipMemsetAsync(q->d_sum, 0, words * 8, st));
        }
        // (the wrap log's cursor and k_part_hist's item counter: zeroed here, not between k_emit and k_part_hist)
        SYBL_HIP(hipMemsetAsync(q->pplan.wrap_log, 0, 8, st));
        SYBL_HIP(hipEventRecord(q->ev[0], st));
        q->eplan.sum_out = q->d_sum;
        // (the 64-byte chunk stores are written once and read once, by another kernel: the non-temporal hint is worth 1-2 % of
        // the scan on every placement tried -- profiles/r05_emit_nt.txt; SYBL_EMIT_PLAIN_STORES=1: without it)
        q->eplan.store_nt = env("SYBL_EMIT_PLAIN_STORES") ? 0 : 1;
        for (auto &pp : q->part_more) pp.E.store_nt = q->eplan.store_nt;
        q->pplan.sum_out = q->d_sum;
        q->pplan.max_out = q->d_max;
        // counting sort: count per (workgroup, bin) -> exact regions -> scatter.  The counts are a function of the table's rows,
        // the query's filters and key columns and the workgroups' row shares -- all fixed for the life of a prepared query (a
        // table that changes makes it stale: SYBL_E_STATE) -- and nothing downstream writes them (k_emit and k_part_hist read
        // boff; wbase is re-derived from it): like the block statistics they are taken once, by the query's first scan, and a
        // rescan skips the pass (config 4: 0.32 ms and a 2 GB read of the key column per step).  Queries of three or four
        // aggregations run the sequence twice over the one table of counts and keep counting.  SYBL_NO_COUNT_CACHE=1: every scan.
        const bool reuse_counts = q->count_cached && q->part_more.empty() && !env("SYBL_NO_COUNT_CACHE");
        if (!reuse_counts) {
            e = q->part_packed ? launch_count_packed(q->eplan, q->part_nf, q->part_ng, q->n_wg, st)
                               : launch_count(q->eplan, q->part_nf, q->part_ng, q->n_wg, st);
            if (e != hipSuccess) return hip_fail(e, "k_count");
            q->count_cached = q->part_more.empty();
        }
        q->stats.count_pass_reused = reuse_counts ? 1 : 0;
        trace.mark("count");
        e = q->part_packed ? launch_emit_packed(q->eplan, q->part_nf, q->part_ng, q->part_na, q->n_wg, st)
                           : launch_emit(q->eplan, q->part_nf, q->part_ng, q->part_na, q->n_wg, st);
        if (e != hipSuccess) return hip_fail(e, "k_emit");
        trace.mark("emit");
        // SYBL_PARTHIST_TRACE=<file> (diagnostic): k_part_hist's phase timestamps of this scan, one line per workgroup
        const char *ph_trace = env("SYBL_PARTHIST_TRACE");
        DevOwner own_trace;
        const size_t trace_words = (size_t)q->pplan.n_parts * (size_t)q->pplan.split * kPartTraceWords;
        q->pplan.trace = nullptr;
        if (ph_trace) {
            SYBL_HIP(hipMalloc(&own_trace.p, trace_words * 8));
            SYBL_HIP(hipMemsetAsync(own_trace.p, 0, trace_words * 8, st));
            q->pplan.trace = (unsigned long long *)own_trace.p;
        }
        e = launch_part_hist(q->pplan, st);
        if (e != hipSuccess) return hip_fail(e, "k_part_hist");
        if (ph_trace) {
            std::vector<unsigned long long> tr(trace_words);
            SYBL_HIP(hipStreamSynchronize(st));
            SYBL_HIP(hipMemcpy(tr.data(), own_trace.p, trace_words * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(ph_trace, "w")) {
                for (size_t w = 0; w < trace_words / kPartTraceWords; w++) {
                    for (int k = 0; k < kPartTraceWords; k++) fprintf(f, "%llu%c", tr[w * kPartTraceWords + k], k + 1 < kPartTraceWords ? ' ' : '\n');
                }
                fclose(f);
            }
            q->pplan.trace = nullptr;
        }
        e = launch_part_fix(q->pplan, st);
        if (e != hipSuccess) return hip_fail(e, "k_part_fix");
        // aggregations 2.. of a query with three or four: the same sequence over the same rows and buffers
        for (auto &pp : q->part_more) {
            SYBL_HIP(hipMemsetAsync(pp.H.wrap_log, 0, 8, st));  // (behind the previous pass's k_part_fix)
            pp.E.sum_out = q->d_sum;
            pp.H.sum_out = q->d_sum;
            pp.H.max_out = q->d_max;
            e = pp.packed ? launch_count_packed(pp.E, q->part_nf, q->part_ng, q->n_wg, st) : launch_count(pp.E, q->part_nf, q->part_ng, q->n_wg, st);
            if (e != hipSuccess) return hip_fail(e, "k_count");
            e = pp.packed ? launch_emit_packed(pp.E, q->part_nf, q->part_ng, pp.na, q->n_wg, st) : launch_emit(pp.E, q->part_nf, q->part_ng, pp.na, q->n_wg, st);
            if (e != hipSuccess) return hip_fail(e, "k_emit");
            e = launch_part_hist(pp.H, st);
            if (e != hipSuccess) return hip_fail(e, "k_part_hist");
            e = launch_part_fix(pp.H, st);
            if (e != hipSuccess) return hip_fail(e, "k_part_fix");
        }
        trace.mark("hist");
        if ((rc = out_log_end(q, ran, st))) return rc;
        SYBL_HIP(hipEventRecord(q->ev[1], st));
        SYBL_HIP(hipEventRecord(q->ev[2], st));
        if ((rc = scan_distinct(q, ran, st))) return rc;
        q->scanned = true;
        q->snapshot_pending = false;
        return SYBL_OK;
    }
    if (q->use_lds && ran && !P.windowed) {
        // the fold overwrites every cell field; only the header and the bucket arrays accumulate
        SYBL_HIP(hipMemsetAsync(q->d_sum, 0, (size_t)kHeaderWords * 8, st));
        if (P.hist_stride > 0)
            SYBL_HIP(hipMemsetAsync(q->d_sum + P.hist_off, 0, (size_t)(P.n_cells * P.hist_stride) * 8, st));
    } else {
        SYBL_HIP(hipMemsetAsync(q->d_sum, 0, (size_t)q->n_sum_words * 8, st));
        e = launch_fill64(q->d_max, q->n_max_words, INT64_MIN, st);
        if (e != hipSuccess) return hip_fail(e, "k_fill64");
    }
    SYBL_HIP(hipEventRecord(q->ev[0], st));
    if (ran && q->pre_n_slots) {
        // the filters the packed bodies do not evaluate: a row bitmap first (planner.cpp: Planner::prefilter)
        bool wrote = false;
        if (q->pre_generic_slots) {
            e = launch_prefilter(q->d_preplan, q->pre_generic_slots, q->d_prebits, q->n_wg, st);
            if (e != hipSuccess) return hip_fail(e, "k_prefilter");
            wrote = true;
        }
        for (size_t k = 0; k < q->pre_fps.size(); k++) {
            e = launch_prefilter_packed(q->pre_fps[k], q->pre_fp_nf[k], q->d_prebits, wrote, q->n_wg, st);
            if (e != hipSuccess) return hip_fail(e, "k_prefilter_packed");
            wrote = true;
        }
    }
    if (ran && q->hash_mode) {
        if (q->hash_fast) {
            q->fplan.sum_out = q->d_sum;
            q->fplan.max_out = q->d_max;
            e = q->hash_packed
                    ? launch_scan_hash_packed(q->fplan, P.hash_keys, q->fast_nf, q->fast_ng, q->fast_na, q->fast_mode, q->time_mode, P.lds_cells,
                                              P.n_sum_fields, P.n_max_fields, q->n_wg, q->lds_bytes, st)
                    : launch_scan_hash_fast(q->fplan, P.hash_keys, q->fast_nf, q->fast_ng, q->fast_na, q->fast_mode, q->time_mode, P.lds_cells,
                                      P.n_sum_fields, P.n_max_fields, q->n_wg, q->lds_bytes, st);
        } else {
            e = launch_scan_hash(q->d_plan, P.n_slots, q->n_wg, q->lds_bytes, st);
        }
        if (e != hipSuccess) return hip_fail(e, "k_scan_hash");
    } else if (ran) {
        if (q->fast) {
            q->fplan.sum_out = q->d_sum;
            q->fplan.max_out = q->d_max;
            if (q->fast_packed_n) {
                e = launch_scan_packed_n(q->fplan, q->fast_nf, q->fast_ng, q->fast_na, q->fast_mode, q->time_mode, q->n_wg, q->lds_bytes, st);
            } else if (q->fast_packed) {
                e = launch_scan_packed(q->fplan, q->fast_nf, q->fast_ng, q->fast_na, q->fast_mode, q->time_mode, q->n_wg, q->lds_bytes, st);
            } else {
                e = launch_scan_fast(q->fplan, q->fast_nf, q->fast_ng, q->fast_na, q->fast_mode, q->time_mode, q->fast_gen, q->n_wg,
                                     q->lds_bytes, st);
            }
            if (e != hipSuccess) return hip_fail(e, q->fast_packed ? "k_scan_packed" : "k_scan_fast");
        } else {
            e = launch_scan(q->d_plan, P.n_slots, q->n_wg, q->use_lds, q->lds_bytes, st);
            if (e != hipSuccess) return hip_fail(e, "k_scan");
        }
    }
    if ((rc = out_log_end(q, ran, st))) return rc;
    SYBL_HIP(hipEventRecord(q->ev[1], st));
    if (q->use_lds && ran && !P.windowed) {
        int64_t wsum = (int64_t)P.n_sum_fields * P.n_cells, wmax = (int64_t)P.n_max_fields * P.n_cells;
        e = launch_fold(q->d_ws_sum, q->d_sum + kHeaderWords, wsum, q->d_ws_max, q->d_max, wmax, q->n_wg, st);
        if (e != hipSuccess) return hip_fail(e, "k_fold");
    }
    SYBL_HIP(hipEventRecord(q->ev[2], st));
    if ((rc = scan_distinct(q, ran, st))) return rc;
    q->scanned = true;
    q->snapshot_pending = false;
    return SYBL_OK;
}

// A partition buffer overflowed (badly skewed keys): redo the scan with per-value atomics.
int query_rescan_without_part_hist(Query *q) {
    q->part_hist = false;
    q->stats.packed_kernel = q->fast && q->fast_packed;
    q->stats.strategy = q->use_lds ? (q->plan.windowed ? (q->fast ? 4 : 3) : (q->fast ? 2 : 0)) : (q->hash_mode ? 7 : 1);
    return scan(q);
}

}  // namespace sybl

using namespace sybl;

// ==================================================================== C ABI

extern "C" {

int sybl_abi_version(void) { return SYBL_ABI_VERSION; }

const char *sybl_last_error(void) { return g_err.c_str(); }

int sybl_init(int device, sybl_ctx **out) {
    if (!out) return fail(SYBL_E_INVAL, "sybl_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SYBL_E_NODEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(SYBL_E_INVAL, "device %d out of range (have %d)", device, n);
    SYBL_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SYBL_HIP(hipGetDeviceProperties(&prop, device));
    sybl_ctx *c = new sybl_ctx();
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    c->dev_name = prop.name;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipStreamCreate");
    }
    c->stream = c->own_stream;
    // The loader's streams, made NOW: the runtime spreads streams over its few hardware queues as they are created, and a
    // process that had made and destroyed many streams before its first load (bench.py after its query configs) got all
    // sixteen onto one or two of them -- the blocks' kernel chains then ran one after the other, a 0.065 s open took 0.14 s
    // (profiles/r06_gpu_varint.txt).  SYBL_LOADER_STREAMS says how many a load uses.
    for (int i = 0; i < 16; i++)
        if (hipStreamCreateWithFlags(&c->load_streams[i], hipStreamNonBlocking) != hipSuccess) {
            c->load_streams[i] = nullptr;
            (void)hipGetLastError();
            break;
        }
    *out = c;
    return SYBL_OK;
}

void sybl_shutdown(sybl_ctx *ctx) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return;
    sybl_comm_free(ctx);
    ctx_free_load_arena(ctx);
    if (ctx->h2d_stage) (void)hipHostFree(ctx->h2d_stage);
    if (ctx->d_copy_digest) (void)hipFree(ctx->d_copy_digest);
    if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    for (auto &ls : ctx->load_streams)
        if (ls) hipStreamDestroy(ls);
    if (ctx->own_stream) {
        hipStreamSynchronize(ctx->own_stream);
        hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
}

int sybl_ctx_set_stream(sybl_ctx *ctx, void *hip_stream) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return fail(SYBL_E_INVAL, "ctx is NULL");
    SYBL_HIP(hipSetDevice(ctx->device));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SYBL_OK;
}

int sybl_ctx_sync(sybl_ctx *ctx) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return fail(SYBL_E_INVAL, "ctx is NULL");
    SYBL_HIP(hipSetDevice(ctx->device));
    SYBL_HIP(hipStreamSynchronize(ctx->stream));
    return SYBL_OK;
}

int sybl_device_info(sybl_ctx *ctx, char *name, size_t name_cap, int *n_cus, int64_t *hbm_bytes) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return fail(SYBL_E_INVAL, "ctx is NULL");
    if (name && name_cap) snprintf(name, name_cap, "%s", ctx->dev_name.c_str());
    if (n_cus) *n_cus = ctx->n_cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return SYBL_OK;
}

// ------------------------------------------------------------------ queries

int sybl_query_prepare(sybl_table *t, const sybl_query_desc *desc, sybl_query **out) {
    SYBL_API_GUARD(t);
    if (!t || !desc || !out) return fail(SYBL_E_INVAL, "sybl_query_prepare: NULL argument");
    *out = nullptr;
    SYBL_HIP(hipSetDevice(t->ctx->device));
    sybl_query *q = new sybl_query();
    q->t = t;
    q->ctx = t->ctx;
    int rc = plan_query(t, desc, q);
    if (rc) {
        free_query(q);
        return rc;
    }
    q->table_version = t->version;  // after planning: building a group dictionary does not count
    t->queries.push_back(q);
    *out = q;
    return SYBL_OK;
}

void sybl_query_free(sybl_query *q) {
    SYBL_API_GUARD(q);
    if (!q) return;
    hipSetDevice(q->ctx->device);
    hipStreamSynchronize(q->ctx->stream);
    free_query(q);
}

int sybl_query_scan(sybl_query *q) {
    SYBL_API_GUARD(q);
    if (!q) return fail(SYBL_E_INVAL, "query is NULL");
    if (q->table_version != q->t->version)
        return fail(SYBL_E_STATE, "the table changed (blocks appended, bounds or dictionaries set) after this query was prepared");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    return scan(q);
}

int sybl_query_partials(sybl_query *q, void **d_sum, int64_t *n_sum_words, void **d_max, int64_t *n_max_words) {
    SYBL_API_GUARD(q);
    if (!q) return fail(SYBL_E_INVAL, "query is NULL");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    if (q->hash_mode) {
        // the canonical (dense, key-ordered) form of the hash table: sizes are known once the scan has run
        if (!q->scanned) return fail(SYBL_E_STATE, "a hash group-by has no partial-table layout before its scan (sybl_query_scan first)");
        int rc = query_hash_compact(q);
        if (rc) return rc;
        if (d_sum) *d_sum = q->d_dense_sum;
        if (d_max) *d_max = q->d_dense_max;
        if (n_sum_words) *n_sum_words = hash_dense_sum_words(q, q->hash_live);
        if (n_max_words) *n_max_words = hash_dense_max_words(q, q->hash_live);
        return SYBL_OK;
    }
    if (d_sum || d_max) {
        int rc = ensure_partials(q);
        if (rc) return rc;
    }
    if (d_sum) *d_sum = q->d_sum;
    if (d_max) *d_max = q->d_max;
    if (n_sum_words) *n_sum_words = q->n_sum_words;
    if (n_max_words) *n_max_words = q->n_max_words;
    return SYBL_OK;
}

int sybl_query_bind_partials(sybl_query *q, void *d_sum, void *d_max) {
    SYBL_API_GUARD(q);
    if (!q || !d_sum || !d_max) return fail(SYBL_E_INVAL, "sybl_query_bind_partials: NULL argument");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    if (q->hash_mode)
        return fail(SYBL_E_INVAL, "a hash group-by keeps its partials in library-owned buffers whose size follows the keys found "
                                  "(sybl_query_partials after the scan)");
    if (q->n_distinct)
        return fail(SYBL_E_INVAL, "a count-distinct query merges its sketches (register-wise maximum) in sybl_query_allreduce: its partials "
                                  "cannot be bound to caller-owned buffers for an external all-reduce");
    if (q->own_partials) {
        SYBL_HIP(hipStreamSynchronize(q->ctx->stream));
        if (q->d_sum) hipFree(q->d_sum);
        if (q->d_max) hipFree(q->d_max);
        q->own_partials = false;
    }
    q->d_sum = (int64_t *)d_sum;
    q->d_max = (int64_t *)d_max;
    q->plan_dirty = true;
    return SYBL_OK;
}

int sybl_query_stats(sybl_query *q, sybl_run_stats *out) {
    SYBL_API_GUARD(q);
    if (!q || !out) return fail(SYBL_E_INVAL, "NULL argument");
    if (q->scanned) {
        SYBL_HIP(hipSetDevice(q->ctx->device));
        SYBL_HIP(hipEventSynchronize(q->ev[2]));
        float a = 0, b = 0;
        SYBL_HIP(hipEventElapsedTime(&a, q->ev[0], q->ev[1]));
        SYBL_HIP(hipEventElapsedTime(&b, q->ev[1], q->ev[2]));
        q->stats.scan_ms = a;
        q->stats.reduce_ms = b;
    }
    *out = q->stats;
    return SYBL_OK;
}

int sybl_debug_regex_match(const char *pattern, const char *text, int64_t text_len) {
    if (!pattern || (!text && text_len > 0)) return fail(SYBL_E_INVAL, "NULL argument");
    Re2Lite re;
    std::string why;
    if (!re.compile(pattern, &why)) {
        set_error("bad regex '%s': %s", pattern, why.c_str());
        return -1;
    }
    return re.search(text ? text : "", (size_t)text_len) ? 1 : 0;
}

const char *sybl_debug_regex_replace(const char *pattern, const char *text, const char *templ) {
    static thread_local std::string out;
    if (!pattern || !text || !templ) {
        set_error("NULL argument");
        return nullptr;
    }
    Re2Lite re;
    std::string why;
    if (!re.compile(pattern, &why)) {
        set_error("bad regex '%s': %s", pattern, why.c_str());
        return nullptr;
    }
    out = re.replace_all(text, templ);
    return out.c_str();
}

int sybl_debug_query_cells(sybl_query *q, int which, int agg, int64_t *out, int64_t cap, int64_t *n_cells) {
    SYBL_API_GUARD(q);
    if (!q || !out || !n_cells) return fail(SYBL_E_INVAL, "NULL argument");
    if (!q->scanned || !q->h_sum || q->snapshot_pending) return fail(SYBL_E_STATE, "sybl_debug_query_cells: no finalized scan");
    if (which < 0 || which > 3 || (which > 0 && (agg < 0 || agg >= (int)q->aggs.size()))) return fail(SYBL_E_INVAL, "bad field");
    const ScanPlan &P = q->plan;
    const int64_t ncell = q->hash_mode ? q->hash_live : P.n_cells, na = (int64_t)q->aggs.size();
    const int64_t *F = q->h_sum + kHeaderWords;
    *n_cells = ncell;
    const int64_t n = std::min<int64_t>(cap, ncell);
    if (which == 0) {
        for (int64_t c = 0; c < n; c++) out[c] = F[c];
        return SYBL_OK;
    }
    const AggDesc &A = q->aggs[(size_t)agg].d;
    if (which == 1) {
        for (int64_t c = 0; c < n; c++) out[c] = F[(int64_t)A.f_sum * ncell + c];
        return SYBL_OK;
    }
    const int f = which == 2 ? A.f_sb : A.f_sb2;
    if (f >= 0) {
        for (int64_t c = 0; c < n; c++) out[c] = F[(int64_t)f * ncell + c];
    } else if (q->hist_summary && q->h_mom && A.hist_full) {
        for (int64_t c = 0; c < n; c++) out[c] = q->h_mom[(c * na + agg) * 2 + (which - 2)];
    } else {
        return fail(SYBL_E_STATE, "the query keeps no bucket moments");
    }
    return SYBL_OK;
}

int sybl_query_collective_finalize(const sybl_query *q) { SYBL_API_GUARD(q); return q && (q->rs_active || q->top_merge) ? 1 : 0; }

int sybl_query_hash_keys(sybl_query *q, const uint64_t **keys, int64_t *n) {
    SYBL_API_GUARD(q);
    if (!q || !keys || !n) return fail(SYBL_E_INVAL, "NULL argument");
    *keys = nullptr;
    *n = 0;
    if (!q->hash_mode) return SYBL_OK;  // direct-mapped: every rank has the same cells already
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_hash_keys before sybl_query_scan");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    int rc = query_hash_compact(q);
    if (rc) return rc;
    *keys = q->h_dense_keys;
    *n = q->hash_live;
    return SYBL_OK;
}

int sybl_query_hash_install_union(sybl_query *q, const uint64_t *keys, int64_t n) {
    SYBL_API_GUARD(q);
    if (!q) return fail(SYBL_E_INVAL, "NULL argument");
    if (!q->hash_mode) return fail(SYBL_E_STATE, "the query is direct-mapped (sybl_query_hash_keys returned no keys)");
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_hash_install_union before sybl_query_scan");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    int rc = query_hash_compact(q);
    if (rc) return rc;
    return query_hash_install_union(q, keys, n);
}

int sybl_query_snapshot(sybl_query *q) {
    SYBL_API_GUARD(q);
    if (!q) return fail(SYBL_E_INVAL, "NULL argument");
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_snapshot before sybl_query_scan");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    return query_snapshot(q);
}

int sybl_query_finalize(sybl_query *q, sybl_result **out) {
    SYBL_API_GUARD(q);
    if (!q || !out) return fail(SYBL_E_INVAL, "NULL argument");
    if (!q->scanned) return fail(SYBL_E_STATE, "sybl_query_finalize before sybl_query_scan");
    SYBL_HIP(hipSetDevice(q->ctx->device));
    Result *r = nullptr;
    int rc = query_finalize(q, &r);
    if (rc) return rc;
    *out = (sybl_result *)r;
    return SYBL_OK;
}

}  // extern "C"

