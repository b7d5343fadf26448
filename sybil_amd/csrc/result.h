// result.h -- the host-side result model shared by finalize (result.cpp), the renderers (render.cpp) and the
// -encode-results writer (encode.cpp).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"

namespace sybl {

struct AggAcc {
    bool tracked_cnt = false;
    const int64_t *pct_gpu = nullptr;  // GetPercentiles computed by k_hist_summary (100 entries)
    bool moments = false;              // sb / sb2 are valid even though the query keeps bucket arrays
    int64_t cnt = 0, smp = 0, pop = 0;
    uint64_t sum = 0;
    int64_t sb = 0, sb2 = 0;
    int64_t n_out = 0;
    uint64_t sum_out = 0;
    uint64_t sq[4] = {0, 0, 0, 0};
    int64_t vmax = INT64_MIN, nmin = INT64_MIN;
    const int64_t *values = nullptr;  // bucket counts (full-hist mode), n_values long
};

struct CellAcc {
    int64_t count = 0, samples = 0;
    bool has_aggs = false;
    AggAcc aggs[kMaxAggs];
};

// Result.BinaryByKey / Result.GroupByKey of the rows: one entry per GROUP CELL, built once per prepared query (a query's
// key space and dictionaries cannot change after prepare) and shared by every result of it -- a time series has
// n_buckets x groups rows but only `groups` keys, and a bench step used to build 360 000 strings per finalize -- or,
// when the key space is too wide for that (hash group-by), one entry per row, owned by the result.
struct KeyStore {
    static constexpr size_t kKeyBytes = SYBL_MAX_GROUPS * SYBL_GROUP_BY_WIDTH;
    std::vector<std::string> gbk;
    std::vector<uint8_t> keys;  // [entry][kKeyBytes]
    void resize(size_t n) {
        gbk.resize(n);
        keys.resize(n * kKeyBytes);
    }
    uint8_t *key(size_t i) { return keys.data() + i * kKeyBytes; }
};

struct RowStore {
    const uint8_t *key = nullptr;       // into Result::keys
    const std::string *gbkp = nullptr;
    const std::string &gbk() const { return *gbkp; }
    int64_t time_bucket = 0, count = 0, samples = 0;
    int64_t agg_off = 0;  // this row's n_aggs entries in Result::agg_pool / val_pool / pctoff_pool
    int64_t cell = -1;    // group cell (rows of Results / TimeResults)
};

// The big per-result arrays, recycled between the results of one query: a time-series or
// high-cardinality result is tens of MB, and allocating it afresh costs more in page faults (and
// in munmap on free) than building its rows does.
struct ResultStore {
    std::vector<RowStore> rows[3];
    std::vector<uint32_t> order0;    // SortResults: position -> index into rows[0] (empty = as built); the rows stay where they were built
    std::vector<sybl_group_row> view[3];
    std::vector<int64_t> pct_pool, pctoff_pool;
    std::vector<sybl_agg_out> agg_pool;
    std::vector<const int64_t *> val_pool;
    std::vector<int64_t> live, alltime, all_count, all_samples;  // finalize scratch
    std::shared_ptr<KeyStore> own_keys;  // per-row keys (hash group-by / very wide key spaces)
    void swap(ResultStore &o) {
        own_keys.swap(o.own_keys);
        for (int w = 0; w < 3; w++) {
            rows[w].swap(o.rows[w]);
            view[w].swap(o.view[w]);
        }
        order0.swap(o.order0);
        pct_pool.swap(o.pct_pool);
        pctoff_pool.swap(o.pctoff_pool);
        agg_pool.swap(o.agg_pool);
        val_pool.swap(o.val_pool);
        live.swap(o.live);
        alltime.swap(o.alltime);
        all_count.swap(o.all_count);
        all_samples.swap(o.all_samples);
    }
};

struct ResultPool {
    std::mutex m;
    bool full = false;
    ResultStore spare;
};

// What building the rows of a result needs, owned by the result: the rows of a big direct-mapped result (a time series of
// 360 000 buckets x groups, 65 536 histogram groups) are built when a caller first asks for them -- sybl_result_rows, a
// renderer, the encoder -- not by sybl_query_finalize, which only finds the live cells, sorts them and keeps the
// snapshot: the reference's Results ARE its accumulators (aggregate.go:186-203), there is nothing to build before the
// printers walk them.  Everything here is a copy or a reference-counted snapshot, so the query may be scanned again
// (or freed) in between.
struct FinCtx {
    int op = 0;
    bool weighted = false, loghist = false, want_percentiles = false, time_mode = false, hashed = false, summary = false;
    bool out_usable = false, keys_cached = false;
    bool top_only = false;  // a printer's result: percentiles / stddev / buckets for the printed rows and Cumulative only
    // -limit pushed into the scan (pushdown.hip): the cells beyond the limit hold their Count only, so Cumulative's sum(v) and
    // max(v) come from what the scan added up over EVERY row (header words kHdrPdSum / kHdrPdMax), not from the cells
    bool pushdown = false;
    int64_t pd_sum[kMaxAggs] = {0}, pd_max[kMaxAggs] = {0};
    std::vector<AggInfo> aggs;
    ScanPlan P;
    int64_t ncell = 0, gcells = 0;
    size_t n_groups = 0;
    const int64_t *F = nullptr, *hm = nullptr, *H = nullptr;  // cell fields / extrema / bucket arrays of the snapshot
    const int64_t *h_pct = nullptr;                           // GPU-computed percentiles (Result::keep_pct)
    std::vector<int64_t> mom, hm_copy;                        // bucket moments / extrema: copied (the query's buffers are reused)
    const Query *q = nullptr;                                 // rows with keys of their own (hash group-by): built while the query lives
    HostPin keys_buf;                                         // the sorted composite keys of a hash group-by (pinned, shared)
    const uint64_t *dense_keys = nullptr;
};

struct Result : ResultStore {
    const RowStore &sorted0(size_t i) const { return rows[0][order0.empty() ? i : order0[i]]; }
    RowStore &sorted0(size_t i) { return rows[0][order0.empty() ? i : order0[i]]; }
    std::shared_ptr<ResultPool> pool;  // where the arrays go back to when the result is freed
    Query *owner = nullptr;  // the query a lazily finalized result with per-row keys is registered with
    std::shared_ptr<std::recursive_mutex> api_m;  // the ctx's entry-point lock (engine.h: Ctx::api_m), alive as long as this result
    ~Result() {
        if (owner) {
            auto &v = owner->lazy_results;
            for (size_t i = 0; i < v.size(); i++)
                if (v[i] == this) {
                    v.erase(v.begin() + (long)i);
                    break;
                }
        }
        if (!pool) return;
        std::lock_guard<std::mutex> lk(pool->m);
        if (!pool->full) {
            pool->spare.swap(*this);
            pool->full = true;
        }
    }
    int64_t matched = 0;
    FinCtx fin;
    bool rows_pending = false;                    // the rows have not been built yet (result_ensure_rows)
    std::mutex rows_m;
    size_t top_n = 0;                             // rows of the sort order whose bucket arrays sit in top_vals
    std::vector<std::pair<int64_t, int64_t>> out_recs;  // outlier values by (pool slot of the row's aggregation, value), sorted
    int64_t row0_cell(size_t i) const {           // the cell behind row i of the sort order (built or not)
        const size_t ix = order0.empty() ? i : order0[i];
        return fin.time_mode ? -1 : live[ix];
    }
    std::shared_ptr<KeyStore> keys;               // what the rows' key / gbkp point into (the query's cache or own_keys)
    HostPin keep_pct;                             // the snapshot of the GPU-computed percentiles the rows point into
    HostPin keep;                                 // the pinned snapshot of the partial table the bucket
                                                  // arrays of the rows point into
    std::vector<std::vector<int64_t>> total_vals; // Cumulative bucket arrays
    std::vector<int64_t> top_vals;                // bucket arrays of the first `limit` rows (GPU summary path)
    std::vector<int64_t> outlier_vals;            // the outliers' values, grouped by (row, aggregation): sybl_agg_out::outlier_values
    // count distinct: the cells' sketches ([n_cells][kHllRegs]) followed by Cumulative's (the union) and an empty one
    // (rows that own none: the all-time Results of a time series); a row finds its own through RowStore::cell
    bool has_distinct = false;
    std::vector<uint8_t> hll;
    int64_t hll_cells = 0;
    std::vector<int64_t> distinct[3];             // Distinct.Cardinality() per row of rows[w] (as built, not as sorted)
    // Distinct.Cardinality() of a row the renderers hold by reference
    int64_t distinct_of(const RowStore &row) const {
        for (int w = 0; w < 3; w++)
            if (!rows[w].empty() && &row >= rows[w].data() && &row < rows[w].data() + rows[w].size())
                return distinct[w][(size_t)(&row - rows[w].data())];
        return 0;
    }
    const uint8_t *row_registers(int w, const RowStore &row) const {
        const int64_t slot = w == 2 ? hll_cells : ((w == 0 && time_mode) || row.cell < 0 || row.cell >= hll_cells ? hll_cells + 1 : row.cell);
        return hll.data() + (size_t)slot * 16384;
    }
    // (ResultStore) pct_pool: 100 entries per (row, agg) with percentiles; agg_pool / val_pool /
    // pctoff_pool: n_aggs entries per row, all row kinds (pctoff: offset into pct_pool, -1 = none)
    // for rendering
    int op = 0;
    bool weighted = false, time_mode = false, want_percentiles = false, loghist = false;
    std::vector<std::vector<sybl_subhist>> subs;  // -loghist: per aggregation, the layout of its values arrays
    int limit = 0;
    int n_aggs = 0;
    std::vector<int64_t> n_values;
    std::string order_by;
    std::vector<std::string> group_names, agg_names;
    std::string rendered[2];
    bool render_refused = false;  // a printed row has outliers whose values are not available
    // -encode-results
    std::vector<std::pair<int64_t, int64_t>> agg_info;  // Info.Min / Info.Max per aggregation
    int64_t time_bucket = 0;
    bool order_asc = false;
    std::string encoded;
};

void result_ensure_rows(Result *R);  // (result.cpp) builds the rows of a lazily finalized result; cheap when they exist

}  // namespace sybl
