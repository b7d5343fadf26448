// result.cpp -- turns the (all-reduced) partial group table back into sybil's result model.
//
// Reference mapping (src/lib/ of logv/sybil):
//   cell -> Result{BinaryByKey, GroupByKey, Count, Samples, Hists}   query_spec.go:85-93,
//                                                                    aggregate.go:125-143,284-324
//   Cumulative ("TOTAL")                                             aggregate.go:422-438
//   TimeResults / all-time Results in time-series mode               aggregate.go:146-183
//   avg / stddev / percentiles from exact integers                   hist_basic.go:153-219
//   sort                                                             aggregate.go:43-54,497-525
// Rendering lives in render.cpp (printer.go), -encode-results in encode.cpp.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

#include "result.h"
#include "hll.h"

namespace sybl {

// GetPercentiles, hist_basic.go:153-183 (same loop as the reference, including the
// percentiles[p] = k overwrite that later iterations repair)
static void percentiles_from_values(const int64_t *values, int64_t n_values, int64_t bucket_size, int64_t hmin,
                                    int64_t count, int64_t *out100) {
    int64_t pct[101];
    memset(pct, 0, sizeof(pct));
    pct[0] = hmin;
    int64_t c = 0, prev_p = 0;
    for (int64_t k = 0; k < n_values; k++) {
        c += values[k];
        int64_t p = (100 * c) / count;
        p = std::min<int64_t>(std::max<int64_t>(p, 0), 100);
        for (int64_t ip = prev_p; ip <= p; ip++) pct[ip] = k * bucket_size + hmin;
        pct[p] = k;
        prev_p = p;
    }
    memcpy(out100, pct, 100 * sizeof(int64_t));
}

// (Q: what the row builders read of the query -- op, weighted, loghist, want_percentiles, aggs: FinCtx, the result's own copy)
template <class Q>
static void agg_finish(const Q *q, Result *R, const AggInfo &ai, const AggAcc &a, int64_t row_count, sybl_agg_out &o,
                       const int64_t *&values_out, int64_t &pct_off, int64_t pct_slot) {
    memset(&o, 0, sizeof(o));
    values_out = nullptr;
    pct_off = -1;
    const AggDesc &A = ai.d;
    int64_t cnt = a.tracked_cnt ? a.cnt : row_count;
    // the hist exists once the group saw ONE populated value for the column, accepted or not
    // (aggregate.go:246-258 creates it before AddWeightedValue can reject)
    if (a.pop <= 0) return;
    o.present = 1;
    o.count = cnt;
    o.samples = q->weighted ? a.smp : 0;  // BasicHist.Samples only moves with a weight column (hist_basic.go:111-116)
    o.sum = (int64_t)a.sum;
    long double avg_l = cnt != 0 ? (long double)(int64_t)a.sum / (long double)cnt : 0.0L;
    o.avg = (double)avg_l;
    int64_t tmax = a.vmax, tmin = a.nmin == INT64_MIN ? INT64_MAX : -a.nmin;
    if (q->op == SYBL_AGG_HIST) {
        o.min = std::min(A.info_min, tmin);   // SetupBuckets: h.Min = Info.Min
        o.max = std::max(ai.info_max, tmax);  //               h.Max = Info.Max
        o.bucket_size = A.bucket_size;
        o.num_buckets = ai.num_buckets;
        o.n_values = A.n_values;
        o.n_outliers = a.n_out;
    } else if (q->loghist) {
        o.min = std::min(A.info_min, tmin);  // a MultiHist starts at Info.Min / Info.Max in avg mode too (hist_multi.go:31-32)
        o.max = std::max(ai.info_max, tmax);
    } else {
        o.min = std::min<int64_t>(0, tmin);  // avg mode: Go zero values (hist_basic.go:72-85)
        o.max = std::max<int64_t>(0, tmax);
    }
    if (q->op != SYBL_AGG_HIST) return;
    if (q->loghist) {
        // MultiHist: percentiles (GetPercentiles, hist_multi.go:93-128) and stddev (GetStdDev, :140-155) over the union
        // of the sub-histograms' sparse buckets (:190-207) -- non-zero buckets keyed by their lower edge, every
        // outlier once more under its own value, equal keys added up.  Mean = exact sum / count; keys in ascending order.
        o.bucket_size = 0;
        o.num_buckets = 0;
        o.n_values = A.n_values;
        o.n_outliers = 0;
        values_out = a.values;
        if (!a.values) return;
        std::vector<std::pair<int64_t, int64_t>> kc;
        for (const sybl_subhist &S : ai.subs) {
            const int64_t *v = a.values + S.offset, *e = a.values + S.ext_offset;
            for (int64_t k = 0; k < S.n_values; k++)
                if (v[k] > 0) kc.emplace_back(k * S.bucket_size + S.info_min, v[k]);
            for (int64_t k = 0; k < S.n_ext; k++)
                if (e[k] > 0) {
                    kc.emplace_back(S.ext_first + k, e[k]);
                    o.n_outliers += e[k];
                }
        }
        std::sort(kc.begin(), kc.end());
        size_t m = 0;
        for (size_t i = 0; i < kc.size(); i++) {
            if (m > 0 && kc[m - 1].first == kc[i].first) kc[m - 1].second += kc[i].second;
            else kc[m++] = kc[i];
        }
        kc.resize(m);
        int64_t total = 0;
        for (auto &p : kc) total += p.second;
        if (cnt != 0) {
            pct_off = pct_slot;
            int64_t pct[101];
            memset(pct, 0, sizeof(pct));
            int64_t prev_p = 0, c = 0;
            for (auto &p : kc) {
                if (total <= 0) break;
                c += p.second;
                const int64_t pp = (100 * c) / total;
                for (int64_t ip = prev_p; ip <= pp; ip++)
                    if (ip <= 100) pct[ip] = p.first;
                if (pp <= 100 && pp >= 0) pct[pp] = p.first;
                prev_p = pp;
            }
            memcpy(R->pct_pool.data() + pct_off, pct, 100 * sizeof(int64_t));
            o.percentiles = R->pct_pool.data() + pct_off;
        }
        long double var = 0;
        for (auto &p : kc) {
            const long double d = (long double)p.first - avg_l;
            var += d * d * ((long double)p.second / (long double)cnt);
        }
        o.stddev = cnt == 0 ? (kc.empty() ? 0.0 : (double)NAN) : (double)sqrtl(var);
        return;
    }

    // outlier term of GetStdDev: sum (o - avg)^2 / Count, from exact n, sum(o), sum(o^2)
    long double out_term = 0;
    if (a.n_out > 0 && cnt != 0) {
        long double sq = (long double)a.sq[0] + ldexpl((long double)a.sq[1], 32) + ldexpl((long double)a.sq[2], 64) +
                         ldexpl((long double)a.sq[3], 96);
        long double A1 = (long double)o.avg;
        out_term = (sq - 2.0L * A1 * (long double)(int64_t)a.sum_out + (long double)a.n_out * A1 * A1) / (long double)cnt;
    }
    if (q->want_percentiles && (a.values || a.pct_gpu)) {
        values_out = a.values;  // nullptr on the GPU-summary path: attached later for the printed rows
        if (cnt != 0) {
            pct_off = pct_slot;  // pre-sized pool: one 100-entry slot per (row, agg)
            if (a.pct_gpu) {
                o.percentiles = a.pct_gpu;  // straight out of the snapshot (Result::keep_pct): 800 B per row not copied
            } else {
                percentiles_from_values(a.values, A.n_values, A.bucket_size, A.hmin, cnt, R->pct_pool.data() + pct_off);
                o.percentiles = R->pct_pool.data() + pct_off;
            }
        }
    }
    if (q->want_percentiles && a.values && !a.moments) {
        // GetStdDev, hist_basic.go:192-219, with Avg = sum/count
        double sum_variance = 0;
        for (int64_t b = 0; b < A.n_values; b++) {
            int64_t val = b * A.bucket_size + A.hmin;
            double delta = (double)val - o.avg;
            double ratio = (double)a.values[b] / (double)cnt;
            sum_variance += (delta * delta) * ratio;
        }
        o.stddev = sqrt(sum_variance + (double)out_term);
    } else {
        // moments form of the same sum: e_b - avg = BS*b + (hmin - avg)
        long double BS = (long double)A.bucket_size, c = (long double)A.hmin - (long double)o.avg;
        long double var = cnt != 0 ? (BS * BS * (long double)a.sb2 + 2.0L * BS * c * (long double)a.sb + c * c * (long double)cnt) /
                                         (long double)cnt
                                   : 0.0L;
        var += out_term;
        o.stddev = cnt == 0 ? (double)NAN : (var > 0 ? (double)sqrtl(var) : 0.0);  // Go: 0/0 ratios -> NaN
        // (a printer's result: a row beyond the printed ones has neither bucket array nor bucket moments)
        if (q->top_only && A.hist_full && !a.values) o.stddev = (double)NAN;
    }
}

static void build_key(const Query *q, int64_t gcell, uint8_t *key, std::string &gbk) {
    size_t ng = q->groups.size();
    memset(key, 0, SYBL_MAX_GROUPS * SYBL_GROUP_BY_WIDTH);
    gbk.clear();
    if (ng == 0) gbk = "total";  // translate_group_by, aggregate.go:294-296
    int64_t rem = gcell, stride = q->group_cells;
    char num[24];
    for (size_t g = 0; g < ng; g++) {
        const GroupInfo &gi = q->groups[g];
        stride /= gi.gcard;
        int64_t digit = rem / stride;
        rem -= digit * stride;
        uint64_t v;
        bool missing = digit >= gi.value_card;  // the separate MISSING digit
        int64_t sv = gi.gmin + digit;
        if (gi.dict && !missing) sv = q->t->cols[(size_t)gi.col]->gdict[(size_t)digit];
        if (!missing) v = (uint64_t)sv;
        if (missing || (gi.type == SYBL_INT_VAL && v == UINT64_MAX)) {
            // MISSING_VALUE (aggregate.go:31).  The int value -1 has the same 8-byte image, shares the
            // group, and translate_group_by prints nothing for it either (aggregate.go:308-316).
            v = UINT64_MAX;
        } else {
            if (gi.type == SYBL_STR_VAL && gi.replaced) {
                if ((size_t)sv < gi.replaced->strs.size()) gbk += gi.replaced->strs[(size_t)sv];  // -str-replace: the rewritten string
            } else if (gi.type == SYBL_STR_VAL) {
                const Column *c = q->t->cols[(size_t)gi.col].get();
                size_t id = (size_t)sv;
                if (id < c->dict.size()) gbk += c->dict[id];
            } else {
                // strconv.FormatInt(v, 10)
                uint64_t uv = sv < 0 ? (uint64_t)0 - (uint64_t)sv : (uint64_t)sv;
                int pos = (int)sizeof(num);
                do {
                    num[--pos] = (char)('0' + uv % 10);
                    uv /= 10;
                } while (uv);
                if (sv < 0) num[--pos] = '-';
                gbk.append(num + pos, sizeof(num) - (size_t)pos);
            }
        }
        for (int b = 0; b < 8; b++) key[g * 8 + b] = (uint8_t)(v >> (8 * b));
        gbk += "\t";
    }
}

// Writes row `row` whose pool slot (agg_off) was assigned by the caller; thread safe because
// every pool is pre-sized and rows own disjoint slots.
template <class Q>
static void finish_row(const Q *q, Result *R, const CellAcc &acc, RowStore &row, bool outliers_logged) {
    row.count = acc.count;
    row.samples = acc.samples;
    size_t na = q->aggs.size();
    for (size_t a = 0; a < na; a++) {
        size_t k = (size_t)row.agg_off + a;
        if (!acc.has_aggs) {
            memset(&R->agg_pool[k], 0, sizeof(sybl_agg_out));
            R->val_pool[k] = nullptr;
            R->pctoff_pool[k] = -1;
            continue;
        }
        sybl_agg_out &o = R->agg_pool[k];
        agg_finish(q, R, q->aggs[a], acc.aggs[a], acc.count, o, R->val_pool[k], R->pctoff_pool[k], (int64_t)k * 100);
        o.values = R->val_pool[k];
        // the outliers' values are attached later when they were logged; -1: wanted but not available
        // (-loghist keeps its sub-histograms' outliers as exact counters inside `values`: nothing is ever missing)
        o.n_outlier_values = (o.n_outliers > 0 && q->want_percentiles && !outliers_logged && !q->loghist) ? -1 : 0;
    }
}

// Persistent finalize workers: starting 20-odd threads per finalize cost more than the rows they built.
// run(nt, fn) executes fn(0) .. fn(nt - 1), the caller taking its share; one job at a time.  The pool is
// never destroyed (workers block on a condition variable and die with the process).
class WorkerPool {
   public:
    static WorkerPool &get() {
        static WorkerPool *p = new WorkerPool();
        return *p;
    }
    static size_t cap() {
        // (the CPUs the process may actually use: 22 ranges on a 16-CPU quota run as two rounds)
        static const size_t usable = usable_cpus();
        size_t c = std::min<size_t>(usable, 32);
        if (const char *e = env("SYBL_FINALIZE_THREADS")) c = (size_t)std::max(1, atoi(e));
        return c;
    }
    template <typename F>
    void run(size_t nt, F fn) {
        if (nt <= 1) {
            fn((size_t)0);
            return;
        }
        std::lock_guard<std::mutex> job(job_m_);  // one job at a time
        std::function<void(size_t)> f = fn;
        {
            std::unique_lock<std::mutex> lk(m_);
            while (workers_ < nt - 1) {
                std::thread(&WorkerPool::loop, this, workers_).detach();
                workers_++;
            }
            fn_ = &f;
            n_ = nt;
            next_ = 1;  // task 0 is the caller's
            left_ = nt - 1;
            gen_++;
        }
        cv_.notify_all();
        fn((size_t)0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return left_ == 0; });
        fn_ = nullptr;
    }

   private:
    void loop(size_t) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return gen_ != seen && fn_ && next_ < n_; });
            seen = gen_;
            while (fn_ && next_ < n_) {
                const size_t k = next_++;
                const std::function<void(size_t)> *f = fn_;
                lk.unlock();
                (*f)(k);
                lk.lock();
                if (--left_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex m_, job_m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, next_ = 0, left_ = 0, workers_ = 0;
    uint64_t gen_ = 0;
};

// f(i0, i1) over [0, n) on the finalize workers when n is large enough to pay for them
template <typename F>
static void parallel_ranges(size_t n, size_t min_per_thread, F f) {
    size_t nt = std::max<size_t>(1, std::min(WorkerPool::cap(), n / std::max<size_t>(min_per_thread, 1)));
    if (nt == 1) {
        f(0, n);
        return;
    }
    const size_t nr = nt * 4;  // (ranges are claimed dynamically: a late worker costs a quarter range, not a whole one)
    std::atomic<size_t> next{0};
    WorkerPool::get().run(nt, [&](size_t) {
        for (size_t j = next++; j < nr; j = next++) f(n * j / nr, n * (j + 1) / nr);
    });
}

// LogLogBeta.Cardinality (hll.h): the sum in register order, as the reference's loop and the oracle's
uint64_t hll_cardinality(const uint8_t *regs) {
    const double m = (double)kHllRegs;
    const double alpha = 0.7213 / (1.0 + 1.079 / m);
    double sum = 0.0, ez = 0.0;
    for (int i = 0; i < kHllRegs; i++) {
        if (regs[i] == 0) ez += 1.0;
        sum += 1.0 / pow(2.0, (double)regs[i]);
    }
    const double zl = log(ez + 1.0);
    const double beta = -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * pow(zl, 2) + 0.16339839 * pow(zl, 3) +
                        -0.09237745 * pow(zl, 4) + 0.03738027 * pow(zl, 5) + -0.005384159 * pow(zl, 6) + 0.00042419 * pow(zl, 7);
    return (uint64_t)(alpha * m * (m - ez) / (beta + sum));
}

static void make_views(Result *R) {
    // (sybl_agg_out::values / percentiles / outlier fields were set when the rows were built)
    for (int w = 0; w < 3; w++) {
        R->view[w].resize(R->rows[w].size());
        parallel_ranges(R->rows[w].size(), 1 << 15, [&, w](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                RowStore &r = w == 0 ? R->sorted0(i) : R->rows[w][i];
                sybl_group_row &v = R->view[w][i];
                v.binary_key = r.key;
                v.group_by_key = r.gbkp->c_str();
                v.time_bucket = r.time_bucket;
                v.count = r.count;
                v.samples = r.samples;
                v.aggs = R->agg_pool.data() + r.agg_off;
            }
        });
    }
}

// Enqueues the device -> host copy of the (reduced) partial tables behind whatever the stream holds.
// Pinned snapshot: results keep a reference to the snapshot their bucket arrays point into; the
// query reuses the buffer for the next finalize unless a live result still holds it (then a fresh
// one is allocated) -- so a 525 MB histogram table is never copied, page-faulted or unmapped per query.
// Many cells with bucket arrays: percentiles / bucket moments / Cumulative buckets come from the GPU.
// A pinned host buffer of at least `words` int64 that no result holds any more (use count 1: only the query's list),
// or a new one: results keep the snapshot their rows point into, and a pipelined host frees a result only after the
// query's next snapshot was queued -- allocating 52 MB of pinned memory per step instead cost 1.8 ms of every step.
int query_acquire_host_buf(Query *q, int64_t words, std::shared_ptr<HostBuf> &cur) {
    cur.reset();  // (the caller's current buffer, if no result holds it, is a candidate again)
    // best fit among the buffers no result holds: a query keeps buffers of several sizes (the cell-table snapshot, 52 MB of
    // percentiles, a hash group-by's keys) -- first fit let a small request take the big buffer, the big request then
    // found none, dropped every free one as "too small" and allocated 52 MB of pinned memory inside a step (8.5 ms, twice
    // in twenty steps of config 4)
    // free = no result's rows point into it (HostBuf::pins) and it is not one of the query's current snapshot targets
    auto is_free = [&](const std::shared_ptr<HostBuf> &b) {
        return b->pins.load() == 0 && b != q->h_sum_buf && b != q->h_pct_buf && b != q->h_keys_buf && b != q->h_spare_buf;
    };
    int best = -1;
    for (size_t i = 0; i < q->host_bufs.size(); i++) {
        auto &b = q->host_bufs[i];
        if (is_free(b) && b->words >= words && (best < 0 || b->words < q->host_bufs[(size_t)best]->words)) best = (int)i;
    }
    if (best >= 0 && q->host_bufs[(size_t)best]->words <= std::max<int64_t>(4 * words, (int64_t)1 << 17)) {
        cur = q->host_bufs[(size_t)best];
        return SYBL_OK;
    }
    if (q->host_bufs.size() >= 12) {  // (a bound on what a query hoards: the free ones go)
        for (size_t i = 0; i < q->host_bufs.size();) {
            if (is_free(q->host_bufs[i])) q->host_bufs.erase(q->host_bufs.begin() + (long)i);
            else i++;
        }
    }
    auto nb = std::make_shared<HostBuf>();
    SYBL_HIP(hipHostMalloc((void **)&nb->p, (size_t)words * 8, hipHostMallocDefault));
    nb->words = words;
    q->host_bufs.push_back(nb);
    cur = nb;
    return SYBL_OK;
}

// The pinned host copy of a hash group-by's sorted keys: a lazily finalized result shares it, so a buffer some result still
// holds is left alone and another one taken.
int query_host_keys(Query *q, int64_t n) {
    const int64_t words = std::max<int64_t>(n, 1);
    if (!q->h_keys_buf || q->h_keys_buf->pins.load() > 0 || q->h_keys_buf->words < words) {
        int rc = query_acquire_host_buf(q, words, q->h_keys_buf);
        if (rc) return rc;
    }
    q->h_dense_keys = (uint64_t *)q->h_keys_buf->p;
    return SYBL_OK;
}

void query_finish_lazy_results(Query *q) {
    std::vector<Result *> pending;
    pending.swap(q->lazy_results);
    for (Result *R : pending) {
        R->owner = nullptr;  // (no unregistering from inside: the list is gone)
        result_ensure_rows(R);
    }
}

// Cumulative's bucket sums (k_hist_total) and their pinned twin: all a printer's query needs of the summary buffers
int query_total_buffers(Query *q) {
    if (q->d_total) return SYBL_OK;
    const ScanPlan &P = q->plan;
    SYBL_HIP(hipMalloc((void **)&q->d_total, (size_t)P.hist_stride * 8));
    SYBL_HIP(hipHostMalloc((void **)&q->h_total, (size_t)P.hist_stride * 8, hipHostMallocDefault));
    return SYBL_OK;
}

int query_summary_buffers(Query *q) {
    if (q->d_pct) return SYBL_OK;
    const ScanPlan &P = q->plan;
    // (sized for the padded cell count an in-place all-gather of equal slices needs)
    const int64_t pairs = ((int64_t)P.n_cells + kMaxScatterRanks) * (int64_t)q->aggs.size();
    SYBL_HIP(hipMalloc((void **)&q->d_pct, (size_t)pairs * 100 * 8));
    SYBL_HIP(hipMalloc((void **)&q->d_mom, (size_t)pairs * 2 * 8));
    SYBL_HIP(hipHostMalloc((void **)&q->h_mom, (size_t)pairs * 2 * 8, hipHostMallocDefault));
    return query_total_buffers(q);
}

bool query_wants_hist_summary(const Query *q) {
    const ScanPlan &P = q->plan;
    if (env("SYBL_NO_HISTSUMMARY") || q->hash_mode || q->loghist) return false;
    if (q->op != SYBL_AGG_HIST || !q->want_percentiles || q->time_mode || P.hist_stride <= 0 || q->aggs.empty()) return false;
    for (auto &a : q->aggs)
        if (!a.d.hist_full) return false;
    return P.n_cells >= 2048;
}

int query_snapshot(Query *q) {
    PhaseTrace trace("snapshot");
    hipStream_t st = q->ctx->stream;
    const ScanPlan &P = q->plan;
    if (q->hash_mode) {
        int rc = query_hash_compact(q);  // (no-op when the all-reduce already did it)
        if (!rc) rc = query_hash_distinct(q);  // (count distinct: the sketches of the final key set, unless the all-reduce made them)
        if (rc) return rc;
    }
    // a hash group-by snapshots its dense, key-ordered arrays, whose size follows the keys found
    const int64_t sum_words = q->hash_mode ? hash_dense_sum_words(q, q->hash_live) : q->n_sum_words;
    const int64_t max_words = q->hash_mode ? hash_dense_max_words(q, q->hash_live) : q->n_max_words;
    // (what the copy below will bring: the whole table, or -- GPU summary path with a row limit -- everything before the
    // bucket arrays.  Config 4's pinned snapshots were sized for the 525 MB table they never receive: 22 ms of pinned
    // allocation for each of the four a pipelined pair of queries takes, 2 GB of pinned host memory)
    const bool will_summarise = query_wants_hist_summary(q);
    const int64_t snap_words = q->hash_mode ? sum_words : ((!will_summarise || q->limit <= 0) ? q->n_sum_words : (int64_t)P.hist_off);
    if (!q->h_sum_buf || q->h_sum_buf->pins.load() > 0 || q->h_sum_buf->words < snap_words) {
        int rc = query_acquire_host_buf(q, snap_words, q->h_sum_buf);
        if (rc) return rc;
    }
    if (!q->h_max || q->h_max_words < max_words) {
        if (q->h_max) SYBL_HIP(hipHostFree(q->h_max));
        q->h_max = nullptr;
        SYBL_HIP(hipHostMalloc((void **)&q->h_max, (size_t)max_words * 8, hipHostMallocDefault));
        q->h_max_words = max_words;
    }
    q->h_sum = q->h_sum_buf->p;
    trace.mark("buffers");
    q->hist_summary = will_summarise;
    // a printer's query (sybl_query_desc.printed_only): only Cumulative's buckets are summed here; the printed rows'
    // percentiles / stddev come from their bucket arrays, fetched after the sort (query_finalize)
    q->top_only = q->hist_summary && q->printed_only && q->limit > 0;
    if (q->top_merge && !q->top_only) return fail(SYBL_E_STATE, "the all-reduce left the bucket table rank-local but the snapshot is not a printer's");
    // the bucket arrays cross PCIe only when every row's are wanted (no limit); otherwise the printed
    // rows' arrays are gathered after the sort (query_finalize)
    q->snap_has_buckets = !q->hist_summary || q->limit <= 0;
    if (q->top_only) {
        int rc = query_total_buffers(q);
        if (rc) return rc;
        if (!(q->pushdown && q->pushdown_ran)) {  // (a pushed-down scan summed Cumulative's buckets itself, from every row: pushdown.hip)
            SYBL_HIP(hipMemsetAsync(q->d_total, 0, (size_t)P.hist_stride * 8, st));
            hipError_t e = launch_hist_total(q->d_sum + P.hist_off, P.hist_stride, 0, P.n_cells, q->d_total, st);
            if (e != hipSuccess) return hip_fail(e, "k_hist_total");
        }
        if (q->top_merge) {
            int rc2 = comm_allreduce_sum(q->ctx, q->d_total, (size_t)P.hist_stride);
            if (rc2) return rc2;
        }
    } else if (q->hist_summary) {
        // (sized for the padded cell count an in-place all-gather of equal slices needs)
        const int64_t pairs = ((int64_t)P.n_cells + kMaxScatterRanks) * (int64_t)q->aggs.size();
        {
            int rc = query_summary_buffers(q);
            if (rc) return rc;
        }
        // the rows' percentiles point straight into this snapshot: a result that is still alive keeps its own
        if (!q->h_pct_buf) {
            // a pipelined host holds the previous result of this query while the next snapshot is queued: two buffers
            // from the start (allocating the second one when it is first missed costs a step 3.5 ms of pinned allocation)
            std::shared_ptr<HostBuf> &spare = q->h_spare_buf;  // (held while the second one is taken: two distinct buffers)
            int rc = query_acquire_host_buf(q, pairs * 100, spare);
            if (rc) return rc;
            rc = query_acquire_host_buf(q, pairs * 100, q->h_pct_buf);
            if (rc) return rc;
            // first device write into each of them now (a 52 MB copy into a pinned buffer the device has not written
            // before was seen to block its hipMemcpyAsync for 13-18 ms: once, but inside a timed region of ten steps)
            SYBL_HIP(hipMemcpyAsync(spare->p, q->d_pct, (size_t)pairs * 100 * 8, hipMemcpyDeviceToHost, st));
            SYBL_HIP(hipMemcpyAsync(q->h_pct_buf->p, q->d_pct, (size_t)pairs * 100 * 8, hipMemcpyDeviceToHost, st));
            SYBL_HIP(hipStreamSynchronize(st));
            spare.reset();  // (back among the query's free buffers)
        }
        if (q->h_pct_buf->pins.load() > 0 || q->h_pct_buf->words < pairs * 100) {
            int rc = query_acquire_host_buf(q, pairs * 100, q->h_pct_buf);
            if (rc) return rc;
        }
        q->h_pct = q->h_pct_buf->p;
        SYBL_HIP(hipMemsetAsync(q->d_pct, 0, (size_t)pairs * 100 * 8, st));
        SYBL_HIP(hipMemsetAsync(q->d_total, 0, (size_t)P.hist_stride * 8, st));
        HistSummaryPlan S;
        memset(&S, 0, sizeof(S));
        S.H = q->d_sum + P.hist_off;
        S.F = q->d_sum + kHeaderWords;
        S.hist_stride = P.hist_stride;
        S.n_cells = P.n_cells;
        S.n_aggs = (int32_t)q->aggs.size();
        for (size_t a = 0; a < q->aggs.size(); a++) {
            const AggDesc &A = q->aggs[a].d;
            S.agg_off[a] = P.hist_agg_off[a];
            S.n_values[a] = A.n_values;
            S.bucket_size[a] = A.bucket_size;
            S.hmin[a] = A.hmin;
            S.f_cnt[a] = A.f_cnt >= 0 ? A.f_cnt : 0;
        }
        S.pct = q->d_pct;
        S.mom = q->d_mom;
        S.cell0 = q->rs_active ? q->rs_cell0 : 0;
        S.cell1 = q->rs_active ? q->rs_cell1 : P.n_cells;
        hipError_t e = launch_hist_summary(S, q->d_total, st);
        if (e != hipSuccess) return hip_fail(e, "k_hist_summary");
        if (q->rs_active) {
            // every rank summarised its slice of cells: gather the slices, add up the Cumulative buckets
            const size_t na = q->aggs.size(), per = (size_t)q->rs_cells_per;
            int rc = comm_allgather_inplace(q->ctx, q->d_pct, per * na * 100);
            if (!rc) rc = comm_allgather_inplace(q->ctx, q->d_mom, per * na * 2);
            if (!rc) rc = comm_allreduce_sum(q->ctx, q->d_total, (size_t)P.hist_stride);
            if (rc) return rc;
        }
    }
    // Big snapshots (config 4: 52 MB of percentiles, 2 ms of PCIe) leave through a copy stream of their own, ordered behind
    // everything queued so far by an event: the copy engine then works under the next query's scan instead of in front
    // of it (config 4, 10 pipelined steps: 7.3 ms per step with the copy on the main stream, 6.4 ms this way).  Small ones (config 3: 57 KB) stay on the main stream -- an event
    // round trip costs more than they do.
    trace.mark("summary-kernels");
    const int64_t real_pairs = q->hist_summary && !q->top_only ? (int64_t)P.n_cells * (int64_t)q->aggs.size() : 0;
    const int64_t main_words = q->hash_mode ? sum_words : (q->snap_has_buckets ? q->n_sum_words : P.hist_off);
    hipStream_t cs = st;
    q->snap_on_aux = false;
    if ((real_pairs * 102 + main_words) * 8 >= ((int64_t)4 << 20) && !env("SYBL_NO_COPY_STREAM")) {
        Ctx *ctx = q->ctx;
        if (!ctx->copy_stream) SYBL_HIP(create_side_stream(&ctx->copy_stream, +1));
        if (!q->ev_ready) SYBL_HIP(hipEventCreateWithFlags(&q->ev_ready, hipEventDisableTiming));
        SYBL_HIP(hipEventRecord(q->ev_ready, st));
        SYBL_HIP(hipStreamWaitEvent(ctx->copy_stream, q->ev_ready, 0));
        cs = ctx->copy_stream;
        q->snap_on_aux = true;
    }
    trace.mark("copy-stream");
    if (q->top_only) {
        SYBL_HIP(hipMemcpyAsync(q->h_total, q->d_total, (size_t)P.hist_stride * 8, hipMemcpyDeviceToHost, cs));
    } else if (q->hist_summary) {
        SYBL_HIP(hipMemcpyAsync(q->h_pct, q->d_pct, (size_t)real_pairs * 100 * 8, hipMemcpyDeviceToHost, cs));
        trace.mark("pct");
        SYBL_HIP(hipMemcpyAsync(q->h_mom, q->d_mom, (size_t)real_pairs * 2 * 8, hipMemcpyDeviceToHost, cs));
        SYBL_HIP(hipMemcpyAsync(q->h_total, q->d_total, (size_t)P.hist_stride * 8, hipMemcpyDeviceToHost, cs));
        trace.mark("mom+total");
    }
    if (q->hash_mode) {
        SYBL_HIP(hipMemcpyAsync(q->h_sum, q->d_dense_sum, (size_t)sum_words * 8, hipMemcpyDeviceToHost, cs));
        if (P.n_max_fields > 0) SYBL_HIP(hipMemcpyAsync(q->h_max, q->d_dense_max, (size_t)max_words * 8, hipMemcpyDeviceToHost, cs));
    } else {
        SYBL_HIP(hipMemcpyAsync(q->h_sum, q->d_sum, (size_t)main_words * 8, hipMemcpyDeviceToHost, cs));
        if (P.n_max_fields > 0)
            SYBL_HIP(hipMemcpyAsync(q->h_max, q->d_max, (size_t)q->n_max_words * 8, hipMemcpyDeviceToHost, cs));
    }
    if (!q->ev_snap) SYBL_HIP(hipEventCreateWithFlags(&q->ev_snap, hipEventDisableTiming));
    SYBL_HIP(hipEventRecord(q->ev_snap, cs));
    q->snapshot_pending = true;
    trace.mark("copies");
    return SYBL_OK;
}

// The bucket arrays of the first `top` rows of Results (the rows a printer shows) -- gathered on
// the GPU into one buffer and copied in one piece.  Runs on the context's auxiliary stream: the
// main stream may already be busy with the scan of another query.
static const size_t kTopDmaRows = 512;

static int fetch_top_values(Query *q, Result *R, size_t top) {
    const ScanPlan &P = q->plan;
    Ctx *ctx = q->ctx;
    if (top == 0) return SYBL_OK;
    if (!ctx->aux_stream) SYBL_HIP(create_side_stream(&ctx->aux_stream, -1));
    if ((int64_t)top > q->top_cap) {
        if (q->d_top) SYBL_HIP(hipFree(q->d_top));
        if (q->d_top_cells) SYBL_HIP(hipFree(q->d_top_cells));
        SYBL_HIP(hipMalloc((void **)&q->d_top, top * (size_t)P.hist_stride * 8));
        SYBL_HIP(hipMalloc((void **)&q->d_top_cells, top * 8));
        q->top_cap = (int64_t)top;
    }
    std::vector<int64_t> cells(top);
    for (size_t i = 0; i < top; i++) cells[i] = R->row0_cell(i);
    R->top_vals.resize(top * (size_t)P.hist_stride);
    if (q->rs_active || q->top_merge) {
        // the printed rows' bucket arrays live on the ranks that own their cells (reduce-scatter), or every rank holds its
        // own part of each (a printer's merge: the table stayed rank-local): every rank gathers what it has (zeros for
        // cells it does not own) and the buffers are summed -- on the main stream, in step with the other collectives
        hipStream_t st = ctx->stream;
        const int64_t own0 = q->rs_active ? q->rs_cell0 : 0, own1 = q->rs_active ? q->rs_cell1 : (int64_t)P.n_cells;
        SYBL_HIP(hipMemcpyAsync(q->d_top_cells, cells.data(), top * 8, hipMemcpyHostToDevice, st));
        hipError_t e = launch_hist_gather(q->d_sum + P.hist_off, P.hist_stride, q->d_top_cells, (int64_t)top, own0, own1, q->d_top, st);
        if (e != hipSuccess) return hip_fail(e, "k_hist_gather");
        int rc = comm_allreduce_sum(ctx, q->d_top, top * (size_t)P.hist_stride);
        if (rc) return rc;
        SYBL_HIP(hipMemcpyAsync(R->top_vals.data(), q->d_top, R->top_vals.size() * 8, hipMemcpyDeviceToHost, st));
        SYBL_HIP(hipStreamSynchronize(st));
    } else if (top <= kTopDmaRows && !env("SYBL_TOP_GATHER_KERNEL")) {
        // A printer's worth of rows: one asynchronous copy per row, straight out of the table into pinned memory.
        // Measured on config 4 (two queries pipelined): the gather kernel below, queued on the auxiliary stream with
        // pageable source / destination buffers, completed only when the other query's k_emit had ended (4.7 ms per
        // finalize, whatever the stream's priority); these copies (the runtime turns each into a one-workgroup
        // __amd_rocclr_copyBuffer blit, profiles/r02c_cfg4_kernel_trace.txt) take 0.3-0.7 ms under the same scan.
        const size_t words = top * (size_t)P.hist_stride;
        if ((int64_t)words > q->h_top_words) {
            if (q->h_top) SYBL_HIP(hipHostFree(q->h_top));
            q->h_top = nullptr;
            SYBL_HIP(hipHostMalloc((void **)&q->h_top, words * 8, hipHostMallocDefault));
            q->h_top_words = (int64_t)words;
        }
        for (size_t i = 0; i < top; i++)
            SYBL_HIP(hipMemcpyAsync(q->h_top + i * (size_t)P.hist_stride, q->d_sum + P.hist_off + cells[i] * P.hist_stride,
                                    (size_t)P.hist_stride * 8, hipMemcpyDeviceToHost, ctx->aux_stream));
        SYBL_HIP(hipStreamSynchronize(ctx->aux_stream));
        memcpy(R->top_vals.data(), q->h_top, words * 8);
    } else {
        SYBL_HIP(hipMemcpyAsync(q->d_top_cells, cells.data(), top * 8, hipMemcpyHostToDevice, ctx->aux_stream));
        hipError_t e = launch_hist_gather(q->d_sum + P.hist_off, P.hist_stride, q->d_top_cells, (int64_t)top, 0, P.n_cells, q->d_top, ctx->aux_stream);
        if (e != hipSuccess) return hip_fail(e, "k_hist_gather");
        SYBL_HIP(hipMemcpyAsync(R->top_vals.data(), q->d_top, R->top_vals.size() * 8, hipMemcpyDeviceToHost, ctx->aux_stream));
        SYBL_HIP(hipStreamSynchronize(ctx->aux_stream));
    }
    R->top_n = top;  // (the rows get their pointers when they are built: result_ensure_rows)
    return SYBL_OK;
}

int query_finalize(Query *q, Result **out) {
    const ScanPlan &P = q->plan;
    PhaseTrace trace;
    if (!q->snapshot_pending) {
        int rc = query_snapshot(q);
        if (rc) return rc;
    }
    // only the snapshot is waited for: work enqueued behind it (the scan of another query) keeps running
    SYBL_HIP(hipEventSynchronize(q->ev_snap));
    q->snapshot_pending = false;
    trace.mark("copy+sync");
    const int64_t *hs = q->h_sum, *hm = q->h_max;
    if (hs[kHdrEmitStall] != 0)
        return fail(SYBL_E_STATE, "k_emit: %lld records were dropped by a stalled staging bin (engine bug)", (long long)hs[kHdrEmitStall]);
    if (hs[kHdrPartOverflow] != 0 && q->part_hist) {
        int rc = query_rescan_without_part_hist(q);
        if (rc) return rc;
        return query_finalize(q, out);
    }
    if (hs[kHdrHashFull] != 0)
        return fail(SYBL_E_NOMEM, "hash group-by: the query has more distinct group keys than the %d-slot table holds (%lld rows lost)",
                    P.n_cells, (long long)hs[kHdrHashFull]);
    if (hs[kHdrOverflow] != 0)
        return fail(SYBL_E_STATE,
                    "%lld rows fell outside the declared column bounds (sybl_table_set_bounds) -- results would be incomplete",
                    (long long)hs[kHdrOverflow]);

    Result *R = new Result();
    std::unique_ptr<Result> r_guard(R);  // (every early return below gives the result back)
    R->api_m = q->ctx->api_m;
    if (!q->rpool) q->rpool = std::make_shared<ResultPool>();
    R->pool = q->rpool;
    {
        std::lock_guard<std::mutex> lk(q->rpool->m);
        if (q->rpool->full) {
            R->swap(q->rpool->spare);
            q->rpool->full = false;
        }
    }
    R->matched = hs[kHdrMatched];
    R->op = q->op;
    R->weighted = q->weighted;
    R->time_mode = q->time_mode;
    R->want_percentiles = q->want_percentiles;
    R->limit = q->limit;
    R->order_by = q->order_by;
    R->order_asc = q->order_asc;
    R->time_bucket = q->time_bucket;
    for (auto &a : q->aggs) R->agg_info.emplace_back(a.d.info_min, a.info_max);
    R->loghist = q->loghist;
    R->subs.clear();
    for (auto &a : q->aggs) R->subs.push_back(a.subs);
    R->n_aggs = (int)q->aggs.size();
    for (auto &g : q->groups) R->group_names.push_back(q->t->cols[(size_t)g.col]->name);
    for (auto &a : q->aggs) {
        R->agg_names.push_back(a.name);
        R->n_values.push_back(a.d.n_values);
    }

    // (hash group-by: "cell" i is the i-th key of the dense, key-ordered arrays)
    const bool hashed = q->hash_mode;
    const int64_t ncell = hashed ? q->hash_live : P.n_cells, gcells = q->group_cells;
    const int64_t *F = hs + kHeaderWords;
    const int64_t *H = nullptr;
    if (P.hist_stride > 0 && q->snap_has_buckets) H = hashed ? F + (int64_t)P.n_sum_fields * ncell : hs + P.hist_off;
    R->keep = q->h_sum_buf;  // the rows are built from (and their bucket arrays live in) the snapshot
    const bool summary = q->hist_summary;
    if (summary && !q->top_only) R->keep_pct = q->h_pct_buf;  // the rows' percentiles live in the snapshot
    const size_t na = q->aggs.size();
    // outlier values (plan.h: outlier log): usable when every one of them was logged
    const bool out_logged = q->d_out_log != nullptr;
    const int64_t n_out_log = out_logged ? hs[kHdrOutLog] : 0;
    const bool out_usable = out_logged && !q->out_log_partial && n_out_log <= q->out_cap;

    FinCtx &C = R->fin;
    C.op = q->op;
    C.weighted = q->weighted;
    C.loghist = q->loghist;
    C.want_percentiles = q->want_percentiles;
    C.time_mode = q->time_mode;
    C.hashed = hashed;
    C.summary = summary;
    C.out_usable = out_usable;
    C.aggs = q->aggs;
    C.P = P;
    C.ncell = ncell;
    C.gcells = gcells;
    C.n_groups = q->groups.size();
    C.F = F;
    C.H = H;
    C.hm = hm;
    C.top_only = q->top_only;
    C.pushdown = q->pushdown && q->pushdown_ran && q->top_only;
    if (C.pushdown)
        for (int a = 0; a < kMaxAggs; a++) {
            C.pd_sum[a] = hs[kHdrPdSum + a];
            C.pd_max[a] = hs[kHdrPdMax + a];
        }
    C.h_pct = summary && !q->top_only ? q->h_pct : nullptr;
    C.q = q;
    C.keys_buf = hashed ? q->h_keys_buf : nullptr;
    C.dense_keys = hashed ? q->h_dense_keys : nullptr;

    R->total_vals.resize(na);
    for (size_t a = 0; a < na; a++) {
        if (!q->time_mode && q->aggs[a].d.hist_full) R->total_vals[a].assign((size_t)q->aggs[a].d.n_values, 0);
        else R->total_vals[a].clear();
    }
    std::vector<int64_t> &all_count = R->all_count, &all_samples = R->all_samples;
    if (q->time_mode && !hashed) {
        all_count.assign((size_t)gcells, 0);
        all_samples.assign((size_t)gcells, 0);
    }
    // pass 1: the live cells (ascending), and in time-series mode the all-time Count/Samples.  Ranges of cells are
    // compacted by worker threads into their own slice of `live` and the slices closed up afterwards (721 x 500 cells
    // took 0.65 ms serially: a quarter of what the scan kernel of config 5 takes).
    std::vector<int64_t> &live = R->live;
    {
        const int64_t *E = P.f_samples >= 0 ? F + (int64_t)P.f_samples * ncell : F;
        live.resize((size_t)ncell);
        int64_t *lv = live.data();
        const size_t grain = (size_t)1 << 16, n_chunks = ((size_t)ncell + grain - 1) / grain;
        std::vector<size_t> found(n_chunks, 0);
        parallel_ranges(n_chunks, 1, [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; c++) {
                const int64_t b = (int64_t)(c * grain), e = std::min<int64_t>(ncell, b + (int64_t)grain);
                size_t n = 0;
                int64_t *dst = lv + b;
                for (int64_t cell = b; cell < e; cell++) {
                    dst[n] = cell;
                    n += E[cell] != 0;  // branch-free compaction
                }
                found[c] = n;
            }
        });
        size_t n_live = 0;
        for (size_t c = 0; c < n_chunks; c++) {
            if (n_live != c * grain) memmove(lv + n_live, lv + c * grain, found[c] * sizeof(int64_t));
            n_live += found[c];
        }
        live.resize(n_live);
    }
    std::vector<int64_t> &alltime = R->alltime;  // group cells with any row (time-series mode)
    alltime.clear();
    if (q->time_mode && hashed) {
        // hash group-by: the all-time Results are the distinct group keys of the [time bucket || key] rows; all_count /
        // all_samples are indexed like `alltime` here, not by group cell (the key space is up to 2^62 wide)
        const int64_t *S = P.f_samples >= 0 ? F + (int64_t)P.f_samples * ncell : F;
        std::vector<std::pair<int64_t, int64_t>> byg(live.size());  // (group key, dense row)
        for (size_t i = 0; i < live.size(); i++) byg[i] = {(int64_t)(q->h_dense_keys[(size_t)live[i]] % (uint64_t)gcells), live[i]};
        std::sort(byg.begin(), byg.end());
        all_count.clear();
        all_samples.clear();
        for (size_t i = 0; i < byg.size();) {
            int64_t c = 0, sm = 0;
            size_t j = i;
            for (; j < byg.size() && byg[j].first == byg[i].first; j++) {
                c += F[byg[j].second];
                sm += S[byg[j].second];
            }
            if (q->weighted ? sm != 0 : c != 0) {
                alltime.push_back(byg[i].first);
                all_count.push_back(c);
                all_samples.push_back(sm);
            }
            i = j;
        }
    } else if (q->time_mode) {
        // (time bucket major, group minor: walk a bucket's cells as one contiguous run per group range)
        const int64_t *S = P.f_samples >= 0 ? F + (int64_t)P.f_samples * ncell : F;
        parallel_ranges((size_t)gcells, 1 << 12, [&](size_t g0, size_t g1) {
            for (int64_t tb = 0; tb * gcells < ncell; tb++) {
                const int64_t *fc = F + tb * gcells, *fs = S + tb * gcells;
                for (size_t g = g0; g < g1; g++) {
                    all_count[g] += fc[g];
                    all_samples[g] += fs[g];
                }
            }
        });
        for (int64_t g = 0; g < gcells; g++)
            if (q->weighted ? all_samples[(size_t)g] != 0 : all_count[(size_t)g] != 0) alltime.push_back(g);
    }
    trace.mark("live");

    // BinaryByKey / GroupByKey per group cell: the query's cache (built by its first finalize), or per row
    const bool keys_cached = !hashed && gcells <= ((int64_t)1 << 18);
    C.keys_cached = keys_cached;
    std::shared_ptr<KeyStore> ks;
    if (keys_cached) {
        if (!q->key_cache) {
            auto kc = std::make_shared<KeyStore>();
            kc->resize((size_t)gcells + 1);
            parallel_ranges((size_t)gcells, 1 << 10, [&](size_t g0, size_t g1) {
                for (size_t g = g0; g < g1; g++) build_key(q, (int64_t)g, kc->key(g), kc->gbk[g]);
            });
            memset(kc->key((size_t)gcells), 0, KeyStore::kKeyBytes);
            kc->gbk[(size_t)gcells] = "TOTAL";
            for (size_t g = 1; g < q->groups.size(); g++) kc->gbk[(size_t)gcells] += "\t";
            q->key_cache = kc;
        }
        ks = q->key_cache;
    } else {
        if (!R->own_keys) R->own_keys = std::make_shared<KeyStore>();
        ks = R->own_keys;  // (sized when the rows are built)
    }
    R->keys = ks;
    trace.mark("alloc");

    // Lazy rows: a direct-mapped result whose keys are cached needs nothing of the query to build its rows later.  What
    // the query's next snapshot would overwrite is copied (bucket moments, extrema); the snapshots themselves are
    // reference counted.  Small results are built right away (the threshold only keeps trivial results simple to debug),
    // count-distinct results as well (their sketches are fetched here).
    // (SYBL_LAZY_ROWS=1: whatever the size -- the test suite runs once that way)
    // Rows with keys of their own (hash group-by, very wide key spaces) are built from the query's group columns and the
    // table's dictionaries: such a result registers with its query, which builds the rows before it goes away.
    const bool lazy = !q->n_distinct && (live.size() >= 2048 || env("SYBL_LAZY_ROWS")) && !env("SYBL_EAGER_ROWS");
    if (summary) {
        if (!q->top_only) C.mom.assign(q->h_mom, q->h_mom + (size_t)P.n_cells * na * 2);
        else C.mom.clear();
        for (size_t a = 0; a < na; a++)
            if (!R->total_vals[a].empty())
                memcpy(R->total_vals[a].data(), q->h_total + P.hist_agg_off[a], R->total_vals[a].size() * sizeof(int64_t));
    }
    if (lazy && P.n_max_fields > 0) {
        // (worker threads: 9.5e7 hash groups are 0.76 GB of extrema, 150 ms for one thread)
        const size_t words = (size_t)(hashed ? hash_dense_max_words(q, ncell) : q->n_max_words);
        C.hm_copy.resize(words);
        int64_t *dst = C.hm_copy.data();
        parallel_ranges(words, (size_t)1 << 20, [&](size_t i0, size_t i1) { memcpy(dst + i0, hm + i0, (i1 - i0) * sizeof(int64_t)); });
        C.hm = C.hm_copy.data();
    }
    R->rows_pending = true;
    R->out_recs.clear();
    R->top_n = 0;

    // ---- outlier values (plan.h: outlier log) -> (pool slot of the row's aggregation, value), sorted: slots ascending,
    // values ascending inside a slot (cell rows own the first live.size() * na pool slots); attached when the rows exist
    R->outlier_vals.clear();
    if (out_usable && n_out_log > 0) {
        const int64_t n_log = n_out_log;
        std::vector<int64_t> log((size_t)n_log * kOutLogWords);
        SYBL_HIP(hipMemcpy(log.data(), q->d_out_log, log.size() * 8, hipMemcpyDeviceToHost));
        std::vector<std::pair<int64_t, int64_t>> &recs = R->out_recs;
        recs.reserve((size_t)n_log);
        for (int64_t i = 0; i < n_log; i++) {
            int64_t cell = log[(size_t)i * kOutLogWords];
            const int64_t a = log[(size_t)i * kOutLogWords + 1];
            if (hashed) {  // the log names the group by its composite key
                auto it = std::lower_bound(q->h_dense_keys, q->h_dense_keys + ncell, (uint64_t)cell);
                if (it == q->h_dense_keys + ncell || *it != (uint64_t)cell) continue;
                cell = (int64_t)(it - q->h_dense_keys);
            }
            auto lv = std::lower_bound(live.begin(), live.end(), cell);
            if (lv == live.end() || *lv != cell || a < 0 || a >= (int64_t)na) continue;
            recs.emplace_back((int64_t)((size_t)(lv - live.begin()) * na + (size_t)a), log[(size_t)i * kOutLogWords + 2]);
        }
        std::sort(recs.begin(), recs.end());
    }

    // SortResults, aggregate.go:497-525 (stable over the canonical key order), straight from the cell fields: row i of
    // Results is live cell i (a time series: all-time group i, Count only -- no aggregation is present there)
    if (!q->order_by.empty()) {
        int by = -1;
        if (q->order_by != "$COUNT") {
            for (size_t a = 0; a < na; a++)
                if (q->aggs[a].name == q->order_by) by = (int)a;
            if (by < 0) return fail(SYBL_E_INVAL, "order_by '%s' is neither $COUNT nor an aggregated column", q->order_by.c_str());
        }
        const size_t n = q->time_mode ? alltime.size() : live.size();
        auto row_count = [&](size_t i) -> int64_t { return q->time_mode ? all_count[hashed ? i : (size_t)alltime[i]] : F[live[i]]; };
        auto row_mean = [&](size_t i) -> double {  // agg_finish's avg of aggregation `by`; -inf when the row has no such hist
            if (q->time_mode) return -INFINITY;
            const AggDesc &A = q->aggs[(size_t)by].d;
            const int64_t cell = live[i], count = F[cell];
            const int64_t samples = P.f_samples >= 0 ? F[(int64_t)P.f_samples * ncell + cell] : count;
            const int64_t pop = A.f_pop >= 0 ? F[(int64_t)A.f_pop * ncell + cell] : (P.f_samples >= 0 ? samples : count);
            if (pop <= 0) return -INFINITY;
            const int64_t cnt = A.f_cnt >= 0 ? F[(int64_t)A.f_cnt * ncell + cell] : count;
            const long double avg_l = cnt != 0 ? (long double)F[(int64_t)A.f_sum * ncell + cell] / (long double)cnt : 0.0L;
            return (double)avg_l;
        };
        std::vector<uint32_t> &order = R->order0;
        order.resize(n);
        if (n < 8192) {
            std::vector<int64_t> kc(by < 0 ? n : 0);
            std::vector<double> km(by < 0 ? 0 : n);
            for (size_t i = 0; i < n; i++) {
                order[i] = (uint32_t)i;
                if (by < 0) kc[i] = row_count(i);
                else km[i] = row_mean(i);
            }
            auto less = [&](uint32_t ix, uint32_t iy) { return by < 0 ? kc[ix] > kc[iy] : km[ix] > km[iy]; };
            std::stable_sort(order.begin(), order.end(), less);
        } else {
            // many groups: stable LSD radix sort (16-bit digits) of (key, index) pairs, the key's ascending unsigned
            // order being the descending order of Count / of the mean.  Digits every key agrees on are skipped
            // (counts of a uniform 65536-group table differ in their low 16 bits only: one pass).
            struct KeyIx {
                uint64_t key;
                uint64_t ix;
            };
            std::vector<KeyIx> cur(n), nxt(n);
            uint64_t differ = 0;
            for (size_t i = 0; i < n; i++) {
                uint64_t u;
                if (by < 0) {
                    u = (uint64_t)row_count(i) ^ 0x8000000000000000ull;
                } else {
                    double m = row_mean(i);
                    if (m == 0.0) m = 0.0;  // -0.0 and +0.0 compare equal
                    uint64_t b;
                    memcpy(&b, &m, 8);
                    u = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
                }
                cur[i].key = ~u;
                cur[i].ix = i;
                differ |= cur[i].key ^ cur[0].key;
            }
            std::vector<uint32_t> cnt(65536);
            for (int shift = 0; shift < 64; shift += 16) {
                if (((differ >> shift) & 0xFFFFu) == 0) continue;
                std::fill(cnt.begin(), cnt.end(), 0u);
                for (size_t i = 0; i < n; i++) cnt[(cur[i].key >> shift) & 0xFFFFu]++;
                uint32_t run = 0;
                for (size_t d = 0; d < 65536; d++) {
                    uint32_t c2 = cnt[d];
                    cnt[d] = run;
                    run += c2;
                }
                for (size_t i = 0; i < n; i++) nxt[cnt[(cur[i].key >> shift) & 0xFFFFu]++] = cur[i];
                cur.swap(nxt);
            }
            for (size_t i = 0; i < n; i++) order[i] = (uint32_t)cur[i].ix;
        }
        if (q->order_asc) std::reverse(order.begin(), order.end());
    } else {
        R->order0.clear();
    }
    trace.mark("sort");
    // the bucket arrays of the rows a printer shows (GPU summary path): fetched now, while the table is the scan's
    R->top_vals.clear();
    if (summary && !q->snap_has_buckets) {
        int rc = fetch_top_values(q, R, std::min<size_t>((size_t)q->limit, q->time_mode ? alltime.size() : live.size()));
        if (rc) return rc;
        trace.mark("top-values");
    }
    if (!lazy) {
        result_ensure_rows(R);
        trace.mark("rows");
        // ---- count distinct (query_spec.go:87,180-188): every row's sketch and its Cardinality()
        R->has_distinct = q->n_distinct > 0;
        if (R->has_distinct) {
            // (a hashed group-by: a sketch per dense key -- a row's cell is its key's place in the sorted list)
            const int64_t hll_cells = q->hash_mode ? q->hash_live : (int64_t)P.n_cells;
            R->hll_cells = hll_cells;
            R->hll.resize((size_t)(hll_cells + 2) * kHllRegs);
            // the pass that fills the sketches (and their all-reduce) was queued on the context's stream: copy behind it
            SYBL_HIP(hipMemcpyAsync(R->hll.data(), q->d_hll, (size_t)q->hll_bytes, hipMemcpyDeviceToHost, q->ctx->stream));
            SYBL_HIP(hipStreamSynchronize(q->ctx->stream));
            uint8_t *total_regs = R->hll.data() + (size_t)hll_cells * kHllRegs;
            memset(total_regs, 0, 2 * (size_t)kHllRegs);
            if (!q->time_mode) {
                // Cumulative combines every Result (aggregate.go:431-434): the register-wise maximum.  (In a time series it
                // combines the all-time Results, whose sketches stay empty: aggregate.go:156-169 only counts there.)
                for (const RowStore &row : R->rows[0]) {
                    const uint8_t *regs = R->row_registers(0, row);
                    for (int k = 0; k < kHllRegs; k++) total_regs[k] = std::max(total_regs[k], regs[k]);
                }
            }
            for (int w = 0; w < 3; w++) {
                R->distinct[w].resize(R->rows[w].size());
                parallel_ranges(R->rows[w].size(), 256, [&, w](size_t i0, size_t i1) {
                    for (size_t i = i0; i < i1; i++) R->distinct[w][i] = (int64_t)hll_cardinality(R->row_registers(w, R->rows[w][i]));
                });
            }
        }
    } else {
        R->has_distinct = false;
    }
    if (lazy && !keys_cached) {
        R->owner = q;
        q->lazy_results.push_back(R);
    } else {
        C.q = nullptr;  // (nothing built later may look at the query)
    }
    *out = r_guard.release();
    return SYBL_OK;
}

// The rows of a result, from its own context (FinCtx) and snapshots: one row per live cell (worker threads when there are
// enough of them), the all-time Results of a time series, Cumulative; then the outliers' values, the printed rows' bucket
// arrays and the sybl_group_row views.
void result_ensure_rows(Result *R) {
    std::lock_guard<std::mutex> lk(R->rows_m);
    if (!R->rows_pending) return;
    R->rows_pending = false;
    const FinCtx &C = R->fin;
    const FinCtx *q = &C;
    const ScanPlan &P = C.P;
    const bool hashed = C.hashed, summary = C.summary, keys_cached = C.keys_cached, out_usable = C.out_usable;
    const int64_t ncell = C.ncell, gcells = C.gcells;
    const int64_t *F = C.F, *H = C.H, *hm = C.hm;
    const size_t na = C.aggs.size();
    std::vector<int64_t> &live = R->live, &alltime = R->alltime, &all_count = R->all_count, &all_samples = R->all_samples;
    KeyStore *ks = R->keys.get();
    if (!keys_cached) ks->resize(live.size() + alltime.size() + 1);
    auto load_cell = [&](int64_t cell, CellAcc &acc) -> bool {
        acc.count = F[cell];
        acc.samples = P.f_samples >= 0 ? F[(int64_t)P.f_samples * ncell + cell] : acc.count;
        bool exists = C.weighted ? acc.samples != 0 : acc.count != 0;
        if (!exists) return false;
        acc.has_aggs = true;
        for (size_t a = 0; a < na; a++) {
            const AggDesc &A = C.aggs[a].d;
            AggAcc &x = acc.aggs[a];
            x = AggAcc();
            x.sum = (uint64_t)F[(int64_t)A.f_sum * ncell + cell];
            // when the count is not tracked per aggregation it equals the row count of the cell
            x.tracked_cnt = true;
            x.cnt = A.f_cnt >= 0 ? F[(int64_t)A.f_cnt * ncell + cell] : acc.count;
            if (A.f_smp >= 0) x.smp = F[(int64_t)A.f_smp * ncell + cell];
            x.pop = A.f_pop >= 0 ? F[(int64_t)A.f_pop * ncell + cell] : (P.f_samples >= 0 ? acc.samples : acc.count);
            if (A.f_sb >= 0) x.sb = F[(int64_t)A.f_sb * ncell + cell];
            if (A.f_sb2 >= 0) x.sb2 = F[(int64_t)A.f_sb2 * ncell + cell];
            if (A.f_out >= 0) {
                x.n_out = F[(int64_t)A.f_out * ncell + cell];
                x.sum_out = (uint64_t)F[(int64_t)(A.f_out + 1) * ncell + cell];
                for (int k = 0; k < 4; k++) x.sq[k] = (uint64_t)F[(int64_t)(A.f_out + 2 + k) * ncell + cell];
            }
            if (A.m_max >= 0) x.vmax = hm[(int64_t)A.m_max * ncell + cell];
            if (A.m_nmin >= 0) x.nmin = hm[(int64_t)A.m_nmin * ncell + cell];
            if (A.hist_full && H) x.values = H + cell * P.hist_stride + P.hist_agg_off[a];
            if (A.hist_full && summary && !C.top_only) {
                const int64_t pair = cell * (int64_t)na + (int64_t)a;
                x.pct_gpu = C.h_pct + pair * 100;
                x.sb = C.mom[(size_t)pair * 2];
                x.sb2 = C.mom[(size_t)pair * 2 + 1];
                x.moments = true;
            }
        }
        return true;
    };

    CellAcc total;
    total.has_aggs = !C.time_mode;
    std::vector<std::vector<int64_t>> summary_totals;
    if (summary) summary_totals = R->total_vals;  // (k_hist_total's Cumulative buckets: put back below)
    for (size_t a = 0; a < na; a++) {
        total.aggs[a].tracked_cnt = true;
        if (!R->total_vals[a].empty()) {
            std::fill(R->total_vals[a].begin(), R->total_vals[a].end(), 0);
            total.aggs[a].values = R->total_vals[a].data();
        }
    }
    std::vector<RowStore> &cell_rows = R->rows[C.time_mode ? 1 : 0];
    if (!C.time_mode) R->rows[1].clear();
    cell_rows.resize(live.size());
    const size_t n_all_rows = live.size() + alltime.size() + 1;
    R->agg_pool.resize(n_all_rows * na);
    R->val_pool.resize(n_all_rows * na);
    R->pctoff_pool.resize(n_all_rows * na);
    R->pct_pool.resize(C.want_percentiles ? n_all_rows * na * 100 : 0);
    // pass 2: one row per live cell.  Rows own disjoint pool slots, so ranges of cells are
    // finished by worker threads when there are enough of them to pay for the threads.
    // a printer's result (FinCtx::top_only): the rows of the sort order's first top_n places get percentiles and stddev
    // from their bucket arrays (fetch_top_values), exactly as a small result's rows do; the others have neither
    std::vector<int32_t> top_ix;
    if (C.top_only) {
        top_ix.assign(live.size(), -1);
        for (size_t i = 0; i < R->top_n && i < live.size(); i++) top_ix[R->order0.empty() ? i : R->order0[i]] = (int32_t)i;
    }
    auto work = [&](size_t i0, size_t i1, CellAcc *tot, std::vector<std::vector<int64_t>> *tot_vals) {
        CellAcc acc;
        for (size_t i = i0; i < i1; i++) {
            const int64_t cell = live[i];
            load_cell(cell, acc);
            if (C.top_only && top_ix[i] >= 0)
                for (size_t a = 0; a < na; a++)
                    if (C.aggs[a].d.hist_full) acc.aggs[a].values = R->top_vals.data() + (size_t)top_ix[i] * (size_t)P.hist_stride + P.hist_agg_off[a];
            // (hash group-by: the composite key is [time bucket || group key], the dense arrays are in key order)
            const int64_t ckey = hashed ? (int64_t)C.dense_keys[(size_t)cell] : cell;
            const int64_t tbi = hashed && !C.time_mode ? 0 : ckey / gcells, gcell = ckey - tbi * gcells;
            RowStore &row = cell_rows[i];
            row.agg_off = (int64_t)(i * na);
            row.cell = cell;
            if (keys_cached) {
                row.key = ks->key((size_t)gcell);
                row.gbkp = &ks->gbk[(size_t)gcell];
            } else {
                build_key(C.q, gcell, ks->key(i), ks->gbk[i]);
                row.key = ks->key(i);
                row.gbkp = &ks->gbk[i];
            }
            row.time_bucket = C.time_mode ? (P.tb_min + tbi) * P.time_bucket : 0;  // (rows are recycled: assign every field)
            finish_row(q, R, acc, row, out_usable);
            if (!C.time_mode) {
                for (size_t a = 0; a < na; a++) {
                    AggAcc &d = tot->aggs[a];
                    const AggAcc &s = acc.aggs[a];
                    d.cnt += s.cnt;
                    d.smp += s.smp;
                    d.pop += s.pop;
                    d.sum += s.sum;
                    d.sb += s.sb;
                    d.sb2 += s.sb2;
                    d.n_out += s.n_out;
                    d.sum_out += s.sum_out;
                    for (int k = 0; k < 4; k++) d.sq[k] += s.sq[k];
                    d.vmax = std::max(d.vmax, s.vmax);
                    d.nmin = std::max(d.nmin, s.nmin);
                    if (s.values) {
                        int64_t *tv = (*tot_vals)[a].data();
                        for (int64_t k = 0; k < C.aggs[a].d.n_values; k++) tv[k] += s.values[k];
                    }
                }
            }
            tot->count += acc.count;
            tot->samples += acc.samples;
        }
    };
    size_t n_threads = 1;
    {
        // cost estimate: buckets touched per row dominate in full-histogram mode
        size_t per_row = 64 + (size_t)(C.want_percentiles ? P.hist_stride * 3 : 0);
        size_t cost = live.size() * per_row;
        n_threads = std::max<size_t>(1, std::min(WorkerPool::cap(), cost / (1u << 20)));
    }
    if (n_threads <= 1) {
        work(0, live.size(), &total, &R->total_vals);
    } else {
        std::vector<CellAcc> part(n_threads);
        std::vector<std::vector<std::vector<int64_t>>> part_vals(n_threads);
        for (size_t k = 0; k < n_threads; k++) {
            part[k].has_aggs = total.has_aggs;
            part_vals[k].resize(na);
            for (size_t a = 0; a < na; a++) {
                part[k].aggs[a].tracked_cnt = true;
                if (!R->total_vals[a].empty()) part_vals[k][a].assign(R->total_vals[a].size(), 0);
            }
        }
        // many small ranges claimed as threads get to them: with one range per thread a worker that wakes late (or shares
        // its CPU) doubles the phase -- the row build of config 5 took 2.0 or 5.6 ms from one finalize to the next
        const size_t n_ranges = n_threads * 8;
        std::atomic<size_t> next_range{0};
        WorkerPool::get().run(n_threads, [&](size_t k) {
            for (size_t j = next_range++; j < n_ranges; j = next_range++)
                work(live.size() * j / n_ranges, live.size() * (j + 1) / n_ranges, &part[k], &part_vals[k]);
        });
        for (size_t k = 0; k < n_threads; k++) {
            total.count += part[k].count;
            total.samples += part[k].samples;
            for (size_t a = 0; a < na; a++) {
                AggAcc &d = total.aggs[a];
                const AggAcc &s2 = part[k].aggs[a];
                d.cnt += s2.cnt;
                d.smp += s2.smp;
                d.pop += s2.pop;
                d.sum += s2.sum;
                d.sb += s2.sb;
                d.sb2 += s2.sb2;
                d.n_out += s2.n_out;
                d.sum_out += s2.sum_out;
                for (int j = 0; j < 4; j++) d.sq[j] += s2.sq[j];
                d.vmax = std::max(d.vmax, s2.vmax);
                d.nmin = std::max(d.nmin, s2.nmin);
                for (size_t j = 0; j < part_vals[k][a].size(); j++) R->total_vals[a][j] += part_vals[k][a][j];
            }
        }
    }
    if (summary) R->total_vals = summary_totals;
    for (size_t a = 0; a < na; a++)
        if (!R->total_vals[a].empty()) total.aggs[a].values = R->total_vals[a].data();
    if (C.pushdown)  // (the cells beyond the limit carry no sums: Cumulative's come from the scan's own totals over every row)
        for (size_t a = 0; a < na; a++) {
            total.aggs[a].sum = (uint64_t)C.pd_sum[a];
            total.aggs[a].vmax = C.aggs[a].d.m_max >= 0 ? C.pd_max[a] : total.aggs[a].vmax;
        }
    size_t next_slot = live.size();
    if (C.time_mode) {
        // all-time Results carry Count/Samples only (aggregate.go:156-169)
        R->rows[0].resize(alltime.size());
        for (size_t i = 0; i < alltime.size(); i++) {
            const int64_t g = alltime[i];
            RowStore &row = R->rows[0][i];
            row.agg_off = (int64_t)((next_slot + i) * na);
            row.time_bucket = 0;
            row.cell = -1;  // (rows are recycled)
            const size_t e = keys_cached ? (size_t)g : next_slot + i;
            if (!keys_cached) build_key(C.q, g, ks->key(e), ks->gbk[e]);
            row.key = ks->key(e);
            row.gbkp = &ks->gbk[e];
            CellAcc a2;
            a2.count = all_count[hashed ? i : (size_t)g];
            a2.samples = all_samples[hashed ? i : (size_t)g];
            finish_row(q, R, a2, row, out_usable);
        }
        next_slot += alltime.size();
    }
    // Cumulative, aggregate.go:422-438
    {
        R->rows[2].resize(1);
        RowStore &row = R->rows[2].back();
        row.agg_off = (int64_t)(next_slot * na);
        row.time_bucket = 0;
        const size_t e = keys_cached ? (size_t)gcells : next_slot;
        if (!keys_cached) {
            memset(ks->key(e), 0, KeyStore::kKeyBytes);
            ks->gbk[e] = "TOTAL";
            for (size_t g = 1; g < C.n_groups; g++) ks->gbk[e] += "\t";
        }
        row.key = ks->key(e);
        row.gbkp = &ks->gbk[e];
        row.cell = -1;
        finish_row(q, R, total, row, out_usable);
    }
    // the outliers' values -> the rows that own them
    {
        const std::vector<std::pair<int64_t, int64_t>> &recs = R->out_recs;
        R->outlier_vals.resize(recs.size());
        for (size_t i = 0; i < recs.size(); i++) R->outlier_vals[i] = recs[i].second;
        for (size_t i = 0; i < recs.size();) {
            size_t j = i;
            while (j < recs.size() && recs[j].first == recs[i].first) j++;
            sybl_agg_out &o = R->agg_pool[(size_t)recs[i].first];  // (cell rows own the first live.size() * na pool slots)
            o.outlier_values = R->outlier_vals.data() + i;
            o.n_outlier_values = (int64_t)(j - i);
            i = j;
        }
    }
    // the printed rows' bucket arrays (fetch_top_values)
    for (size_t i = 0; i < R->top_n; i++)
        for (size_t a = 0; a < na; a++)
            if (R->agg_pool[(size_t)R->sorted0(i).agg_off + a].present) {
                const size_t k = (size_t)R->sorted0(i).agg_off + a;
                R->val_pool[k] = R->top_vals.data() + i * (size_t)P.hist_stride + P.hist_agg_off[a];
                R->agg_pool[k].values = R->val_pool[k];
            }
    make_views(R);
    if (R->owner) {
        auto &v = R->owner->lazy_results;
        v.erase(std::remove(v.begin(), v.end(), R), v.end());
        R->owner = nullptr;
    }
    R->fin.q = nullptr;
}

}  // namespace sybl

using namespace sybl;

extern "C" {

int sybl_result_rows(const sybl_result *r, int which, const sybl_group_row **rows, int64_t *n) {
    SYBL_API_GUARD(r);
    if (r) result_ensure_rows((Result *)r);
    const Result *R = (const Result *)r;
    if (!R || which < 0 || which > 2) return fail(SYBL_E_INVAL, "sybl_result_rows: bad argument");
    if (rows) *rows = R->view[which].data();
    if (n) *n = (int64_t)R->view[which].size();
    return SYBL_OK;
}

int sybl_result_subhists(const sybl_result *r, int agg, const sybl_subhist **subs, int64_t *n) {
    SYBL_API_GUARD(r);
    const Result *R = (const Result *)r;
    if (!R || !subs || !n || agg < 0 || agg >= R->n_aggs) return fail(SYBL_E_INVAL, "sybl_result_subhists: bad argument");
    *subs = (size_t)agg < R->subs.size() && !R->subs[(size_t)agg].empty() ? R->subs[(size_t)agg].data() : nullptr;
    *n = (size_t)agg < R->subs.size() ? (int64_t)R->subs[(size_t)agg].size() : 0;
    return SYBL_OK;
}

int sybl_result_distinct(const sybl_result *r, int which, int64_t row, int64_t *cardinality, const uint8_t **registers) {
    SYBL_API_GUARD(r);
    const Result *R = (const Result *)r;
    if (!R || !R->has_distinct) return fail(SYBL_E_INVAL, "not a count-distinct result");
    if (which < 0 || which > 2 || row < 0 || row >= (int64_t)R->rows[which].size()) return fail(SYBL_E_INVAL, "no such row");
    // (Results are handed out in sorted order: Result::order0)
    const size_t built = which == 0 && !R->order0.empty() ? (size_t)R->order0[(size_t)row] : (size_t)row;
    if (cardinality) *cardinality = R->distinct[which][built];
    if (registers) *registers = R->row_registers(which, R->rows[which][built]);
    return SYBL_OK;
}

int sybl_debug_hll_ints(const int64_t *values, const uint8_t *populated, int64_t n_rows, int32_t n_cols, uint8_t *registers) {
    if (!values || !registers || n_cols < 1 || n_cols > kMaxDistinct) return fail(SYBL_E_INVAL, "bad arguments");
    for (int64_t i = 0; i < n_rows; i++) {
        uint64_t w[kMaxDistinct];
        for (int c = 0; c < kMaxDistinct; c++)
            w[c] = c < n_cols && (!populated || populated[i * n_cols + c]) ? (uint64_t)values[i * n_cols + c] : ~(uint64_t)0;
        uint32_t reg, rank;
        hll_place(metro64_words(w, n_cols, kHllSeed), reg, rank);
        if (registers[reg] < rank) registers[reg] = (uint8_t)rank;
    }
    return SYBL_OK;
}

uint64_t sybl_debug_hll_bytes(const uint8_t *bytes, int64_t len, uint8_t *registers) {
    uint64_t h = metro64_bytes(bytes, (size_t)len, kHllSeed);
    // (the piecewise form the device's slow path over several columns uses must agree, whatever the length: a difference
    // is reported as a hash no checker will accept)
    Metro64Stream S;
    S.init(kHllSeed);
    for (int64_t i = 0; i < len; i++) S.put(bytes[i]);
    if (S.finish() != h) h = ~h;
    if (registers) {
        uint32_t reg, rank;
        hll_place(h, reg, rank);
        if (registers[reg] < rank) registers[reg] = (uint8_t)rank;
    }
    return h;
}

int64_t sybl_debug_hll_cardinality(const uint8_t *registers) { return registers ? (int64_t)hll_cardinality(registers) : -1; }

}  // extern "C"
namespace sybl {
std::shared_ptr<std::recursive_mutex> api_mutex_of(const sybl_result *r) { return r ? ((const Result *)r)->api_m : nullptr; }
}
extern "C" {

int64_t sybl_result_matched(const sybl_result *r) { SYBL_API_GUARD(r); return r ? ((const Result *)r)->matched : 0; }

void sybl_result_free(sybl_result *r) { SYBL_API_GUARD(r); delete (Result *)r; }

}  // extern "C"
