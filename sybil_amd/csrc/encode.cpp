// encode.cpp -- `-encode-results`: the result as encoding/gob of NodeResults (printer.go:284-289).
#include <string.h>

#include "gobenc.h"
#include "result.h"

using namespace sybl;

extern "C" {

// ------------------------------------------------------------------ -encode-results (printer.go:284-289)
// encoding/gob of NodeResults{QuerySpec{QueryParams, QueryResults}} (node_aggregator.go:8-13,
// query_spec.go:16-93) -- what `sybil query -encode-results` prints and `sybil aggregate` / src/api read.
// gob matches struct fields by NAME, so only the fields the engine fills are defined and sent; the
// reference's golden NodeResults (testdata/TestDecodeGoldenFiles/node_results.golden.gob) carries the
// same type and field names (tests/test_gpu_cli.py decodes both with the same decoder).
namespace {

using GobW = gobenc::Buf;
using gobenc::Fields;

enum GobId : int {  // builtin ids, then ours in definition order
    G_BOOL = 1, G_INT = 2, G_FLOAT = 4, G_STRING = 6, G_IFACE = 8,
    T_NODE = 65, T_QSPEC, T_QPARAMS, T_GROUPING, T_GROUPINGS, T_AGG, T_AGGS, T_QRESULTS, T_RESULT, T_HISTMAP, T_RESULTMAP,
    T_TIMEMAP, T_SORTED, T_HISTCOMPAT, T_BASICHIST, T_CACHED, T_I64S, T_INTINFO, T_F64S, T_MULTICOMPAT, T_MULTI, T_SUBHISTS, T_LOGLOG,
};

struct GobStream {
    std::string out;
    void message(const std::string &payload) {
        GobW h;
        h.u(payload.size());
        out += h.b;
        out += payload;
    }
    static void common(GobW &w, const char *name, int id) {  // CommonType{Name, Id}
        if (name && name[0]) {
            w.u(1);
            w.s(name);
            w.u(1);
        } else {
            w.u(2);
        }
        w.i(id);
        w.u(0);
    }
    void def_struct(int id, const char *name, std::initializer_list<std::pair<const char *, int>> fields) {
        GobW w;
        w.i(-id);
        w.u(3);  // wireType.StructT
        w.u(1);
        common(w, name, id);
        w.u(1);
        w.u(fields.size());
        for (auto &fd : fields) {
            w.u(1);
            w.s(fd.first);
            w.u(1);
            w.i(fd.second);
            w.u(0);
        }
        w.u(0);
        w.u(0);
        message(w.b);
    }
    void def_slice(int id, const char *name, int elem) {
        GobW w;
        w.i(-id);
        w.u(2);  // wireType.SliceT
        w.u(1);
        common(w, name, id);
        w.u(1);
        w.i(elem);
        w.u(0);
        w.u(0);
        message(w.b);
    }
    // a type that marshals itself (wireType.BinaryMarshalerT: gobEncoderType{CommonType}); its values travel as byte strings
    void def_binary_marshaler(int id, const char *name) {
        GobW w;
        w.i(-id);
        w.u(6);  // wireType.BinaryMarshalerT
        w.u(1);
        common(w, name, id);
        w.u(0);
        w.u(0);
        message(w.b);
    }
    void def_map(int id, const char *name, int key, int elem) {
        GobW w;
        w.i(-id);
        w.u(4);  // wireType.MapT
        w.u(1);
        common(w, name, id);
        w.u(1);
        w.i(key);
        w.u(1);
        w.i(elem);
        w.u(0);
        w.u(0);
        message(w.b);
    }
};

// one HistCompat{BasicHist{BasicHistCachedInfo{...}}} value
struct BasicFields {
    int64_t num_buckets = 0, bucket_size = 0;
    const int64_t *values = nullptr;
    int64_t n_values = 0;
    bool percentile_mode = false;
    const int64_t *outliers = nullptr;  // ascending; those below info_min are the Underliers
    int64_t n_outliers = 0;
    const int64_t *out_counts = nullptr;  // non-NULL: outlier k is the value out_first + k, out_counts[k] times
    int64_t out_first = 0;
    int64_t max = 0, min = 0, samples = 0, count = 0;
    double avg = 0;
    int64_t info_min = 0, info_max = 0;
};

static void gob_histcompat(GobW &v, const BasicFields &b) {
    Fields hc(v);
    hc.at(0);  // HistCompat.BasicHist
    Fields bh(v);
    bh.at(0);  // BasicHist.BasicHistCachedInfo
    Fields ci(v);
    ci.put_int(0, b.num_buckets);
    ci.put_int(1, b.bucket_size);
    if (b.values && b.n_values > 0) {
        ci.at(2);  // Values []int64
        v.u((uint64_t)b.n_values);
        for (int64_t k = 0; k < b.n_values; k++) v.i(b.values[k]);
    }
    // (Averages []float64, field 3: the per-bucket running means are written by AddWeightedValue and read by nothing
    // (hist_basic.go:144-150; Combine, GetPercentiles, GetStdDev and the printers ignore them).  They are not tracked:
    // omitted, i.e. nil to a Go decoder.)
    if (b.percentile_mode) {
        ci.at(4);
        v.u(1);
    }
    if (b.out_counts) {
        // a sub-histogram of a MultiHist: its Outliers from the exact per-value counters
        int64_t n = 0;
        for (int64_t k = 0; k < b.n_outliers; k++) n += b.out_counts[k];
        if (n > 0) {
            ci.at(5);
            v.u((uint64_t)n);
            for (int64_t k = 0; k < b.n_outliers; k++)
                for (int64_t j = 0; j < b.out_counts[k]; j++) v.i(b.out_first + k);
        }
    } else if (b.n_outliers > 0) {
        // Outliers (beyond the last bucket) / Underliers (below hist Min), hist_basic.go:132-142; values ascending
        int64_t n_under = 0;
        while (n_under < b.n_outliers && b.outliers[n_under] < b.info_min) n_under++;
        if (b.n_outliers > n_under) {
            ci.at(5);
            v.u((uint64_t)(b.n_outliers - n_under));
            for (int64_t k = n_under; k < b.n_outliers; k++) v.i(b.outliers[k]);
        }
        if (n_under > 0) {
            ci.at(6);
            v.u((uint64_t)n_under);
            for (int64_t k = 0; k < n_under; k++) v.i(b.outliers[k]);
        }
    }
    ci.put_int(7, b.max);
    ci.put_int(8, b.min);
    ci.put_int(9, b.samples);
    ci.put_int(10, b.count);
    if (b.avg != 0.0) {
        ci.at(11);
        v.f(b.avg);
    }
    {
        ci.at(12);  // Info IntInfo{Min, Max}
        Fields in(v);
        in.put_int(0, b.info_min);
        in.put_int(1, b.info_max);
        in.end();
    }
    ci.end();
    bh.end();
    hc.end();
}

// MultiHist{Max, Min, Samples, Count, Avg, PercentileMode, Subhists []*HistCompat, Info *IntInfo} (hist_multi.go:6-19).
// Of a sub-histogram only what the reference reads back is known here: its buckets, its outliers, Count = their sum
// and the range; its own Avg / Samples (read by nothing: MultiHist keeps the mean) stay zero.
static void gob_multihist(GobW &v, const Result *R, const sybl_agg_out &o, int a) {
    Fields mh(v);
    mh.put_int(0, o.max);
    mh.put_int(1, o.min);
    mh.put_int(2, o.samples);
    mh.put_int(3, o.count);
    if (o.avg != 0.0) {
        mh.at(4);
        v.f(o.avg);
    }
    const std::vector<sybl_subhist> &subs = R->subs[(size_t)a];
    if (R->op == SYBL_AGG_HIST) {
        mh.at(5);  // PercentileMode
        v.u(1);
        if (!subs.empty() && o.values) {
            mh.at(6);
            v.u(subs.size());
            for (const sybl_subhist &S : subs) {
                BasicFields b;
                b.num_buckets = S.num_buckets;
                b.bucket_size = S.bucket_size;
                b.values = o.values + S.offset;
                b.n_values = S.n_values;
                b.percentile_mode = true;
                b.out_counts = o.values + S.ext_offset;
                b.n_outliers = S.n_ext;
                b.out_first = S.ext_first;
                b.max = S.info_max;
                b.min = S.info_min;
                for (int64_t k = 0; k < S.n_values; k++) b.count += b.values[k];
                b.info_min = S.info_min;
                b.info_max = S.info_max;
                gob_histcompat(v, b);
            }
        }
    }
    {
        mh.at(7);  // Info *IntInfo
        Fields in(v);
        in.put_int(0, R->agg_info[(size_t)a].first);
        in.put_int(1, R->agg_info[(size_t)a].second);
        in.end();
    }
    mh.end();
}

static void gob_hist(GobW &w, const Result *R, const sybl_agg_out &o, int a) {
    // Histogram interface value: registered name, concrete type id, byte count, the value
    GobW v;
    if (R->loghist) {
        // MultiHistCompat{*MultiHist; Histogram *MultiHist} (hist_compat.go:50-54): both point at the same histogram and
        // gob flattens pointers, so it travels twice
        Fields mc(v);
        mc.at(0);
        gob_multihist(v, R, o, a);
        mc.at(1);
        gob_multihist(v, R, o, a);
        mc.end();
        w.s("*sybil.MultiHistCompat");
        w.i(T_MULTICOMPAT);
        w.u(v.b.size());
        w.b += v.b;
        return;
    }
    BasicFields b;
    b.num_buckets = o.num_buckets;
    b.bucket_size = o.bucket_size;
    b.values = o.values;
    b.n_values = o.n_values;
    b.percentile_mode = R->op == SYBL_AGG_HIST;
    b.outliers = o.outlier_values;
    b.n_outliers = o.n_outlier_values > 0 ? o.n_outlier_values : 0;
    b.max = o.max;
    b.min = o.min;
    b.samples = o.samples;
    b.count = o.count;
    b.avg = o.avg;
    b.info_min = R->agg_info[(size_t)a].first;
    b.info_max = R->agg_info[(size_t)a].second;
    gob_histcompat(v, b);
    w.s("*sybil.HistCompat");
    w.i(T_HISTCOMPAT);
    w.u(v.b.size());
    w.b += v.b;
}

static void gob_result(GobW &w, const Result *R, const RowStore &row, size_t n_groups, const uint8_t *regs = nullptr) {
    Fields f(w);
    bool any = false;
    for (int a = 0; a < R->n_aggs; a++) any = any || R->agg_pool[(size_t)row.agg_off + a].present;
    if (any) {
        f.at(0);  // Hists map[string]Histogram
        size_t n = 0;
        for (int a = 0; a < R->n_aggs; a++) n += R->agg_pool[(size_t)row.agg_off + a].present ? 1 : 0;
        w.u(n);
        for (int a = 0; a < R->n_aggs; a++) {
            const sybl_agg_out &o = R->agg_pool[(size_t)row.agg_off + a];
            if (!o.present) continue;
            w.s(R->agg_names[(size_t)a]);
            gob_hist(w, R, o, a);
        }
    }
    f.put_str(1, row.gbk());
    if (n_groups > 0) {
        f.at(2);  // BinaryByKey: 8 little-endian bytes per group column
        w.s((const char *)row.key, n_groups * SYBL_GROUP_BY_WIDTH);
    }
    f.put_int(3, row.count);
    f.put_int(4, row.samples);
    if (R->has_distinct && regs) {
        // Result.Distinct *hll.LogLogBeta (query_spec.go:87) as a self-marshalling value: one byte of precision (14) and the
        // 16384 registers, in register order.  PARITY UNPINNED: github.com/logv/loglogbeta is not in the reference tree and
        // no reference test holds an encoded sketch; a `sybil aggregate` built against the real dependency may expect
        // another blob.  The registers are the ones sybl_result_distinct hands out (merge = register-wise maximum).
        f.at(5);
        w.u((uint64_t)(1 + SYBL_HLL_REGISTERS));
        const char prec = 14;
        w.b.append(&prec, 1);
        w.b.append((const char *)regs, (size_t)SYBL_HLL_REGISTERS);
    }
    f.end();
}

static void gob_result_map(GobW &w, const Result *R, const std::vector<RowStore> &rows, size_t i0, size_t i1, size_t n_groups, int which = 0) {
    w.u(i1 - i0);
    for (size_t i = i0; i < i1; i++) {
        w.s(rows[i].gbk());
        gob_result(w, R, rows[i], n_groups, R->has_distinct ? R->row_registers(which, rows[i]) : nullptr);
    }
}

}  // namespace

const void *sybl_result_encode(sybl_result *r, int64_t *n_bytes) {
    SYBL_API_GUARD(r);
    Result *R = (Result *)r;
    if (R) result_ensure_rows(R);
    if (!R || !n_bytes) {
        set_error("sybl_result_encode: NULL argument");
        return nullptr;
    }
    if (R->loghist) {
        // (every outlier of every sub-histogram is written out as a value: bound it)
        int64_t total = 0;
        for (auto &o : R->agg_pool) total += o.present ? o.n_outliers : 0;
        if (total > ((int64_t)1 << 26)) {
            set_error("-encode-results of this -loghist result would list %lld outlier values", (long long)total);
            return nullptr;
        }
    }
    for (auto &o : R->agg_pool)
        if (o.present && o.n_outlier_values < 0) {
            set_error("histograms with outliers cannot be encoded: their values were not kept (outlier log overflow, or a result merged "
                      "across ranks)");
            return nullptr;
        }
    GobStream S;
    S.def_struct(T_NODE, "NodeResults", {{"QuerySpec", T_QSPEC}});
    S.def_struct(T_QSPEC, "QuerySpec", {{"QueryParams", T_QPARAMS}, {"QueryResults", T_QRESULTS}});
    S.def_struct(T_QPARAMS, "QueryParams",
                 {{"Groups", T_GROUPINGS}, {"Aggregations", T_AGGS}, {"OrderBy", G_STRING}, {"OrderAsc", G_BOOL}, {"Limit", G_INT},
                  {"TimeBucket", G_INT}});
    S.def_struct(T_GROUPING, "Grouping", {{"Name", G_STRING}});
    S.def_slice(T_GROUPINGS, "[]sybil.Grouping", T_GROUPING);
    S.def_struct(T_AGG, "Aggregation", {{"Op", G_STRING}, {"Name", G_STRING}, {"HistType", G_STRING}});
    S.def_slice(T_AGGS, "[]sybil.Aggregation", T_AGG);
    S.def_struct(T_QRESULTS, "QueryResults",
                 {{"Cumulative", T_RESULT}, {"Results", T_RESULTMAP}, {"TimeResults", T_TIMEMAP}, {"MatchedCount", G_INT},
                  {"Sorted", T_SORTED}});
    if (R->has_distinct) {
        S.def_struct(T_RESULT, "Result",
                     {{"Hists", T_HISTMAP}, {"GroupByKey", G_STRING}, {"BinaryByKey", G_STRING}, {"Count", G_INT}, {"Samples", G_INT},
                      {"Distinct", T_LOGLOG}});
        S.def_binary_marshaler(T_LOGLOG, "LogLogBeta");
    } else {
        S.def_struct(T_RESULT, "Result",
                     {{"Hists", T_HISTMAP}, {"GroupByKey", G_STRING}, {"BinaryByKey", G_STRING}, {"Count", G_INT}, {"Samples", G_INT}});
    }
    S.def_map(T_HISTMAP, "map[string]sybil.Histogram", G_STRING, G_IFACE);
    S.def_map(T_RESULTMAP, "ResultMap", G_STRING, T_RESULT);
    S.def_map(T_TIMEMAP, "map[int]sybil.ResultMap", G_INT, T_RESULTMAP);
    S.def_slice(T_SORTED, "[]*sybil.Result", T_RESULT);
    S.def_struct(T_HISTCOMPAT, "HistCompat", {{"BasicHist", T_BASICHIST}});
    S.def_struct(T_BASICHIST, "BasicHist", {{"BasicHistCachedInfo", T_CACHED}});
    S.def_struct(T_CACHED, "BasicHistCachedInfo",
                 {{"NumBuckets", G_INT}, {"BucketSize", G_INT}, {"Values", T_I64S}, {"Averages", T_F64S}, {"PercentileMode", G_BOOL},
                  {"Outliers", T_I64S}, {"Underliers", T_I64S}, {"Max", G_INT}, {"Min", G_INT}, {"Samples", G_INT}, {"Count", G_INT},
                  {"Avg", G_FLOAT}, {"Info", T_INTINFO}});
    S.def_slice(T_I64S, "[]int64", G_INT);
    S.def_slice(T_F64S, "[]float64", G_FLOAT);
    S.def_struct(T_INTINFO, "IntInfo", {{"Min", G_INT}, {"Max", G_INT}});
    if (R->loghist) {
        S.def_struct(T_MULTICOMPAT, "MultiHistCompat", {{"MultiHist", T_MULTI}, {"Histogram", T_MULTI}});
        S.def_struct(T_MULTI, "MultiHist",
                     {{"Max", G_INT}, {"Min", G_INT}, {"Samples", G_INT}, {"Count", G_INT}, {"Avg", G_FLOAT}, {"PercentileMode", G_BOOL},
                      {"Subhists", T_SUBHISTS}, {"Info", T_INTINFO}});
        S.def_slice(T_SUBHISTS, "[]*sybil.HistCompat", T_HISTCOMPAT);
    }

    const size_t ng = R->group_names.size();
    GobW w;
    w.i(T_NODE);
    Fields node(w);
    node.at(0);  // NodeResults.QuerySpec
    Fields qs(w);
    qs.at(0);  // QueryParams
    {
        Fields qp(w);
        if (ng > 0) {
            qp.at(0);
            w.u(ng);
            for (auto &g : R->group_names) {
                Fields gf(w);
                gf.put_str(0, g);
                gf.end();
            }
        }
        if (R->n_aggs > 0) {
            qp.at(1);
            w.u((uint64_t)R->n_aggs);
            for (auto &a : R->agg_names) {
                Fields af(w);
                af.put_str(0, R->op == SYBL_AGG_HIST ? "hist" : "avg");
                af.put_str(1, a);
                af.put_str(2, R->loghist ? "multi" : "basic");  // query_spec.go:225-231
                af.end();
            }
        }
        qp.put_str(2, R->order_by);
        if (R->order_asc) {
            qp.at(3);
            w.u(1);
        }
        qp.put_int(4, R->limit);
        qp.put_int(5, R->time_mode ? R->time_bucket : 0);
        qp.end();
    }
    qs.at(1);  // QueryResults
    {
        Fields qr(w);
        qr.at(0);  // Cumulative
        gob_result(w, R, R->rows[2][0], 0, R->has_distinct ? R->row_registers(2, R->rows[2][0]) : nullptr);
        if (!R->rows[0].empty()) {
            qr.at(1);  // Results
            gob_result_map(w, R, R->rows[0], 0, R->rows[0].size(), ng);
        }
        if (!R->rows[1].empty()) {
            qr.at(2);  // TimeResults map[int]ResultMap: rows[1] is ordered by bucket
            const std::vector<RowStore> &tr = R->rows[1];
            size_t n_buckets = 0;
            for (size_t i = 0; i < tr.size(); i++) n_buckets += i == 0 || tr[i].time_bucket != tr[i - 1].time_bucket;
            w.u(n_buckets);
            for (size_t i = 0; i < tr.size();) {
                size_t j = i;
                while (j < tr.size() && tr[j].time_bucket == tr[i].time_bucket) j++;
                w.i(tr[i].time_bucket);
                gob_result_map(w, R, tr, i, j, ng, 1);
                i = j;
            }
        }
        qr.put_int(3, R->matched);
        if (!R->rows[0].empty() && !R->order_by.empty()) {
            qr.at(4);  // Sorted []*Result (SortResults, aggregate.go:497-525): the rows are already in that order
            w.u(R->rows[0].size());
            for (size_t i = 0; i < R->rows[0].size(); i++)
                gob_result(w, R, R->sorted0(i), ng, R->has_distinct ? R->row_registers(0, R->sorted0(i)) : nullptr);
        }
        qr.end();
    }
    qs.end();
    node.end();
    S.message(w.b);
    R->encoded.swap(S.out);
    *n_bytes = (int64_t)R->encoded.size();
    return R->encoded.data();
}

}  // extern "C"
