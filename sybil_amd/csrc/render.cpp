// render.cpp -- `sybil query` output formats over a finished result (printer.go:25-308): the text table
// (printSortedResults / printResult), -json (toResultJSON) and the tabwriter-aligned time-series table.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <map>

#include "result.h"

namespace sybl {

// ------------------------------------------------------------------ rendering (printer.go)

static void json_escape(const std::string &s, std::string &o) {
    o += '"';
    for (unsigned char ch : s) {
        switch (ch) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        case '<': o += "\\u003c"; break;  // encoding/json escapes HTML by default
        case '>': o += "\\u003e"; break;
        case '&': o += "\\u0026"; break;
        default:
            if (ch < 0x20) {
                char b[8];
                snprintf(b, sizeof(b), "\\u%04x", ch);
                o += b;
            } else {
                o += (char)ch;
            }
        }
    }
    o += '"';
}

// encoding/json float formatting: shortest repr that round-trips, 'e' form outside [1e-6,1e21)
static std::string go_float(double f) {
    if (f == 0) return signbit(f) ? "-0" : "0";
    if (!isfinite(f)) return "null";  // json.Marshal would fail; the reference prints nothing useful
    char buf[64];
    int prec = 1;
    for (; prec <= 17; prec++) {
        snprintf(buf, sizeof(buf), "%.*e", prec - 1, f);
        if (strtod(buf, nullptr) == f) break;
    }
    double af = fabs(f);
    if (af < 1e-6 || af >= 1e21) {
        // mantissa 'e' exponent with at least... Go: strconv 'e' then trims "e-07" -> "e-7"
        std::string s(buf);
        size_t e = s.find('e');
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        int exv = atoi(ex.c_str());
        char eb[16];
        snprintf(eb, sizeof(eb), "e%s%02d", exv < 0 ? "-" : "+", abs(exv));
        std::string r = mant + eb;
        // encoding/json: clean up e-09 to e-9
        size_t n = r.size();
        if (n >= 4 && r[n - 4] == 'e' && r[n - 3] == '-' && r[n - 2] == '0') {
            r[n - 2] = r[n - 1];
            r.resize(n - 1);
        }
        return r;
    }
    // 'f' form with the same digits
    int decimals = 0;
    {
        std::string s(buf);
        size_t e = s.find('e');
        int exv = atoi(s.c_str() + e + 1);
        decimals = std::max(0, (prec - 1) - exv);
    }
    snprintf(buf, sizeof(buf), "%.*f", decimals, f);
    return buf;
}

static void json_agg(Result *R, const RowStore &r, size_t a, std::string &o) {
    const size_t pk = (size_t)r.agg_off + a;
    const sybl_agg_out &g = R->agg_pool[pk];
    const int64_t *vals = R->val_pool[pk];
    const int64_t poff = R->pctoff_pool[pk];
    if (R->op == SYBL_AGG_AVG) {
        o += g.present ? go_float(g.avg) : "null";
        return;
    }
    o += "{";
    if (g.present) {
        // keys in the order encoding/json emits a map: sorted
        o += "\"avg\":" + go_float(g.avg);
        if (R->want_percentiles && vals) {
            // GetStrBuckets + getSparseBuckets: non-zero buckets keyed by their lower edge, sorted as strings
            // -- plus one per outlier / underlier under its own value (hist_basic.go:246-254)
            std::map<std::string, int64_t> bmap;
            if (R->loghist) {
                // MultiHist.GetStrBuckets (hist_multi.go:175-188): every sub-histogram's GetStrBuckets in turn, a later
                // one REPLACING an equal key of an earlier one (where GetSparseBuckets, behind percentiles and stddev, adds)
                for (const sybl_subhist &S : R->subs[a]) {
                    std::map<std::string, int64_t> sub;
                    for (int64_t k = 0; k < S.n_values; k++) sub[std::to_string((long long)(k * S.bucket_size + S.info_min))] = vals[S.offset + k];
                    for (int64_t k = 0; k < S.n_ext; k++)
                        if (vals[S.ext_offset + k] > 0) sub[std::to_string((long long)(S.ext_first + k))] += vals[S.ext_offset + k];
                    for (auto &kv : sub) bmap[kv.first] = kv.second;
                }
                for (auto it = bmap.begin(); it != bmap.end();) it = it->second > 0 ? std::next(it) : bmap.erase(it);
            } else {
                for (size_t b = 0; b < (size_t)R->n_values[a]; b++)
                    if (vals[b] > 0) bmap[std::to_string((long long)((int64_t)b * g.bucket_size + g.min))] += vals[b];
                if (g.n_outlier_values < 0) R->render_refused = true;  // outliers exist but their values were not kept
                for (int64_t k = 0; k < g.n_outlier_values; k++) bmap[std::to_string((long long)g.outlier_values[k])] += 1;
            }
            std::vector<std::pair<std::string, int64_t>> bk(bmap.begin(), bmap.end());
            o += ",\"buckets\":{";
            for (size_t k = 0; k < bk.size(); k++) {
                if (k) o += ",";
                o += "\"" + bk[k].first + "\":" + std::to_string((long long)bk[k].second);
            }
            o += "}";
            o += ",\"percentiles\":[";
            for (size_t k = 0; poff >= 0 && k < 100; k++) {
                if (k) o += ",";
                o += std::to_string((long long)g.percentiles[k]);
            }
            o += "]";
        }
        o += ",\"samples\":" + std::to_string((long long)g.count);  // "samples" = TotalCount() (printer.go:123)
        o += ",\"stddev\":" + go_float(g.stddev);
        o += ",\"sum\":" + go_float(g.avg * (double)g.count);  // Mean()*TotalCount(), printer.go:122
    }
    o += "}";
}

static void json_row(Result *R, const RowStore &r, std::string &o) {
    // ResultJSON is a map: keys are emitted sorted
    std::vector<std::pair<std::string, std::string>> kv;
    for (size_t a = 0; a < R->agg_names.size(); a++) {
        std::string v;
        json_agg(R, r, a, v);
        kv.emplace_back(R->agg_names[a], v);
    }
    size_t pos = 0;
    for (size_t g = 0; g < R->group_names.size(); g++) {
        size_t e = r.gbk().find('\t', pos);
        std::string part = r.gbk().substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        pos = e == std::string::npos ? r.gbk().size() : e + 1;
        std::string v;
        json_escape(part, v);
        kv.emplace_back(R->group_names[g], v);
    }
    if (R->has_distinct) {
        // printer.go:142-144: the cardinality stands in for Count as well, and Samples is not written
        kv.emplace_back("Distinct", std::to_string((long long)R->distinct_of(r)));
        kv.emplace_back("Count", std::to_string((long long)R->distinct_of(r)));
    } else {
        kv.emplace_back("Count", std::to_string((long long)r.count));
        kv.emplace_back("Samples", std::to_string((long long)r.samples));
    }
    std::stable_sort(kv.begin(), kv.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    // later duplicates of a key overwrite earlier ones in a Go map
    o += "{";
    bool first = true;
    for (size_t k = 0; k < kv.size(); k++) {
        if (k + 1 < kv.size() && kv[k + 1].first == kv[k].first) continue;
        if (!first) o += ",";
        first = false;
        json_escape(kv[k].first, o);
        o += ":" + kv[k].second;
    }
    o += "}";
}

static void text_row(const Result *R, const RowStore &r, std::string &o) {
    // printResult, printer.go:183-232
    std::string gk = r.gbk();
    std::replace(gk.begin(), gk.end(), '\t', ',');
    while (!gk.empty() && gk.back() == ',') gk.pop_back();
    char b[64];
    snprintf(b, sizeof(b), "%-20s", gk.c_str());
    std::string pad(b);
    o += pad.substr(0, 20);
    if (r.count != 0) o += std::to_string((long long)r.count);  // "%.0d" prints nothing for 0
    if (R->weighted) o += " (" + std::to_string((long long)r.samples) + ")";
    if (R->has_distinct) o += " Distinct: " + std::to_string((long long)R->distinct_of(r));  // printer.go:204-205
    o += "\n";
    for (size_t a = 0; a < R->agg_names.size(); a++) {
        const sybl_agg_out &g = R->agg_pool[(size_t)r.agg_off + a];
        const int64_t poff = R->pctoff_pool[(size_t)r.agg_off + a];
        snprintf(b, sizeof(b), "  %5s", R->agg_names[a].c_str());
        std::string col = b;
        if (R->op == SYBL_AGG_HIST) {
            if (!g.present) continue;
            if (poff >= 0) {
                const int64_t *p = g.percentiles;
                char line[512];
                snprintf(line, sizeof(line), "%s | %lld %lld | %.2f | %lld %lld %lld %lld %lld | %.2f\n", col.c_str(),
                         (long long)p[0], (long long)p[99], g.avg, (long long)p[0], (long long)p[25], (long long)p[50],
                         (long long)p[75], (long long)p[99], g.stddev);
                o += line;
            } else if (!R->want_percentiles) {
                // moments-only result: no percentile columns to print
                char line[256];
                snprintf(line, sizeof(line), "%s | %.2f | %.2f\n", col.c_str(), g.avg, g.stddev);
                o += line;
            } else {
                o += col + " No Data\n";
            }
        } else {
            char line[128];
            snprintf(line, sizeof(line), "%s %.2f\n", col.c_str(), g.present ? g.avg : 0.0);
            o += line;
        }
    }
}

}  // namespace sybl

using namespace sybl;

extern "C" {

const char *sybl_result_render(sybl_result *r, int format) {
    SYBL_API_GUARD(r);
    Result *R = (Result *)r;
    if (R) result_ensure_rows(R);
    if (!R || (format != 0 && format != 1)) {
        set_error("sybl_result_render: bad argument");
        return nullptr;
    }
    std::string &o = R->rendered[format];
    o.clear();
    R->render_refused = false;
    size_t lim = R->rows[0].size();
    if (R->limit > 0 && (size_t)R->limit < lim) lim = (size_t)R->limit;
    if (R->time_mode) {
        // printTimeResults, printer.go:25-107
        std::vector<const RowStore *> top;
        for (size_t i = 0; i < lim; i++) top.push_back(&R->sorted0(i));
        auto is_top = [&](const RowStore &x) {
            for (auto *t : top)
                if (t->gbk() == x.gbk()) return true;
            return false;
        };
        if (format == 1) {
            // map[string][]ResultJSON keyed by the bucket as a decimal string (sorted as strings)
            std::vector<std::pair<std::string, std::string>> kv;
            size_t i = 0;
            while (i < R->rows[1].size()) {
                int64_t tb = R->rows[1][i].time_bucket;
                std::string arr = "[";
                bool first = true;
                for (; i < R->rows[1].size() && R->rows[1][i].time_bucket == tb; i++) {
                    if (!is_top(R->rows[1][i])) continue;
                    if (!first) arr += ",";
                    first = false;
                    json_row(R, R->rows[1][i], arr);
                }
                arr += "]";
                kv.emplace_back(std::to_string((long long)tb), arr);
            }
            std::sort(kv.begin(), kv.end());
            o += "{";
            for (size_t k = 0; k < kv.size(); k++) {
                if (k) o += ",";
                o += "\"" + kv[k].first + "\":" + kv[k].second;
            }
            o += "}";
        } else {
            // printTimeResults text form (printer.go:64-107): every row is written through a
            // text/tabwriter (minwidth 0, tabwidth 1, padding 0, padchar ' ', AlignRight) as
            //   Fprintln(w, time_str, "\t", Count, "\t", GroupByKey, "\t"[, agg, "\t", avg, "\t"])
            // (Fprintln puts a space between operands); time_str = time.Unix(bucket, 0) in
            // OPTS.TIME_FORMAT "2006-01-02 15:04:05.999999999 -0700 MST" (config.go:127), local zone.
            std::vector<std::string> lines;
            auto time_str = [](int64_t tb) {
                time_t tt = (time_t)tb;
                struct tm tmv;
                localtime_r(&tt, &tmv);
                char b[96];
                strftime(b, sizeof(b), "%Y-%m-%d %H:%M:%S %z %Z", &tmv);
                return std::string(b);
            };
            for (auto &row : R->rows[1]) {
                if (R->has_distinct) {  // printer.go:79-80: the cardinality, and no aggregation columns
                    lines.push_back(time_str(row.time_bucket) + " \t " + std::to_string((long long)R->distinct_of(row)) + " \t " + row.gbk() + " \t");
                    continue;
                }
                std::string head = time_str(row.time_bucket) + " \t " + std::to_string((long long)row.count) + " \t " + row.gbk() + " \t";
                bool any = false;
                for (size_t a = 0; a < R->agg_names.size(); a++) {
                    const sybl_agg_out &g = R->agg_pool[(size_t)row.agg_off + a];
                    if (!g.present) continue;
                    char avg[64];
                    snprintf(avg, sizeof(avg), "%.2f", g.avg);
                    lines.push_back(head + " " + R->agg_names[a] + " \t " + avg + " \t");
                    any = true;
                }
                if (!any) lines.push_back(head);  // len(r.Hists) == 0
            }
            // tabwriter: cells end at a tab; a column's width is the widest cell of the
            // contiguous run of lines that have that column; AlignRight pads on the left
            std::vector<std::vector<std::string>> cells(lines.size());
            std::vector<std::string> tail(lines.size());
            size_t maxcols = 0;
            for (size_t i = 0; i < lines.size(); i++) {
                size_t pos = 0;
                for (;;) {
                    size_t e = lines[i].find('\t', pos);
                    if (e == std::string::npos) break;
                    cells[i].push_back(lines[i].substr(pos, e - pos));
                    pos = e + 1;
                }
                tail[i] = lines[i].substr(pos);
                maxcols = std::max(maxcols, cells[i].size());
            }
            std::vector<std::vector<size_t>> width(lines.size());
            for (size_t i = 0; i < lines.size(); i++) width[i].assign(cells[i].size(), 0);
            for (size_t c = 0; c < maxcols; c++) {
                size_t i = 0;
                while (i < lines.size()) {
                    if (cells[i].size() <= c) { i++; continue; }
                    size_t j = i, w = 0;
                    while (j < lines.size() && cells[j].size() > c) { w = std::max(w, cells[j][c].size()); j++; }
                    for (size_t k = i; k < j; k++) width[k][c] = w;
                    i = j;
                }
            }
            for (size_t i = 0; i < lines.size(); i++) {
                for (size_t c = 0; c < cells[i].size(); c++) {
                    o.append(width[i][c] - cells[i][c].size(), ' ');
                    o += cells[i][c];
                }
                o += tail[i];
                o += "\n";
            }
        }
        if (R->render_refused) {
            set_error("rows with outliers cannot be printed as -json: their values were not kept (outlier log overflow, or a result merged across ranks)");
            return nullptr;
        }
        return o.c_str();
    }
    if (format == 1) {
        o += "[";
        for (size_t i = 0; i < lim; i++) {
            if (i) o += ",";
            json_row(R, R->sorted0(i), o);
        }
        o += "]";
    } else {
        // printSortedResults / printResults: the cumulative row first when there is more than one group
        if (lim > 1 && !R->rows[2].empty()) text_row(R, R->rows[2][0], o);
        for (size_t i = 0; i < lim; i++) text_row(R, R->sorted0(i), o);
    }
    if (R->render_refused) {
        set_error("rows with outliers cannot be printed as -json: their values were not kept (outlier log overflow, or a result merged across ranks)");
        return nullptr;
    }
    return o.c_str();
}

}  // extern "C"
