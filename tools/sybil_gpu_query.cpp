// sybil-gpu-query -- `sybil query` on the MI355X engine.
//
// A stand-in for the reference CLI's query subcommand (src/cmd/cmd_query.go:19-372) for hosts
// without a Go toolchain: same flag names and defaults, same output formats, but the scan runs
// through the sybl_* C ABI (include/sybilgpu.h) instead of LoadAndQueryRecords.  With a Go
// toolchain the real CLI gets the same effect through the cgo shim in INTEGRATION.md.
//
//   sybil-gpu-query -dir db -table events -group browser,device -int pageload -op hist
//       -int-filter "pageload:gt:100" -json
//
// Several GPUs: one process per GPU, each started with the same query plus  -gpu-rank i -gpu-ranks n -gpu-id-file path
// (-device picks the GPU).  Rank i loads the i-th contiguous share of the table's block directories; rank 0 writes the
// communicator's 128-byte id to the id file (as the reference ships its encoded flags to other hosts as files,
// scripts/basic_aggregation_test.sh:12-21, config.go -encode-flags), the others wait for it; the ranks agree on bounds and
// dictionaries (sybl_table_agree), scan, merge over RCCL (sybl_query_allreduce) and rank 0 prints -- the whole result,
// where the reference's `sybil aggregate` (node_aggregator.go:147-177, cmd_aggregate.go:10) re-reads per-host gob files.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <execinfo.h>
#include <signal.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "../include/sybilgpu.h"

static std::vector<std::string> split(const std::string &s, const std::string &sep) {
    std::vector<std::string> out;
    if (s.empty()) return out;
    size_t pos = 0;
    for (;;) {
        size_t e = sep.empty() ? std::string::npos : s.find(sep, pos);
        out.push_back(s.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
        if (e == std::string::npos) break;
        pos = e + sep.size();
    }
    return out;
}

// -stats: where a cold run's wall time goes (stderr), phase by phase
static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
struct Phases {
    double t0 = now_s(), last = t0;
    std::string line;
    void mark(const char *what) {
        const double t = now_s();
        char b[64];
        snprintf(b, sizeof(b), " %s=%.1fms", what, (t - last) * 1e3);
        line += b;
        last = t;
    }
};

static int die(const char *what) {
    fprintf(stderr, "sybil-gpu-query: %s: %s\n", what, sybl_last_error());
    return 1;
}

// SYBL_CLI_BACKTRACE=1: a crash prints where it happened (module + offset per frame) before the process dies -- the multi-rank
// tests set it, so a rank that falls over says more than "-11"
static void crash_trace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "sybil-gpu-query: fatal signal; frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char **argv) {
    if (getenv("SYBL_CLI_BACKTRACE")) {
        signal(SIGSEGV, crash_trace);
        signal(SIGBUS, crash_trace);
        signal(SIGABRT, crash_trace);
    }
    // flag defaults: src/cmd/cmd_query.go:19-74, src/lib/config.go:147-176
    std::map<std::string, std::string> f = {
        {"dir", "./db/"}, {"table", ""}, {"op", "avg"}, {"limit", "100"}, {"print", "true"}, {"json", "false"},
        {"sort", "$COUNT"}, {"sort-asc", "false"}, {"time", "false"}, {"time-col", "time"}, {"time-bucket", "3600"},
        {"weight-col", ""}, {"int-filter", ""}, {"str-filter", ""}, {"set-filter", ""}, {"int-bucket", "0"},
        {"int", ""}, {"str", ""}, {"set", ""}, {"group", ""}, {"field-separator", ","}, {"filter-separator", ":"},
        {"device", "0"}, {"gpu-rank", "0"}, {"gpu-ranks", "1"}, {"gpu-id-file", ""}, {"block-skip", "true"}, {"stats", "false"}, {"encode-results", "false"}, {"loghist", "false"}, {"str-replace", ""}, {"distinct", ""},
        // accepted so that a command line written for `sybil query` runs unchanged; without effect here:
        //  -prune-sort / -distinct-limit bound the reference's intermediate results (aggregate.go:347-359, table_query.go:258-279:
        //   which groups survive depends on block completion order) -- this engine aggregates every group exactly and cuts at
        //   -limit after the sort (note 9);
        //  -recycle-mem / -fast-recycle / -shorten-key-table / -cache-queries / -debug steer the Go runtime's memory, caches and log
        {"prune-sort", "$COUNT"}, {"distinct-limit", "-1"}, {"recycle-mem", "true"}, {"fast-recycle", "true"}, {"shorten-key-table", "true"},
        {"cache-queries", "false"}, {"debug", "false"}};
    const std::set<std::string> bools = {"print", "json", "sort-asc", "time", "block-skip", "stats", "encode-results", "loghist",
                                         "recycle-mem", "fast-recycle", "shorten-key-table", "cache-queries", "debug"};
    // flags of `sybil query` whose work is not the aggregation path (samples and exports read the row store, -tables / -info /
    // -update-info the table directory, the flag codecs feed the reference's own multi-host scripts): named, not "undefined"
    const std::set<std::string> elsewhere = {"samples", "sample-cols", "export", "read-log", "tables", "info", "update-info", "encode-flags", "decode-flags", "tdigest"};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() < 2 || a[0] != '-') {
            fprintf(stderr, "unexpected argument %s\n", a.c_str());
            return 2;
        }
        a = a.substr(a[1] == '-' ? 2 : 1);
        std::string val;
        size_t eq = a.find('=');
        bool has_val = eq != std::string::npos;
        if (has_val) {
            val = a.substr(eq + 1);
            a = a.substr(0, eq);
        }
        if (elsewhere.count(a)) {
            fprintf(stderr, "-%s is a flag of `sybil query` that this binary does not serve (it runs the filter / group / aggregate path only)\n", a.c_str());
            return 2;
        }
        if (!f.count(a)) {
            fprintf(stderr, "flag provided but not defined: -%s\n", a.c_str());
            return 2;
        }
        if (bools.count(a)) {
            f[a] = has_val ? val : "true";
        } else {
            if (!has_val) {
                if (i + 1 >= argc) {
                    fprintf(stderr, "flag needs an argument: -%s\n", a.c_str());
                    return 2;
                }
                val = argv[++i];
            }
            f[a] = val;
        }
    }
    auto on = [&](const char *k) { return f[k] == "true" || f[k] == "1"; };
    if (f["table"].empty()) {
        fprintf(stderr, "no table specified (-table)\n");
        return 2;
    }
    const std::string fs = f["field-separator"], ps = f["filter-separator"];
    std::vector<std::string> ints = split(f["int"], fs), groups = split(f["group"], fs);
    // -distinct a,b (cmd_query.go:61,124-126,216-219); -op distinct counts the -group columns instead of grouping by
    // them (:221-224) -- its aggregations are computed but printed by neither printer, so they are not requested here
    std::vector<std::string> distincts = split(f["distinct"], fs);
    if (f["op"] == "distinct") {
        distincts = groups;
        groups.clear();
        ints.clear();
    }

    // ---- filters (BuildFilters, filter.go:59-139)
    struct Filt {
        std::string col, op, val;
        int kind;
    };
    std::vector<Filt> filts;
    const char *fkeys[3] = {"int-filter", "str-filter", "set-filter"};
    for (int k = 0; k < 3; k++)
        for (auto &spec : split(f[fkeys[k]], fs)) {
            std::vector<std::string> tok = split(spec, ps);
            if (tok.size() < 3) {
                fprintf(stderr, "bad filter '%s' (want col%sop%sval)\n", spec.c_str(), ps.c_str(), ps.c_str());
                return 2;
            }
            filts.push_back({tok[0], tok[1], tok[2], k});
        }
    const int64_t time_bucket = atoll(f["time-bucket"].c_str());

    // ---- the columns the query references (the LoadSpec, table_load_spec.go:59-73)
    std::vector<std::string> cols;
    auto use = [&](const std::string &c) {
        for (auto &x : cols)
            if (x == c) return;
        if (!c.empty()) cols.push_back(c);
    };
    for (auto &c : ints) use(c);
    for (auto &c : groups) use(c);
    for (auto &c : distincts) use(c);
    for (auto &x : filts) use(x.col);
    if (on("time")) use(f["time-col"]);
    use(f["weight-col"]);

    const int rank = atoi(f["gpu-rank"].c_str()), nranks = atoi(f["gpu-ranks"].c_str());
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        fprintf(stderr, "-gpu-rank %d is not one of -gpu-ranks %d\n", rank, nranks);
        return 2;
    }
    if (nranks > 1 && f["gpu-id-file"].empty()) {
        fprintf(stderr, "-gpu-ranks %d needs -gpu-id-file (rank 0 writes the communicator id there, the others read it)\n", nranks);
        return 2;
    }
    Phases ph;
    sybl_ctx *ctx = nullptr;
    if (sybl_init(atoi(f["device"].c_str()), &ctx)) return die("init");
    ph.mark("init");
    if (nranks > 1) {
        unsigned char id[128];
        const std::string path = f["gpu-id-file"], tmp = path + ".tmp";
        if (rank == 0) {
            if (sybl_comm_unique_id(id)) return die("communicator id");
            FILE *fp = fopen(tmp.c_str(), "wb");
            if (!fp || fwrite(id, 1, sizeof(id), fp) != sizeof(id) || fclose(fp) != 0 || rename(tmp.c_str(), path.c_str()) != 0) {
                fprintf(stderr, "sybil-gpu-query: cannot write %s\n", path.c_str());
                return 1;
            }
        } else {
            // (the file appears whole: rank 0 renames it into place)
            const char *to = getenv("SYBIL_GPU_ID_TIMEOUT_S");
            const time_t give_up = time(nullptr) + (to ? atoi(to) : 300);
            size_t got = 0;
            while (got != sizeof(id)) {
                if (FILE *fp = fopen(path.c_str(), "rb")) {
                    got = fread(id, 1, sizeof(id), fp);
                    fclose(fp);
                }
                if (got != sizeof(id)) {
                    if (time(nullptr) > give_up) {
                        fprintf(stderr, "sybil-gpu-query: rank %d: no communicator id in %s\n", rank, path.c_str());
                        return 1;
                    }
                    usleep(2000);
                }
            }
        }
        if (sybl_comm_init(ctx, id, nranks, rank)) return die("communicator");
    }
    sybl_table *tab = nullptr;
    std::vector<const char *> cptr;
    for (auto &c : cols) cptr.push_back(c.c_str());
    // (compact storage: the columns at the narrowest width that holds their range -- what the packed scan kernels read)
    if (sybl_table_open_flags(ctx, f["dir"].c_str(), f["table"].c_str(), cptr.empty() ? nullptr : cptr.data(), (int32_t)cptr.size(), rank, nranks,
                              SYBL_OPEN_COMPACT, &tab))
        return die("open table");
    ph.mark("open");
    // the ranks' bounds, dictionaries and sparse-key dictionaries become one (collective); one rank: dictionaries are sorted,
    // so the output is the same whatever the number of GPUs
    if (!getenv("SYBL_CLI_SKIP_AGREE")) {  // (the switch: tests/test_gpu_cli_multirank.py shows what the layout check then says)
        std::vector<const char *> gp;
        for (auto &g : groups) gp.push_back(g.c_str());
        if (sybl_table_agree(tab, gp.empty() ? nullptr : gp.data(), (int32_t)gp.size())) return die("agree");
    }

    static const std::map<std::string, int> opcode = {{"gt", SYBL_OP_GT}, {"lt", SYBL_OP_LT}, {"eq", SYBL_OP_EQ},
                                                      {"neq", SYBL_OP_NEQ}, {"re", SYBL_OP_RE}, {"nre", SYBL_OP_NRE},
                                                      {"in", SYBL_OP_IN}, {"nin", SYBL_OP_NIN}};
    std::vector<sybl_filter> cf(filts.size());
    for (size_t i = 0; i < filts.size(); i++) {
        memset(&cf[i], 0, sizeof(sybl_filter));
        cf[i].col = filts[i].col.c_str();
        auto it = opcode.find(filts[i].op);
        cf[i].op = it == opcode.end() ? -1 : it->second;  // unknown op: the filter matches nothing, like the reference
        if (filts[i].kind == 0) {
            int64_t v = strtoll(filts[i].val.c_str(), nullptr, 10);
            // a time filter is aligned down to the bucket in a time-series query (filter.go:86-95)
            if (on("time") && filts[i].col == f["time-col"] && time_bucket > 0) v = v / time_bucket * time_bucket;
            cf[i].int_value = v;
        } else {
            cf[i].str_value = filts[i].val.c_str();
        }
    }
    std::vector<const char *> gptr, aptr;
    for (auto &g : groups) gptr.push_back(g.c_str());
    for (auto &a : ints) aptr.push_back(a.c_str());
    sybl_query_desc d;
    memset(&d, 0, sizeof(d));
    d.n_filters = (int32_t)cf.size();
    d.filters = cf.empty() ? nullptr : cf.data();
    d.n_groups = (int32_t)gptr.size();
    d.groups = gptr.empty() ? nullptr : gptr.data();
    d.n_aggs = (int32_t)aptr.size();
    d.aggs = aptr.empty() ? nullptr : aptr.data();
    d.op = f["op"] == "hist" ? SYBL_AGG_HIST : SYBL_AGG_AVG;
    d.hist_bucket = atoll(f["int-bucket"].c_str());
    d.want_percentiles = 1;  // both printers output percentiles in hist mode
    if (on("time")) {
        d.time_col = f["time-col"].c_str();
        d.time_bucket = time_bucket;
    }
    d.weight_col = f["weight-col"].empty() ? nullptr : f["weight-col"].c_str();
    d.order_by = f["sort"].c_str();
    d.order_asc = on("sort-asc");
    d.limit = atoi(f["limit"].c_str());
    d.block_skip = on("block-skip");
    d.loghist = on("loghist");  // FLAGS.LOG_HIST (cmd_query.go:43)
    // a printer looks at `limit` rows and Cumulative (printer.go:291-308); -encode-results ships every Result whole (:263-289)
    // (2: ... and the rows beyond -limit need nothing but their Count -- with -sort $COUNT the library pushes the limit into the scan)
    d.printed_only = on("encode-results") ? 0 : 2;
    // -str-replace col:find:replace[,col:find:replace...]  (cmd_query.go:51, table_query.go:33-46: split on ':' whatever the
    // filter separator is; fewer than three tokens = ignored)
    std::vector<std::vector<std::string>> sr_tok;
    for (auto &spec : split(f["str-replace"], fs)) {
        std::vector<std::string> tok = split(spec, ":");
        if (tok.size() > 2) sr_tok.push_back(tok);
    }
    std::vector<sybl_str_replace> sr(sr_tok.size());
    for (size_t i = 0; i < sr_tok.size(); i++) {
        memset(&sr[i], 0, sizeof(sybl_str_replace));
        sr[i].col = sr_tok[i][0].c_str();
        sr[i].pattern = sr_tok[i][1].c_str();
        sr[i].replace = sr_tok[i][2].c_str();
    }
    d.n_str_replace = (int32_t)sr.size();
    d.str_replace = sr.empty() ? nullptr : sr.data();
    std::vector<const char *> dptr;
    for (auto &c : distincts) dptr.push_back(c.c_str());
    d.n_distincts = (int32_t)dptr.size();
    d.distincts = dptr.empty() ? nullptr : dptr.data();

    ph.mark("agree");
    sybl_query *q = nullptr;
    if (sybl_query_prepare(tab, &d, &q)) return die("prepare");
    ph.mark("prepare");
    if (sybl_query_scan(q)) return die("scan");
    if (nranks > 1 && sybl_query_allreduce(q)) return die("allreduce");
    // (every rank finalizes: after a reduce-scatter or a printer's merge the finalize is itself collective --
    // sybl_query_collective_finalize -- and a rank that only merged has nothing else left to do)
    sybl_result *res = nullptr;
    if (sybl_query_finalize(q, &res)) return die("finalize");
    ph.mark("scan+finalize");
    if (rank != 0) {
        // rank 0 prints
    } else if (on("encode-results")) {
        // PrintResults: -encode-results wins over -print (printer.go:291-297); gob NodeResults on stdout
        int64_t n = 0;
        const void *bytes = sybl_result_encode(res, &n);
        if (!bytes) return die("encode");
        fwrite(bytes, 1, (size_t)n, stdout);
    } else if (on("print")) {
        const char *out = sybl_result_render(res, on("json") ? 1 : 0);
        if (!out) return die("render");
        fputs(out, stdout);
    }
    ph.mark("print");
    // A one-shot CLI has nothing to give back that the process exit does not: the result is printed -- leave.  Freeing 1-2 GB of
    // HBM buffer by buffer, unpinning the loader's arena and tearing the HIP runtime down cost a cold query ~0.1 s of its
    // ~0.7 (profiles/r06_cold_cli_phases.txt); the reference's `sybil query` frees nothing either.  -stats and the multi-rank
    // runs (the communicator is torn down in step with the other ranks) take the orderly way out.
    if (!on("stats") && nranks == 1 && !getenv("SYBIL_GPU_ORDERLY_EXIT")) {
        fflush(stdout);
        fflush(stderr);
        _exit(0);
    }
    if (on("stats")) {
        sybl_run_stats st;
        sybl_query_stats(q, &st);
        if (nranks > 1) fprintf(stderr, "rank %d/%d: ", rank, nranks);
        fprintf(stderr, "rows=%lld blocks=%lld skipped=%lld scan_ms=%.3f GB/s=%.1f strategy=%d\n", (long long)st.rows_scanned,
                (long long)st.blocks_scanned, (long long)st.blocks_skipped, st.scan_ms,
                st.scan_ms > 0 ? st.algorithmic_bytes / (st.scan_ms * 1e-3) / 1e9 : 0.0, st.strategy);
    }
    sybl_result_free(res);
    sybl_query_free(q);
    sybl_table_free(tab);
    if (nranks > 1) sybl_comm_free(ctx);
    sybl_shutdown(ctx);
    if (on("stats")) {
        ph.mark("free");
        fprintf(stderr, "phases (since main):%s total=%.1fms\n", ph.line.c_str(), (now_s() - ph.t0) * 1e3);
    }
    return 0;
}
