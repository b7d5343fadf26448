// engine.h -- host-side objects behind the C ABI (include/sybilgpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/sybilgpu.h"
#include "plan.h"
#include "scan_fast.h"
#include "scan_packed.h"

namespace sybl {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
#define SYBL_HIP(expr)                                         \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);    \
    } while (0)

// kernels.hip
hipError_t launch_scan(const ScanPlan *d_plan, int n_slots, int n_wg, bool use_lds, size_t lds_bytes, hipStream_t st);
hipError_t launch_fold(const int64_t *ws_sum, int64_t *out_sum, int64_t words_sum, const int64_t *ws_max, int64_t *out_max,
                       int64_t words_max, int n_wg, hipStream_t st);
hipError_t launch_fill64(int64_t *p, int64_t n, int64_t v, hipStream_t st);
hipError_t launch_synth(int64_t *out, int64_t n, int64_t row0, int64_t total_rows, int kind, int64_t a, int64_t b,
                        uint64_t col_seed, hipStream_t st);
hipError_t launch_block_minmax(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *blocks, int n_blocks,
                               int64_t *out_min, int64_t *out_max, int64_t *out_pop, hipStream_t st);
hipError_t launch_repack(const void *src, int sw, int64_t sbase, void *dst, int dw, int64_t dbase, int64_t n, hipStream_t st);

hipError_t launch_distinct(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *blocks, int n_blocks,
                           int64_t *keys, uint32_t mask, unsigned long long *n_distinct, unsigned long long limit, hipStream_t st);
hipError_t launch_pack32(const int64_t *src, int32_t *dst, int64_t n, hipStream_t st);
hipError_t launch_outlog_gather(const int64_t *stage, int64_t cap, int64_t *log, int64_t *header, hipStream_t st);
hipError_t launch_unpack32(const int32_t *src, int64_t *dst, int64_t n, hipStream_t st);
hipError_t launch_decode_bins(const void *recs, int rec_width, const int64_t *bin_off, const int64_t *bin_val, int n_bins,
                              bool delta_encoded, void *col, int out_width, int64_t vbase, uint32_t *valid, uint32_t nrows, hipStream_t st);
hipError_t launch_decode_delta(const void *deltas, int val_width, int64_t n, bool value_encoded, void *col, int out_width, int64_t vbase,
                               hipStream_t st);
hipError_t launch_remap_ids(const void *local, int local_width, const int32_t *lut, int32_t n_lut, int64_t n, int32_t *col, hipStream_t st);
// one block's bucket-encoded / value-encoded columns, a launch each (loader.cpp: DecodeBatches); arguments as above
hipError_t launch_decode_bins_multi(const DecodeBinsBatch &B, hipStream_t st);
hipError_t launch_decode_delta_multi(const DecodeDeltaBatch &B, const GobBinsBatch &G, hipStream_t st);  // (G: gob_bins.h's jobs ride along)
hipError_t launch_gob_values(const GobValuesBatch &B, hipStream_t st);  // gobgpu.hip: the varint walk of int column files ...

hipError_t launch_hist_summary(const HistSummaryPlan &S, int64_t *total, hipStream_t st);
hipError_t launch_rank_column(const void *col, int width, int64_t vbase, const uint32_t *valid, const int64_t *dkeys, const int32_t *dranks, uint32_t dmask,
                              int64_t n, void *out, int ow, uint32_t miss, hipStream_t st);
hipError_t launch_weight_carry(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *d_blocks, int n_blocks, int64_t *out, hipStream_t st);
hipError_t launch_hist_total(const int64_t *H, int64_t hist_stride, int64_t cell0, int64_t cell1, int64_t *total, hipStream_t st);
hipError_t launch_hist_gather(const int64_t *H, int64_t hist_stride, const int64_t *d_cells, int64_t n, int64_t cell0, int64_t cell1,
                              int64_t *out, hipStream_t st);

hipError_t launch_copy_digest(const void *p, int64_t n_words, unsigned long long *out, hipStream_t st);  // SYBL_VERIFY_COPIES
hipError_t create_side_stream(hipStream_t *out, int toward);  // engine.cpp: a stream on a priority level (and so hardware queues) of its own

struct Ctx {
    // Every sybl_* entry point that takes a handle of this ctx (the ctx, its tables, queries and results) holds this lock
    // for the length of the call (SYBL_API_GUARD): calls on one ctx from arbitrary OS threads -- 16 goroutines in the
    // reference, table_query.go:110,230-231 -- serialise HERE, not in the caller.  Recursive: entry points call each
    // other.  Shared: a result may outlive its ctx, and sybl_shutdown holds the lock while it deletes the ctx.
    std::shared_ptr<std::recursive_mutex> api_m = std::make_shared<std::recursive_mutex>();
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in force (own or the host framework's)
    int n_cus = 0;
    int64_t hbm_bytes = 0;
    std::string dev_name;
    void *comm = nullptr;  // ncclComm_t (rccl.cpp)
    int comm_rank = 0, comm_nranks = 1;
    hipStream_t aux_stream = nullptr;  // small finalize-side copies that must not queue behind another query's scan
    hipStream_t copy_stream = nullptr; // big snapshots (result.cpp: query_snapshot); not the aux stream: a finalize's gather
                                       // must not queue behind the NEXT query's snapshot, which waits for that query's scan
    // Table loads spread consecutive blocks over several streams (loader.cpp): a block's decode kernels are a handful of
    // tiny launches, and on one stream the GPU ran them strictly one after another.  While load_multi is set,
    // `stream` is one of load_streams; code that touches state shared between blocks (a column's staging block, a
    // reallocation) calls load_sync_all first.
    static constexpr int kMaxLoadStreams = 32;
    hipStream_t load_streams[kMaxLoadStreams] = {};
    int n_load_streams = 0;
    bool load_multi = false;
    // a load in progress keeps the decode launches of a few consecutive blocks back to issue them together: whoever is about to
    // wait for the load's streams (load_sync_all -- a column about to be reallocated, a widening repack) has them issued first
    std::function<int()> load_flush;
    // the loader's staging arena (loader.cpp: SlabPool), kept between loads: pinning and unpinning a few hundred MB cost
    // every sybl_table_open / sybl_table_refresh tens of milliseconds.  Freed by sybl_shutdown (SYBL_LOADER_KEEP_ARENA=0:
    // by the load that allocated it).
    // (round 6: it grows in CHUNKS of a few slabs, each pinned when the load first needs it -- a cold open used to pin all
    // ~280 MB before its first worker had a slab to parse into: ~0.1 s of a 0.28 s open in a fresh process)
    struct ArenaChunk {
        char *h = nullptr, *d = nullptr;
        size_t slabs = 0;
    };
    std::vector<ArenaChunk> load_chunks;
    size_t load_scratch_bytes = 0;  // device-only bytes behind every slab's twin
    size_t load_slab_bytes = 0;  // bytes per slab the chunks were carved for (another size: the arena starts over)
    // host_to_device (table.cpp): the pinned staging buffer every copy of CALLER memory -- sybl_table_append_block's
    // columns, dictionaries, look-up tables -- goes through (round 6; they were pageable hipMemcpyAsync before), and the
    // device word SYBL_VERIFY_COPIES digests into
    char *h2d_stage = nullptr;
    size_t h2d_stage_bytes = 0;
    unsigned long long *d_copy_digest = nullptr;
};
// dst (device) <- src (any host memory), `bytes` a multiple of 4, on the ctx stream, complete on return; under
// SYBL_VERIFY_COPIES=1 the bytes in HBM are digested and compared with the host's: SYBL_E_NODEVICE when they differ
int host_to_device(Ctx *ctx, void *dst, const void *src, size_t bytes, const char *what);
void ctx_free_load_arena(Ctx *ctx);
int load_sync_all(Ctx *ctx);  // waits for every load stream (no-op outside a multi-stream load)

struct Column {
    std::string name;
    int type = SYBL_INT_VAL;
    int elem = 8;    // bytes per stored value: canonical (8 = int64 values, 4 = int32 dictionary ids) or,
                     // after sybl_table_compact, the narrowest of 1/2/4 that holds max - min
    int64_t vbase = 0;  // value = vbase + zero-extended stored bits (0 for canonical storage)
    int canon() const { return type == SYBL_INT_VAL ? 8 : 4; }
    bool packed() const { return elem != canon() || vbase != 0; }
    bool info_given = false;
    int64_t info_min = 0, info_max = 0;
    void *d_data = nullptr;
    int64_t cap_rows = 0;
    uint32_t *d_valid = nullptr;  // bit per physical row; nullptr = fully populated
    int64_t valid_cap_words = 0;
    bool has_missing = false;
    // per-block statistics (exact, over populated rows)
    std::vector<int64_t> blk_min, blk_max, blk_pop;
    int64_t stats_blocks = 0;  // blocks covered by blk_*
    int64_t exact_min = INT64_MAX, exact_max = INT64_MIN, n_pop = 0;
    // bounds declared by a multi-rank host (global over all ranks)
    bool bounds_set = false;
    int64_t bound_lo = 0, bound_hi = 0;
    // group dictionary for sparse key ranges: sorted distinct values + device value->rank map
    std::vector<int64_t> gdict;        // sorted distinct values (host)
    int64_t gdict_blocks = -1;         // blocks covered when it was built (-1: none); set by the host = -2
    int64_t *d_gdict_keys = nullptr;
    int32_t *d_gdict_ranks = nullptr;
    uint32_t gdict_mask = 0;
    bool gdict_refused = false;        // sybl_table_agree: some rank holds more distinct values than a dictionary may -- every rank's planner hashes
    int64_t gdict_gen = 0;             // bumped whenever gdict / its device map change (column_install_gdict)
    // The column as RANKS in its group dictionary (table.cpp: column_build_rank): a narrow derived column that a group-by on
    // a sparse int key direct-maps through -- rank_col->d_valid is BORROWED from this column.  Valid for (gdict_gen, Table::version).
    std::unique_ptr<Column> rank_col;
    int64_t rank_gen = -1, rank_version = -1;
    // A weight column with unpopulated rows: the weights IN FORCE, row by row, as a dense int64 column (k_weight_carry;
    // planner.cpp: Planner::weight), shared by the prepared queries that weigh by this column and valid for one Table::version
    // (round 6: every prepare used to build its own -- a pass over the column and 8 B/row of HBM per weighted query).
    std::shared_ptr<Column> carried_weight;
    int64_t carried_version = -1;
    // table-global dictionary (str / set)
    std::vector<std::string> dict;
    std::unordered_map<std::string, int32_t> dict_ix;
    std::vector<const char *> dict_view;  // sybl_table_column_dict
    // set columns: CSR over physical rows
    int64_t *d_set_off = nullptr;
    int32_t *d_set_vals = nullptr;
    int64_t set_vals_cap = 0, set_vals_n = 0;
    std::vector<int64_t> h_set_off;  // host mirror: CSR offsets per physical row (+1)
    std::vector<int32_t> h_set_vals; // host mirror: member ids (table-global)
    bool set_dirty = false;          // host mirror newer than the device copy
    // compact mode: writers fill this canonical-width staging block, block_commit packs it in place
    void *d_stage = nullptr;
    int64_t stage_cap = 0;           // bytes
};

// a block directory the loader has seen (sybl_table_open / sybl_table_refresh)
struct LoadedBlock {
    std::string name;
    int64_t mtime_ns = 0, size = 0;  // of <block>/info.db when it was read
    int64_t index = -1;              // into Table::blocks; -1: skipped as broken / unreadable
};

struct Query;
struct Table {
    Ctx *ctx = nullptr;
    std::string name;
    std::vector<std::unique_ptr<Column>> cols;
    std::map<std::string, int> col_ix;
    int64_t phys_rows = 0;     // physical rows incl. per-block padding to 32
    int64_t reserve_hint_rows = 0;  // a loader's estimate of the rows to come: the first growth of a column goes straight there
    int64_t logical_rows = 0;
    std::vector<Segment> blocks;  // physical start / logical row count
    Segment *d_blocks = nullptr;
    int64_t d_blocks_n = 0;
    bool compact_mode = false;  // sybl_table_compact was called: appended blocks are packed in place
    int64_t *d_scratch = nullptr;  // one Segment + min/max/pop per staged column of a block
    int64_t scratch_words = 0;
    int64_t version = 0;        // bumped by every change a prepared query would not know about
    int64_t broken_blocks = 0;  // blocks the loader skipped (unreadable info / column unpack error)
    sybl_load_stats load_stats{};  // of the sybl_table_open / sybl_table_refresh that last loaded blocks
    std::string src_dir;           // <dir>/<table> the table was opened from ("" = built through the ABI)
    int src_rank = 0, src_nranks = 1;
    std::vector<LoadedBlock> loaded;
    std::vector<Query *> queries;  // prepared queries that are alive (a table that goes away first has their pending lazy results built)
    Column *find(const char *name) const;
};

int table_ensure_stats(Table *t);
int table_reserve(Table *t, Column *c, int64_t phys_rows);
int valid_reserve(Table *t, Column *c, int64_t phys_rows);
int64_t table_drop_dead_tail(Table *t, int64_t keep_blocks);  // trailing blocks without rows give their row range back
int table_upload_blocks(Table *t);                          // host segments -> t->d_blocks (before k_block_minmax / k_distinct)
int table_reclaim_dead_rows(Table *t, bool force);           // rows of dead blocks in the MIDDLE of the table: the live blocks close up
void column_free(Column *c);
int32_t dict_intern(Column *c, const std::string &s);
int column_upload_set(Table *t, Column *c);
int column_build_gdict(Table *t, Column *c);           // distinct values of the resident rows
int column_build_rank(Table *t, Column *c);            // Column::rank_col for the current dictionary and table version
int column_install_gdict(Table *t, Column *c);         // sorted gdict -> device value->rank map
int column_repack(Table *t, Column *c, int width, int64_t vbase);  // change the stored width in place

struct BlockWriter {
    Table *t = nullptr;
    int64_t start = 0, nrows = 0, new_phys = 0;
    // compact mode: columns whose block sits in Column::d_stage, with the block's extrema when the writer
    // already knows them (block_col_stats) -- else block_commit computes them on the GPU and waits
    struct Staged {
        Column *c = nullptr;
        bool have_stats = false;
        int64_t mn = 0, mx = 0, pop = 0;
    };
    std::vector<Staged> staged;
    std::vector<Staged> direct;  // columns written in their stored form at the block's final place (block_col_direct)
    bool serial = false;         // multi-stream load: this block used state shared between blocks; its stream is drained at commit
};
int block_begin(Table *t, int64_t nrows, BlockWriter *w);
int block_col_device(BlockWriter &w, Column *c, bool all_populated, void **col, uint32_t **valid);
// Compact mode, block extrema known up front (mn / mx over the pop populated rows): when the column's current (width,
// base) holds them, *ok = true and the writer produces the STORED form (Column::elem bytes per row, offsets from
// Column::vbase) at *col, the block's final place -- no staging block, no k_repack.
int block_col_direct(BlockWriter &w, Column *c, bool all_populated, int64_t mn, int64_t mx, int64_t pop, void **col, uint32_t **valid, bool *ok);
int block_col_absent(BlockWriter &w, Column *c);
void block_col_stats(BlockWriter &w, Column *c, int64_t mn, int64_t mx, int64_t pop);  // min / max over the pop populated rows
int block_col_int_host(BlockWriter &w, Column *c, const int64_t *vals, const uint8_t *populated);
int block_col_str_host(BlockWriter &w, Column *c, const int32_t *global_ids, const uint8_t *populated);
int block_col_set_host(BlockWriter &w, Column *c, const int64_t *off, const int32_t *global_ids, const uint8_t *populated);
int block_commit(BlockWriter &w);

// pinned host memory owned jointly by a query and the results that point into it
// CPUs this process can actually use: hardware threads, capped by a cgroup v2 / v1 CPU quota
inline size_t usable_cpus() {
    size_t n = std::max<unsigned>(1, std::thread::hardware_concurrency());
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f2, "%lld", &period) != 1) period = 0;
            fclose(f2);
        }
    }
    if (quota > 0 && period > 0) n = std::min<size_t>(n, (size_t)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

// SYBL_FINALIZE_TRACE=1: per-phase host timings of scan / snapshot / finalize on stderr
struct PhaseTrace {
    const char *what_ = "finalize";
    PhaseTrace() {}
    explicit PhaseTrace(const char *w) : what_(w) {}
    bool on = env("SYBL_FINALIZE_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string line;
    void mark(const char *what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        char b[64];
        snprintf(b, sizeof(b), " %s=%.1fus", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        line += b;
        t0 = t1;
    }
    ~PhaseTrace() {
        if (on) fprintf(stderr, "%s:%s\n", what_, line.c_str());
    }
};

// a device allocation that is freed on every exit unless ownership was passed on (release())
struct DevOwner {
    void *p = nullptr;
    ~DevOwner() {
        if (p) (void)hipFree(p);
    }
    template <typename T>
    T *release() {
        T *x = (T *)p;
        p = nullptr;
        return x;
    }
};

struct HostBuf {
    int64_t *p = nullptr;
    int64_t words = 0;
    std::atomic<int> pins{0};  // results whose rows point into this buffer (HostPin): the query leaves it alone while > 0
    ~HostBuf() {
        if (p) (void)hipHostFree(p);
    }
};

// A result's hold on a pinned snapshot buffer.  The query that owns the buffer decides whether it may write the next
// snapshot into it from HostBuf::pins -- an explicit count of these holders -- not from shared_ptr::use_count(), which
// also counts the query's own list, its current-buffer pointers and any temporary.
struct HostPin {
    std::shared_ptr<HostBuf> b;
    HostPin() = default;
    HostPin(const HostPin &o) : b(o.b) {
        if (b) b->pins.fetch_add(1);
    }
    HostPin &operator=(const std::shared_ptr<HostBuf> &nb) {
        if (nb) nb->pins.fetch_add(1);
        if (b) b->pins.fetch_sub(1);
        b = nb;
        return *this;
    }
    HostPin &operator=(const HostPin &o) { return *this = o.b; }
    ~HostPin() {
        if (b) b->pins.fetch_sub(1);
    }
    HostBuf *operator->() const { return b.get(); }
    explicit operator bool() const { return (bool)b; }
};

// -str-replace on one str column: dictionary id -> id of the rewritten string (first id that rewrites to it)
struct StrReplaced {
    std::vector<std::string> strs;  // [new id]
    std::vector<int32_t> remap;     // [table-global dictionary id] -> new id
    int64_t *d_keys = nullptr;      // device map old id -> new id (the kSlotDict layout)
    int32_t *d_ranks = nullptr;
    uint32_t mask = 0;
};

struct GroupInfo {
    int col;
    int type;
    int64_t gmin;
    int64_t gcard;       // digits of this key column incl. a separate MISSING digit, if any
    int64_t value_card;  // digits that are real values
    int64_t missing_digit;  // digit missing rows map to (-1: column has no missing rows)
    bool dict = false;      // digits are ranks in the column's sorted distinct values (Column::gdict)
    bool rank = false;      // ... read from the column's derived rank column (Column::rank_col), not probed per row
    int slot = -1;          // the slot the scan reads the key digit from
    const StrReplaced *replaced = nullptr;  // -str-replace: digits are ids of the rewritten strings
    bool has_missing;
};

struct AggInfo {
    int col;
    std::string name;
    AggDesc d;
    int32_t f_out;  // base of 6 outlier fields (n, sum, sq limb0..3) or -1
    int64_t num_buckets;
    int64_t info_max;
    std::vector<sybl_subhist> subs;  // -loghist: the sub-histograms (layout of the aggregation's bucket words)
};

struct Result;
struct ResultPool;  // result.cpp: result arrays recycled between the finalizes of one query

struct Query {
    Table *t = nullptr;
    Ctx *ctx = nullptr;
    int64_t table_version = 0;  // Table::version at prepare time
    // copied descriptor
    int op = SYBL_AGG_AVG;
    int64_t hist_bucket = 0;
    bool want_percentiles = false;
    bool weighted = false;
    bool loghist = false;          // FLAGS.LOG_HIST: MultiHist (hist_multi.go)
    std::map<int, std::unique_ptr<StrReplaced>> replaced;  // -str-replace, by table column index
    std::vector<MultiSub> h_multi; // every aggregation's sub-histograms, as the kernels see them
    MultiSub *d_multi = nullptr;
    // count distinct (hll.h, distinct.hip): a pass of its own after the scan, reading dplan = the scan's plan + the
    // distinct columns' slots
    int n_distinct = 0;
    ScanPlan dplan;
    ScanPlan *d_dplan = nullptr;
    uint8_t *d_hll = nullptr;          // [n_cells][kHllRegs]
    uint64_t *d_hll_idhash = nullptr;  // one str column: hash per dictionary id
    char *d_hll_chars = nullptr;       // several columns, a str column among them: the dictionaries' strings + offsets (plan.h: hll_mixed)
    int64_t *d_hll_stroff = nullptr;
    int64_t hll_bytes = 0;
    bool distinct_pending = false;     // hashed group-by: the sketch pass has yet to run over the final key set (query_hash_distinct)
    bool time_mode = false;
    int64_t time_bucket = 0;
    std::string order_by;
    bool order_asc = false;
    int limit = 0;
    // a weight column with unpopulated rows: the weight in force at every row (k_weight_carry), a dense int64 column of this
    // query's own that the scan reads in the weight column's place (planner.cpp: Planner::weight)
    std::shared_ptr<Column> eff_weight;  // (= the weight column's Column::carried_weight)
    int printed_level = 0;      // sybl_query_desc.printed_only as given (2: the rows beyond the limit may carry their Count alone)
    bool printed_only = false;  // sybl_query_desc.printed_only: percentiles / stddev / bucket arrays for the printed rows + Cumulative only
    bool top_only = false;      // ... and this query is one it applies to (query_snapshot: summary shape, limit > 0)
    bool top_merge = false;     // ... across ranks: the bucket table stayed rank-local, finalize sums the printed rows' arrays (rccl.cpp)
    std::vector<GroupInfo> groups;
    std::vector<AggInfo> aggs;
    // plan
    ScanPlan plan;
    ScanPlan *d_plan = nullptr;
    bool plan_dirty = true;
    int n_wg = 0;
    bool use_lds = false;
    size_t lds_bytes = 0;
    int64_t group_cells = 0;
    std::vector<Segment> segs;
    std::vector<int32_t> wg_seg_begin;
    Segment *d_segs = nullptr;
    int32_t *d_wg_seg_begin = nullptr;
    int32_t *d_wg_cell_base = nullptr;
    std::vector<void *> d_idmasks;
    int64_t n_sum_words = 0, n_max_words = 0;
    int64_t *d_sum = nullptr, *d_max = nullptr;
    bool own_partials = false;
    int64_t *d_ws_sum = nullptr, *d_ws_max = nullptr;
    std::shared_ptr<ResultPool> rpool;
    std::shared_ptr<HostBuf> h_sum_buf;          // pinned snapshot of the SUM section (shared with results)
    int64_t *h_sum = nullptr, *h_max = nullptr;  // h_sum = h_sum_buf->p; h_max: pinned staging
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // GPU-side histogram summaries (percentiles, bucket moments, Cumulative buckets) for queries with
    // many cells: only they, not the bucket arrays, cross PCIe unless every row's buckets are wanted
    bool hist_summary = false;
    bool snap_has_buckets = true;   // the last snapshot carried the bucket arrays
    int64_t *d_pct = nullptr, *d_mom = nullptr, *d_total = nullptr;
    std::shared_ptr<HostBuf> h_pct_buf;          // pinned snapshot of the GPU-computed percentiles (shared with results)
    std::shared_ptr<HostBuf> h_spare_buf;        // (held only while query_snapshot takes its first two percentile buffers)
    std::vector<std::shared_ptr<HostBuf>> host_bufs;  // pinned snapshot buffers of this query, reused once no result holds them
    int64_t *h_pct = nullptr, *h_mom = nullptr, *h_total = nullptr;  // pinned (h_pct = h_pct_buf->p)
    std::shared_ptr<struct KeyStore> key_cache;  // the group cells' BinaryByKey / GroupByKey (result.h), built by the first finalize
    int64_t *d_top_cells = nullptr, *d_top = nullptr;               // bucket arrays of the printed rows
    int64_t top_cap = 0;
    hipEvent_t ev_snap = nullptr;   // the device -> host snapshot of the partial tables has landed
    int64_t *h_top = nullptr;       // pinned: the printed rows' bucket arrays (result.cpp: attach_top_values)
    int64_t h_top_words = 0;
    hipEvent_t ev_ready = nullptr;  // the tables are ready to be copied (big snapshots leave through Ctx::aux_stream)
    bool snap_on_aux = false;       // the last snapshot was queued on the auxiliary stream: a rescan must wait for it
    bool snapshot_pending = false;
    // multi-GPU merge of big bucket tables (rccl.cpp): the bucket arrays were reduce-SCATTERED over cell ranges, this
    // rank holds the reduced arrays of cells [rs_cell0, rs_cell1) only; percentiles / moments are derived per slice
    // and all-gathered.  snapshot and finalize are then collective calls (every rank makes them).
    bool hash_fast = false;     // hash group-by through k_scan_hash_fast (fplan, fast_nf / ng / na / mode)
    bool hash_packed = false;   // ... through k_scan_hash_packed (compact storage, 32-bit composite key)
    bool rs_active = false;
    int rs_int32 = -1;          // -1: not decided yet (first collective of the query), 0 / 1: the bucket slices travel as int64 / int32
    int32_t *d_h32 = nullptr;   // int32 staging of the bucket table + room for the reduced slice
    int64_t rs_cells_per = 0, rs_cell0 = 0, rs_cell1 = 0;
    // outlier log (plan.h): values of the outliers / underliers of queries that keep bucket arrays
    int64_t *d_out_log = nullptr;    // the dense outlier log (k_outlog_gather), read by finalize and the multi-rank merge
    int64_t *d_out_stage = nullptr;  // what the scan kernels append to: cursors + stripes (plan.h)
    int64_t out_cap = 0;
    bool out_log_partial = false;  // the partial tables were merged across ranks: the log only holds this rank's values
    bool scanned = false;
    bool layout_checked = false;   // the ranks compared their partial-table layouts (agree.cpp: query_check_layout, first collective of the query)
    sybl_run_stats stats{};
    bool never_matches = false;
    // hash group-by (strategy 7, hashgroup.hip): the cell table is an open-addressing table over the composite key; after
    // the scan the live slots are compacted into dense arrays in key order (query_hash_compact), which is what the
    // all-reduce and finalize see
    bool hash_mode = false;
    uint64_t *d_hash_keys = nullptr;        // [n_cells] slot -> composite key
    bool hash_compacted = false;
    int64_t hash_live = 0;                  // keys of the dense form (live slots; after a union install: the union)
    uint64_t *d_pair_keys = nullptr;        // (key, slot) of the live slots, unsorted
    uint32_t *d_pair_slots = nullptr;
    uint64_t *d_dense_keys = nullptr;       // sorted composite keys
    uint32_t *d_dense_slots = nullptr;      // their slots
    int64_t *d_dense_sum = nullptr;         // [header][F][hash_live], then [hash_live][hist_stride]
    int64_t *d_dense_max = nullptr;         // [M][hash_live]
    int64_t pair_cap = 0, pair_slots_cap = 0, dense_keys_cap = 0, dense_slots_cap = 0, dense_sum_cap = 0, dense_max_cap = 0;
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint64_t *d_hash_count = nullptr;
    std::shared_ptr<HostBuf> h_keys_buf;    // pinned host copy of the sorted keys (finalize, multi-rank union), shared with results
    uint64_t *h_dense_keys = nullptr;       // = h_keys_buf->p
    // filter pre-pass (planner.cpp: Planner::prefilter): the filters the packed bodies cannot evaluate run first, as a
    // kernel of their own that writes a row bitmap the scan then reads like a validity word
    ScanPlan preplan;
    ScanPlan *d_preplan = nullptr;
    uint32_t *d_prebits = nullptr;
    int pre_n_slots = 0;            // slots of the pre-pass in all
    int pre_generic_slots = 0;      // ... of them evaluated by the generic k_prefilter (set members, 8-byte columns): preplan's slots
    std::vector<FastPlan> pre_fps;  // ... the others: <= kFastMaxF filter columns per launch of k_prefilter_packed
    std::vector<int> pre_fp_nf;
    bool table_gone = false;       // sybl_table_free ran before sybl_query_free: nothing of q->t may be touched any more
    std::vector<Result *> lazy_results;     // results whose rows are still to be built and need this query for it (result.cpp)
    int64_t h_max_words = 0;                // capacity of h_max
    // role-specialised kernel (scan_fast.h)
    bool fast = false, fast_gen = false, fast_packed = false;
    bool fast_packed_n = false;  // ... the run-time-column-count form of the packed kernel (3-4 group columns)
    int fast_nf = 0, fast_ng = 0, fast_na = 0, fast_mode = 0;
    FastPlan fplan;
    // partitioned histograms (strategy 5)
    bool part_hist = false, part_packed = false;
    int part_nf = 0, part_ng = 0, part_na = 0;
    EmitPlan eplan;
    PartHistPlan pplan;
    // a query with three or four aggregations: the kernels of strategy 5 are instantiated for one or two, so aggregations
    // 2.. go through the same count -> emit -> k_part_hist sequence again (same rows, same record buffers, their own
    // bucket arrays and sum fields of the one cell table)
    struct PartPass {
        EmitPlan E;
        PartHistPlan H;
        int na = 0;
        bool packed = false;
    };
    std::vector<PartPass> part_more;
    uint32_t *d_recs = nullptr, *d_cursor = nullptr;
    // -limit pushed into the scan (pushdown.hip): sybl_query_desc.printed_only = 2 on a query strategy 5 would take, one
    // direct-mapped key of <= 65536 cells, no filter, sorted by $COUNT descending, one GPU
    bool pushdown = false;
    bool pushdown_ran = false;  // the last scan went through it (a query that matches nothing, an empty table: the ordinary zeroed tables)
    PushdownPlan dplan_pd;
    uint32_t *d_pd = nullptr;  // ws | carry | cnt | bitmap | top_cells | n_top in one allocation
    bool pd_static = false;  // several ranks, and the query alone does not rule the pushdown out: the ranks agree at the first scan ...
    int pd_agreed = -1;      // ... 1: every rank planned it, 0: some rank did not (none takes it), -1: not asked yet
    int64_t *d_pd_max = nullptr;  // across ranks: the header's Cumulative maxima set aside for their MAX all-reduce (rccl.cpp)
    bool count_cached = false;  // d_cursor holds the count pass's regions for this query's rows (engine.cpp: a rescan skips k_count)
};

int plan_query(Table *t, const sybl_query_desc *d, Query *q);  // planner.cpp
int query_rescan_without_part_hist(Query *q);

constexpr int kMaxScatterRanks = 64;  // the SUM section is padded so that a reduce-scatter over up to this many ranks fits in place
bool query_wants_hist_summary(const Query *q);
int query_acquire_host_buf(Query *q, int64_t words, std::shared_ptr<HostBuf> &cur);  // (result.cpp) a pinned buffer no result holds
int query_host_keys(Query *q, int64_t n);  // (result.cpp) q->h_dense_keys with room for n keys, not shared with a live result
void query_finish_lazy_results(Query *q);  // (result.cpp) builds the rows of results that still need the query (before it goes away)
int query_summary_buffers(Query *q);  // (result.cpp) d_pct / d_mom / d_total + their pinned twins
int query_total_buffers(Query *q);    // (result.cpp) d_total + its pinned twin alone: a printer's query
// rccl.cpp: collectives on the ctx communicator and stream (SYBL_E_STATE without a communicator)
int comm_allgather_inplace(Ctx *ctx, int64_t *buf, size_t words_per_rank);
int comm_allreduce_sum(Ctx *ctx, int64_t *buf, size_t words);
int comm_all_agree(Ctx *ctx, bool mine, bool *all);  // (a blocking MIN over the ranks: rccl.cpp)
int comm_allreduce_u32_sum(Ctx *ctx, uint32_t *buf, size_t n);  // (the pushed-down scan's group counts between its passes)
// agree.cpp
bool group_key_wants_dict(unsigned __int128 card, int64_t cells);  // the planner's test, shared with sybl_table_agree
int query_check_layout(Query *q);                                  // collective; an error on EVERY rank when the layouts differ
// hashgroup.hip
hipError_t launch_scan_hash(const ScanPlan *d_plan, int n_slots, int n_wg, size_t lds_bytes, hipStream_t st);
hipError_t launch_scan_packed_n(const FastPlan &P, int nf, int ng, int na, int mode, bool time, int n_wg, size_t lds_bytes, hipStream_t st);
hipError_t launch_prefilter_packed(const FastPlan &P, int nf, uint32_t *bits, bool and_into, int n_wg, hipStream_t st);
hipError_t launch_scan_hash_packed(const FastPlan &P, uint64_t *keys, int nf, int ng, int na, int mode, bool time, int L, int F, int M, int n_wg,
                                   size_t lds_bytes, hipStream_t st);
hipError_t launch_scan_hash_fast(const FastPlan &P, uint64_t *keys, int nf, int ng, int na, int mode, bool time, int L, int F, int M, int n_wg,
                                 size_t lds_bytes, hipStream_t st);
// distinct.hip
hipError_t launch_scan_distinct(const ScanPlan *d_plan, int n_slots, int n_wg, hipStream_t st);
int query_hash_reset(Query *q);     // every slot free (before a scan)
int query_hash_compact(Query *q);   // live slots -> dense arrays in key order
int query_hash_distinct(Query *q);  // (engine.cpp) count distinct over a hashed group-by: the sketch pass, once the dense keys are final
int query_hash_install_union(Query *q, const uint64_t *keys, int64_t n);                // host keys
int query_hash_install_union_device(Query *q, const uint64_t *d_union, int64_t n);      // device keys
int hash_union_of_lists(Query *q, const uint64_t *d_lists, int64_t total, uint64_t **out, int64_t *n_out);
int64_t hash_dense_sum_words(const Query *q, int64_t n);
int64_t hash_dense_max_words(const Query *q, int64_t n);
void query_hash_free(Query *q);
int query_snapshot(Query *q);
int query_finalize(Query *q, Result **out);

}  // namespace sybl

// opaque C types are the C++ objects
struct sybl_ctx : sybl::Ctx {};
struct sybl_table : sybl::Table {};
struct sybl_query : sybl::Query {};

namespace sybl {
struct ApiGuard {
    std::shared_ptr<std::recursive_mutex> m;
    explicit ApiGuard(std::shared_ptr<std::recursive_mutex> mm) : m(std::move(mm)) {
        if (m) m->lock();
    }
    ~ApiGuard() {
        if (m) m->unlock();
    }
    ApiGuard(const ApiGuard &) = delete;
    ApiGuard &operator=(const ApiGuard &) = delete;
};
inline std::shared_ptr<std::recursive_mutex> api_mutex_of(const sybl_ctx *c) { return c ? c->api_m : nullptr; }
inline std::shared_ptr<std::recursive_mutex> api_mutex_of(const sybl_table *t) { return t && t->ctx ? t->ctx->api_m : nullptr; }
inline std::shared_ptr<std::recursive_mutex> api_mutex_of(const sybl_query *q) { return q && q->ctx ? q->ctx->api_m : nullptr; }
std::shared_ptr<std::recursive_mutex> api_mutex_of(const sybl_result *r);  // (result.cpp: the ctx's lock, kept alive by the result)
}  // namespace sybl
#define SYBL_API_GUARD(handle) sybl::ApiGuard api_guard__(sybl::api_mutex_of(handle))
