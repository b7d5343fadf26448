/*
 * sybilgpu.h -- C ABI of the MI355X-native scan/aggregate engine for logv/sybil.
 *
 * This is the drop-in boundary for sybil's query hot path.  The reference has no
 * FFI/plugin interface; the seam this library replaces is the body of the per-block
 * loop in  src/lib/table_query.go:111-220  (LoadBlockFromDir -> CopyQuerySpec ->
 * FilterAndAggRecords -> block_specs[...]) together with the merge that follows
 * (MultiCombineResults/CombineResults, table_query.go:230-257,397-404 and
 * aggregate.go:361-467).  One call level up, the whole of
 *     count = t.LoadAndQueryRecords(&loadSpec, &querySpec)   (src/cmd/cmd_query.go:362)
 * maps to  sybl_query_prepare + sybl_query_scan (+ all-reduce) + sybl_query_finalize.
 * INTEGRATION.md shows the cgo shim a sybil maintainer would add.
 *
 * Conventions:
 *  - plain C types only; every entry point returns int (0 = ok, <0 = SYBL_E_*),
 *    sybl_last_error() gives the text for the calling thread; no exceptions or abort()
 *    cross the boundary.
 *  - the library never keeps a caller pointer after a call returns (cgo rule); inputs
 *    are consumed during the call, outputs live in library-owned memory until the
 *    matching *_free.
 *  - no globals: everything the reference reads from FLAGS/OPTS (config.go:119-120)
 *    is an explicit field of sybl_query_desc.
 *  - one sybl_ctx per GPU (one process per GPU under RCCL).  Thread-safe for concurrent
 *    calls from arbitrary OS threads (the reference queries blocks from 16 goroutines,
 *    table_query.go:110,230-231): calls on different ctx objects run concurrently; calls
 *    that take a handle of ONE ctx -- the ctx, its tables, queries and results -- are
 *    serialised by the library (a per-ctx lock held for the length of each call), so a
 *    thread may read a result's rows while another scans or frees the query behind it.
 *    What stays the caller's business is ORDER, not exclusion: a handle must not be used
 *    after its *_free, and scan -> [allreduce] -> [snapshot] -> finalize of one query are
 *    issued in that order.
 *  - there is NO CPU fallback: without a usable HIP device every compute call fails
 *    with SYBL_E_NODEVICE.
 */
#ifndef SYBILGPU_H
#define SYBILGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYBL_ABI_VERSION 5

enum {
    SYBL_OK = 0,
    SYBL_E_INVAL = -1,    /* bad argument / unknown column / unsupported query shape */
    SYBL_E_NODEVICE = -2, /* no HIP device, or a HIP call failed */
    SYBL_E_NOMEM = -3,
    SYBL_E_IO = -4,       /* table directory / gob decode problem */
    SYBL_E_STATE = -5,    /* call order violated (e.g. finalize before scan) */
    SYBL_E_BLOCK = -6     /* a block was rejected (the reference skips such blocks) */
};

/* record.go:14-19 */
enum { SYBL_NO_VAL = 0, SYBL_INT_VAL = 1, SYBL_STR_VAL = 2, SYBL_SET_VAL = 3 };

/* filter ops: filter.go:176-190 (int), :213-245 (str), :268-283 (set) */
enum {
    SYBL_OP_GT = 0, SYBL_OP_LT = 1, SYBL_OP_EQ = 2, SYBL_OP_NEQ = 3,
    SYBL_OP_RE = 4, SYBL_OP_NRE = 5, SYBL_OP_IN = 6, SYBL_OP_NIN = 7
};

/* FLAGS.OP (aggregate.go:24-29) */
enum { SYBL_AGG_AVG = 0, SYBL_AGG_HIST = 1 };

#define SYBL_MAX_GROUPS 8
#define SYBL_MAX_AGGS 6
#define SYBL_MAX_FILTERS 16
#define SYBL_GROUP_BY_WIDTH 8 /* aggregate.go:16 */
#define SYBL_BLOCK_ROWS 65536 /* table.go:44 CHUNK_SIZE */

typedef struct sybl_ctx sybl_ctx;
typedef struct sybl_table sybl_table;
typedef struct sybl_query sybl_query;
typedef struct sybl_result sybl_result;

/* ------------------------------------------------------------------ context */

int sybl_abi_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char *sybl_last_error(void);

/* device: HIP device ordinal of this process' GPU (LOCAL_RANK under torchrun). */
int sybl_init(int device, sybl_ctx **out);
void sybl_shutdown(sybl_ctx *ctx);
/* Run all work of this ctx on an existing hipStream_t (e.g. the host framework's
 * current stream); NULL restores the ctx's own stream. */
int sybl_ctx_set_stream(sybl_ctx *ctx, void *hip_stream);
int sybl_ctx_sync(sybl_ctx *ctx);
/* Gives back what the ctx keeps between calls for speed and a long-running host may not want to pay for: the loader's
 * staging arena (up to ~512 MB of pinned host memory and as much HBM, kept from one sybl_table_open / sybl_table_refresh to
 * the next because pinning it costs every load tens of milliseconds).  A host that opens its tables once and then only
 * queries them calls this after the last open; the next load allocates the arena again.  ABI 4. */
int sybl_ctx_trim(sybl_ctx *ctx);
int sybl_device_info(sybl_ctx *ctx, char *name, size_t name_cap, int *n_cus, int64_t *hbm_bytes);

/* ------------------------------------------------------------------ tables
 * A table is the HBM-resident form of a sybil table: one dense array per column
 * (reference value types: IntField int64, StrField int32 dictionary id, SetField
 * []int32 -- record_fields.go:8-10), a validity bitmap for columns with missing
 * rows, and a table-global string dictionary per str/set column (the reference's
 * are block-local, table_column.go:27-48; SURVEY.md 8a note 8).                  */

int sybl_table_create(sybl_ctx *ctx, const char *name, sybl_table **out);
void sybl_table_free(sybl_table *t);

/* Declare a column before the first block is appended.  info_min/info_max are the
 * table-level IntInfo.Min/Max (table_column_info.go:18-24, loaded from info.db by
 * the host) that hist geometry and the outlier gate use (aggregate.go:254,
 * hist_basic.go:104).  Pass min > max to let the library use the exact extrema. */
int sybl_table_add_column(sybl_table *t, const char *name, int type, int64_t info_min, int64_t info_max);

/* One host-decoded column of one block (what unpackIntCol/unpackStrCol/unpackSetCol
 * produce, column_store_io.go:493-780), handed over columnar. */
typedef struct {
    const char *name;
    int32_t type;             /* SYBL_INT_VAL | SYBL_STR_VAL | SYBL_SET_VAL */
    const int64_t *ints;      /* INT: nrows values */
    const int32_t *str_ids;   /* STR: nrows ids into `strings` (block-local dictionary) */
    const int64_t *set_off;   /* SET: nrows+1 CSR offsets */
    const int32_t *set_ids;   /* SET: member ids into `strings` */
    const uint8_t *populated; /* nrows bytes 0/1 (Record.Populated != _NO_VAL); NULL = all rows */
    const char *const *strings; /* STR/SET: block StringTable */
    int32_t n_strings;
} sybl_col_view;

/* Appends one block (<= 65536 rows in the reference, any size here).  Columns of the
 * table that are absent from `cols` are unpopulated for the whole block. */
int sybl_table_append_block(sybl_table *t, int64_t nrows, int32_t ncols, const sybl_col_view *cols);

/* Device-side deterministic generator for benchmarks and parity tests (SURVEY.md 8d;
 * the formula is restated in oracle/sybil_oracle.c:orc_synth_fill).  Every column is a
 * fully populated INT column. */
enum { SYBL_SYN_UNIFORM = 0, SYBL_SYN_TIME = 1, SYBL_SYN_BELL = 2 };
typedef struct {
    const char *name;
    int32_t kind;      /* SYBL_SYN_* */
    int32_t col_index; /* salt of the per-column hash stream */
    int64_t a, b;      /* UNIFORM: [a, a+b)   TIME: a + floor(i*b/N)   BELL: a + 4 x [0,b) */
    int64_t info_min, info_max; /* IntInfo given to hists; min > max = exact extrema */
} sybl_synth_col;
/* Rows [row0, row0+nrows) of a virtual table of total_rows rows (a rank's shard). */
int sybl_table_create_synth(sybl_ctx *ctx, const char *name, uint64_t seed, int64_t total_rows,
                            int64_t row0, int64_t nrows, int32_t ncols, const sybl_synth_col *cols,
                            sybl_table **out);

/* Native loader: reads a sybil table directory written by the reference
 * (<dir>/<table>/info.db and <block>/{info.db,int_*.db,str_*.db,set_*.db}[.gz],
 * table_io.go:132, table_block_io.go:225-310) straight into HBM.  `columns` limits
 * the load to the referenced columns like LoadSpec does (table_load_spec.go:59-73);
 * NULL loads every column.  rank/nranks shard the block list contiguously. */
int sybl_table_open(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns,
                    int32_t n_columns, int32_t rank, int32_t nranks, sybl_table **out);

/* sybl_table_open with options.  SYBL_OPEN_COMPACT: the table is in compact storage from the first
 * block on (each decoded block is packed into place, columns widen when a block needs it), so a table
 * whose canonical form would not fit in HBM can still be loaded. */
#define SYBL_OPEN_COMPACT 1
int sybl_table_open_flags(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns,
                          int32_t n_columns, int32_t rank, int32_t nranks, int32_t flags, sybl_table **out);

/* Writes the resident table under <dir>/<table name>/ in the reference's on-disk format (block
 * directories with info.db + int_/str_/set_<col>.db gob files, table info.db: column_store_io.go:64-358,
 * table_io.go:40-78) -- bucket encoded at <= 5000 distinct values per block, else value encoded -- so a
 * `sybil` binary or sybl_table_open can read it back. */
int sybl_table_save(sybl_table *t, const char *dir);

/* Test hook (no GPU needed): the gob bytes sybl_table_save writes for one block of an INT column (kind
 * SYBL_INT_VAL) or a STR column (SYBL_STR_VAL; vals = ids into dict).  Library-owned, valid until the next call
 * on the thread. */
const void *sybl_debug_encode_column(int kind, const char *name, const int64_t *vals, const uint8_t *populated, int64_t n,
                                     const char *const *dict, int64_t n_dict, int64_t *n_bytes);

/* Where the time of the last sybl_table_open of this table went (disk -> HBM, the "TableBlock load" half of the hot
 * path).  Worker threads read and gob-decode column files a window of blocks ahead; the calling thread interns
 * dictionaries, copies the compact decoded pieces through a pinned ring and launches the decode kernels in block
 * order. */
typedef struct {
    double wall_s;          /* sybl_table_open, start to finish */
    double parse_cpu_s;     /* CPU time (CLOCK_THREAD_CPUTIME_ID) summed over worker threads: file read + gob decode + flatten */
    double wait_s;          /* calling thread blocked on the next block's worker */
    double apply_s;         /* calling thread: dictionaries, pinned-ring copies, kernel launches, block commit */
    int64_t file_bytes;     /* bytes read from column / info files (after gunzip) */
    int64_t h2d_bytes;      /* bytes that crossed PCIe */
    int32_t workers, blocks;
    /* SYBL_LOADER_GPU_VARINT=1 (ABI 5): column files whose varints were walked on the GPU (csrc/gobgpu.hip), and blocks the
     * host parser then loaded again because a walk reported damage or values outside the block's info.db bounds */
    int32_t gpu_varint_cols, gpu_varint_redone;
} sybl_load_stats;
int sybl_table_load_stats(const sybl_table *t, sybl_load_stats *out);

/* Makes a table opened with sybl_table_open* follow its directory: block directories that appeared since are loaded
 * behind the resident ones, blocks that vanished leave the scan, blocks whose <block>/info.db changed (digest rewrote
 * them) are dropped and loaded again, and the columns' IntInfo is re-read from the table's info.db.  The resident
 * table replaces the reference's per-query re-listing of the directory (table_query.go:40-106) and its per-block
 * result cache (query_cache.go:30-64); this call is what keeps it current.  Prepared queries must be prepared again
 * (SYBL_E_STATE otherwise).  Any of the counters may be NULL.  Dropped rows stay in HBM, unreferenced, until the table
 * is reopened. */
int sybl_table_refresh(sybl_table *t, int64_t *n_added, int64_t *n_dropped, int64_t *n_reloaded);

/* Blocks sybl_table_open skipped the way the reference does: unreadable block info.db, NumRecords
 * <= 0, or a column file whose record ids / value count exceed NumRecords ("BLOCK SIZE CHANGED
 * DURING QUERY", column_store_io.go:524-526,572-574,733-735; table_query.go:134-139). */
int64_t sybl_table_broken_blocks(const sybl_table *t);

int64_t sybl_table_rows(const sybl_table *t);
int64_t sybl_table_blocks(const sybl_table *t);
int64_t sybl_table_hbm_bytes(const sybl_table *t);
/* Exact extrema of an INT column over resident rows and the IntInfo in force. */
int sybl_table_column_info(const sybl_table *t, const char *name, int *type, int64_t *exact_min,
                           int64_t *exact_max, int64_t *info_min, int64_t *info_max, int *has_missing);
/* Multi-rank hosts: declare bounds that hold on EVERY rank (all-reduce the exact extrema
 * first) so that the direct-mapped group layout, and therefore the partial tables, are
 * identical across ranks.  has_missing != 0 reserves the MISSING_VALUE key slot.  lo > hi (no rank holds a
 * value) declares no bounds but still applies has_missing. */
int sybl_table_set_bounds(sybl_table *t, const char *name, int64_t lo, int64_t hi, int has_missing);
/* Group-by on a key column whose value RANGE is too wide for direct mapping (more than 2^22
 * values, or more than 2^27 cells together with the other keys) goes through a dictionary of the
 * column's DISTINCT values (at most 2^22), built on the GPU on first use.  Multi-rank hosts make
 * the dictionaries identical: gather sybl_table_column_distinct from every rank, install the
 * union with sybl_table_set_group_dict on every rank.  `values` is library-owned (valid until the
 * column changes). */
int sybl_table_column_distinct(sybl_table *t, const char *name, const int64_t **values, int64_t *n);
int sybl_table_set_group_dict(sybl_table *t, const char *name, const int64_t *values, int64_t n);
/* Str / set dictionaries (table-global ids are assigned in first-seen order per process).  Ranks of
 * a multi-GPU job gather every rank's dictionary and install the same union everywhere; resident
 * ids are remapped in place.  The new dictionary must contain every resident value. */
int sybl_table_column_dict(sybl_table *t, const char *name, const char *const **strings, int64_t *n);
int sybl_table_set_dict(sybl_table *t, const char *name, const char *const *strings, int64_t n);
/* Compact storage.  Canonical device storage mirrors the reference's Record fields: int64 per INT
 * value (IntField, record_fields.go:8), int32 per STR id (StrField).  sybl_table_compact re-encodes
 * every INT / STR column as unsigned offsets from the column's exact minimum at the narrowest of
 * 1, 2, 4 bytes that holds max - min (8 when it does not fit) -- the in-HBM counterpart of the
 * reference's bucket / delta encoded column files (column_store_io.go:64-358).  Every kernel decodes
 * while loading, so query results are identical; a scan streams the compact bytes.  The table stays
 * in compact mode: blocks appended later are packed into place (a column is re-encoded wider when a
 * block does not fit its width / base; call compact again to re-narrow).  On an empty table the call
 * only switches the mode on.  Installing a str dictionary returns that column to int32 ids.
 * Prepared queries must be prepared again afterwards (SYBL_E_STATE otherwise). */
int sybl_table_compact(sybl_table *t);
/* bytes per stored value and the value base of a column as currently laid out in HBM */
int sybl_table_column_storage(const sybl_table *t, const char *name, int32_t *width, int64_t *base);
/* Copies rows [row0,row0+n) of an INT column back to the host (tests, samples). */
int sybl_table_read_int(const sybl_table *t, const char *name, int64_t row0, int64_t n, int64_t *out);

/* ------------------------------------------------------------------ queries
 * sybl_query_desc mirrors QueryParams (query_spec.go:25-41) plus the FLAGS/OPTS the
 * hot loop reads (aggregate.go:100, hist_basic.go:79,111, hist.go:29-31).         */

typedef struct {
    const char *col;
    int32_t op;            /* SYBL_OP_* ; the column's type selects Int/Str/SetFilter */
    int64_t int_value;     /* IntFilter.Value */
    const char *str_value; /* StrFilter/SetFilter.Value (literal, or regex for RE/NRE) */
    /* optional: RE/NRE evaluated by the host per dictionary entry (Go's regexp), one
     * byte per table-global id; overrides str_value when non-NULL */
    const uint8_t *id_match;
    int64_t id_match_len;
} sybl_filter;

/* -str-replace col:pattern:replacement (table_query.go:33-46, column_store_io.go:517-545): when a block is unpacked, every
 * string of the column's StringTable is rewritten with regexp.ReplaceAllString and strings that become equal share one
 * id -- so str filters compare, and group-by groups, the REWRITTEN strings.  Here the dictionaries are resident and
 * table-global: the rewrite is a property of the query.  Either `pattern` + `replace` (the library's RE2 engine,
 * $1 / ${name} / $$ expanded as regexp.Expand does) or, when `replaced` is non-NULL, the host's own result: one
 * rewritten string per table-global dictionary id (sybl_table_column_dict order).  Only str columns are rewritten. */
typedef struct {
    const char *col;
    const char *pattern;
    const char *replace;
    const char *const *replaced;
    int64_t n_replaced;
} sybl_str_replace;

typedef struct {
    int32_t n_filters;
    const sybl_filter *filters; /* ANDed (aggregate.go:105-116) */
    int32_t n_groups;
    const char *const *groups;  /* int or str columns (set group-by is rejected, cmd_query.go:254) */
    int32_t n_aggs;
    const char *const *aggs;    /* int columns */
    int32_t op;                 /* SYBL_AGG_AVG | SYBL_AGG_HIST */
    int64_t hist_bucket;        /* FLAGS.HIST_BUCKET, 0 = auto */
    /* HIST only: 1 = keep every bucket array (percentiles and buckets can be output);
     * 0 = moments only: count/sum/avg/min/max and the reference's bucket-quantised
     * stddev are produced from four exact integer accumulators. */
    int32_t want_percentiles;
    const char *time_col;       /* with time_bucket > 0: time-series query */
    int64_t time_bucket;        /* QuerySpec.TimeBucket */
    const char *weight_col;     /* OPTS.WEIGHT_COL */
    const char *order_by;       /* "$COUNT", an aggregated column, or NULL/"" = unsorted */
    int32_t order_asc;
    int32_t limit;              /* FLAGS.LIMIT; <= 0 = all groups */
    int32_t block_skip;         /* ShouldLoadBlockFromDir min/max pruning (table_block_io.go:110-182) */
    /* FLAGS.LOG_HIST (-loghist): MultiHist instead of BasicHist (hist.go:27-38, hist_multi.go) -- with op HIST a chain
     * of sub-histograms whose ranges halve from Info.Max downwards; percentiles and stddev come from the union of their
     * buckets.  Bucket arrays are always kept (want_percentiles is implied); sybl_result_subhists describes them. */
    int32_t loghist;
    int32_t n_str_replace;              /* QueryParams.StrReplace (query_spec.go:30) */
    const sybl_str_replace *str_replace;
    /* -distinct (QuerySpec.Distincts, query_spec.go:29; aggregate.go:205-243): every Result also keeps a count-distinct
     * sketch of the listed columns' combined value; sybl_result_distinct reads it and the renderers print it where the
     * reference prints Distinct.Cardinality() (printer.go:79-80,142-144,204-205).  Either int columns (at most
     * SYBL_MAX_GROUPS; the reference's fast path) or exactly one str column (its slow path over one column); a list
     * mixing str with other columns is refused (SYBL_E_INVAL).  The sketch is the reference's dependency
     * github.com/logv/loglogbeta restated from the published algorithms (PARITY UNPINNED: DESIGN.md section 7). */
    int32_t n_distincts;
    const char *const *distincts;
    /* ABI 4.  1 = the caller is a PRINTER (printSortedResults / printResults / -json: printer.go:25-308): of Results it looks
     * at the first `limit` rows of the sort order and at Cumulative, as the reference does -- GetPercentiles / GetStdDev are
     * print-time calls there (printer.go:60-76), made for the rows that are printed.  With limit > 0, op HIST, bucket arrays
     * kept and 2048 group cells or more, percentiles / stddev / bucket arrays are then produced for those rows and
     * Cumulative only: the other rows of sybl_result_rows carry count / samples / sum / avg / min / max, percentiles = NULL,
     * values = NULL and stddev = NaN.  What it saves: the summary pass over every group's bucket array per query (config 4:
     * 0.28 ms and 52 MB of percentiles per step) and, across ranks, the reduce-scatter of the whole bucket table -- only the
     * cell fields, Cumulative's buckets and the printed rows' arrays travel (~1 MB instead of 263 MB per rank;
     * sybl_query_collective_finalize is then 1: snapshot and finalize are collective).  0 (the default, and what
     * -encode-results needs: every Result travels whole, printer.go:263-289): every row carries everything.
     * 2 (ABI 5): as 1, and the rows beyond `limit` need nothing but their Count -- what both printers of the reference look at
     * (printer.go:154-158: sorted[:Limit]).  When the order is $COUNT descending the limit is then pushed INTO the scan
     * (csrc/pushdown.hip): group counts from the key column alone, the printed groups chosen on the device, one pass over key
     * and value columns for Cumulative and those groups -- no bucket array exists for any other group.  Taken for a single
     * direct-mapped key of at most 65 536 cells without filters (sybl_run_stats.strategy = 8); any other query answers as
     * with 1.  The rows beyond the limit then report sum 0 / avg 0 / min, max at their initial values.
     * Across ranks (sybl_comm_init): sybl_query_scan of such a query is a COLLECTIVE call -- at the first scan the ranks ask
     * each other whether every one of them planned the pushed-down scan (a rank without rows cannot; then none takes it), and
     * every pushed-down scan all-reduces the groups' counts between its two passes so that all ranks print the same groups.
     * (A host that merges the ranks' partial tables with collectives of its own -- sybl_query_partials -- must pass 0 or 1: the
     * library cannot make its ranks agree on the printed groups.) */
    int32_t printed_only;
} sybl_query_desc;

int sybl_query_prepare(sybl_table *t, const sybl_query_desc *desc, sybl_query **out);
void sybl_query_free(sybl_query *q);

/* Launches the scan of this rank's resident rows on the ctx stream; asynchronous. */
int sybl_query_scan(sybl_query *q);

/* The rank-local partial group table, laid out identically on every rank of a job:
 * `sum` words combine with SUM, `max` words with MAX (minima are stored negated).
 * Multi-GPU hosts all-reduce both buffers in place (RCCL over xGMI) between
 * sybl_query_scan and sybl_query_finalize; single-GPU hosts skip this. */
int sybl_query_partials(sybl_query *q, void **d_sum, int64_t *n_sum_words, void **d_max, int64_t *n_max_words);
/* Optional: make the scan write its partials into caller-owned device buffers
 * (e.g. torch tensors) of at least the sizes sybl_query_partials reports. */
int sybl_query_bind_partials(sybl_query *q, void *d_sum, void *d_max);

/* In-library RCCL path for hosts without a collective runtime of their own (the Go
 * host): unique_id is the 128-byte ncclUniqueId produced by sybl_comm_unique_id on
 * rank 0 and distributed by the host. */
int sybl_comm_unique_id(void *id128);
int sybl_comm_init(sybl_ctx *ctx, const void *id128, int32_t nranks, int32_t rank);
int sybl_comm_free(sybl_ctx *ctx);
/* COLLECTIVE (every rank, same order).  The first call on a query also compares the ranks' partial-table layouts and
 * fails with SYBL_E_STATE on EVERY rank when they differ (bounds or dictionaries were not agreed: sybl_table_agree) --
 * instead of summing unrelated words or hanging.  Collectives hold the ctx's lock while they wait for the other ranks. */
int sybl_query_allreduce(sybl_query *q);
/* rank / number of ranks of the ctx's communicator (0 / 1 without one).  ABI 5. */
int sybl_comm_info(const sybl_ctx *ctx, int32_t *rank, int32_t *nranks);
/* COLLECTIVE over the ctx's communicator (sybl_comm_init), called by every rank once its shard of the table is resident
 * (sybl_table_open with rank / nranks, or appended blocks) and again after a sybl_table_refresh that loaded blocks: the
 * whole agreement a multi-GPU job needs before its partial tables line up, so that a host without a collective runtime
 * of its own -- the Go host -- runs N ranks through this library alone.  (1) the ranks must hold the same columns
 * (SYBL_E_STATE on every rank otherwise); (2) INT columns: bounds = the extrema over ALL ranks' rows (one MAX all-reduce,
 * minima complemented), has_missing = any rank's -- what sybl_table_set_bounds declares by hand; (3) STR / SET columns:
 * every rank installs the SORTED union of the ranks' dictionaries (sybl_table_set_dict; resident ids renumbered in place)
 * and the union of has_missing; (4) for each int column of `group_cols` (the columns the host is about to group by, in
 * -group order; NULL / 0 = none) whose agreed range the planner would group through a dictionary of distinct values: the
 * union of the ranks' distinct values becomes every rank's dictionary (sybl_table_set_group_dict); when some rank -- or the
 * union -- holds more values than a dictionary may, every rank's planner takes the hash table instead.  Without a
 * communicator (one GPU) the call only sorts the str / set dictionaries, so that the order of equal-count groups in the
 * output does not depend on how many GPUs ran the query.  Prepared queries must be prepared again.  Replaces the
 * reference's key translation at merge time (aggregate.go:284-324,414-467; node_aggregator.go:147-177).  ABI 5. */
int sybl_table_agree(sybl_table *t, const char *const *group_cols, int32_t n_group_cols);

/* Hash group-by (group keys that do not direct-map: more than 2^27 possible cells, or a key column with more than 2^22
 * distinct values -- the reference's map[string]*Result, aggregate.go:186-200).  The scan aggregates into an
 * open-addressing table; afterwards the partial table is the DENSE, key-ordered list of the groups this rank found, so
 * its size is only known after the scan (sybl_query_partials then reports it; sybl_query_bind_partials is refused).
 * sybl_query_allreduce runs the whole multi-rank protocol.  Hosts with their own collective runtime do it in three
 * steps: gather every rank's sybl_query_hash_keys (ascending 62-bit composite keys; n = 0 and keys = NULL for a
 * direct-mapped query), install the sorted union on every rank with sybl_query_hash_install_union (the partial tables
 * then have identical layouts), all-reduce the buffers sybl_query_partials returns.  `keys` is library-owned, valid
 * until the next scan / union install of the query. */
int sybl_query_hash_keys(sybl_query *q, const uint64_t **keys, int64_t *n);
int sybl_query_hash_install_union(sybl_query *q, const uint64_t *keys, int64_t n);

/* 1 when the last sybl_query_allreduce left part of the merge to the finalize: big histogram tables with a row limit
 * are either reduce-SCATTERED (every rank keeps the reduced arrays of a slice of the cells, derives its percentiles there
 * and the summaries are all-gathered) or, for a printer (sybl_query_desc.printed_only), not sent at all -- the ranks
 * all-reduce the cell fields, derive the same sort order, and sum Cumulative's buckets and the printed rows' arrays
 * only.  sybl_query_snapshot and sybl_query_finalize are then COLLECTIVE: every rank calls them, in the same order;
 * every rank gets the full result.  0: rank-local calls as usual. */
int sybl_query_collective_finalize(const sybl_query *q);

/* Optional: enqueue the device -> host copy of the (reduced) partials now (after the all-reduce on
 * multi-GPU hosts).  A following sybl_query_finalize then waits for that copy only, not for work
 * enqueued behind it -- a host serving a stream of queries overlaps the finalize of one query with
 * the scan of the next (two prepared queries, alternating). */
int sybl_query_snapshot(sybl_query *q);
/* Copies the (reduced) partials to the host (unless sybl_query_snapshot already did), finds the groups that exist and
 * sorts them (SortResults, aggregate.go:497-525).  Waits for the copy (hence for the scan and the all-reduce before
 * it).  The ROWS of a result with 2048 groups or more -- avg / stddev / percentiles, GroupByKey strings, the
 * sybl_group_row array -- are built when first asked for (sybl_result_rows, sybl_result_render, sybl_result_encode),
 * from the result's own reference-counted snapshot: the reference's Results are its accumulators (aggregate.go:186-203),
 * nothing has to be built before a printer walks them, and a time series of 360 000 rows or a 65 536-group histogram
 * costs the step that finalizes it nothing it does not print.  The query may be scanned again in between.  A result
 * whose rows carry keys of their own (hash group-by, key spaces beyond 2^18 cells) builds them from the query's group
 * columns and the table's dictionaries: it stays registered with its query, and sybl_query_free builds the rows of
 * such results that are still alive before the query goes away (the table must outlive its queries, as always). */
int sybl_query_finalize(sybl_query *q, sybl_result **out);

/* ------------------------------------------------------------------ results */

typedef struct {
    int32_t present;      /* 0: this group never saw an INT value for the aggregation */
    int64_t count;        /* BasicHist.Count (weighted) */
    int64_t samples;      /* BasicHist.Samples */
    int64_t sum;          /* exact  sum(v*w)  over accepted values */
    double avg;           /* sum/count      (reference: running mean, <=1e-6 rel) */
    double stddev;        /* GetStdDev semantics (hist_basic.go:192-219); 0 in AVG mode */
    int64_t min, max;     /* BasicHist.Min/Max incl. the reference's initial values */
    int64_t bucket_size;  /* HIST: BasicHist.BucketSize */
    int64_t num_buckets;  /* HIST: BasicHist.NumBuckets */
    int64_t n_values;     /* HIST: len(Values) */
    const int64_t *values;      /* HIST + want_percentiles: bucket counts, else NULL */
    const int64_t *percentiles; /* HIST + want_percentiles: 100 entries (GetPercentiles), else NULL */
    int64_t n_outliers;   /* accepted values clipped into an edge bucket (BasicHist.Outliers + Underliers, all blocks) */
    /* HIST + want_percentiles: the outliers' values (ascending; those below hist Min are the reference's Underliers),
     * as the reference remembers them (hist_basic.go:132-142) and prints them as buckets of their own (:239-257).
     * n_outlier_values = -1: not available -- the query's outlier log overflowed (SYBL_OUTLIER_LOG_CAP, default 2^20
     * values per query) or the result was merged across ranks (the values stay on the rank that saw them); the
     * renderers then refuse -json / -encode-results for rows with outliers instead of printing wrong buckets.
     * The reference itself keeps only ONE block's list per histogram after merging (BasicHist.Combine does not merge
     * them, hist_basic.go:259-279) and none in Cumulative: this is the deterministic superset. */
    const int64_t *outlier_values;
    int64_t n_outlier_values;
} sybl_agg_out;

typedef struct {
    const uint8_t *binary_key; /* Result.BinaryByKey: 8 LE bytes per group column */
    const char *group_by_key;  /* Result.GroupByKey: values joined and terminated by '\t' */
    int64_t time_bucket;       /* TimeResults key (0 for all-time results) */
    int64_t count;             /* Result.Count */
    int64_t samples;           /* Result.Samples */
    const sybl_agg_out *aggs;  /* n_aggs entries */
} sybl_group_row;

/* -loghist results: how the `values` array of aggregation `agg` is laid out.  Sub-histogram k (Subhists[k] of the
 * reference's MultiHist, hist_multi.go:223-257) owns values[offset .. offset + n_values): its BasicHist.Values; and
 * values[ext_offset .. ext_offset + n_ext): one exact counter per value from ext_first upwards -- how often that value
 * was an Outlier of the sub-histogram (beyond its last bucket: clipped into it AND remembered, hist_basic.go:132-135;
 * counted per occurrence, not per weight).  sybl_agg_out.bucket_size / num_buckets are 0 for such rows and
 * n_outliers is the sum of the ext counters.  The array is library-owned (valid until the result is freed); n = 0
 * for a query without loghist. */
typedef struct {
    int64_t info_min, info_max;   /* the sub-histogram's Info range (inclusive) */
    int64_t bucket_size, num_buckets, n_values;
    int64_t offset;               /* of its Values inside sybl_agg_out.values */
    int64_t ext_first, n_ext, ext_offset;
} sybl_subhist;
int sybl_result_subhists(const sybl_result *r, int agg, const sybl_subhist **subs, int64_t *n);

/* (the first call on a lazily finalized result builds its rows: see sybl_query_finalize)
 * which: 0 = Results (every group, sorted by order_by; the limit is applied when rendering,
 *            as printSortedResults does), 1 = TimeResults (by bucket, then key),
 *        2 = Cumulative ("TOTAL", one row) */
int sybl_result_rows(const sybl_result *r, int which, const sybl_group_row **rows, int64_t *n);
int64_t sybl_result_matched(const sybl_result *r); /* QuerySpec.MatchedCount */

/* Count-distinct queries (sybl_query_desc.distincts): Result.Distinct of row `row` of sybl_result_rows(which) --
 * *cardinality = Distinct.Cardinality(); *registers (may be NULL) = the sketch's SYBL_HLL_REGISTERS bytes, library-owned
 * until the result is freed (a host that merges results of several nodes takes the register-wise maximum,
 * query_spec.go:180-188).  SYBL_E_INVAL for a query without distincts or a row out of range. */
#define SYBL_HLL_REGISTERS 16384
int sybl_result_distinct(const sybl_result *r, int which, int64_t row, int64_t *cardinality, const uint8_t **registers);
void sybl_result_free(sybl_result *r);

typedef struct {
    int64_t rows_scanned;     /* rows of blocks that were not skipped */
    int64_t blocks_scanned, blocks_skipped;
    int64_t algorithmic_bytes;/* rows_scanned x sum of stored widths of referenced columns (as laid out in HBM) */
    int64_t canonical_bytes;  /* the same with canonical widths (8 per INT value, 4 per STR id) */
    double scan_ms;           /* hipEvent time of the scan kernel(s) of the last sybl_query_scan */
    double reduce_ms;         /* partial-table fold kernels */
    int32_t n_cells;          /* direct-mapped group cells */
    int32_t strategy;         /* 0 = LDS cell table (generic kernel), 1 = global atomics, 2 = LDS cell table
                               * (role-specialised kernel), 3 / 4 = per-workgroup LDS time window (generic /
                               * role-specialised kernel), 5 = partitioned histograms, 6 = cell table AND
                               * bucket arrays in LDS, 7 = hash group-by (open-addressing table in HBM behind an
                               * LDS staging table; n_cells = slots of the table), 8 = -limit pushed into the scan of a
                               * printer's histogram query (printed_only = 2; csrc/pushdown.hip) */
    int32_t lds_bytes, n_workgroups, replicas;
    int32_t n_sum_fields;     /* int64 fields per cell in the SUM section */
    int32_t n_max_fields;     /* fields per cell in the MAX section; 0 = nothing to MAX-reduce */
    int32_t packed_kernel;    /* 1: the scan ran k_scan_packed (compact storage, 32-bit offset domain) */
    int32_t count_pass_reused;/* strategy 5: 1 = the last scan reused the per-(workgroup, bin) record counts its query's first scan
                               * took (they depend on the table's rows, the filters and the key columns only) and skipped the
                               * counting pass over the key column.  ABI 5. */
} sybl_run_stats;
/* Valid after the stream has been synchronised (sybl_query_finalize / sybl_ctx_sync). */
int sybl_query_stats(sybl_query *q, sybl_run_stats *out);

/* Test hook: the exact integer accumulators behind the last finalized result, by direct-mapped cell
 * (cell = time-bucket index x group cells + sum of (key - min) x stride, first group column most
 * significant): which = 0 Count, 1 sum(v), 2 sum(b), 3 sum(b^2) of aggregation `agg` (b = the reference's
 * bucket index, hist_basic.go:130; 2 and 3 exist in moments mode and when percentiles were summarised on
 * the GPU).  Writes min(cap, n_cells) values; *n_cells = cells of the query.  Full-size parity tests compare
 * these with the CPU oracle bit for bit. */
int sybl_debug_query_cells(sybl_query *q, int which, int agg, int64_t *out, int64_t cap, int64_t *n_cells);

/* reference output surface (printer.go:109-232,291-308): renders a result the way
 * `sybil query` prints it.  format: 0 = text table, 1 = -json.  Returns a
 * library-owned NUL-terminated buffer valid until the result is freed. */
const char *sybl_result_render(sybl_result *r, int format);

/* `-encode-results` (printer.go:284-289): the result as encoding/gob of NodeResults{QuerySpec{QueryParams,
 * QueryResults{Cumulative, Results, TimeResults, MatchedCount, Sorted}}} with HistCompat histograms --
 * what `sybil aggregate` (node_aggregator.go) and src/api consume.  Library-owned buffer valid until the
 * result is freed; bucket arrays are included for the rows that carry them. */
const void *sybl_result_encode(sybl_result *r, int64_t *n_bytes);

/* Test hook (no GPU needed): the library's regular-expression engine for re / nre str filters -- Go regexp
 * (RE2) syntax, unanchored search like regexp.MatchString (filter.go:213-236).  1 = match, 0 = no match,
 * -1 = the pattern does not compile (sybl_last_error says why). */
int sybl_debug_regex_match(const char *pattern, const char *text, int64_t text_len);

/* Test hook (no GPU needed): regexp.ReplaceAllString(text, templ) of the same engine -- what -str-replace applies to
 * every dictionary string (column_store_io.go:517-530).  NULL = the pattern does not compile.  Library-owned buffer,
 * valid until the next call on the thread. */
const char *sybl_debug_regex_replace(const char *pattern, const char *text, const char *templ);

/* Test hooks (no GPU needed) for the count-distinct sketch: the host build of the very functions the kernel uses
 * (csrc/hll.h).  sybl_debug_hll_ints: rows of n_cols int64 values (row-major; populated: one byte per value, NULL = all
 * populated) pushed into `registers` (SYBL_HLL_REGISTERS bytes, updated in place) the way the int fast path does;
 * sybl_debug_hll_bytes: MetroHash64(seed 1337) of a byte string pushed the way the str path does, returns the hash;
 * sybl_debug_hll_cardinality: Cardinality() of a register array. */
int sybl_debug_hll_ints(const int64_t *values, const uint8_t *populated, int64_t n_rows, int32_t n_cols, uint8_t *registers);
uint64_t sybl_debug_hll_bytes(const uint8_t *bytes, int64_t len, uint8_t *registers);
int64_t sybl_debug_hll_cardinality(const uint8_t *registers);

/* Test/diagnostic hook: decodes one gob file (info.db, int_/str_/set_*.db, optionally .gz) to JSON
 * with the library's gob reader.  Library-owned buffer, valid until the next call on the thread. */
const char *sybl_debug_gob_to_json(const char *path);

/* Test/diagnostic hook: what the loader's worker half (no GPU involved) makes of ONE block directory -- the staging
 * slab it would send across PCIe -- as a text summary: the block's row count, per column its kind / element widths /
 * counts / extrema, and an FNV-1a digest of every region of the slab (bin values, bin offsets, record ids, values,
 * block-local ids, validity prefix) and of the block's string table.  types[i]: SYBL_INT_VAL / SYBL_STR_VAL /
 * SYBL_SET_VAL.  The CPU test suite holds the decode variants (AVX-512 windows / scalar loops, narrow / wide slices)
 * against each other with it.  NULL (and sybl_last_error) when the arguments are bad; an unreadable or broken block
 * is reported in the text.  Library-owned buffer, valid until the next call on the thread. */
const char *sybl_debug_block_layout(const char *block_dir, const char *const *columns, const int32_t *types, int32_t n_columns);

#ifdef __cplusplus
}
#endif
#endif /* SYBILGPU_H */
