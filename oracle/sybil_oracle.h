/*
 * sybil_oracle.h -- CPU ORACLE for the sybil scan hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (logv/sybil, Go) for
 * the per-block filter -> group -> aggregate -> merge path.  It is the checker
 * the HIP path is compared against; it is never the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Reference files restated (paths relative to the reference repo, src/lib/):
 *   aggregate.go:56-282   FilterAndAggRecords (row loop, key format, time buckets)
 *   aggregate.go:414-467  CombineResults
 *   aggregate.go:43-54, 497-525  sort order
 *   filter.go:171-195     IntFilter.Filter
 *   filter.go:199-250     StrFilter.Filter (eq/neq; re/nre through a per-id table)
 *   filter.go:252-285     SetFilter.Filter
 *   hist_basic.go:34-70   SetupBuckets
 *   hist_basic.go:101-151 AddWeightedValue
 *   hist_basic.go:153-183 GetPercentiles
 *   hist_basic.go:192-219 GetStdDev
 *   hist_basic.go:259-279 BasicHist.Combine
 *   query_spec.go:107-193 ResultMap.Combine / Result.Combine
 *   table_block_io.go:110-182 ShouldLoadBlockFromDir (min/max block skip)
 *
 * Parity pinning: the reference binary cannot be built here (no Go toolchain),
 * so the oracle is pinned against (a) the reference's golden NodeResults file
 * (testdata/TestDecodeGoldenFiles/node_results.golden.json: Combine identity,
 * bucket geometry, key format) and (b) the hand-derived known-answer tests of
 * SURVEY.md section 8c.  See tests/test_oracle_*.py.
 *
 * Two arithmetic modes are produced by one run:
 *   reference-order: per-block float64 running means, blocks merged in index
 *                    order with the count-weighted float merge (what Go does,
 *                    with goroutine completion order fixed to block order);
 *   exact:           int64 sums/counts/buckets (what the HIP path computes).
 */
#ifndef SYBIL_ORACLE_H
#define SYBIL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Count-distinct sketch.  PARITY UNPINNED against the reference's dependency: the reference imports
 * github.com/logv/loglogbeta (query_spec.go:8; no go.mod / vendor directory, so no pinned version; a fork of
 * github.com/seiflotfy/loglogbeta) and holds no test or golden value for -distinct.  Restated from the published
 * algorithms: LogLog-Beta (Qin, Kim, Tung 2016), precision 14, the paper's beta(14) polynomial, and MetroHash64
 * (J. A. Rogers 2015) with seed 1337 as go-metro's Hash64 -- the hash is pinned on MetroHash64's published test
 * vectors (tests/test_oracle_distinct.py). */
#define ORC_LLB_P 14
#define ORC_LLB_M (1 << ORC_LLB_P)
uint64_t orc_metro64(const uint8_t *p, int64_t len, uint64_t seed);
void orc_llb_add_hash(uint8_t *regs, uint64_t x);                /* LogLogBeta.AddHash */
void orc_llb_add(uint8_t *regs, const uint8_t *value, int64_t len); /* LogLogBeta.Add: metro64(value, 1337) */
void orc_llb_merge(uint8_t *regs, const uint8_t *other);         /* LogLogBeta.Merge: register-wise max */
uint64_t orc_llb_cardinality(const uint8_t *regs);               /* LogLogBeta.Cardinality */

/* record.go:14-19 value tags */
enum { ORC_NO_VAL = 0, ORC_INT_VAL = 1, ORC_STR_VAL = 2, ORC_SET_VAL = 3 };

/* filter ops: filter.go:176-190 (int), :213-245 (str), :268-283 (set) */
enum {
    ORC_OP_GT = 0, ORC_OP_LT = 1, ORC_OP_EQ = 2, ORC_OP_NEQ = 3, /* int, str(eq/neq) */
    ORC_OP_RE = 4, ORC_OP_NRE = 5,                               /* str via id table */
    ORC_OP_IN = 6, ORC_OP_NIN = 7                                /* set */
};

enum { ORC_AGG_AVG = 0, ORC_AGG_HIST = 1 }; /* FLAGS.OP "avg" | "hist" */

#define ORC_MAX_GROUPS 8
#define ORC_MAX_AGGS 8
#define ORC_GROUP_BY_WIDTH 8 /* aggregate.go:16 */

typedef struct {
    int32_t type;             /* ORC_INT_VAL | ORC_STR_VAL | ORC_SET_VAL */
    const int64_t *ints;      /* INT: one value per row */
    const int32_t *strs;      /* STR: one (table-global) dictionary id per row */
    const int64_t *set_off;   /* SET: CSR offsets, nrows+1 */
    const int32_t *set_vals;  /* SET: CSR member ids */
    const uint8_t *populated; /* per row 0/1, NULL = every row populated */
} orc_col;

typedef struct {
    int32_t col;            /* index into cols[] */
    int32_t op;             /* ORC_OP_* */
    int64_t value;          /* int constant, or str/set target id (-1: value not in dictionary) */
    const uint8_t *idtable; /* RE/NRE: idtable[id] = regex matched that dictionary entry */
    int64_t idtable_len;
} orc_filter;

typedef struct {
    int32_t col;
    int64_t info_min, info_max; /* table-level IntInfo.Min/Max (aggregate.go:254) */
} orc_agg;

typedef struct {
    int32_t n_filters;
    const orc_filter *filters;
    int32_t n_groups;
    int32_t group_cols[ORC_MAX_GROUPS];
    int32_t n_aggs;
    orc_agg aggs[ORC_MAX_AGGS];
    int32_t op;          /* ORC_AGG_AVG | ORC_AGG_HIST */
    int64_t hist_bucket; /* FLAGS.HIST_BUCKET (-int-bucket), 0 = auto */
    int32_t time_col;    /* -1 = no time series */
    int64_t time_bucket; /* QuerySpec.TimeBucket, 0 = off */
    int32_t weight_col;  /* -1 = none (OPTS.WEIGHT_COL) */
    int32_t block_skip;  /* apply ShouldLoadBlockFromDir using exact per-block min/max */
    int64_t block_rows;  /* CHUNK_SIZE, 65536 in production (table.go:44) */
    int32_t n_threads;   /* blocks scanned in parallel; merge is always in block order */
    int32_t loghist;     /* FLAGS.LOG_HIST: MultiHist instead of BasicHist (hist.go:27-38, hist_multi.go) */
    /* -distinct (query_spec.go:29 Distincts): columns whose combined value feeds each Result's LogLog-Beta sketch
     * (aggregate.go:205-243).  distinct_dicts[i]: the strings of a STR column's dictionary ids (the slow path,
     * aggregate.go:225-239); NULL for INT columns. */
    int32_t n_distincts;
    int32_t distinct_cols[ORC_MAX_GROUPS];
    const char *const *distinct_dicts[ORC_MAX_GROUPS];
    int64_t distinct_dict_len[ORC_MAX_GROUPS];
} orc_query;

/* One merged Result (query_spec.go:85-93) + its hists, both arithmetic modes. */
typedef struct orc_result orc_result;
typedef struct orc_results orc_results;

/* Runs the scan over nrows rows of ncols columns; never returns NULL on valid input. */
orc_results *orc_query_run(const orc_query *q, const orc_col *cols, int32_t ncols, int64_t nrows);
void orc_results_free(orc_results *r);

int64_t orc_matched_count(const orc_results *r);
int64_t orc_blocks_scanned(const orc_results *r);
int64_t orc_blocks_skipped(const orc_results *r);

/* which: 0 = Results (all-time), 1 = TimeResults entries, 2 = Cumulative ("TOTAL", 1 entry) */
int64_t orc_num_results(const orc_results *r, int which);
/* Results are returned in a canonical order: (time bucket, key bytes as unsigned LE ints). */
int orc_result_get(const orc_results *r, int which, int64_t idx,
                   uint8_t *key /* 8*n_groups */, int64_t *time_bucket,
                   int64_t *count, int64_t *samples);
/* Per-aggregation state of result idx.  present=0 when the result has no hist
 * for that aggregation (no INT value was ever added, aggregate.go:246-259). */
typedef struct {
    int32_t present;
    int32_t percentile_mode;
    int64_t num_buckets;  /* BasicHist.NumBuckets */
    int64_t bucket_size;  /* BasicHist.BucketSize */
    int64_t n_values;     /* len(Values) */
    int64_t count;        /* h.Count (weighted) */
    int64_t samples;      /* h.Samples */
    int64_t min, max;     /* h.Min / h.Max incl. the reference's initial values */
    double avg;           /* reference-order running / merged mean */
    int64_t sum_exact;    /* exact Σ v*w over accepted values (wraps mod 2^64) */
    int64_t true_min, true_max; /* extrema over accepted values only */
    int64_t n_outliers;   /* accepted values clipped into the last bucket (all blocks) */
    int64_t n_underliers;
    double stddev_ref;    /* GetStdDev on the merged hist as the reference would hold it
                             (outlier list of the first block only, hist_basic.go:259-279) */
    double stddev_exact;  /* GetStdDev with avg = sum_exact/count and every outlier */
} orc_hist_info;
int orc_result_hist(const orc_results *r, int which, int64_t idx, int agg, orc_hist_info *out);
/* Result.Distinct.Cardinality() (printer.go:79-80,142-144,204-205); regs_out (ORC_LLB_M bytes, may be NULL) receives
 * the sketch's registers.  -1: no such result. */
int64_t orc_result_distinct(const orc_results *r, int which, int64_t idx, uint8_t *regs_out);
/* Copies len(Values) bucket counts; returns n_values or <0. */
int64_t orc_result_hist_values(const orc_results *r, int which, int64_t idx, int agg, int64_t *out, int64_t cap);
/* GetPercentiles (hist_basic.go:153-183): writes up to 100 entries, returns how many (0 if Count==0). */
int orc_result_percentiles(const orc_results *r, int which, int64_t idx, int agg, int64_t *out100);

/* ---- stand-alone pieces, exposed so known-answer tests can pin them ---- */

/* hist_basic.go:34-70 */
void orc_setup_buckets(int64_t info_min, int64_t info_max, int64_t hist_bucket,
                       int64_t *bucket_size, int64_t *num_buckets, int64_t *n_values);
/* GetPercentiles over an explicit Values array */
int orc_percentiles_from_values(const int64_t *values, int64_t n_values, int64_t bucket_size,
                                int64_t hmin, int64_t count, int64_t *out100);
/* GetStdDev over explicit state */
double orc_stddev_from_values(const int64_t *values, int64_t n_values, int64_t bucket_size,
                              int64_t hmin, int64_t count, double avg,
                              const int64_t *outliers, int64_t n_out,
                              const int64_t *underliers, int64_t n_under);
/* BasicHist.Combine mean merge (hist_basic.go:264-265) */
double orc_combine_avg(double avg_a, int64_t count_a, double avg_b, int64_t count_b);
/* aggregate.go:174 */
int64_t orc_time_bucket(int64_t t, int64_t bucket);

/* A tiny single-hist driver for KATs: feeds (v,w) pairs through AddWeightedValue. */
typedef struct orc_hist orc_hist;
orc_hist *orc_hist_new(int64_t info_min, int64_t info_max, int op, int64_t hist_bucket, int weight_col_mode);
void orc_hist_add(orc_hist *h, int64_t v, int64_t w);
void orc_hist_combine(orc_hist *h, const orc_hist *other);
void orc_hist_info_get(const orc_hist *h, orc_hist_info *out);
int64_t orc_hist_values(const orc_hist *h, int64_t *out, int64_t cap);
int orc_hist_percentiles(const orc_hist *h, int64_t *out100);
int64_t orc_hist_outliers(const orc_hist *h, int64_t *out, int64_t cap);

/* -loghist: MultiHist (hist_multi.go): an outer Count / Avg / Min / Max behind the same Info.Min .. Info.Max*10 gate and,
 * in hist mode, a chain of BasicHist sub-histograms whose ranges halve from Info.Max downwards (TrackPercentiles,
 * :223-257); a value goes to the FIRST sub-histogram whose [Info.Min, Info.Max] holds it (:84-89).  An orc_hist made
 * by orc_hist_new_multi answers orc_hist_add / orc_hist_combine / orc_hist_info_get like any other; its percentiles
 * and standard deviation are GetPercentiles (:93-128) and GetStdDev (:140-155) over the union of the sub-histograms'
 * sparse buckets -- "exact" variant: every block's outliers, mean = exact sum / count, keys summed in ascending
 * order (the reference keeps one block's outlier lists and sums in map order). */
orc_hist *orc_hist_new_multi(int64_t info_min, int64_t info_max, int op, int64_t hist_bucket, int weight_mode);
int orc_hist_n_sub(const orc_hist *h);
/* sub-histogram k: out6 = {Info.Min, Info.Max, BucketSize, NumBuckets, len(Values), offset of its Values in the
 * concatenated array orc_hist_values returns} */
int orc_hist_sub(const orc_hist *h, int k, int64_t *out6);
/* the union of the sub-histograms' sparse buckets (GetSparseBuckets, :190-207), keys ascending; returns the number of
 * keys (nothing written beyond cap) */
int64_t orc_hist_sparse(const orc_hist *h, int64_t *keys, int64_t *counts, int64_t cap);
/* all outliers + underliers of a result's hist over every block, ascending; returns the count (nothing written if > cap) */
int64_t orc_result_outliers(const orc_results *R, int which, int64_t idx, int agg, int64_t *out, int64_t cap);
/* -loghist results: see orc_hist_n_sub / orc_hist_sub / orc_hist_sparse */
int orc_result_n_sub(const orc_results *R, int which, int64_t idx, int agg);
int orc_result_sub(const orc_results *R, int which, int64_t idx, int agg, int k, int64_t *out6);
int64_t orc_result_sparse(const orc_results *R, int which, int64_t idx, int agg, int64_t *keys, int64_t *counts, int64_t cap);
void orc_hist_free(orc_hist *h);

/* ---- synthetic table generator (OURS, not the reference's; DESIGN.md "Synthetic table") ---- */
enum {
    ORC_SYN_UNIFORM = 0, /* uniform in [a, a+b)                     */
    ORC_SYN_TIME = 1,    /* a + floor(i * b / N)   (non-decreasing)  */
    ORC_SYN_BELL = 2     /* sum of four uniform [0, b) + a           */
};
uint64_t orc_splitmix64(uint64_t x);
void orc_synth_fill(int kind, int64_t a, int64_t b, uint64_t seed, int32_t col_index,
                    int64_t row0, int64_t n, int64_t total_rows, int64_t *out);

/* ---- columnar CPU baseline (OURS: BASELINE.md section 2, variant ii) --------------------------------
 * What a straightforward columnar engine does on the host for the config-3 query shape -- no Record rows,
 * no maps: nf inclusive int ranges, ng direct-mapped int group columns, na aggregation columns with the
 * moments Count / sum(v) / sum(b) / sum(b^2) of the reference's bucket index b = (v - hmin) / bucket_size
 * (hist_basic.go:130) per cell; one thread-local cell table per OpenMP thread, summed at the end.
 * out: [1 + 3 * na][cells] int64 (Count, then sum(v), sum(b), sum(b^2) per aggregation).  Returns the
 * matched-row count, or -1 when a key falls outside [gmin, gmin + gcard).  Test / bench infrastructure
 * only, like everything in this directory. */
int64_t orc_columnar_scan(int64_t nrows, int32_t nf, const int64_t *const *fcols, const int64_t *lo, const int64_t *hi,
                          int32_t ng, const int64_t *const *gcols, const int64_t *gmin, const int64_t *gcard, int32_t na,
                          const int64_t *const *acols, const int64_t *hmin, const int64_t *bucket_size, int32_t n_threads,
                          int64_t *out);

/* ---- full-size checker (OURS): the columnar scan above, fused with the synthetic generator ----------------
 * The BASELINE.json configurations are 10^9 rows: too many to materialise on the host or to push through the
 * per-row hash maps of orc_query_run in test time.  orc_synth_scan regenerates every referenced column of the
 * synthetic table block by block (orc_synth_fill, one 65536-row block per OpenMP task, nothing bigger than a
 * block is ever held) and runs the reference's row loop in its direct-mapped columnar form:
 *   filters   lo <= x <= hi                                   (filter.go:171-195, gt/lt/eq folded)
 *   time      tb = t / time_bucket * time_bucket              (aggregate.go:174)
 *   cell      ((tb / time_bucket - tb_min) * cells_g) + sum over group columns of (x - gmin) radix gcard
 *             -- first group column most significant          (aggregate.go:125-143 key order)
 *   per agg   b = (v - hmin) / bucket_size, clipped into [0, n_values)   (hist_basic.go:130-150)
 * out_fields: [1 + 3 * na][cells] int64 = Count, then per aggregation sum(v), sum(b), sum(b^2).
 * out_hist:   NULL, or [cells][na][nv_max] int64 bucket counts (relaxed atomic adds on one shared table).
 * Returns the matched-row count, -1 when a key or time bucket falls outside the declared ranges.
 * tests/test_oracle_query.py checks it against orc_query_run on the same (materialised) table. */
typedef struct {
    int32_t kind, col_index;
    int64_t a, b;
} orc_synth_col;
typedef struct {
    uint64_t seed;
    int64_t total_rows, row0, nrows;
    int32_t nf, ng, na, has_time;
    orc_synth_col fcol[4], gcol[4], acol[4], tcol;
    int64_t lo[4], hi[4];
    int64_t gmin[4], gcard[4];
    int64_t hmin[4], bucket_size[4], n_values[4];
    int64_t time_bucket, tb_min, n_tb;
    int64_t nv_max; /* stride of out_hist per (cell, agg) */
    int32_t n_threads, pad_;
} orc_synth_query;
int64_t orc_synth_scan(const orc_synth_query *q, int64_t *out_fields, int64_t *out_hist);

#ifdef __cplusplus
}
#endif
#endif
