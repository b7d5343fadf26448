// plan.h -- the device-visible description of one scan (shared by engine.cpp and kernels.hip).
//
// A "slot" is one referenced column as the scan kernel sees it: every role the
// reference's row loop gives that column (filter operand, group-by key part,
// aggregation input, time column, weight column -- aggregate.go:96-263) is a flag on
// the slot, so each referenced column is streamed from HBM exactly once per row.
#pragma once
#include <stdint.h>

namespace sybl {

// The diagnostic switches (SYBL_*).  The process environment is read ONCE -- the first time the library asks -- into a
// table of the library's own, and every later question is answered from it: no entry point calls getenv() at query time
// (getenv is not safe against a setenv from another thread of a multi-threaded Go host).  SYBL_ENV_LIVE=1
// in that first snapshot (the Python wrapper sets it: the test suite and the tools flip switches between queries) makes
// every question a getenv() again.
const char *env(const char *name);


constexpr int kMaxSlots = 12;
constexpr int kMaxAggs = 6;   // == SYBL_MAX_AGGS
constexpr int kMaxNeq = 4;    // neq constants folded per slot
constexpr int kWgThreads = 1024;
constexpr int kRowsPerThread = 2;                        // one 16-byte load per int64 column
constexpr int kTileRows = kWgThreads * kRowsPerThread;   // rows per workgroup iteration
constexpr int kHeaderWords = 64;                         // SUM-section header
constexpr int kLdsBudgetBytes = 152 * 1024;              // of 160 KiB per CU
constexpr int64_t kDictEmpty = INT64_MIN;                // free slot of a group dictionary
constexpr int64_t kDictMaxDistinct = 1 << 22;            // distinct values a dictionary may hold

enum SlotFlags : uint32_t {
    kSlotRange = 1u << 0,   // lo <= x <= hi   (gt/lt/eq int filters folded, filter.go:171-195)
    kSlotNeq = 1u << 1,     // x != neq[i]
    kSlotIdMask = 1u << 2,  // str filter as a bit per dictionary id (filter.go:199-250)
    kSlotGroup = 1u << 3,   // key part (aggregate.go:125-143)
    kSlotAgg = 1u << 4,     // aggregation input (aggregate.go:246-261)
    kSlotTime = 1u << 5,    // time column (aggregate.go:146-183)
    kSlotWeight = 1u << 6,  // weight column (aggregate.go:100-102)
    kSlotSet = 1u << 8,     // set column: base = CSR offsets per physical row, filters in setp[]
    kSlotDict = 1u << 9,    // group key through a value -> rank hash map (sparse / wide key ranges)
    kSlotFilter = kSlotRange | kSlotNeq | kSlotIdMask,
};

struct SlotDesc {
    const void *base;        // column values: `width` bytes per row, value = vbase + zero-extended raw
                             // (canonical int columns: width 8, vbase 0; str ids: width 4, vbase 0;
                             // compacted columns: the narrowest width holding max - min)
    const uint32_t *valid;   // bit per physical row, nullptr = every row populated
    const uint32_t *idmask;  // kSlotIdMask: bit per dictionary id, 1 = row passes
    int64_t lo, hi;          // kSlotRange (inclusive)
    int64_t neq[kMaxNeq];    // kSlotNeq
    int64_t gmin;            // kSlotGroup: cell += (x - gmin) * gstride
    uint32_t flags;
    int32_t n_neq;
    int32_t idmask_bits;
    int32_t gcard;           // digits incl. a separate MISSING digit
    int32_t gvalues;         // digits that are real values (x - gmin must be below this)
    int32_t gstride;
    int32_t gmissing;        // cell offset of missing rows (their own digit, or the digit of the value
                             // -1 whose 8-byte image equals MISSING_VALUE), -1 = column has no missing rows
    int32_t agg_index;
    int32_t pad_;
    // kSlotSet (filter.go:252-285): predicate p passes when (any member == set_id[p]) == set_in[p]
    int32_t n_setp;
    const int32_t *set_vals;  // CSR member ids (table-global dictionary)
    int32_t set_id[kMaxNeq];
    int32_t set_in[kMaxNeq];
    // kSlotDict: open-addressing map of the column's distinct values to their rank in sorted
    // order; dkeys[h] == kDictEmpty marks a free slot
    const int64_t *dkeys;
    const int32_t *dranks;
    uint32_t dmask;
    int32_t width;           // bytes per stored value: 1, 2, 4 or 8
    int64_t vbase;           // added to the raw (unsigned) stored value
    // the key digit's weight in the 64-bit composite key (== gstride / gmissing / gvalues unless the query groups
    // through the hash table, where the key space may be as wide as 2^62)
    int64_t gstride64, gmissing64, gvalues64;
};

struct AggDesc {
    int64_t info_min;     // hist_basic.go:104  reject v < Info.Min
    int64_t max10;        //                    reject v > Info.Max*10 (wrapping like Go)
    int64_t hmin;         // BasicHist.Min at setup (= Info.Min in hist mode)
    int64_t bucket_size;  // BasicHist.BucketSize
    double inv_bucket;    // 1.0 / bucket_size
    int32_t n_values;     // len(Values)
    int32_t big_div;      // value range too wide for the double-reciprocal divide
    // field indices into the cell table (-1 = not tracked)
    int32_t f_sum;        // sum(v*w)
    int32_t f_cnt;        // sum(w) over accepted values (only if rejection/missing/weights possible)
    int32_t f_smp;        // accepted values (weighted queries)
    int32_t f_pop;        // populated values incl. rejected ones: the group owns a hist for the
                          // aggregation as soon as one INT value was seen (aggregate.go:246-258);
                          // tracked only for columns with missing rows (else == row count)
    int32_t f_sb;         // moments: sum(b*w)
    int32_t f_sb2;        // moments: sum(b*b*w)
    int32_t f_out;        // 6 fields: n, sum(o), sum(o^2) as four 32-bit limbs -- only when the
                          // column's bounds allow a value to land beyond the last bucket
    int32_t m_max;        // MAX-section field: max(v)
    int32_t m_nmin;       // MAX-section field: max(-v)
    int32_t hist_full;    // 1: bucket arrays in the SUM section (global atomics)
    int32_t multi_n;      // -loghist: sub-histograms of this aggregation (0 = BasicHist), ScanPlan::multi[multi_off ..]
    int32_t multi_off;
};

// -loghist (MultiHist, hist_multi.go): one sub-histogram of an aggregation.  A value goes to the FIRST sub-histogram
// whose [mn, mx] holds it (:84-89), through that BasicHist's own AddWeightedValue: its reject gate (v > Info.Max*10 of
// the SUB-range: a negative maximum rejects most of its own range), b = (v - mn) / bs, and when b reaches len(Values)
// the value is clipped into the last bucket and remembered -- here counted in an exact per-value counter, since the
// outliers of a sub-histogram all lie in [ext_first, mx].
struct MultiSub {
    int64_t mn, mx, max10, bs;
    double inv_bs;
    int32_t nv, big_div;
    int64_t off;        // of the sub-histogram's Values inside the aggregation's bucket words
    int64_t ext_off;    // of its outlier counters
    int64_t ext_first;  // value of the first outlier counter (= mn + nv * bs)
    int64_t n_ext;
};

struct Segment {           // a run of physical rows one workgroup scans
    int64_t start;         // even
    int64_t n;
};

struct ScanPlan {
    SlotDesc slot[kMaxSlots];
    AggDesc agg[kMaxAggs];
    int32_t n_slots, n_aggs;
    int32_t hist_mode;       // FLAGS.OP == "hist"
    int32_t weighted;        // OPTS.WEIGHT_COL
    int32_t time_slot;       // -1 = none
    int32_t weight_slot;     // -1 = none
    int64_t time_bucket;     // QuerySpec.TimeBucket
    double inv_time_bucket;
    int64_t tb_min;          // trunc(tmin / time_bucket)
    int32_t n_tb;
    int32_t tb_stride;       // = number of group cells (direct-mapped kernels)
    int64_t tb_stride64;     // the same as the time bucket's weight in the 64-bit composite key (hash group-by)
    int32_t tb_big_div;
    int32_t n_cells;         // n_tb * group cells
    int32_t n_sum_fields;    // F: [F][n_cells] int64, field 0 = Count, (field 1 = Samples)
    int32_t n_max_fields;    // M: [M][n_cells] int64 (MAX-combined)
    int32_t f_samples;       // -1 unless weighted
    int32_t rep_shift;       // LDS replicas = 1 << rep_shift
    // LDS-window strategy (time-sorted tables): each workgroup's rows span only a few time
    // buckets, so its LDS table covers cells [wg_cell_base[wg], +lds_cells) of the global table
    // and is flushed with atomics at the end instead of being folded.
    int32_t windowed;
    int32_t lds_cells;       // cells per LDS table (== n_cells unless windowed)
    const int32_t *wg_cell_base;
    // Hash group-by (strategy 7): the composite key of a row -- sum of digit x gstride64, below 2^62 -- does not index the
    // cell table but is looked up / inserted in an open-addressing table of n_cells slots (a power of two); the slot
    // number then plays the role of the cell.  hash_keys[slot] == kHashEmpty marks a free slot.
    int32_t hash_mode, pad_hash_;
    uint64_t *hash_keys;
    // Outlier log (queries that keep bucket arrays and whose column bounds allow a value beyond the last bucket):
    // every outlier / underlier is appended as (cell or composite key, aggregation, value) -- the reference remembers
    // the values themselves (hist_basic.go:132-142) and prints them as buckets of their own (GetStrBuckets, :239-257).
    // The kernels append to a STAGING log of kOutStripes stripes, each behind a cursor of its own (a wave picks its
    // stripe from its workgroup and wave number and reserves the places of all its outlier lanes with one atomic:
    // scan_generic.h, log_outlier); k_outlog_gather closes the stripes up into the query's dense log behind the scan and
    // writes the number of records appended to header word kHdrOutLog (more than out_cap: some were only counted).
    int64_t *out_log;        // staging: [kOutStripes][kOutCursorWords] cursors, then [kOutStripes][out_cap / kOutStripes] records
    int64_t out_cap;         // records the log holds (a multiple of kOutStripes)
    const MultiSub *multi;   // -loghist: every aggregation's sub-histograms
    int64_t hist_off;        // word offset of bucket arrays in the SUM section
    int64_t hist_stride;     // words per cell = sum of n_values over full-hist aggs
    int64_t hist_agg_off[kMaxAggs];
    // outputs
    int64_t *sum_out;        // SUM section: [header][F*n_cells][hist]
    int64_t *max_out;        // MAX section: [M*n_cells]
    int64_t *ws_sum;         // LDS strategy: [n_wg][F*n_cells]
    int64_t *ws_max;         // LDS strategy: [n_wg][M*n_cells]
    // work
    const Segment *segs;
    const int32_t *wg_seg_begin;  // [n_wg+1]
    // Count distinct (hll.h; only in the copy of the plan that k_scan_distinct reads, Query::dplan): every matched row
    // raises one register of its cell's sketch.  Int columns: the hashed buffer holds 8 bytes per column in list order
    // (distinct_slot[i] = the slot of the i-th column).  One str column: the hash of (string + "\t") was computed per
    // dictionary id on the host (hll_idhash; hll_missing for a row without the column).
    uint8_t *hll;                // [n_cells][kHllRegs]; hashed group-by: [hll_nkeys][kHllRegs], a sketch per dense key
    // hashed group-by (round 6): the query's sorted composite keys as they stand when the pass runs -- after the scan's
    // compaction and, across ranks, the key union --; a row's sketch is the one of its key's place in that list
    const uint64_t *hll_keys;
    int64_t hll_nkeys;
    const uint64_t *hll_idhash;
    uint64_t hll_missing;
    int64_t hll_ids;
    int32_t n_distinct;
    int32_t distinct_slot[8];
    // The slow path over SEVERAL columns with a str column among them (aggregate.go:224-239): the hashed buffer is, per
    // column, the decimal digits of an int / the dictionary string of a str id / nothing for a row without the column,
    // each followed by "\t" -- assembled and hashed per row (hll.h: Metro64Stream).  hll_chars[i] / hll_stroff[i]: the i-th
    // column's dictionary strings back to back (as -str-replace left them) and their [ids + 1] offsets; NULL = int column.
    int32_t hll_mixed;
    const char *hll_chars[8];
    const int64_t *hll_stroff[8];
    int64_t hll_nids[8];
};

// finalize-side histogram summaries (kernels.hip: k_hist_summary / k_hist_total)
struct HistSummaryPlan {
    const int64_t *H;          // [cell][hist_stride]
    const int64_t *F;          // cell fields [field][n_cells]
    int64_t hist_stride, n_cells;
    int32_t n_aggs, pad_;
    int64_t agg_off[kMaxAggs], n_values[kMaxAggs], bucket_size[kMaxAggs], hmin[kMaxAggs];
    int32_t f_cnt[kMaxAggs];   // field holding the aggregation's count (0 = Result.Count)
    int64_t cell0, cell1;      // cells to summarise (a rank's slice after a reduce-scatter; else all of them)
    int64_t *pct;              // [cell * n_aggs + a][100], zeroed
    int64_t *mom;              // [cell * n_aggs + a][2]: sum(b * Values[b]), sum(b^2 * Values[b])
};

constexpr int kOutLogWords = 3;                  // int64 words per outlier record: cell / key, aggregation, value
constexpr int kOutStripes = 64;                  // staging stripes of the outlier log (one cursor each)
constexpr int kOutCursorWords = 16;              // a 128-byte line per cursor
constexpr int64_t kOutLogDefaultCap = 1 << 20;   // records
constexpr uint64_t kHashEmpty = ~(uint64_t)0;   // free slot of the group hash table (composite keys are below 2^62)
constexpr int64_t kHashMaxSlots = (int64_t)1 << 27;

// SUM-section header words
enum Header : int {
    kHdrMatched = 0,     // QuerySpec.MatchedCount
    kHdrOverflow = 1,    // rows whose key / bucket fell outside the declared bounds (must be 0)
    kHdrPartOverflow = 2, // partitioned histograms: records that did not fit their partition buffer
    kHdrEmitStall = 3,    // partitioned histograms: a lane gave up waiting for a staging chunk (must be 0: a bug)
    kHdrHashFull = 4,     // hash group-by: rows whose key found no free slot (more distinct keys than the table holds)
    kHdrOutLog = 5,       // outlier log: records appended (k_outlog_gather; more than the log's capacity: some were dropped)
    kHdrPdSum = 8,        // .. 8 + kMaxAggs: a pushed-down printer's query (pushdown.hip): sum(v) of EVERY row, per aggregation --
    kHdrPdMax = 16,       // .. and max(v): Cumulative's, the cell fields of the rows beyond the limit being empty in that mode
};

// ---- table load, round 6: the int columns of one block whose `Values` varints are walked on the GPU (gobgpu.hip)
constexpr int kGobWgThreads = 256;                       // one 64-byte chunk per thread: a workgroup walks 16 KB of a file
constexpr int kGobWgBytes = kGobWgThreads * 64;
constexpr int kGobMaxWgs = 64;                           // per file (the look-back is one wave wide): 1 MB of values
// a job's state words (zeroed before the launch): flags | values found | zero values found | records in the bins (k_gob_bins) |
// the workgroups' exit maps | their value counts
// (Tail: `Values` -- the three values behind the announced ones, each + 1, 0 = there is none: what the struct ends with)
constexpr int kGobStateFlags = 0, kGobStateFound = 1, kGobStateZeros = 2, kGobStateRecs = 3, kGobStateTail = 4, kGobStateMaps = 8,
              kGobStateCounts = 8 + kGobMaxWgs, kGobStateWords = 8 + 2 * kGobMaxWgs;
struct GobValuesJob {
    const uint8_t *bytes;        // device, 16-byte aligned: the first value of the file's slice (behind its count) ...
    uint32_t n_bytes, n;         // ... to the end of the file's value message; how many values to take from it at most
    long long *out;              // [n] `Values`: the signed values as the file holds them (deltas when the column is value-encoded);
                                 // `Bins` (zpos != null): the unsigned form of every value of the region
    unsigned long long *state;   // [kGobStateWords]
    uint32_t *zpos;              // `Bins`: [n_zpos] the ranks of the values that are 0 (the buckets' terminators, mostly), in order
    uint32_t n_zpos;
    int32_t n_wgs;
};
constexpr uint32_t kGobBadByte = 1u;    // flags: a value starts with a byte 0x80..0xF7
constexpr uint32_t kGobShort = 2u;      // fewer values in the region than the slice header announced
constexpr uint32_t kGobTruncated = 4u;  // a value reaches beyond the region
constexpr uint32_t kGobOutOfBounds = 8u;  // a column value outside the bounds the block was placed by (k_decode_delta)
constexpr uint32_t kGobGaveUp = 16u;    // a look-back word did not arrive
constexpr uint32_t kGobBadBins = 32u;   // the buckets do not parse (k_gob_bins), or hold another number of records than announced
constexpr int kGobBatchMax = 16;
struct GobValuesBatch {
    GobValuesJob job[kGobBatchMax];
    int32_t n;
};
// the buckets of one bucket-encoded int column file, from the values k_gob_values found in its `Bins` region (k_gob_bins)
constexpr int kGobMaxBins = 8192;
struct GobBinsJob {
    const unsigned long long *tok;  // the region's values, unsigned form
    const uint32_t *zpos;           // ranks of the zero values
    unsigned long long *state;      // the walk's state words (values / zeros found; flags and the record total go back there)
    long long *bin_val;             // [n_bins] out
    long long *bin_rng;             // [2 * n_bins] out: first and one-past-last rank of every bucket's records in tok
    uint32_t n_bins, tok_cap, zpos_cap, pad_;
    long long n_recs, chk_min, chk_max;  // what the block's info.db announced: records in all buckets; bounds of the values
};
struct GobBinsBatch {
    GobBinsJob job[kGobBatchMax];
    int32_t n;
};

// ---- table load: one block's bucket-encoded / value-encoded columns, a launch each (loader.cpp, kernels.hip:
// k_decode_bins_multi / k_decode_delta_multi); the fields are the single-column launchers' arguments (engine.h)
constexpr int kDecodeBatchMax = 16;
struct DecodeBinsJob {
    const void *recs;
    const int64_t *bin_off, *bin_val;
    void *col;
    uint32_t *valid;
    int64_t vbase;
    int32_t n_bins;
    uint8_t rec_w, out_w, delta, pad;
    uint32_t nrows, pad2_;  // of the job's block (a launch may hold the columns of several blocks)
    // rec_w == 8 (the GPU varint walk's columns): recs are the walk's 64-bit values and bin_off holds a [first, one-past-last)
    // pair per bucket (GobBinsJob::bin_rng); kGobOutOfBounds is set in *chk_flags for a record id that is no row of the block
    unsigned long long *chk_flags;
};
struct DecodeBinsBatch {
    DecodeBinsJob job[kDecodeBatchMax];
    int32_t n;
};
struct DecodeDeltaJob {
    const void *deltas;
    void *col;
    int64_t n, vbase;
    uint8_t val_w, out_w, venc, pad[5];
    // non-null (the GPU varint walk's columns): kGobOutOfBounds is set there when a column value lies outside [chk_min, chk_max]
    unsigned long long *chk_flags;
    int64_t chk_min, chk_max;
};
struct DecodeDeltaBatch {
    DecodeDeltaJob job[kDecodeBatchMax];
    int32_t n;
};

}  // namespace sybl
