// scan_fast.h -- role-specialised scan kernels (the hot path of the hot path).
//
// k_scan (kernels.hip) interprets a per-slot flag word at run time; that generality costs
// ~260 VALU + ~240 SALU instructions per row and hundreds of scalar reloads of the plan.
// The common query shape -- NF int-range filter columns, NG int group columns, NA int
// aggregation columns, every column a fully populated int64 column whose bounds rule out
// rejects and outliers -- is compiled here as k_scan_fast<NF,NG,NA,MODE>: slot roles are
// template parameters, the plan is a small by-value kernel argument that stays in SGPRs,
// and the row body is ~50 VALU instructions.  Same reference semantics
// (aggregate.go:96-263, hist_basic.go:101-151), same cell-table layout, so k_fold, the
// all-reduce and finalize are shared with the generic kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"
#include "outlog.h"
#include "wg_header.h"

namespace sybl {

constexpr int kFastMaxF = 4, kFastMaxG = 4, kFastMaxA = 4;
// (the kernels with compile-time column counts -- k_scan_fast, k_scan_packed, k_emit*, k_scan_hash_fast -- are instantiated
// for <= 2 group and <= 2 aggregation columns; the run-time-count packed body, hashpacked.hip, takes kFastMaxG / kFastMaxA)
constexpr int kFastTemplatedG = 2, kFastTemplatedA = 2;

enum FastMode : int {
    kFastAvg = 0,      // op avg:  Count, sum(v)                        (extrema provably == initial values)
    kFastAvgMax = 1,   // op avg:  Count, sum(v), max(v)                (values >= 0 with a positive maximum)
    kFastMoments = 2,  // op hist, moments: Count, sum(v), sum(b), sum(b^2)
    kFastHist = 3,     // op hist, full:    Count, sum(v) in LDS + bucket arrays in HBM
    kFastModes = 4,
};

struct FastPlan {
    const int64_t *fcol[kFastMaxF];
    const int64_t *gcol[kFastMaxG];
    const int64_t *acol[kFastMaxA];
    int64_t lo[kFastMaxF], hi[kFastMaxF];  // inclusive
    int64_t gmin[kFastMaxG];
    uint32_t gcard[kFastMaxG];
    int32_t gstride[kFastMaxG];
    int64_t hmin[kFastMaxA];
    double inv_bucket[kFastMaxA];
    uint32_t bucket_size[kFastMaxA];
    int32_t n_values[kFastMaxA];
    int32_t f_sum[kFastMaxA], f_sb[kFastMaxA], f_sb2[kFastMaxA], m_max[kFastMaxA];
    // avg mode over a column with negative values tracks a MINIMUM too (BasicHist.Min starts at Go's zero value,
    // hist_basic.go:72-85), as max(-v); and an aggregation whose values never exceed 0 tracks no maximum.  ext_general says
    // that some aggregation of a kFastAvgMax query departs from "every aggregation tracks a maximum and nothing else":
    // the row bodies then test m_max / m_nmin per aggregation (wave-uniform) instead of assuming it (round 5: such
    // queries used to run the plan interpreter).
    int32_t m_nmin[kFastMaxA];
    int32_t ext_general;
    int64_t hist_agg_off[kFastMaxA];
    int64_t hist_off, hist_stride;
    int32_t n_cells, n_sum_fields, n_max_fields, rep_shift;
    // time series (aggregate.go:146-183) and the per-workgroup LDS window over time buckets
    const int64_t *tcol;
    int64_t time_bucket, tb_min;
    double inv_time_bucket;
    int32_t n_tb, tb_stride;
    int32_t windowed, lds_cells;   // lds_cells == n_cells unless windowed
    int32_t hist_lds;              // kFastHist: bucket arrays as uint32 in LDS (few cells), flushed once
    int32_t pad_;
    // GEN kernels only: validity bitmaps (nullptr = fully populated), int32 group ids (str columns),
    // missing-key cells, and the reject gate / per-aggregation counts of hist_basic.go:104
    const uint32_t *fvalid[kFastMaxF], *gvalid[kFastMaxG], *avalid[kFastMaxA], *tvalid;
    int32_t gmissing[kFastMaxG], gvalues[kFastMaxG];
    // hash group-by (k_scan_hash_fast, hashgroup.hip): the digits' weights in the 64-bit composite key
    int64_t gstride64[kFastMaxG], gmissing64[kFastMaxG], gvalues64[kFastMaxG], tb_stride64;
    // str filters (filter.go:199-250) as one bit per dictionary id: fmask != nullptr replaces the range
    const uint32_t *fmask[kFastMaxF];
    int32_t fmask_bits[kFastMaxF];
    // int `neq` filters (filter.go:171-195: up to kMaxNeq constants per column, next to or without a range);
    // pneq: the same constants as stored offsets for k_scan_packed<NUL> (one that no offset can equal is dropped)
    int32_t nneq[kFastMaxF], npneq[kFastMaxF];
    int64_t neq[kFastMaxF][kMaxNeq];
    uint32_t pneq[kFastMaxF][kMaxNeq];
    int32_t f_cnt[kFastMaxA], f_pop[kFastMaxA], f_smp[kFastMaxA], f_out[kFastMaxA];
    int64_t info_min[kFastMaxA], max10[kFastMaxA];
    const int64_t *wcol;           // weight column (OPTS.WEIGHT_COL, aggregate.go:100-102), fully populated
    int32_t f_samples;             // Result.Samples field when weighted, else -1
    // k_scan_packed, avg modes, every value populated: Result.Count rides in the high bits of aggregation 0's sum word --
    // the row adds (1 << cshift) + offset with ONE LDS atomic instead of two (the LDS atomic unit is what bounds these bodies:
    // profiles/r06_ldsrate.txt) -- and fast_finish takes the words apart replica by replica.  The planner sets it when a
    // replica's rows and their offsets' sum provably fit (planner.cpp: plan_count_packing); 0 = Count has a word of its own.
    int32_t cshift;
    // stored width (bytes: 1, 2, 4, 8) and value base per column: value = base + zero-extended raw
    // (canonical int64 columns: width 8, base 0 -- loaded as they are)
    int32_t fwid[kFastMaxF], gwid[kFastMaxG], awid[kFastMaxA], twid, wwid;
    int64_t fbase[kFastMaxF], gbase[kFastMaxG], abase[kFastMaxA], tbase, wbase;
    // k_scan_packed (scan_packed.h) works on the stored offsets: filter bounds, key digits, bucket
    // numerators and the time value rebased by the planner
    uint32_t plo[kFastMaxF], phi[kFastMaxF];   // offset range that passes (plo > phi: nothing does)
    uint32_t gdoff[kFastMaxG];                 // gbase - gmin   (mod 2^32)
    uint32_t adoff[kFastMaxA];                 // abase - h.Min  (mod 2^32)
    uint32_t tdoff;                            // tbase - tb_min * time_bucket
    // k_scan_hash_packed: take the late path (filter columns a tile ahead, the other columns loaded only for waves with a
    // passing row) -- it wins below ~0.5 % of the rows passing and costs 4-17 % of the kernel when most rows pass
    // (profiles/r06_selectivity_hash.txt), so the planner turns it on from the filters' ranges against the columns' extrema
    // (fill_packed; uniform values assumed: a wrong guess costs that much, never a result)
    int32_t late;
    double pinv_bucket[kFastMaxA], pinv_time;  // reciprocals scaled by (1 - 2^-40): never above the true quotient
    // k_scan_packed<NUL>: columns with missing rows / str ids / the Info.Min..Max*10 reject gate, rebased:
    // a value is accepted iff alo <= offset <= ahi
    uint32_t alo[kFastMaxA], ahi[kFastMaxA];
    int32_t nul, pad4_;
    const int32_t *wg_cell_base;
    // The row bitmap of the filter pre-pass (k_prefilter, kernels.hip): bit per physical row, 1 = the row passes the
    // filters the packed bodies do not evaluate themselves (set members, a fifth filter column); nullptr = none.  Read
    // by the NUL variants next to the validity words: a row whose bit is 0 fails like a row that fails a filter.
    const uint32_t *xvalid;
    int64_t *out_log;   // outlier log (plan.h), nullptr = not kept
    int64_t out_cap;
    int64_t *sum_out, *max_out, *ws_sum, *ws_max;
    const Segment *segs;
    const int32_t *wg_seg_begin;
};

hipError_t launch_scan_fast(const FastPlan &P, int nf, int ng, int na, int mode, bool time, bool gen, int n_wg,
                            size_t lds_bytes, hipStream_t st);

// ---- partitioned histograms (strategy 5) -------------------------------------------------
// Full-histogram queries over many cells cannot keep [cell][agg][bucket] in LDS, and one
// device-scope atomic per value runs at ~23 G atomics/s (config 4: 129 ms per 1e9 rows).
// Instead k_emit writes one 4-byte record per accepted (row, agg) into the buffer of the
// PARTITION that owns the value's (cell, agg) pair -- kPartCells pairs per partition -- and
// k_part_hist then lets one workgroup own a partition: its 32 bucket arrays, counts and sums
// live in LDS, are updated with LDS atomics and are written out with plain stores.
//
// The pass is a counting sort.  Records leave k_emit in CHUNKS of 16 (64 bytes, 64-byte aligned), one
// staging chunk per BIN in LDS (a bin = a partition, or one of 1 << sub_shift sub-bins of a partition
// chosen by the lane number when there are few partitions).  k_count first counts every workgroup's
// records per bin (it reads only the filter and key columns) and lays the workgroup's output out as one
// exactly sized REGION per bin, bins in order (boff); k_part_bases puts the workgroups' outputs behind
// each other (wbase).  A bin's g-th full chunk therefore has a fixed place -- region start + g -- and
// k_emit needs neither cursors nor device-scope atomics; the same bytes land in the same places on
// every run; a region's last chunk is padded with kRecSentinel.  k_part_hist walks the n_wg regions of
// its partition (the sub-bins of a partition are neighbours in a workgroup's output: one range).
//
// History (all measured on config 4, 1e9 rows): round 1 appended per-bin runs of ~9 records behind
// device-scope cursors (112 M atomics, 77 % partial-line writes, 10.3 GB written for 4 GB of records:
// 6.1 ms); round 2 made it a counting sort with partition-major regions, every lane copying the chunks
// it completed with four 16-byte stores (3.9 ms); round 3 found what that waited for (profiles/
// r03_cfg4_before_pmc.txt): 251 M 16-byte write requests from the L1s (four per chunk) with the L1's
// write path saturated 80 % of the time, and -- loads and stores retiring in order on one counter --
// a `s_waitcnt vmcnt(6)` for the column loads issued four tiles earlier that in fact waited for the
// stores of the previous tile.  Now four neighbouring lanes write one chunk (one 64-byte request), a
// wave writes its completed chunks with a FIXED number of store instructions per tile (lanes without a
// chunk aim past the end of the workgroup's buffer descriptor: dropped by the range check), so the
// compiler's count of younger operations is exact again and the column loads stay four tiles ahead
// (3.3 ms).  That left the vector ALUs 80 % busy (profiles/r03_cfg4_v3_pmc.txt), 40 % of it on records
// that found their bin's single staging chunk complete but not yet copied out (5 % of all records: one
// retry pass per tile and a carried record per lane).  So a bin now stages TWO chunks, generations
// alternating between them: a record only waits when its bin receives sixteen more records while a
// chunk is being copied out, and the retry path is cold.  The LDS for that comes from halving the bins:
// a partition is 64 (cell, agg) pairs, whose bucket arrays k_part_hist keeps as 16-bit counters.
constexpr int kPartCells = 64;       // (cell, agg) pairs per partition
constexpr int kPartCellBits = 6;
constexpr int kBucketBits = 10;      // len(Values) <= 1024
constexpr int kRecValueBits = 26;    // v - h.Min + BucketSize < (len(Values) + 1) * BucketSize < 2^26 (planner: select_part_hist)
constexpr int kMaxParts = 1024;      // LDS staging in k_emit: two 16-record chunks per bin
constexpr int kEmitMaxBins = 1024;   // bins = n_parts << sub_shift
constexpr uint32_t kEmitChunk = 16;  // records per chunk
constexpr uint32_t kEmitQueue = 32;  // completed chunks a wave copies out per drain (two store instructions) ...
constexpr uint32_t kEmitQueueMax = 64;  // ... four when a push holds eight records per lane (k_emit_packed, two tiles at a time); LDS is sized for this
constexpr int kEmitBinWords = 4 + 2 * (int)kEmitChunk;  // LDS words per bin: {cnt, wr0, wr1, region start} + two chunks
constexpr uint32_t kRecSentinel = 0u;               // padding record: never a real one -- a record's value part is v - h.Min + BucketSize >= 1
                                                    // -- and what a range-checked load past a region's end returns (k_part_hist counts it nowhere)
constexpr uint32_t kEmitDropOffset = 0x80000000u;   // byte offset past any workgroup's output (< 2 GB: planner)

struct EmitPlan {
    FastPlan fp;                     // columns, filters, group mapping, hmin / bucket geometry
    uint32_t *recs;                  // the workgroups' outputs, back to back (wbase)
    uint32_t *boff;                  // [n_wg][nb + 1] first chunk of a bin's region inside its workgroup's output
                                     // ([nb]: the output's chunks): k_count
    uint32_t *wbase;                 // [n_wg + 1] first chunk of a workgroup's output in recs: k_part_bases
    int32_t n_parts, n_aggs, n_wg;
    int32_t sub_shift;               // bins per partition = 1 << sub_shift (a lane's sub-bin follows from its lane
                                     // number, in k_count and k_emit alike, so that few partitions do not serialise
                                     // on a handful of LDS counters)
    int32_t quiet, store_nt;         // quiet: a later pass over the same rows (the third / fourth aggregation of a query,
                                     // Query::part_more): matched / overflow were counted by the first
                                     // store_nt: the chunk stores carry the non-temporal hint (round 5; SYBL_EMIT_PLAIN_STORES=1: off)
    int64_t *sum_out;                // header: matched / overflow
};

struct PartHistPlan {
    const uint32_t *recs;
    const uint32_t *boff, *wbase;    // as in EmitPlan
    int32_t n_parts, n_aggs, n_cells, nv_max;
    int32_t n_wg, sub_shift;
    int32_t split;                   // shares of a partition (> 1: results are combined with atomics)
    int32_t n_cus, tail_mode;        // (tail_mode: SYBL_PARTHIST_TAIL, diagnostic) the persistent kernel's grid: one workgroup per compute unit claims (partition, share) items
    uint32_t *wrap_log;              // [0]: entries used, [1]: items claimed (k_part_hist's work counter), then {pair, bucket | kind << 16}:
                                     // 16-bit counters that wrapped
    uint32_t wrap_cap;               // entries the log holds
    int32_t no_count;                // a later pass of a query with three or four aggregations: Result.Count of the cells
                                     // was written by the first (every aggregation accepts every row here)
    int32_t n_values[kFastMaxA], f_sum[kFastMaxA], m_max[kFastMaxA];
    int32_t f_out[kFastMaxA];        // outliers (a value beyond the last bucket, hist_basic.go:132-135): base of the cell's six
                                     // fields n, sum(o), sum(o^2) as four 32-bit limbs; -1: the column's bounds rule them out
    int32_t agg0, pad_;              // number of this pass's first aggregation in the query (the outlier log names it)
    int64_t *out_log;                // outlier log (plan.h), nullptr = not kept
    int64_t out_cap;
    int64_t hmin[kFastMaxA], bucket_size[kFastMaxA], hist_agg_off[kFastMaxA];
    double pinv_bucket[kFastMaxA];   // 1 / BucketSize scaled by (1 - 2^-40): the quotient estimate is never above the true one
    int64_t hist_off, hist_stride;
    int64_t *sum_out, *max_out;
    unsigned long long *trace;       // SYBL_PARTHIST_TRACE (diagnostic): [workgroup][kPartTraceWords] timestamps of the phases
};
constexpr int kPartTraceWords = 32;

// ---- -limit pushed into the scan (pushdown.hip): a printer's histogram query sorted by $COUNT
struct PushdownPlan {
    FastPlan fp;             // the key column (gcol / gwid / gdoff / gcard [0]), the value columns and their bucket geometry
    int32_t n_cells, n_aggs, n_wg, limit;
    uint32_t *ws;            // [n_wg][(n_cells + 1) / 2] the workgroups' counter tables, two 15-bit fields to a word
    uint32_t *carry;         // [n_cells] units of 32768 taken out of a field that filled up
    uint32_t *cnt;           // [n_cells] the folded counts
    uint32_t *bitmap;        // [(n_cells + 31) / 32] the printed cells
    int32_t *top_cells;      // [limit] ... as a list (any order)
    int32_t *n_top;
    int64_t *sum_out, *max_out, *total;  // the query's SUM / MAX sections and Cumulative's buckets (Query::d_total)
    int64_t hist_off, hist_stride;
};
hipError_t launch_pushdown_count(const PushdownPlan &D, hipStream_t st);  // pass 1: the groups' counts ...
hipError_t launch_pushdown_scan(const PushdownPlan &D, hipStream_t st);   // ... the printed cells chosen from them, pass 2

// tiles of column loads a lane keeps in flight: a tile is only 8..16 bytes per column and lane, and a CU needs
// ~64 KB on the way to keep HBM busy; bounded by registers (one 16-byte register quad per column and tile)
constexpr int emit_depth(int n_cols) { return n_cols <= 2 ? 4 : n_cols <= 4 ? 2 : 1; }
constexpr int count_depth(int n_cols) { return n_cols <= 1 ? 8 : n_cols <= 2 ? 4 : n_cols <= 4 ? 2 : 1; }
inline size_t emit_lds_bytes(const EmitPlan &E) {
    return ((size_t)E.n_parts << E.sub_shift) * kEmitBinWords * 4 + (size_t)(kWgThreads / 64) * kEmitQueueMax * 8;
}
inline size_t count_lds_bytes(const EmitPlan &E) { return (((size_t)E.n_parts << E.sub_shift) + 64) * 4; }

hipError_t launch_count(const EmitPlan &P, int nf, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_packed(const EmitPlan &P, int nf, int ng, int n_wg, hipStream_t st);
hipError_t launch_emit(const EmitPlan &P, int nf, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_part_hist(const PartHistPlan &P, hipStream_t st);
hipError_t launch_prefilter(const ScanPlan *d_plan, int n_slots, uint32_t *bits, int n_wg, hipStream_t st);
hipError_t launch_part_fix(const PartHistPlan &P, hipStream_t st);

#ifdef __HIPCC__

typedef long long fll2 __attribute__((ext_vector_type(2)));

template <int N>
struct FastTile {
    fll2 v[N > 0 ? N : 1];
    uint32_t pop[N > 0 ? N : 1];  // GEN kernels: validity bits of the two rows
};

typedef unsigned int fu32x2 __attribute__((ext_vector_type(2)));

typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kFastRsrcWord3 = 0x00020000;  // raw buffer descriptor, 32-bit data format (gfx9 family)

// GEN kernels read any stored width.  The loaded bits of a tile (two rows per lane) are kept raw and
// decoded only when the tile is consumed: the load instruction is the same 16-byte buffer load for
// every width -- its descriptor spans exactly the wave's 128 rows, so a narrow column's lanes read
// nothing beyond them -- which keeps the issue branch-free and the loads of a tile back to back.  (A
// width switch around the loads made the compiler merge the results through copies that wait for the
// data: the same query ran at 3.5 TB/s instead of 5.6.)
template <int N>
struct FastRaw {
    fu32x4 v[N > 0 ? N : 1];
    uint32_t pw[N > 0 ? N : 1];  // validity word holding the two rows' bits (all ones: fully populated)
};

__device__ __forceinline__ void fast_issue(const int64_t *col, int width, const uint32_t *valid, int64_t row0, uint32_t lane_row,
                                           fu32x4 &raw, uint32_t &pw) {
    const int ws = width == 8 ? 3 : width >> 1;  // log2(width)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)col + (row0 << ws)), 0,
                                                                          (int)((64u * kRowsPerThread) << ws), (int)kFastRsrcWord3);
    // (dword-aligned offset: two 1-byte rows are half a dword, picked apart in fast_decode)
    raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((lane_row << ws) & ~3u), 0, 2);
    pw = valid ? valid[(row0 + lane_row) >> 5] : 0xFFFFFFFFu;
}

__device__ __forceinline__ void fast_decode(int width, int64_t base, const fu32x4 &raw, uint32_t pw, int64_t row, fll2 &v, uint32_t &pop) {
    switch (width) {
    case 8:
        v.x = (long long)(((unsigned long long)raw.y << 32) | raw.x);
        v.y = (long long)(((unsigned long long)raw.w << 32) | raw.z);
        break;
    case 4:
        v.x = base + (long long)raw.x;
        v.y = base + (long long)raw.y;
        break;
    case 2:
        v.x = base + (long long)(raw.x & 0xFFFFu);
        v.y = base + (long long)(raw.x >> 16);
        break;
    default: {
        const uint32_t x = raw.x >> ((uint32_t)(row & 2) * 8u);  // rows 4k+2, 4k+3 sit in the upper half
        v.x = base + (long long)(x & 0xFFu);
        v.y = base + (long long)((x >> 8) & 0xFFu);
        break;
    }
    }
    pop = (pw >> (row & 31)) & 3u;
}

// GEN bodies name ~150 plan words; hoisted out of the tile loop they do not fit the 100 scalar registers and come back
// through v_readlane.  plan_fresh() hands out the kernel argument again behind a compiler barrier, so the words a stage
// uses are s_load'ed (scalar cache) where they are used.
template <bool FRESH>
__device__ __forceinline__ const FastPlan &plan_fresh(const FastPlan &P) {
    if (!FRESH) return P;
    typedef const char __attribute__((address_space(4))) *KP;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const FastPlan *)p;
}

template <int NF, int NG, int NA, bool TIME>
__device__ __forceinline__ void fast_issue_all(const FastPlan &P, int64_t row, FastRaw<NF> &f, FastRaw<NG> &g, FastRaw<NA> &a,
                                               FastRaw<1> &t, FastRaw<1> &w) {
    // the wave's first row (its first lane's) is wave-uniform: descriptor base; the lane's offset is 32-bit
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)row), hi = __builtin_amdgcn_readfirstlane((uint32_t)(row >> 32));
    const int64_t row0 = (int64_t)(((uint64_t)hi << 32) | lo);
    const uint32_t lane_row = (uint32_t)(row - row0);
    const FastPlan &Q = plan_fresh<true>(P);
    if (Q.wcol) fast_issue(Q.wcol, Q.wwid, nullptr, row0, lane_row, w.v[0], w.pw[0]);
    if (TIME) fast_issue(Q.tcol, Q.twid, Q.tvalid, row0, lane_row, t.v[0], t.pw[0]);
#pragma unroll
    for (int c = 0; c < NF; c++) fast_issue(Q.fcol[c], Q.fwid[c], Q.fvalid[c], row0, lane_row, f.v[c], f.pw[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) fast_issue(Q.gcol[c], Q.gwid[c], Q.gvalid[c], row0, lane_row, g.v[c], g.pw[c]);
#pragma unroll
    for (int c = 0; c < NA; c++) fast_issue(Q.acol[c], Q.awid[c], Q.avalid[c], row0, lane_row, a.v[c], a.pw[c]);
}

template <int NF, int NG, int NA, bool TIME>
__device__ __forceinline__ void fast_decode_all(const FastPlan &P, int64_t row, const FastRaw<NF> &rf, const FastRaw<NG> &rg,
                                                const FastRaw<NA> &ra, const FastRaw<1> &rt, const FastRaw<1> &rw, FastTile<NF> &f,
                                                FastTile<NG> &g, FastTile<NA> &a, FastTile<1> &t, FastTile<1> &w) {
    const FastPlan &Q = plan_fresh<true>(P);
    if (Q.wcol) fast_decode(Q.wwid, Q.wbase, rw.v[0], rw.pw[0], row, w.v[0], w.pop[0]);
    if (TIME) fast_decode(Q.twid, Q.tbase, rt.v[0], rt.pw[0], row, t.v[0], t.pop[0]);
#pragma unroll
    for (int c = 0; c < NF; c++) fast_decode(Q.fwid[c], Q.fbase[c], rf.v[c], rf.pw[c], row, f.v[c], f.pop[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) fast_decode(Q.gwid[c], Q.gbase[c], rg.v[c], rg.pw[c], row, g.v[c], g.pop[c]);
#pragma unroll
    for (int c = 0; c < NA; c++) fast_decode(Q.awid[c], Q.abase[c], ra.v[c], ra.pw[c], row, a.v[c], a.pop[c]);
}

// the plain kernels (and k_emit) are compiled for canonical, fully populated int64 columns: two rows = one
// 16-byte non-temporal load, nothing to decode
template <int NF, int NG, int NA, bool TIME>
__device__ __forceinline__ void fast_load(const FastPlan &P, int64_t row, FastTile<NF> &f, FastTile<NG> &g, FastTile<NA> &a,
                                          FastTile<1> &t) {
    if (TIME) t.v[0] = __builtin_nontemporal_load((const fll2 *)(P.tcol + row));
#pragma unroll
    for (int c = 0; c < NF; c++) f.v[c] = __builtin_nontemporal_load((const fll2 *)(P.fcol[c] + row));
#pragma unroll
    for (int c = 0; c < NG; c++) g.v[c] = __builtin_nontemporal_load((const fll2 *)(P.gcol[c] + row));
#pragma unroll
    for (int c = 0; c < NA; c++) a.v[c] = __builtin_nontemporal_load((const fll2 *)(P.acol[c] + row));
}

__device__ __forceinline__ void lds_add64(int64_t *lds, uint32_t idx, int64_t v) {
    __hip_atomic_fetch_add(lds + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One accumulator word: an LDS atomic (the workgroup's cell / staging table) or a device-scope one (the global table of
// a hash group-by, hashgroup.hip).
template <bool LDS>
__device__ __forceinline__ void fast_add64(int64_t *tab, uint64_t idx, int64_t v) {
    if (LDS) __hip_atomic_fetch_add(tab + (uint32_t)idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(tab + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS>
__device__ __forceinline__ void fast_max64(int64_t *tab, uint64_t idx, int64_t v) {
    int64_t *m = LDS ? tab + (uint32_t)idx : tab + idx;
    if (v > *m) {
        if (LDS) __hip_atomic_fetch_max(m, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_max(m, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// filters (aggregate.go:105-116), group key (aggregate.go:125-143), time bucket (:146-183) of one row.
// Returns 0: the row does not count any further (a filter failed, or -- after it was counted as matched -- it has no
// time value); 1: `cell` is the row's cell (KEY64: its composite key, hash group-by); 2: key / time bucket outside the
// declared bounds.  ng: the group columns actually present (== NG for the direct-mapped kernels).
template <int NF, int NG, bool TIME, bool GEN, bool KEY64>
__device__ __forceinline__ int fast_prepare(const FastPlan &P0, const FastTile<NF> &f, const FastTile<NG> &g, const FastTile<1> &t,
                                            const int r, const int nf, const int ng, uint64_t &cell, uint32_t &matched) {
    bool pass = true;
#pragma unroll
    for (int c = 0; c < NF; c++) {
        if (c >= nf) break;
        const FastPlan &P = plan_fresh<GEN>(P0);
        const int64_t x = r == 0 ? f.v[c].x : f.v[c].y;
        if (GEN && P.fmask[c]) {
            // StrFilter eq / neq / re / nre, evaluated per dictionary id on the host (filter.go:199-250)
            bool ok = false;
            if ((uint64_t)x < (uint64_t)P.fmask_bits[c]) ok = (P.fmask[c][x >> 5] >> (x & 31)) & 1u;
            pass = pass && ok;
        } else {
            pass = pass && x >= P.lo[c] && x <= P.hi[c];  // filter.go:171-195, folded to a range
        }
        if (GEN) {
            for (int k = 0; k < P.nneq[c]; k++) pass = pass && x != P.neq[c][k];
            pass = pass && ((f.pop[c] >> r) & 1u);  // an unpopulated value fails every filter
        }
    }
    if (!pass) return 0;
    matched += 1;  // aggregate.go:117
    cell = 0;
    bool inb = true;
#pragma unroll
    for (int c = 0; c < NG; c++) {
        if (c >= ng) break;
        const FastPlan &P = plan_fresh<GEN>(P0);
        const int64_t x = r == 0 ? g.v[c].x : g.v[c].y;
        if (GEN && !((g.pop[c] >> r) & 1u)) {
            // MISSING_VALUE key (aggregate.go:138): its own digit, or the digit of the value -1
            if (KEY64) {
                inb = inb && P.gmissing64[c] >= 0;
                cell += (uint64_t)P.gmissing64[c];
            } else {
                inb = inb && P.gmissing[c] >= 0;
                cell += (uint32_t)P.gmissing[c];
            }
            continue;
        }
        const uint64_t d = (uint64_t)x - (uint64_t)P.gmin[c];
        if (KEY64) {
            inb = inb && d < (uint64_t)P.gvalues64[c];
            cell += d * (uint64_t)P.gstride64[c];
        } else {
            inb = inb && d < (uint64_t)(GEN ? (uint32_t)P.gvalues[c] : P.gcard[c]);
            cell = (uint32_t)cell + (uint32_t)d * (uint32_t)P.gstride[c];  // aggregate.go:125-143 as a direct-mapped index
        }
    }
    const FastPlan &P = plan_fresh<GEN>(P0);
    if (TIME && GEN && !((t.pop[0] >> r) & 1u)) return 0;  // no time value: dropped after it was counted (aggregate.go:147-153)
    if (TIME) {
        // val = int(val) / TimeBucket * TimeBucket, truncating (aggregate.go:174); |t| < 2^51 here
        const int64_t tv = r == 0 ? t.v[0].x : t.v[0].y;
        const uint64_t ut = tv < 0 ? (uint64_t)0 - (uint64_t)tv : (uint64_t)tv;
        uint64_t qd = (uint64_t)((double)ut * P.inv_time_bucket);
        const int64_t rem = (int64_t)(ut - qd * (uint64_t)P.time_bucket);
        if (rem < 0) {
            qd -= 1;
        } else if (rem >= P.time_bucket) {
            qd += 1;
        }
        const int64_t tb = (tv < 0 ? -(int64_t)qd : (int64_t)qd) - P.tb_min;
        inb = inb && (uint64_t)tb < (uint64_t)P.n_tb;
        if (KEY64) cell += (uint64_t)tb * (uint64_t)P.tb_stride64;
        else cell = (uint32_t)cell + (uint32_t)tb * (uint32_t)P.tb_stride;
    }
    return inb ? 1 : 2;
}

// Count / Samples and the aggregations of one matched row (aggregate.go:202-261, hist_basic.go:101-151) into a table
// laid out [field][ncell << rs] (+ the MAX fields from max_base on); cidx = (local cell << rs) + lane replica.  gcell:
// the cell's number in the global table (bucket arrays, [gcell][hist_stride]); lcell: its place in the workgroup's LDS
// bucket arrays (hist_lds); logkey: what the outlier log calls the group.
template <int NA, int MODE, bool GEN, bool LDS>
__device__ __forceinline__ void fast_accumulate(const FastPlan &P0, const FastTile<NA> &a, const FastTile<1> &w, const int r,
                                                int64_t *tab, int64_t *maxtab, const uint64_t ncell, const uint32_t rs, const uint64_t cidx,
                                                const int64_t gcell, const uint32_t lcell, const int64_t logkey, uint32_t *hist32,
                                                uint32_t &overflow) {
    // weight := r.Ints[WEIGHT_COL] (aggregate.go:100-102); 1 without a weight column -- and then no 64-bit multiply
    // per accumulated word (the branch is wave-uniform; the GEN body spent three of them per aggregation on wt == 1)
    const FastPlan &Ph = plan_fresh<GEN>(P0);
    const bool weighted = GEN && Ph.wcol != nullptr;
    const int64_t wt = weighted ? (r == 0 ? w.v[0].x : w.v[0].y) : 1;
    auto times_w = [&](int64_t v) -> int64_t { return weighted ? (int64_t)((uint64_t)v * (uint64_t)wt) : v; };
    fast_add64<LDS>(tab, cidx, wt);  // Result.Count += weight (aggregate.go:203)
    if (GEN && Ph.f_samples >= 0) fast_add64<LDS>(tab, (((uint64_t)(uint32_t)Ph.f_samples * ncell) << rs) + cidx, 1);  // Result.Samples++
#pragma unroll
    for (int c = 0; c < NA; c++) {
        const FastPlan &P = plan_fresh<GEN>(P0);
        const int64_t x = r == 0 ? a.v[c].x : a.v[c].y;
        if (GEN) {
            if (!((a.pop[c] >> r) & 1u)) continue;
            if (P.f_pop[c] >= 0) fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_pop[c] * ncell) << rs) + cidx, 1);
            if (P.f_cnt[c] >= 0) {
                if (x > P.max10[c] || x < P.info_min[c]) continue;  // hist_basic.go:104
                fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_cnt[c] * ncell) << rs) + cidx, wt);  // h.Count += weight
            }
            if (P.f_smp[c] >= 0) fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_smp[c] * ncell) << rs) + cidx, 1);  // h.Samples++
        }
        fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_sum[c] * ncell) << rs) + cidx, times_w(x));
        if (MODE == kFastAvgMax) {
            if (!P.ext_general) {
                fast_max64<LDS>(maxtab, (((uint64_t)(uint32_t)P.m_max[c] * ncell) << rs) + cidx, x);
            } else {
                if (P.m_max[c] >= 0) fast_max64<LDS>(maxtab, (((uint64_t)(uint32_t)P.m_max[c] * ncell) << rs) + cidx, x);
                if (P.m_nmin[c] >= 0) fast_max64<LDS>(maxtab, (((uint64_t)(uint32_t)P.m_nmin[c] * ncell) << rs) + cidx, x == INT64_MIN ? INT64_MAX : -x);
            }
        }
        if (MODE == kFastMoments || MODE == kFastHist) {
            // bucket_value := (value - h.Min) / BucketSize, hist_basic.go:130.  The planner only
            // selects this kernel when 0 <= value - h.Min < 2^32 and no value can reach
            // len(Values), so one f64 multiply + a correction step is exact.
            const uint32_t n = (uint32_t)((uint64_t)x - (uint64_t)P.hmin[c]);
            uint32_t b = (uint32_t)((double)n * P.inv_bucket[c]);
            const int32_t rem = (int32_t)(n - b * P.bucket_size[c]);
            if (rem < 0) {
                b -= 1;
            } else if ((uint32_t)rem >= P.bucket_size[c]) {
                b += 1;
            }
            if (GEN) {
                // h.Max starts at Info.Max: track only when the column can exceed it
                if (P.m_max[c] >= 0) fast_max64<LDS>(maxtab, (((uint64_t)(uint32_t)P.m_max[c] * ncell) << rs) + cidx, x);
                if (b >= (uint32_t)P.n_values[c]) {
                    // Outlier (hist_basic.go:132-135): clipped into the last bucket AND remembered as
                    // exact n, sum(o), sum(o^2) (four 32-bit limbs)
                    if (P.f_out[c] >= 0) {
                        const uint64_t step = ncell << rs;
                        const uint64_t fi = (((uint64_t)(uint32_t)P.f_out[c] * ncell) << rs) + cidx;
                        const unsigned __int128 sq = (unsigned __int128)((__int128)x * (__int128)x);
                        fast_add64<LDS>(tab, fi, 1);
                        fast_add64<LDS>(tab, fi + step, x);
                        fast_add64<LDS>(tab, fi + 2 * step, (int64_t)(uint64_t)(sq & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fi + 3 * step, (int64_t)(uint64_t)((sq >> 32) & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fi + 4 * step, (int64_t)(uint64_t)((sq >> 64) & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fi + 5 * step, (int64_t)(uint64_t)(sq >> 96));
                        if (P.out_log) log_outlier(P.out_log, P.out_cap, logkey, c, x);
                    } else {
                        overflow += 1;
                    }
                    b = (uint32_t)P.n_values[c] - 1;
                }
            }
            if (MODE == kFastMoments) {
                fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_sb[c] * ncell) << rs) + cidx, times_w((int64_t)b));
                fast_add64<LDS>(tab, (((uint64_t)(uint32_t)P.f_sb2[c] * ncell) << rs) + cidx, times_w((int64_t)((uint64_t)b * (uint64_t)b)));
            } else if (LDS && P.hist_lds) {
                __hip_atomic_fetch_add(hist32 + lcell * (uint32_t)P.hist_stride + (uint32_t)P.hist_agg_off[c] + b, 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_fetch_add(P.sum_out + P.hist_off + gcell * P.hist_stride + P.hist_agg_off[c] + b,
                                       (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int NF, int NG, int NA, int MODE, bool TIME, bool GEN>
__device__ __forceinline__ void fast_row(const FastPlan &P, const FastTile<NF> &f, const FastTile<NG> &g,
                                         const FastTile<NA> &a, const FastTile<1> &t, const FastTile<1> &w, const int r,
                                         int64_t *lds,
                                         const uint32_t rep, const uint32_t max_base, const uint32_t cell_base,
                                         uint32_t *hist32, uint32_t &matched, uint32_t &overflow) {
    uint64_t cell64;
    const int st = fast_prepare<NF, NG, TIME, GEN, false>(P, f, g, t, r, NF, NG, cell64, matched);
    if (st == 0) return;
    const uint32_t cell = (uint32_t)cell64;
    const uint32_t ncell = (uint32_t)P.lds_cells;
    const uint32_t lcell = cell - cell_base;  // position inside this workgroup's LDS table
    if (st == 2 || lcell >= ncell) {
        overflow += 1;
        return;
    }
    const uint32_t rs = (uint32_t)P.rep_shift;
    const uint32_t cidx = (lcell << rs) + rep;
    fast_accumulate<NA, MODE, GEN, true>(P, a, w, r, lds, lds + max_base, (uint64_t)ncell, rs, (uint64_t)cidx, (int64_t)cell, lcell, (int64_t)cell,
                                         hist32, overflow);
}

// LDS layout of one workgroup: [sum fields][max fields][uint32 bucket arrays (hist_lds)], every
// field replicated 1 << rep_shift times per cell.
struct FastLds {
    uint32_t tab_cells, words_sum, words_max, max_base, cell_base, hist_words, rep;
    uint32_t *hist32;
};

// (T: threads of the workgroup -- the strides of the table loops)
template <int MODE, int T = kWgThreads>
__device__ __forceinline__ FastLds fast_begin(const FastPlan &P, int64_t *lds, const bool max32 = false) {
    FastLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t R = 1u << P.rep_shift;
    L.tab_cells = (uint32_t)P.lds_cells;
    L.words_sum = (uint32_t)P.n_sum_fields * L.tab_cells;
    L.words_max = (uint32_t)P.n_max_fields * L.tab_cells;
    L.max_base = L.words_sum << P.rep_shift;
    L.cell_base = P.windowed ? (uint32_t)P.wg_cell_base[blockIdx.x] : 0u;
    for (uint32_t i = tid; i < L.words_sum * R; i += T) lds[i] = 0;
    // (max32, k_scan_packed in avg mode: the MAX words hold uint32 OFFSETS in their low halves -- see fast_finish)
    for (uint32_t i = tid; i < L.words_max * R; i += T) lds[L.max_base + i] = max32 ? 0 : INT64_MIN;
    L.hist32 = (uint32_t *)(lds + L.max_base + (L.words_max << P.rep_shift));
    L.hist_words = (MODE == kFastHist && P.hist_lds) ? L.tab_cells * (uint32_t)P.hist_stride : 0u;
    for (uint32_t i = tid; i < L.hist_words; i += T) L.hist32[i] = 0;
    L.rep = tid & (R - 1);
    __syncthreads();
    return L;
}

// Publishes the workgroup's results: matched / overflow counters, LDS bucket arrays, and the cell
// table -- flushed with atomics (LDS window) or stored to the workgroup's slice for k_fold.
// max32 (k_scan_packed, avg mode): the MAX words hold the largest stored OFFSET of the field's aggregation in their low half; the
// maximum is abase[a] + that offset where the cell's Count is not zero, and untouched (INT64_MIN) where it is.  FastPlan::cshift: Count and aggregation 0's sum of offsets share a word per replica.
template <int T = kWgThreads>
__device__ __forceinline__ void fast_finish(const FastPlan &P, int64_t *lds, const FastLds &L, uint32_t matched, uint32_t overflow, const bool max32 = false) {
    const uint32_t tid = threadIdx.x;
    const uint32_t R = 1u << P.rep_shift;
    const uint32_t words_sum = L.words_sum, words_max = L.words_max, tab_cells = L.tab_cells;
    const uint32_t cshift = (uint32_t)P.cshift, f_sum0 = (uint32_t)P.f_sum[0];
    auto sum_of = [&](uint32_t i) -> int64_t {  // word i of the SUM section, replicas folded
        if (cshift) {
            const uint32_t fi = i / L.tab_cells, c = i - fi * L.tab_cells;
            if (fi == 0 || fi == f_sum0) {
                // (replica by replica: the replicas' sums added up may carry into the count's bits)
                uint64_t cnt = 0, sm = 0;
                const uint64_t mask = ((uint64_t)1 << cshift) - 1;
                for (uint32_t k = 0; k < R; k++) {
                    const uint64_t w = (uint64_t)lds[((f_sum0 * L.tab_cells + c) << P.rep_shift) + k];
                    cnt += w >> cshift;
                    sm += w & mask;
                }
                return fi == 0 ? (int64_t)cnt : (int64_t)(sm + cnt * (uint64_t)P.abase[0]);
            }
        }
        int64_t acc = 0;
        for (uint32_t k = 0; k < R; k++) acc += lds[(i << P.rep_shift) + k];
        return acc;
    };
    auto max_of = [&](uint32_t i) -> int64_t {  // word i of the MAX section
        if (max32) {
            const uint32_t fi = i / L.tab_cells, c = i - fi * L.tab_cells;
            if (sum_of(c) == 0) return INT64_MIN;  // (field 0: the cell's Count)
            int64_t base = 0;
#pragma unroll
            for (int a = kFastMaxA - 1; a >= 0; a--)  // (downwards: the entries behind the query's aggregations are zeros)
                if ((uint32_t)P.m_max[a] == fi) base = P.abase[a];
            uint32_t m = 0;
            for (uint32_t k = 0; k < R; k++) m = max(m, (uint32_t)(uint64_t)lds[L.max_base + (i << P.rep_shift) + k]);
            return (int64_t)((uint64_t)base + (uint64_t)m);
        }
        int64_t acc = INT64_MIN;
        for (uint32_t k = 0; k < R; k++) {
            const int64_t b = lds[L.max_base + (i << P.rep_shift) + k];
            acc = b > acc ? b : acc;
        }
        return acc;
    };
    // matched rows: one add per wave (64-wide butterfly)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        matched += __shfl_xor(matched, o, 64);
        overflow += __shfl_xor(overflow, o, 64);
    }
    {
        const int slot[2] = {kHdrMatched, kHdrOverflow};
        const int64_t v[2] = {(int64_t)matched, (int64_t)overflow};
        wg_header_add<2>(P.sum_out, slot, v);  // (one atomic per workgroup and counter, not per wave: scan_generic.h)
    }

    __syncthreads();
    // LDS bucket arrays: one atomic per touched bucket per workgroup into the zeroed global table
    for (uint32_t i = tid; i < L.hist_words; i += T) {
        const uint32_t x = L.hist32[i];
        if (x) __hip_atomic_fetch_add(P.sum_out + P.hist_off + (int64_t)L.cell_base * P.hist_stride + i, (int64_t)x, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.windowed) {
        // flush the touched cells of this workgroup's window into the global table
        int64_t *gs = P.sum_out + kHeaderWords;
        for (uint32_t i = tid; i < words_sum; i += T) {
            const int64_t acc = sum_of(i);
            if (acc != 0) {
                const uint32_t fi = i / tab_cells, c = i - fi * tab_cells;
                __hip_atomic_fetch_add(gs + (int64_t)fi * P.n_cells + L.cell_base + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        for (uint32_t i = tid; i < words_max; i += T) {
            const int64_t acc = max_of(i);
            if (acc != INT64_MIN) {
                const uint32_t fi = i / tab_cells, c = i - fi * tab_cells;
                __hip_atomic_fetch_max(P.max_out + (int64_t)fi * P.n_cells + L.cell_base + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // fold the lane replicas and publish this workgroup's table (plain stores)
    int64_t *ws = P.ws_sum + (int64_t)blockIdx.x * words_sum;
    for (uint32_t i = tid; i < words_sum; i += T) ws[i] = sum_of(i);
    int64_t *wm = P.ws_max + (int64_t)blockIdx.x * words_max;
    for (uint32_t i = tid; i < words_max; i += T) wm[i] = max_of(i);
}

template <int NF, int NG, int NA, int MODE, bool TIME, bool GEN>
__global__ __launch_bounds__(kWgThreads, 4) void k_scan_fast(const FastPlan P) {
    extern __shared__ int64_t lds[];
    const uint32_t tid = threadIdx.x;
    const FastLds L = fast_begin<MODE>(P, lds);
    const uint32_t rep = L.rep, max_base = L.max_base, cell_base = L.cell_base;
    uint32_t *hist32 = L.hist32;

    uint32_t matched = 0, overflow = 0;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        int64_t row = seg.start + (int64_t)tid * kRowsPerThread;
        FastTile<NF> f0, f1;
        FastTile<NG> g0, g1;
        FastTile<NA> a0, a1;
        FastTile<1> t0, t1, w0;
        if (GEN) {
            // the next tile's raw bits are in flight while this one is consumed; they are decoded (the
            // first use of the loaded registers) only after the rows below
            FastRaw<NF> rf;
            FastRaw<NG> rg;
            FastRaw<NA> ra;
            FastRaw<1> rt, rw;
            if (row < end) {
                fast_issue_all<NF, NG, NA, TIME>(P, row, rf, rg, ra, rt, rw);
                fast_decode_all<NF, NG, NA, TIME>(P, row, rf, rg, ra, rt, rw, f0, g0, a0, t0, w0);
            }
            for (; row < end; row += kTileRows) {
                const int64_t nrow = row + kTileRows;
                if (nrow < end) fast_issue_all<NF, NG, NA, TIME>(P, nrow, rf, rg, ra, rt, rw);
                fast_row<NF, NG, NA, MODE, TIME, GEN>(P, f0, g0, a0, t0, w0, 0, lds, rep, max_base, cell_base, hist32, matched, overflow);
                if (row + 1 < end)
                    fast_row<NF, NG, NA, MODE, TIME, GEN>(P, f0, g0, a0, t0, w0, 1, lds, rep, max_base, cell_base, hist32, matched, overflow);
                if (nrow < end) fast_decode_all<NF, NG, NA, TIME>(P, nrow, rf, rg, ra, rt, rw, f0, g0, a0, t0, w0);
            }
            continue;
        }
        // register double buffer: the next tile's loads are in flight while this one is consumed
        if (row < end) fast_load<NF, NG, NA, TIME>(P, row, f0, g0, a0, t0);
        for (; row < end; row += kTileRows) {
            const int64_t nrow = row + kTileRows;
            if (nrow < end) fast_load<NF, NG, NA, TIME>(P, nrow, f1, g1, a1, t1);
            fast_row<NF, NG, NA, MODE, TIME, GEN>(P, f0, g0, a0, t0, w0, 0, lds, rep, max_base, cell_base, hist32, matched, overflow);
            if (row + 1 < end)
                fast_row<NF, NG, NA, MODE, TIME, GEN>(P, f0, g0, a0, t0, w0, 1, lds, rep, max_base, cell_base, hist32, matched, overflow);
            f0 = f1;
            g0 = g1;
            a0 = a1;
            t0 = t1;
        }
    }
    fast_finish(P, lds, L, matched, overflow);
}

// k_emit: filters + cell index exactly as k_scan_fast, but instead of accumulating it appends
// rec = local pair << 27 | (v - h.Min) to the owning partition (the bucket divide happens in k_part_hist).
//
// LDS staging of k_emit / k_emit_packed, per bin (nb = n_parts << sub_shift bins):
//   cnt    records pushed so far: a push takes slot s = cnt++, generation g = s / 16, which uses chunk g & 1
//          of the bin for the (g / 2 + 1)-th time
//   wr[h]  of chunk h: 17 * (generations copied out of it) + records of its current generation in place: a
//          generation-g record may be written once wr[g & 1] >= 17 (g / 2) (generation g - 2 is out of the chunk),
//          each writer then bumps it, and the one that finds 17 (g / 2) + 15 has completed the chunk: its wave
//          copies it to chunk g of the bin's region and bumps wr once more (-> 17 (g / 2 + 1))
//   start  the bin's region inside the workgroup's output (in chunks; {cnt, wr0, wr1, start} are one 16-byte read)
//   chunks 2 x 16 records; a record's place is rotated by the bin number so bins that fill in step spread over
//          the LDS banks (the order of records inside a chunk is immaterial)
// and per wave a QUEUE of the chunks its last writes completed.
// There is no workgroup barrier between the prologue and the final drain: waves run free.  A record whose chunk
// still holds generation g - 2 (its bin received 16 more records while that was being copied out: rare) makes its
// wave poll -- every lane of the wave keeps retrying everything it holds; the oldest incomplete generation of a bin
// can always be written and its sixteenth writer's wave copies it out at once, so the protocol cannot deadlock.
// (Tried and dropped, measured on config 4: dedicated store waves polling wr -- 4.9 ms against 3.8: the
// polling costs more LDS bandwidth and issue slots than the stores cost the scanning waves; a barrier per
// tile instead of the publication counters -- 5.0 ms.)
struct EmitLds {
    uint32_t *chunk;        // [nb][2][16]
    uint32_t *cnt, *start;  // [nb] each: arrays of their own, so that the bins spread over all 32 LDS banks (as one
    uint2 *wr;              // [nb]   16-byte struct per bin the counters of 64 lanes fell into 8 banks: 71 % of the LDS
                            //        pipe's cycles were bank conflicts, profiles/r03_cfg4_v4_pmc.txt)
    uint2 *queue;           // [kEmitQueueMax] of this wave: {bin << 1 | chunk of the bin, generation = the chunk's number in the bin's region}
    uint32_t *out;          // the workgroup's output in recs
    uint32_t out_bytes;
    uint32_t nb, ss, sub;
    uint32_t nt;            // EmitPlan::store_nt
};

__device__ __forceinline__ EmitLds emit_begin(const EmitPlan &E, uint32_t *elds) {
    EmitLds S;
    const uint32_t tid = threadIdx.x;
    S.ss = (uint32_t)E.sub_shift;
    S.nt = (uint32_t)E.store_nt;
    S.nb = (uint32_t)E.n_parts << S.ss;      // staging bins
    S.sub = tid & ((1u << S.ss) - 1);        // this lane's sub-bin
    S.chunk = elds;                          // [nb][2][16]  (128-byte rows)
    S.wr = (uint2 *)(elds + S.nb * 2 * kEmitChunk);
    uint2 *queues = S.wr + S.nb;             // [waves][kEmitQueueMax]
    S.queue = queues + (tid >> 6) * kEmitQueueMax;
    S.cnt = (uint32_t *)(queues + (kWgThreads / 64) * kEmitQueueMax);
    S.start = S.cnt + S.nb;
    const uint32_t *boff = E.boff + (size_t)blockIdx.x * (S.nb + 1);
    for (uint32_t i = tid; i < S.nb; i += kWgThreads) {
        S.cnt[i] = 0;
        S.wr[i] = make_uint2(0u, 0u);
        S.start[i] = boff[i];
    }
    if (tid < (kWgThreads / 64) * kEmitQueueMax) queues[tid] = make_uint2(0u, 0u);
    // wbase[w] = the chunks of the workgroups before w (boff[w'][nb] is w's own total: k_count).  Every workgroup adds them
    // up for itself -- a few hundred words -- and leaves the result where k_part_hist looks for it (round 3: a one-workgroup
    // kernel of its own between k_count and k_emit, i.e. two more kernel boundaries per scan).
    uint32_t before = 0;
    for (uint32_t w = tid; w < blockIdx.x; w += kWgThreads) before += E.boff[(size_t)w * (S.nb + 1) + S.nb];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    uint32_t *wave_part = (uint32_t *)S.queue;  // (this wave's queue, still unused: one word of it)
    if ((tid & 63u) == 0) wave_part[0] = before;
    __syncthreads();
    before = 0;
#pragma unroll
    for (uint32_t v = 0; v < kWgThreads / 64; v++) before += ((const uint32_t *)(queues + v * kEmitQueueMax))[0];
    __syncthreads();
    if ((tid & 63u) == 0) S.queue[0] = make_uint2(0u, 0u);
    if (tid == 0) E.wbase[blockIdx.x] = before;
    S.out = E.recs + (size_t)before * kEmitChunk;
    S.out_bytes = boff[S.nb] * (kEmitChunk * 4u);
    __syncthreads();
    return S;
}

// LDS operations of a wave are carried out in issue order, so ordering two of them only takes keeping the
// compiler from moving them and -- where a returned value gates the next step -- the wait for the LDS
// counter.  (A workgroup-scope fence also waits for vmcnt(0), i.e. for the prefetched column loads.)
__device__ __forceinline__ void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }  // compiler-only
__device__ __forceinline__ void lds_wait() {
    lds_order();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // gfx9 encoding of lgkmcnt(0) alone (vmcnt / expcnt fields at their maxima)
    lds_order();
}

// Copies out the chunks this wave's writes completed (bit i of `full`: the lane's record i was the sixteenth of its
// generation; which[i] = bin << 1 | chunk of the bin, dest[i] = its place in the workgroup's output).  The owners queue
// {which, dest}; then lanes 4g .. 4g+3 copy queue entry g: each reads one 16-byte quarter of the staged chunk and the
// four stores of a quad are ONE 64-byte write request (the lanes' own stores of round 2 were four 16-byte requests per
// chunk, and their number saturated the L1's write path).  The stores are raw buffer stores into the workgroup's
// output: a lane without an entry aims at kEmitDropOffset and the descriptor's range check drops it, so every pass
// through here issues exactly kEmitQueue / 16 store instructions -- with loads and stores retiring in order on one
// counter, a data-dependent number of stores between the column loads and their use makes the compiler's
// `s_waitcnt vmcnt(N)` wait for the stores of the current tile instead of the loads of four tiles ago.
template <int M, uint32_t Q = kEmitQueue>
__device__ __forceinline__ void emit_copy_full(const EmitLds &S, const uint32_t (&which)[M], const uint32_t (&gen)[M], uint32_t full) {
    const uint32_t lane = threadIdx.x & 63u, g = lane >> 2, j = lane & 3u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)S.out, 0, (int)S.out_bytes, (int)0x00020000);
    for (;;) {
        uint32_t nq = 0;  // wave-uniform
        for (;;) {
            const bool has = full != 0;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
            if (!m) break;
            const uint32_t at = nq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (has && at < Q) {
                const uint32_t i = (uint32_t)__builtin_ctz(full);
                full &= full - 1;
                uint32_t b = which[0], d = gen[0];
#pragma unroll
                for (int k = 1; k < M; k++) {
                    b = i == (uint32_t)k ? which[k] : b;
                    d = i == (uint32_t)k ? gen[k] : d;
                }
                S.queue[at] = make_uint2(b, d);
            }
            nq += (uint32_t)__builtin_popcountll(m);
            if (nq >= Q) {
                nq = Q;
                break;
            }
        }
        lds_order();
        // (two queue entries per lane at a time: with four the staged pieces alone are 16 registers)
#pragma unroll
        for (uint32_t k0 = 0; k0 < Q / 16; k0 += 2) {
            uint2 qe[2];
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) qe[k] = S.queue[(k0 + k) * 16 + g];
            lds_wait();
            fu32x4 piece[2];
            uint32_t st[2];
            bool valid[2];
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                valid[k] = (k0 + k) * 16 + g < nq;
                const uint32_t c = valid[k] ? qe[k].x : 0u;
                piece[k] = ((const fu32x4 *)(S.chunk + c * kEmitChunk))[j];
                // the chunk's place: the bin's region start + its generation (read here, once per chunk, not once per record)
                st[k] = S.start[c >> 1];
            }
            lds_wait();  // (also: the chunks have been read before their next generation may write)
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                const uint32_t off = valid[k] ? (st[k] + qe[k].y) * (kEmitChunk * 4u) + j * 16u : kEmitDropOffset;
                if (S.nt) __builtin_amdgcn_raw_buffer_store_b128(piece[k], rsrc, (int)off, 0, 2);  // (wave-uniform; aux 2 = nt)
                else __builtin_amdgcn_raw_buffer_store_b128(piece[k], rsrc, (int)off, 0, 0);
            }
#pragma unroll
            for (uint32_t k = 0; k < 2; k++)
                if (valid[k] && j == 0)
                    __hip_atomic_fetch_add((uint32_t *)S.wr + qe[k].x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lds_order();
        if (!__builtin_amdgcn_ballot_w64(full != 0)) break;
    }
}

// The store instructions of one drain with every lane aimed past the end: k_emit's prologue issues them behind each
// tile of column loads, so that the ring of loads and stores the compiler counts on the way into the tile loop is the
// one every later round has (otherwise the waits of the whole loop are sized for the first round: three tiles of
// loads and no stores between a tile's loads and their use).
template <uint32_t Q = kEmitQueue>
__device__ __forceinline__ void emit_pad_stores(const EmitLds &S) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)S.out, 0, (int)S.out_bytes, (int)0x00020000);
    const fu32x4 none = {0u, 0u, 0u, 0u};
#pragma unroll
    for (uint32_t k = 0; k < Q / 16; k++) __builtin_amdgcn_raw_buffer_store_b128(none, rsrc, (int)(kEmitDropOffset + 16u * k), 0, 0);  // (distinct, or they merge)
}

// Pushes the lane's records i with act[i] set, rec[i] into bin[i].  The first pass -- all there is for nearly every
// tile -- keeps its predicates as booleans (lane masks in scalar registers); only a wave with a record whose chunk still
// holds the generation before last goes on to the bit-mask bookkeeping of the retry loop.
template <int N, uint32_t Q = kEmitQueue>
__device__ __forceinline__ void emit_push_all(const EmitPlan &E, const EmitLds &S, const uint32_t (&bin)[N], const uint32_t (&rec)[N],
                                              const bool (&act)[N]) {
    uint32_t slot[N], gens[N], which[N], thr[N], pos[N];
    unsigned long long w[N];  // {wr0, wr1}
    bool pend[N];
    // one LDS round trip in the common case: the slots and the bins' {wr0, wr1} (wr only grows: a value read early errs
    // on the side of waiting)
#pragma unroll
    for (int i = 0; i < N; i++)
        slot[i] = act[i] ? __hip_atomic_fetch_add(S.cnt + bin[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
    lds_order();
#pragma unroll
    for (int i = 0; i < N; i++) {
        w[i] = act[i] ? __hip_atomic_load((const unsigned long long *)(S.wr + bin[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
    }
    // ---- first pass
    lds_order();
    bool ok[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t gen = slot[i] >> 4, h = gen & 1u;
        thr[i] = 17u * (gen >> 1);
        which[i] = bin[i] << 1 | h;
        gens[i] = gen;
        pos[i] = which[i] * kEmitChunk + ((slot[i] + bin[i]) & (kEmitChunk - 1));
        ok[i] = act[i] && (uint32_t)(w[i] >> (h << 5)) >= thr[i];
        if (ok[i]) S.chunk[pos[i]] = rec[i];
    }
    lds_order();  // the records go to LDS before wr says so (issue order)
    uint32_t old[N];
#pragma unroll
    for (int i = 0; i < N; i++)
        old[i] = ok[i] ? __hip_atomic_fetch_add((uint32_t *)S.wr + which[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
    uint32_t full = 0;
    bool left = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        full |= ok[i] && old[i] == thr[i] + (kEmitChunk - 1) ? 1u << i : 0u;
        pend[i] = act[i] && !ok[i];
        left = left || pend[i];
    }
    lds_order();  // the chunk is read after wr showed the other 15 records in place
    emit_copy_full<N, Q>(S, which, gens, full);
    if (!__builtin_amdgcn_ballot_w64(left)) return;
    // ---- some lane of the wave holds a record whose chunk is still waiting to be copied out (rare)
    uint32_t pmask = 0;
#pragma unroll
    for (int i = 0; i < N; i++) pmask |= pend[i] ? 1u << i : 0u;
    uint32_t passes = 0;
    while (__builtin_amdgcn_ballot_w64(pmask != 0)) {
        // poll (bounded: a protocol bug must surface as an error from finalize, not as a hung GPU)
        if (++passes > (1u << 22)) {
            __hip_atomic_fetch_add(E.sum_out + kHdrEmitStall, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        uint32_t wr[N];
#pragma unroll
        for (int i = 0; i < N; i++)
            wr[i] = (pmask >> i) & 1u ? __hip_atomic_load((uint32_t *)S.wr + which[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
        lds_order();
        uint32_t okm = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (((pmask >> i) & 1u) && wr[i] >= thr[i]) {
                S.chunk[pos[i]] = rec[i];
                okm |= 1u << i;
            }
        }
        lds_order();
#pragma unroll
        for (int i = 0; i < N; i++)
            old[i] = (okm >> i) & 1u ? __hip_atomic_fetch_add((uint32_t *)S.wr + which[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
        full = 0;
#pragma unroll
        for (int i = 0; i < N; i++) full |= ((okm >> i) & 1u) && old[i] == thr[i] + (kEmitChunk - 1) ? 1u << i : 0u;
        lds_order();
        emit_copy_full<N, Q>(S, which, gens, full);
        pmask &= ~okm;
    }
}

__device__ __forceinline__ uint32_t emit_record(uint32_t pair, uint32_t n32) { return ((pair & (kPartCells - 1)) << kRecValueBits) | n32; }
__device__ __forceinline__ uint32_t emit_bin(const EmitLds &S, uint32_t pair) { return ((pair >> kPartCellBits) << S.ss) | S.sub; }

// Final drain: every bin's incomplete chunk goes out as the last chunk of its region, padded with sentinels.
__device__ __forceinline__ void emit_finish(const EmitPlan &E, const EmitLds &S, uint32_t matched, uint32_t overflow) {
    __syncthreads();
    for (uint32_t bin = threadIdx.x; bin < S.nb; bin += kWgThreads) {
        const uint32_t n = S.cnt[bin], left = n & (kEmitChunk - 1), gen = n >> 4;
        if (left) {
            const uint32_t *c = S.chunk + (bin << 1 | (gen & 1u)) * kEmitChunk;
            uint32_t r[kEmitChunk];
#pragma unroll
            for (uint32_t k = 0; k < kEmitChunk; k++) {
                // the record in place k was pushed as number (k - rotation) mod 16 of its generation
                const uint32_t logical = (k - bin) & (kEmitChunk - 1);
                r[k] = logical < left ? c[k] : kRecSentinel;
            }
            fu32x4 *dst = (fu32x4 *)(S.out + (size_t)(S.start[bin] + gen) * kEmitChunk);
            dst[0] = fu32x4{r[0], r[1], r[2], r[3]};
            dst[1] = fu32x4{r[4], r[5], r[6], r[7]};
            dst[2] = fu32x4{r[8], r[9], r[10], r[11]};
            dst[3] = fu32x4{r[12], r[13], r[14], r[15]};
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        matched += __shfl_xor(matched, o, 64);
        overflow += __shfl_xor(overflow, o, 64);
    }
    if (!E.quiet) {  // (wave-uniform, the whole workgroup alike)
        const int slot[2] = {kHdrMatched, kHdrOverflow};
        const int64_t v[2] = {(int64_t)matched, (int64_t)overflow};
        wg_header_add<2>(E.sum_out, slot, v);
    }
}

// k_count: the first pass of the counting sort -- the same rows, filters, cell computation and lane -> sub-bin
// mapping as k_emit for the same workgroup, but only the filter and key columns are read and a workgroup's
// records are only counted per bin.  Its epilogue lays the workgroup's output out: boff[workgroup][bin] = the
// chunks of the bins before it (every bin's region is a whole number of chunks), boff[workgroup][nb] = all of them.
__device__ __forceinline__ uint32_t *count_begin(const EmitPlan &E, uint32_t *clds) {
    const uint32_t nb = (uint32_t)E.n_parts << E.sub_shift;
    for (uint32_t i = threadIdx.x; i < nb + 64; i += kWgThreads) clds[i] = 0;
    __syncthreads();
    return clds + (threadIdx.x & ((1u << E.sub_shift) - 1));  // counter of bin (part, this lane's sub-bin): mine[part << ss]
}
__device__ __forceinline__ void count_finish(const EmitPlan &E, uint32_t *clds) {
    __syncthreads();
    const uint32_t nb = (uint32_t)E.n_parts << E.sub_shift, tid = threadIdx.x;
    static_assert(kEmitMaxBins <= 2 * kWgThreads, "two bins per thread");
    uint32_t *wave_tot = clds + nb;  // [16]
    // exclusive scan of the bins' chunk counts: two consecutive bins per thread, wave scan, wave totals
    const uint32_t c0 = 2 * tid < nb ? (clds[2 * tid] + kEmitChunk - 1) / kEmitChunk : 0u;
    const uint32_t c1 = 2 * tid + 1 < nb ? (clds[2 * tid + 1] + kEmitChunk - 1) / kEmitChunk : 0u;
    uint32_t x = c0 + c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if ((int)(tid & 63) >= o) x += y;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = x;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t v = 0; v < kWgThreads / 64; v++) {
        const uint32_t t = wave_tot[v];
        before += v < (tid >> 6) ? t : 0u;
        total += t;
    }
    const uint32_t excl = before + x - (c0 + c1);
    uint32_t *out = E.boff + (size_t)blockIdx.x * (nb + 1);
    if (2 * tid < nb) out[2 * tid] = excl;
    if (2 * tid + 1 < nb) out[2 * tid + 1] = excl + c0;
    if (tid == 0) out[nb] = total;
}

template <int NF, int NG>
__global__ __launch_bounds__(kWgThreads, 4) void k_count(const EmitPlan E) {
    extern __shared__ uint32_t elds[];
    const FastPlan &P = E.fp;
    const uint32_t tid = threadIdx.x;
    uint32_t *mine = count_begin(E, elds);
    const uint32_t na = (uint32_t)E.n_aggs, ss = (uint32_t)E.sub_shift;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        const int64_t n_tiles = (seg.n + kTileRows - 1) / kTileRows;
        const int64_t row_first = seg.start + (int64_t)tid * kRowsPerThread;
        constexpr int D = count_depth(NF + NG);
        FastTile<NF> fr[D];
        FastTile<NG> gr[D];
        FastTile<0> a0;
        FastTile<1> t0;
        auto in_seg = [&](int64_t r) { return r < end ? r : seg.start; };
#pragma unroll
        for (int d = 0; d < D; d++) {
                fast_load<NF, NG, 0, false>(P, in_seg(row_first + (int64_t)d * kTileRows), fr[d], gr[d], a0, t0);
                __builtin_amdgcn_sched_barrier(0);  // oldest tile first: the ring is consumed in this order
            }
        for (int64_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
          for (int d = 0; d < D; d++) {
            if (it0 + d >= n_tiles) break;
            const int64_t row = row_first + (it0 + d) * kTileRows;
            const FastTile<NF> f0 = fr[d];
            const FastTile<NG> g0 = gr[d];
            fast_load<NF, NG, 0, false>(P, in_seg(row + (int64_t)D * kTileRows), fr[d], gr[d], a0, t0);
#pragma unroll
            for (int r = 0; r < kRowsPerThread; r++) {
                bool pass = row + r < end;
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    const int64_t x = r == 0 ? f0.v[c].x : f0.v[c].y;
                    pass = pass && x >= P.lo[c] && x <= P.hi[c];
                }
                uint32_t cell = 0;
#pragma unroll
                for (int c = 0; c < NG; c++) {
                    const int64_t x = r == 0 ? g0.v[c].x : g0.v[c].y;
                    const uint64_t d = (uint64_t)x - (uint64_t)P.gmin[c];
                    pass = pass && d < (uint64_t)P.gcard[c];
                    cell += (uint32_t)d * (uint32_t)P.gstride[c];
                }
                // the row's records (one per aggregation) are consecutive pairs of one partition: kPartCells is even
                if (pass) __hip_atomic_fetch_add(mine + (((cell * na) >> kPartCellBits) << ss), na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
    }
    count_finish(E, elds);
}

template <int NF, int NG, int NA>
__global__ __launch_bounds__(kWgThreads, 4) void k_emit(const EmitPlan E) {
    extern __shared__ uint32_t elds[];
    const FastPlan &P = E.fp;
    const uint32_t tid = threadIdx.x;
    const EmitLds S = emit_begin(E, elds);

    uint32_t matched = 0, overflow = 0;
    constexpr int64_t kTile = kTileRows;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        const int64_t n_tiles = (seg.n + kTile - 1) / kTile;
        const int64_t row_first = seg.start + (int64_t)tid * kRowsPerThread;
        // D tiles of loads in flight per lane (narrow queries move few bytes per tile)
        constexpr int D = emit_depth(NF + NG + NA);
        FastTile<NF> fr[D];
        FastTile<NG> gr[D];
        FastTile<NA> ar[D];
        FastTile<1> t0;
        // (every lane always issues its loads -- rows past the end re-read the segment's first rows and are ignored --
        // so that the compiler's load counting, and with it the depth of the pipeline, is exact: see packed_issue_always)
        auto in_seg = [&](int64_t r) { return r < end ? r : seg.start; };
#pragma unroll
        for (int d = 0; d < D; d++) {
                fast_load<NF, NG, NA, false>(P, in_seg(row_first + (int64_t)d * kTile), fr[d], gr[d], ar[d], t0);
                emit_pad_stores(S);  // (the ring as the steady state has it: see emit_pad_stores)
                __builtin_amdgcn_sched_barrier(0);  // oldest tile first: the ring is consumed in this order
            }
        for (int64_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
          for (int d = 0; d < D; d++) {
            // (no early exit inside a round: the rows of a tile past the end fail `row + r < end`; see k_emit_packed)
            const int64_t row = row_first + (it0 + d) * kTile;
            const FastTile<NF> f0 = fr[d];
            const FastTile<NG> g0 = gr[d];
            const FastTile<NA> a0 = ar[d];
            fast_load<NF, NG, NA, false>(P, in_seg(row + (int64_t)D * kTile), fr[d], gr[d], ar[d], t0);
            uint32_t bin[kRowsPerThread * NA], rec[kRowsPerThread * NA];
            bool act[kRowsPerThread * NA];
#pragma unroll
            for (int r = 0; r < kRowsPerThread; r++) {
                bool pass = row + r < end;
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    const int64_t x = r == 0 ? f0.v[c].x : f0.v[c].y;
                    pass = pass && x >= P.lo[c] && x <= P.hi[c];
                }
                uint32_t cell = 0;
                bool inb = true;
#pragma unroll
                for (int c = 0; c < NG; c++) {
                    const int64_t x = r == 0 ? g0.v[c].x : g0.v[c].y;
                    const uint64_t d = (uint64_t)x - (uint64_t)P.gmin[c];
                    inb = inb && d < (uint64_t)P.gcard[c];
                    cell += (uint32_t)d * (uint32_t)P.gstride[c];
                }
                matched += pass ? 1u : 0u;
                overflow += (pass && !inb) ? 1u : 0u;
#pragma unroll
                for (int c = 0; c < NA; c++) {
                    const int64_t x = r == 0 ? a0.v[c].x : a0.v[c].y;
                    const uint32_t n = (uint32_t)((uint64_t)x - (uint64_t)P.hmin[c]);
                    const uint32_t pair = cell * (uint32_t)NA + (uint32_t)c;
                    bin[r * NA + c] = emit_bin(S, pair);
                    rec[r * NA + c] = emit_record(pair, n);
                    act[r * NA + c] = pass && inb;
                }
            }
            emit_push_all<kRowsPerThread * NA>(E, S, bin, rec, act);
          }
        }
    }
    emit_finish(E, S, matched, overflow);
}

template <int NF>
static hipError_t count_launch_nf(const EmitPlan &E, int ng, int n_wg, hipStream_t st) {
    const size_t lds = count_lds_bytes(E);
    switch (ng) {
    case 0: hipLaunchKernelGGL((k_count<NF, 0>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    case 1: hipLaunchKernelGGL((k_count<NF, 1>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    case 2: hipLaunchKernelGGL((k_count<NF, 2>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int NF>
static hipError_t emit_launch_nf(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    const size_t lds = emit_lds_bytes(E);
#define SYBL_EMIT_CASE(G, A)                                                                               \
    case (G)*3 + (A): {                                                                                    \
        auto k = k_emit<NF, G, A>;                                                                         \
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(kWgThreads), lds, st, E);                                   \
        return hipGetLastError();                                                                          \
    }
    switch (ng * 3 + na) {
        SYBL_EMIT_CASE(0, 1)
        SYBL_EMIT_CASE(0, 2)
        SYBL_EMIT_CASE(1, 1)
        SYBL_EMIT_CASE(1, 2)
        SYBL_EMIT_CASE(2, 1)
        SYBL_EMIT_CASE(2, 2)
    default: return hipErrorInvalidValue;
    }
#undef SYBL_EMIT_CASE
}

// One translation unit per NF keeps the build parallel (Makefile: kernels_fast_<NF>.o).
template <int NF, int NG, int NA, int MODE, bool TIME, bool GEN>
static hipError_t fast_launch_k(const FastPlan &P, int n_wg, size_t lds_bytes, hipStream_t st) {
    auto k = k_scan_fast<NF, NG, NA, MODE, TIME, GEN>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(kWgThreads), lds_bytes, st, P);
    return hipGetLastError();
}

template <int NF, int NG, int NA, int MODE>
static hipError_t fast_launch_one(const FastPlan &P, bool time, bool gen, int n_wg, size_t lds_bytes, hipStream_t st) {
    if (time) return gen ? fast_launch_k<NF, NG, NA, MODE, true, true>(P, n_wg, lds_bytes, st)
                         : fast_launch_k<NF, NG, NA, MODE, true, false>(P, n_wg, lds_bytes, st);
    return gen ? fast_launch_k<NF, NG, NA, MODE, false, true>(P, n_wg, lds_bytes, st)
               : fast_launch_k<NF, NG, NA, MODE, false, false>(P, n_wg, lds_bytes, st);
}

template <int NF, int NG, int NA>
static hipError_t fast_launch_mode(const FastPlan &P, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st) {
    if (NA == 0) return fast_launch_one<NF, NG, 0, kFastAvg>(P, time, gen, n_wg, lds, st);
    switch (mode) {
    case kFastAvg: return fast_launch_one<NF, NG, NA, kFastAvg>(P, time, gen, n_wg, lds, st);
    case kFastAvgMax: return fast_launch_one<NF, NG, NA, kFastAvgMax>(P, time, gen, n_wg, lds, st);
    case kFastMoments: return fast_launch_one<NF, NG, NA, kFastMoments>(P, time, gen, n_wg, lds, st);
    case kFastHist: return fast_launch_one<NF, NG, NA, kFastHist>(P, time, gen, n_wg, lds, st);
    default: return hipErrorInvalidValue;
    }
}

template <int NF>
static hipError_t fast_launch_nf(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds,
                                 hipStream_t st) {
    switch (ng * 3 + na) {
    case 0: return fast_launch_mode<NF, 0, 0>(P, mode, time, gen, n_wg, lds, st);
    case 1: return fast_launch_mode<NF, 0, 1>(P, mode, time, gen, n_wg, lds, st);
    case 2: return fast_launch_mode<NF, 0, 2>(P, mode, time, gen, n_wg, lds, st);
    case 3: return fast_launch_mode<NF, 1, 0>(P, mode, time, gen, n_wg, lds, st);
    case 4: return fast_launch_mode<NF, 1, 1>(P, mode, time, gen, n_wg, lds, st);
    case 5: return fast_launch_mode<NF, 1, 2>(P, mode, time, gen, n_wg, lds, st);
    case 6: return fast_launch_mode<NF, 2, 0>(P, mode, time, gen, n_wg, lds, st);
    case 7: return fast_launch_mode<NF, 2, 1>(P, mode, time, gen, n_wg, lds, st);
    case 8: return fast_launch_mode<NF, 2, 2>(P, mode, time, gen, n_wg, lds, st);
    default: return hipErrorInvalidValue;
    }
}

#endif  // __HIPCC__

// per-NF entry points (kernels_fast_<NF>.hip)
hipError_t launch_count_nf0(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_nf1(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_nf2(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_nf3(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_nf4(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_emit_nf0(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_nf1(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_nf2(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_nf3(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_nf4(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_scan_fast_nf0(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_fast_nf1(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_fast_nf2(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_fast_nf3(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_fast_nf4(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds, hipStream_t st);

}  // namespace sybl
