// planner.cpp -- sybl_query_desc + table statistics -> the device plan of a query: slots, folded filters,
// the direct-mapped cell layout, histogram geometry, the work list, and which kernel family runs it
// (k_scan_packed / k_scan_fast / k_emit + k_part_hist / k_scan).
//
// Reference mapping (src/lib/ of logv/sybil):
//   filters / groupings / aggregations   <- BuildFilters, Grouping, Aggregation (filter.go:59-139,
//                                           query_spec.go:214-219), cmd_query.go:204-333
//   histogram geometry                   <- SetupBuckets (hist_basic.go:34-70)
//   block skipping                       <- ShouldLoadBlockFromDir (table_block_io.go:110-182)
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "engine.h"
#include "hll.h"
#include "re2lite.h"

namespace sybl {

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------ planner

struct HostFilterFold {
    bool has_range = false;
    int64_t lo = INT64_MIN, hi = INT64_MAX;
    std::vector<int64_t> neq;
    bool has_mask = false;
    std::vector<uint8_t> mask;  // per dictionary id, ANDed over the column's str filters
    std::vector<std::pair<int32_t, int32_t>> setp;  // set predicates: (member id, 1 = in / 0 = nin)
};

// hist_basic.go:34-70
static void setup_buckets(int64_t info_min, int64_t info_max, int64_t hist_bucket, int64_t *bucket_size,
                          int64_t *num_buckets, int64_t *n_values) {
    int64_t size = info_max - info_min;
    int64_t nb = 1000;  // NUM_BUCKETS, hist.go:3
    int64_t bs = size / nb;
    if (hist_bucket > 0) bs = hist_bucket;
    if (bs == 0) {
        if (size < 100) {
            bs = 1;
            nb = size;
        } else {
            bs = size / 100;
            nb = size / bs;
        }
    }
    nb += 1;
    *bucket_size = bs;
    *num_buckets = nb;
    *n_values = nb + 1;
}


// ShouldLoadBlockFromDir (table_block_io.go:110-182) on exact per-block extrema: a gt/lt
// filter that is false on BOTH the block minimum and maximum, or an eq constant outside
// [min,max], skips the block; a filter column without a single populated row in the block
// fails on both pseudo-records as well.
static bool should_scan_block(const Table *t, const sybl_query_desc *d, int64_t b) {
    for (int i = 0; i < d->n_filters; i++) {
        const sybl_filter &f = d->filters[i];
        const Column *c = t->find(f.col);
        if (!c || c->type != SYBL_INT_VAL) continue;
        if (f.op != SYBL_OP_GT && f.op != SYBL_OP_LT && f.op != SYBL_OP_EQ) continue;
        if (c->blk_pop[(size_t)b] == 0) return false;
        int64_t mn = c->blk_min[(size_t)b], mx = c->blk_max[(size_t)b], v = f.int_value;
        if (f.op == SYBL_OP_GT && !(mn > v) && !(mx > v)) return false;
        if (f.op == SYBL_OP_LT && !(mn < v) && !(mx < v)) return false;
        if (f.op == SYBL_OP_EQ && (mn > v || mx < v)) return false;
    }
    return true;
}

// k_scan_packed works on stored offsets: rebases filter bounds, key digits, bucket numerators and
// the time value onto each column's base.  False when a quantity does not fit the 32-bit domain.
// A slot's column: a column of the table, or -- for the slot of a weight column with unpopulated rows -- the query's own
// dense column of the weights in force (Query::eff_weight; Planner::weight).
constexpr int kEffWeightCol = -2;
constexpr int kRankColBase = -100;  // slot_col = kRankColBase - i: the rank column of table column i (Column::rank_col)
static inline Column *slot_column(const Table *t, const Query *q, int ci) {
    if (ci <= kRankColBase) return t->cols[(size_t)(kRankColBase - ci)]->rank_col.get();
    return ci == kEffWeightCol ? q->eff_weight.get() : t->cols[(size_t)ci].get();
}

static bool fill_packed(Table *t, Query *q, const std::vector<int> &slot_col, FastPlan &FP, int nf, int ng, int na) {
    const ScanPlan &P = q->plan;
    if (env("SYBL_NO_PACKED")) return false;
    if (P.n_cells >= (1 << 24)) return false;  // 24-bit multiplies build the cell index
    for (int c = 0; c < nf; c++) {
        const __int128 umax = ((__int128)1 << (8 * FP.fwid[c])) - 1;
        const __int128 L = (__int128)FP.lo[c] - FP.fbase[c], H = (__int128)FP.hi[c] - FP.fbase[c];
        if (H < 0 || L > umax || L > H) {
            FP.plo[c] = 1;
            FP.phi[c] = 0;
        } else {
            FP.plo[c] = (uint32_t)(L < 0 ? 0 : L);
            FP.phi[c] = (uint32_t)(H > umax ? umax : H);
        }
        FP.npneq[c] = 0;
        for (int k = 0; k < FP.nneq[c]; k++) {
            const __int128 off = (__int128)FP.neq[c][k] - FP.fbase[c];
            if (off >= 0 && off <= umax) FP.pneq[c][FP.npneq[c]++] = (uint32_t)off;  // (else: no stored value equals it)
        }
    }
    {
        // (FastPlan::late) the share of rows the range filters let through, were the columns' values uniform over their extrema
        double pass = 1.0;
        for (int c = 0; c < nf; c++) {
            const Column *col = nullptr;
            for (auto &cp : t->cols) {
                if (cp->d_data == (const void *)FP.fcol[c]) col = cp.get();
                if (cp->rank_col && cp->rank_col->d_data == (const void *)FP.fcol[c]) col = cp->rank_col.get();
            }
            if (FP.plo[c] > FP.phi[c]) {
                pass = 0.0;
            } else if (col && col->n_pop > 0 && col->exact_max >= col->vbase && !FP.fmask[c]) {
                const double top = (double)((__int128)col->exact_max - (__int128)col->vbase);
                const double hi = std::min((double)FP.phi[c], top), lo = (double)FP.plo[c];
                pass *= hi >= lo ? std::min(1.0, (hi - lo + 1.0) / (top + 1.0)) : 0.0;
            }
        }
        FP.late = nf > 0 && pass < 0.005 ? 1 : 0;
        if (const char *e = env("SYBL_LATE_PATH")) FP.late = atoi(e) != 0;  // (A/B: 0 / 1 whatever the estimate)
    }
    for (int c = 0; c < ng; c++) FP.gdoff[c] = (uint32_t)((uint64_t)FP.gbase[c] - (uint64_t)FP.gmin[c]);
    for (int c = 0; c < na; c++) {
        // Info.Min <= v <= Info.Max*10 (hist_basic.go:104) as a range of offsets; only looked at when the
        // aggregation tracks its own count
        const __int128 L = (__int128)FP.info_min[c] - FP.abase[c], H = (__int128)FP.max10[c] - FP.abase[c];
        const __int128 umax = ((__int128)1 << 32) - 1;
        if (H < 0 || L > umax || L > H) {
            FP.alo[c] = 1;
            FP.ahi[c] = 0;
        } else {
            FP.alo[c] = (uint32_t)(L < 0 ? 0 : L);
            FP.ahi[c] = (uint32_t)(H > umax ? umax : H);
        }
    }
    const double shave = 1.0 - 1.0 / (double)((int64_t)1 << 40);
    for (int c = 0; c < na; c++) {
        FP.adoff[c] = (uint32_t)((uint64_t)FP.abase[c] - (uint64_t)FP.hmin[c]);
        FP.pinv_bucket[c] = FP.bucket_size[c] ? (1.0 / (double)FP.bucket_size[c]) * shave : 0.0;
    }
    if (q->time_mode) {
        const SlotDesc &ts = P.slot[P.time_slot];
        const Column *tc = slot_column(t, q, slot_col[(size_t)P.time_slot]);
        if (P.tb_big_div || P.time_bucket >= ((int64_t)1 << 32)) return false;
        if (tc->n_pop > 0 && tc->exact_min < 0) return false;  // truncation == floor only for val >= 0
        const __int128 t0 = (__int128)P.tb_min * P.time_bucket;
        const __int128 off = (__int128)ts.vbase - t0;
        // offsets are at most exact_max - vbase: the rebased time value stays below 2^32
        const __int128 top = tc->n_pop > 0 ? (__int128)tc->exact_max - t0 : off;
        if (off < 0 || top >= ((__int128)1 << 32)) return false;
        FP.tdoff = (uint32_t)off;
        FP.pinv_time = (1.0 / (double)P.time_bucket) * shave;
    }
    return true;
}

// Fills the column / filter / group / bucket part of a FastPlan when the query has the shape the
// role-specialised kernels cover: <= 4 range-filter, <= 2 group, <= 2 aggregation columns, all
// fully populated int64, one role per column, no rejects / outliers / minima to track.
// SYBL_PLAN_TRACE=1 (diagnostic): says on stderr which test sent a query away from the role-specialised row bodies
#define FF_REJECT(...)                                                                                  \
    do {                                                                                                \
        if (env("SYBL_PLAN_TRACE")) fprintf(stderr, "fast path: rejected at planner.cpp:%d\n", __LINE__); \
        return __VA_ARGS__;                                                                             \
    } while (0)

static bool fill_fast_columns(Table *t, Query *q, const std::vector<int> &slot_col, FastPlan &FP, int *pnf, int *png,
                              int *pna, bool *any_max, bool *all_max, bool allow_gen, bool *gen, bool *packed = nullptr,
                              int max_groups = kFastTemplatedG, int max_aggs = kFastTemplatedA, bool part = false) {
    const ScanPlan &P = q->plan;
    memset(&FP, 0, sizeof(FP));
    int nf = 0, ng = 0, na = 0;
    *gen = false;
    // compact storage: k_scan_packed when every column is a plain int column of <= 4 stored bytes,
    // else the GEN kernels (any width); the plain kernels read canonical int64 only
    bool any_packed = false, all_narrow = true;
    bool heavy = false;  // GEN features k_scan_packed<NUL> leaves out: weights, h.Max
    // what the NUL variants of the packed bodies exist for: validity bits, id masks, neq constants, per-aggregation counts,
    // outliers.  A fully populated str column as a group key needs none of it (its ids are offsets like any int's): such
    // queries run the plain packed body -- with late materialisation -- instead of paying for the NUL one (round 4)
    bool nul_needed = false;
    if (packed) *packed = false;
    for (size_t s = 0; s < slot_col.size(); s++) {
        const SlotDesc &sd = P.slot[s];
        if (sd.flags == 0 && q->pre_n_slots > 0) continue;  // (a column only the filter pre-pass reads: Planner::prefilter)
        const Column *c = slot_column(t, q, slot_col[s]);
        bool plain = c->type == SYBL_INT_VAL && !c->d_valid && !c->has_missing;
        any_packed = any_packed || c->packed();
        all_narrow = all_narrow && c->elem <= 4;
        if (!plain) {
            // GEN kernels: int columns with missing rows in any role, str columns as group keys
            if (!allow_gen) FF_REJECT(false);
            uint32_t r2 = sd.flags & (kSlotFilter | kSlotGroup | kSlotAgg | kSlotTime);
            bool str_ok = c->type == SYBL_STR_VAL && (r2 == kSlotGroup || r2 == kSlotIdMask || r2 == (kSlotGroup | kSlotIdMask));
            if (c->type != SYBL_INT_VAL && !str_ok) FF_REJECT(false);
            *gen = true;
            if (c->d_valid || c->has_missing || (sd.flags & kSlotIdMask)) nul_needed = true;
        }
        uint32_t roles = sd.flags & (kSlotFilter | kSlotGroup | kSlotAgg);
        if (sd.flags & (kSlotSet | kSlotDict)) FF_REJECT(false);
        if ((sd.flags & kSlotNeq) && (!allow_gen || c->type != SYBL_INT_VAL)) FF_REJECT(false);
        if ((sd.flags & kSlotIdMask) && !allow_gen) FF_REJECT(false);
        if ((sd.flags & kSlotWeight) && (roles != 0 || (sd.flags & kSlotTime) || !allow_gen)) FF_REJECT(false);
        // a column may be filtered AND be a key / an aggregation input / the time column (it is then
        // streamed once per role; the second read hits L1/L2), but not key and aggregation input at once
        uint32_t fpart = roles & kSlotFilter, rest = roles & (kSlotGroup | kSlotAgg);
        if (fpart != 0 && fpart != kSlotRange && fpart != kSlotIdMask && fpart != kSlotNeq && fpart != (kSlotRange | kSlotNeq)) FF_REJECT(false);
        if (rest == (kSlotGroup | kSlotAgg)) FF_REJECT(false);
        if ((sd.flags & kSlotTime) && rest != 0) FF_REJECT(false);
    }
    for (size_t s = 0; s < slot_col.size(); s++) {
        const SlotDesc &sd = P.slot[s];
        if (!(sd.flags & (kSlotRange | kSlotIdMask | kSlotNeq))) continue;
        if (nf >= kFastMaxF) FF_REJECT(false);
        FP.fcol[nf] = (const int64_t *)sd.base;
        FP.fwid[nf] = sd.width;
        FP.fbase[nf] = sd.vbase;
        FP.fvalid[nf] = sd.valid;
        FP.lo[nf] = (sd.flags & kSlotRange) ? sd.lo : INT64_MIN;
        FP.hi[nf] = (sd.flags & kSlotRange) ? sd.hi : INT64_MAX;
        if (sd.flags & kSlotNeq) {
            FP.nneq[nf] = sd.n_neq;
            for (int k = 0; k < sd.n_neq; k++) FP.neq[nf][k] = sd.neq[k];
            *gen = true;  // the neq constants are compared in the GEN / NUL row bodies
            nul_needed = true;
        }
        if (sd.flags & kSlotIdMask) {
            FP.fmask[nf] = sd.idmask;
            FP.fmask_bits[nf] = sd.idmask_bits;
            *gen = true;
            nul_needed = true;
        }
        nf++;
    }
    for (auto &gi : q->groups) {
        if (ng >= max_groups) FF_REJECT(false);
        int s = gi.slot;
        for (size_t k = 0; s < 0 && k < slot_col.size(); k++)
            if (slot_col[k] == gi.col) s = (int)k;
        const SlotDesc &sd = P.slot[s];
        if (sd.gmissing >= 0 && !allow_gen) FF_REJECT(false);
        if (sd.gmissing >= 0 || sd.gvalues != sd.gcard) nul_needed = true;
        FP.gvalid[ng] = sd.valid;
        FP.gwid[ng] = sd.width;
        FP.gbase[ng] = sd.vbase;
        FP.gmissing[ng] = sd.gmissing;
        FP.gvalues[ng] = sd.gvalues;
        FP.gcol[ng] = (const int64_t *)sd.base;
        FP.gmin[ng] = sd.gmin;
        FP.gcard[ng] = (uint32_t)sd.gcard;
        FP.gstride[ng] = sd.gstride;
        FP.gstride64[ng] = sd.gstride64;
        FP.gmissing64[ng] = sd.gmissing64;
        FP.gvalues64[ng] = sd.gvalues64;
        ng++;
    }
    *any_max = false;
    *all_max = true;
    bool any_min = false;
    for (auto &ai : q->aggs) {
        if (na >= max_aggs) FF_REJECT(false);
        const AggDesc &A = ai.d;
        // (a tracked minimum -- avg mode over negative values -- is the row bodies' business since round 5: ext_general)
        if (A.m_nmin >= 0 && (q->op == SYBL_AGG_HIST || env("SYBL_NO_FAST_MIN"))) FF_REJECT(false);
        any_min = any_min || A.m_nmin >= 0;
        // (part: the scan half of the partitioned histograms only emits v - h.Min; outliers are k_part_hist's business)
        if ((A.f_smp >= 0 || (A.f_out >= 0 && !part)) && !allow_gen) FF_REJECT(false);
        if (A.f_out >= 0 && !part) *gen = true;  // outliers: the GEN body, or the NUL variants of the packed bodies
        if (A.f_out >= 0 && !part) nul_needed = true;
        if (q->op == SYBL_AGG_HIST && A.m_max >= 0) {
            *gen = true;  // h.Max lives in the GEN body
            heavy = true;
        }
        if (A.f_smp >= 0) heavy = true;
        if (A.f_cnt >= 0 || A.f_pop >= 0) {
            if (!allow_gen) FF_REJECT(false);
            *gen = true;  // rejects / missing values: per-aggregation counts
            nul_needed = true;
        }
        if (q->op == SYBL_AGG_HIST) {
            if (A.big_div || A.bucket_size >= ((int64_t)1 << 32)) FF_REJECT(false);
            const Column *c = t->cols[(size_t)ai.col].get();
            int64_t hi = c->bounds_set ? c->bound_hi : c->exact_max;
            if (c->n_pop > 0 || c->bounds_set)
                if ((unsigned __int128)((__int128)hi - (__int128)A.hmin) >= ((unsigned __int128)1 << 32)) FF_REJECT(false);
        }
        *any_max = *any_max || A.m_max >= 0;
        *all_max = *all_max && A.m_max >= 0;
        int s = -1;
        for (size_t k = 0; k < slot_col.size(); k++)
            if (slot_col[k] == ai.col) s = (int)k;
        FP.acol[na] = (const int64_t *)P.slot[s].base;
        FP.awid[na] = P.slot[s].width;
        FP.abase[na] = P.slot[s].vbase;
        FP.avalid[na] = P.slot[s].valid;
        FP.f_cnt[na] = A.f_cnt;
        FP.f_pop[na] = A.f_pop;
        FP.f_smp[na] = A.f_smp;
        FP.f_out[na] = A.f_out;
        FP.info_min[na] = A.info_min;
        FP.max10[na] = A.max10;
        FP.hmin[na] = A.hmin;
        FP.inv_bucket[na] = A.inv_bucket;
        FP.bucket_size[na] = (uint32_t)A.bucket_size;
        FP.n_values[na] = A.n_values;
        FP.f_sum[na] = A.f_sum;
        FP.f_sb[na] = A.f_sb;
        FP.f_sb2[na] = A.f_sb2;
        FP.m_max[na] = A.m_max;
        FP.m_nmin[na] = A.m_nmin;
        FP.hist_agg_off[na] = P.hist_agg_off[na];
        na++;
    }
    // avg mode: "every aggregation tracks a maximum and nothing else" is what the kFastAvgMax bodies assume unless told
    FP.ext_general = (any_min || (*any_max && !*all_max)) ? 1 : 0;
    if (any_min) *any_max = *all_max = true;  // (-> kFastAvgMax; the bodies test m_max / m_nmin per aggregation)
    else if (*any_max && !*all_max && q->op != SYBL_AGG_HIST) *all_max = true;
    FP.f_samples = P.f_samples;
    if (q->weighted) {
        if (!allow_gen) FF_REJECT(false);
        FP.wcol = (const int64_t *)P.slot[P.weight_slot].base;
        FP.wwid = P.slot[P.weight_slot].width;
        FP.wbase = P.slot[P.weight_slot].vbase;
        *gen = true;
        heavy = true;
    }
    FP.out_log = P.out_log;
    FP.out_cap = P.out_cap;
    FP.hist_off = P.hist_off;
    FP.hist_stride = P.hist_stride;
    FP.n_cells = P.n_cells;
    FP.n_sum_fields = P.n_sum_fields;
    FP.n_max_fields = P.n_max_fields;
    FP.rep_shift = P.rep_shift;
    FP.windowed = P.windowed;
    FP.lds_cells = P.lds_cells;
    FP.wg_cell_base = P.wg_cell_base;
    *pnf = nf;
    *png = ng;
    *pna = na;
    if (any_packed) {
        if ((!*gen || (allow_gen && !heavy)) && all_narrow && packed && fill_packed(t, q, slot_col, FP, nf, ng, na)) {
            *packed = true;
            FP.nul = nul_needed || env("SYBL_FORCE_NUL") ? 1 : 0;  // missing rows / id masks / reject gate: k_scan_packed<NUL> (SYBL_FORCE_NUL: A/B)
        } else if (allow_gen) {
            *gen = true;
        } else {
            FF_REJECT(false);
        }
    }
    return true;
}

// Role-specialised kernels (scan_fast.h) cover the common shape; everything else runs k_scan.
static void select_fast_path(Table *t, Query *q, const std::vector<int> &slot_col) {
    q->fast = false;
    const ScanPlan &P = q->plan;
    if (env("SYBL_NO_FAST") || q->loghist) FF_REJECT();  // (MultiHist: the plan-interpreting kernels only)
    if (!q->use_lds) FF_REJECT();
    if (q->time_mode && P.tb_big_div) FF_REJECT();
    FastPlan &FP = q->fplan;
    int nf, ng, na;
    bool any_max, all_max, gen, packed = false;
    q->fast_packed = false;
    q->fast_packed_n = false;
    if (!fill_fast_columns(t, q, slot_col, FP, &nf, &ng, &na, &any_max, &all_max, !env("SYBL_NO_FASTGEN"), &gen, &packed)) {
        // three or four group columns, three or four aggregation columns over compact storage: the packed row body with
        // run-time column counts (k_scan_hash_packed<.., HASH = false>, hashpacked.hip); no bucket arrays
        const bool wide = (int)q->groups.size() > kFastTemplatedG || (int)q->aggs.size() > kFastTemplatedA;
        if (!wide || (int)q->groups.size() > kFastMaxG || (int)q->aggs.size() > kFastMaxA || env("SYBL_NO_PACKED_N") || env("SYBL_NO_FASTGEN")) FF_REJECT();
        if (q->op == SYBL_AGG_HIST && q->want_percentiles) FF_REJECT();
        packed = false;
        if (!fill_fast_columns(t, q, slot_col, FP, &nf, &ng, &na, &any_max, &all_max, true, &gen, &packed, kFastMaxG, kFastMaxA) || !packed) FF_REJECT();
        q->fast_packed_n = true;
    }
    if (q->op == SYBL_AGG_HIST && any_max && !gen) FF_REJECT();
    if (q->weighted && q->op == SYBL_AGG_HIST && q->want_percentiles) FF_REJECT();  // weighted bucket increments: generic kernel
    q->fast_gen = gen;
    if (nf + ng + na == 0 && !q->time_mode) FF_REJECT();  // nothing to stream: the generic kernel picks a driver column
    if (q->time_mode) {
        FP.tcol = (const int64_t *)P.slot[P.time_slot].base;
        FP.tvalid = P.slot[P.time_slot].valid;
        FP.twid = P.slot[P.time_slot].width;
        FP.tbase = P.slot[P.time_slot].vbase;
        FP.time_bucket = P.time_bucket;
        FP.inv_time_bucket = P.inv_time_bucket;
        FP.tb_min = P.tb_min;
        FP.n_tb = P.n_tb;
        FP.tb_stride = P.tb_stride;
        FP.tb_stride64 = P.tb_stride64;
    }
    int mode;
    if (q->op == SYBL_AGG_HIST) {
        mode = q->want_percentiles ? kFastHist : kFastMoments;
    } else {
        if (any_max && !all_max) FF_REJECT();
        mode = any_max ? kFastAvgMax : kFastAvg;
    }
    FP.hist_lds = 0;
    if (mode == kFastHist && !P.windowed && !env("SYBL_NO_LDSHIST")) {
        // few cells: the bucket arrays themselves fit in LDS as uint32 next to the cell table
        // (a workgroup scans far fewer than 2^32 rows); shrink the lane replication to make room
        int64_t hist_bytes = (int64_t)P.n_cells * P.hist_stride * 4;
        int64_t field_bytes = (int64_t)(P.n_sum_fields + P.n_max_fields) * P.n_cells * 8;
        if (hist_bytes + field_bytes <= kLdsBudgetBytes) {
            int rs = 0;
            while (rs < 6 && hist_bytes + (field_bytes << (rs + 1)) <= kLdsBudgetBytes) rs++;
            q->plan.rep_shift = rs;
            FP.rep_shift = rs;
            q->lds_bytes = (size_t)((field_bytes << rs) + hist_bytes + 16);
            FP.hist_lds = 1;
        }
    }
    q->fast = true;
    q->fast_packed = packed;
    q->fast_nf = nf;
    q->fast_ng = ng;
    q->fast_na = na;
    q->fast_mode = mode;
}

#undef FF_REJECT

// Hash group-by (strategy 7) through the role-specialised row body (k_scan_hash_fast, hashgroup.hip) when the query has
// the shape select_fast_path takes -- with up to four group columns; everything else runs the plan-interpreting
// k_scan_hash.
static void select_hash_fast(Table *t, Query *q, const std::vector<int> &slot_col) {
    q->hash_fast = false;
    const ScanPlan &P = q->plan;
    if (!q->hash_mode || env("SYBL_NO_FAST") || env("SYBL_NO_HASH_FAST") || env("SYBL_NO_FASTGEN") || q->loghist) return;
    if (q->time_mode && P.tb_big_div) return;
    FastPlan &FP = q->fplan;
    int nf, ng, na;
    bool any_max, all_max, gen;
    // compact storage, a composite key below 2^32 and no bucket arrays: the offset-domain row body (k_scan_hash_packed)
    bool packed = false;
    q->hash_packed = false;
    const unsigned __int128 key_space = (unsigned __int128)q->group_cells * (unsigned __int128)std::max(P.n_tb, 1);
    const bool try_packed = !env("SYBL_NO_HASH_PACKED") && key_space < ((unsigned __int128)1 << 32) &&
                            !(q->op == SYBL_AGG_HIST && q->want_percentiles) && q->groups.size() <= 2;
    if (!fill_fast_columns(t, q, slot_col, FP, &nf, &ng, &na, &any_max, &all_max, true, &gen, try_packed ? &packed : nullptr,
                           try_packed ? 2 : kFastMaxG, try_packed ? kFastMaxA : kFastTemplatedA))
        return;
    if (!packed && na > kFastTemplatedA) return;  // (k_scan_hash_fast is instantiated for <= 2 aggregations)
    if (q->weighted && q->op == SYBL_AGG_HIST && q->want_percentiles) return;  // weighted bucket increments: generic kernel
    q->hash_packed = packed;
    if (q->time_mode) {
        FP.tcol = (const int64_t *)P.slot[P.time_slot].base;
        FP.tvalid = P.slot[P.time_slot].valid;
        FP.twid = P.slot[P.time_slot].width;
        FP.tbase = P.slot[P.time_slot].vbase;
        FP.time_bucket = P.time_bucket;
        FP.inv_time_bucket = P.inv_time_bucket;
        FP.tb_min = P.tb_min;
        FP.n_tb = P.n_tb;
        FP.tb_stride64 = P.tb_stride64;
    }
    int mode;
    if (q->op == SYBL_AGG_HIST) {
        mode = q->want_percentiles ? kFastHist : kFastMoments;
    } else {
        if (any_max && !all_max) return;
        mode = any_max ? kFastAvgMax : kFastAvg;
    }
    FP.hist_lds = 0;
    q->hash_fast = true;
    q->fast_nf = nf;
    q->fast_ng = ng;
    q->fast_na = na;
    q->fast_mode = mode;
}

// Partitioned histograms (strategy 5, scan_fast.h): full-histogram queries whose (cell, agg)
// pairs fit kMaxParts partitions of kPartCells pairs.
//
// One pass of it over the aggregations [a0, a0 + n) of the query (n <= kFastTemplatedA: what k_emit / k_part_hist are
// instantiated for): column roles, partition / bin geometry and the sizes of its buffers.  ok = false: not eligible.
struct PartGeom {
    int nf = 0, ng = 0, na = 0, ss = 0;
    bool packed = false;
    int64_t n_parts = 0, nb = 0, cap = 0, n_wg = 0;
    size_t wrap_cap = 0, table_words = 0;
};
static bool plan_part_pass(Table *t, Query *q, const std::vector<int> &slot_col, int64_t rows_scanned, int a0, int n, EmitPlan &E, PartGeom &G) {
    const ScanPlan &P = q->plan;
    bool any_max, all_max, gen;
    // (fill_fast_columns walks q->aggs: hand it the slice)
    std::vector<AggInfo> all;
    all.swap(q->aggs);
    q->aggs.assign(all.begin() + a0, all.begin() + a0 + n);
    const bool shape = fill_fast_columns(t, q, slot_col, E.fp, &G.nf, &G.ng, &G.na, &any_max, &all_max, false, &gen, &G.packed,
                                         kFastTemplatedG, kFastTemplatedA, true);
    q->aggs.swap(all);
    if (!shape) return false;
    const int na = G.na;
    // (the emitting kernels subtract h.Min -- canonical storage -- or add adoff = base - h.Min -- compact storage: biased
    // by BucketSize, the record's value part comes out as v - h.Min + BucketSize at no cost)
    for (int c = 0; c < na; c++) {
        E.fp.hmin[c] -= (int64_t)E.fp.bucket_size[c];
        if (G.packed) E.fp.adoff[c] += E.fp.bucket_size[c];
    }
    // a record is (local pair, v - h.Min + BucketSize) -- k_part_hist's quotient is the bucket + 1, and the all-zero word is
    // free to mean "no record" (kRecSentinel) --: the value part must fit kRecValueBits, and k_part_hist's divide multiplies
    // (bucket + 1) x BucketSize in 24 bits
    for (int a = a0; a < a0 + n; a++) {
        const AggDesc &A = q->aggs[(size_t)a].d;
        if (A.n_values > (1 << kBucketBits) || A.bucket_size >= ((int64_t)1 << 24)) return false;
        if (((int64_t)A.n_values + 1) * A.bucket_size >= ((int64_t)1 << kRecValueBits)) return false;
        if (A.f_out >= 0) {
            // outliers: every accepted value, not only the bucket range, must fit the record, and k_part_hist's quotient
            // (a 24-bit multiply checks it) stays below 2^24
            const Column *c = t->cols[(size_t)q->aggs[(size_t)a].col].get();
            const int64_t hi = std::min(c->bounds_set ? c->bound_hi : c->exact_max, A.max10);
            if (hi < A.hmin) continue;
            const unsigned __int128 span = (unsigned __int128)((__int128)hi - (__int128)A.hmin) + (unsigned __int128)A.bucket_size;
            if (span >= ((unsigned __int128)1 << kRecValueBits) - 1 || span / (unsigned __int128)A.bucket_size >= ((unsigned __int128)1 << 24)) return false;
        }
    }
    int64_t pairs = (int64_t)P.n_cells * na;
    G.n_parts = (pairs + kPartCells - 1) / kPartCells;
    if (G.n_parts > kMaxParts) return false;
    // Staging bins: a partition is spread over 1 << ss bins (a lane's bin follows from its lane number) so that
    // few partitions do not serialise on a handful of LDS counters; fewer bins when a workgroup's share of
    // the records would leave most of a bin's chunks padding.
    const int64_t recs_all = rows_scanned * na;
    G.n_wg = std::max(1, q->n_wg);
    if (G.n_wg > 2048) return false;  // k_part_hist keeps one region descriptor per scanning workgroup in LDS
    int ss = 0;
    while ((G.n_parts << (ss + 1)) <= kEmitMaxBins) ss++;
    while (ss > 0 && recs_all / (G.n_wg * (G.n_parts << ss)) < 64) ss--;
    G.ss = ss;
    // the workgroups' outputs are sized exactly by k_count at scan time; this is their upper bound: every record + one
    // partly filled chunk per (workgroup, bin).  Chunk indices are 32-bit; a workgroup's output is addressed with a
    // 32-bit byte offset below kEmitDropOffset (2 GiB).
    G.nb = G.n_parts << ss;
    G.cap = recs_all + G.n_wg * G.nb * (int64_t)kEmitChunk + kEmitChunk;
    if (G.cap >= ((int64_t)1 << 32) - ((int64_t)1 << 22)) return false;
    int64_t wg_rows_max = 0;
    {
        std::vector<int64_t> per_wg((size_t)G.n_wg, 0);
        for (int64_t w = 0; w < G.n_wg && (size_t)w + 1 < q->wg_seg_begin.size(); w++)
            for (int si = q->wg_seg_begin[(size_t)w]; si < q->wg_seg_begin[(size_t)w + 1]; si++) per_wg[(size_t)w] += q->segs[(size_t)si].n;
        for (int64_t v : per_wg) wg_rows_max = std::max(wg_rows_max, v);
    }
    if ((wg_rows_max * na + G.nb * (int64_t)kEmitChunk) * 4 >= (int64_t)kEmitDropOffset) return false;
    // k_part_hist's 16-bit bucket counters log their wraps: at most 3 entries per 65536 records (kernels.hip)
    G.wrap_cap = (size_t)(3 * (recs_all / 65536 + 1) + 4096);
    G.table_words = (size_t)G.n_wg * (size_t)(G.nb + 1) + (size_t)G.n_wg + 1 + 2 + 2 * G.wrap_cap;
    return true;
}

// ... and its plans over the (shared) buffers
static void bind_part_pass(Query *q, int a0, const PartGeom &G, uint32_t *d_recs, uint32_t *d_tables, EmitPlan &E, PartHistPlan &H) {
    const ScanPlan &P = q->plan;
    E.recs = d_recs;
    E.boff = d_tables;
    E.wbase = E.boff + (size_t)G.n_wg * (size_t)(G.nb + 1);
    E.n_parts = (int32_t)G.n_parts;
    E.n_aggs = G.na;
    E.n_wg = (int32_t)G.n_wg;
    E.sub_shift = G.ss;
    E.quiet = a0 > 0 ? 1 : 0;
    memset(&H, 0, sizeof(H));
    H.no_count = a0 > 0 ? 1 : 0;
    H.recs = d_recs;
    H.boff = E.boff;
    H.wbase = E.wbase;
    H.n_wg = (int32_t)G.n_wg;
    H.sub_shift = G.ss;
    H.wrap_log = E.wbase + G.n_wg + 1;
    H.wrap_cap = (uint32_t)G.wrap_cap;
    H.n_parts = (int32_t)G.n_parts;
    H.n_aggs = G.na;
    H.n_cells = P.n_cells;
    H.hist_off = P.hist_off;
    H.hist_stride = P.hist_stride;
    int nv_max = 0;
    for (int a = 0; a < G.na; a++) {
        const AggDesc &A = q->aggs[(size_t)(a0 + a)].d;
        H.pinv_bucket[a] = (1.0 / (double)A.bucket_size) * (1.0 - 0x1p-40);
        H.n_values[a] = A.n_values;
        H.f_sum[a] = A.f_sum;
        H.m_max[a] = A.m_max;
        H.f_out[a] = A.f_out;
        H.hmin[a] = A.hmin;
        H.bucket_size[a] = A.bucket_size;
        H.hist_agg_off[a] = P.hist_agg_off[a0 + a];
        nv_max = std::max(nv_max, A.n_values);
    }
    H.nv_max = nv_max;
    H.agg0 = a0;
    H.out_log = P.out_log;
    H.out_cap = P.out_cap;
    // few partitions: several workgroups share one so the whole chip is busy
    H.split = (int32_t)std::max<int64_t>(1, (int64_t)q->n_wg / G.n_parts);
    H.n_cus = q->ctx->n_cus;
    H.tail_mode = env("SYBL_PARTHIST_TAIL") ? atoi(env("SYBL_PARTHIST_TAIL")) : 1;
}

static int select_part_hist(Table *t, Query *q, const std::vector<int> &slot_col, int64_t rows_scanned) {
    q->part_hist = false;
    q->part_more.clear();
    if (env("SYBL_NO_PARTHIST") || q->hash_mode || q->loghist) return SYBL_OK;
    if (q->op != SYBL_AGG_HIST || !q->want_percentiles || q->time_mode || q->weighted || q->aggs.empty()) return SYBL_OK;
    if (q->fast && q->fplan.hist_lds) return SYBL_OK;  // the bucket arrays already live in LDS
    // k_emit / k_part_hist take one or two aggregations: a query with three or four runs the sequence twice (same rows,
    // same buffers; every pass fills its own aggregations' bucket arrays and sum fields of the one cell table)
    const int n_all = (int)q->aggs.size();
    if (n_all > kFastMaxA) return SYBL_OK;
    const int n_pass = (n_all + kFastTemplatedA - 1) / kFastTemplatedA;
    std::vector<PartGeom> geo((size_t)n_pass);
    std::vector<EmitPlan> eps((size_t)n_pass);
    size_t rec_words = 0, table_words = 0;
    for (int p = 0; p < n_pass; p++) {
        const int a0 = p * kFastTemplatedA, n = std::min(kFastTemplatedA, n_all - a0);
        memset(&eps[(size_t)p], 0, sizeof(EmitPlan));
        if (!plan_part_pass(t, q, slot_col, rows_scanned, a0, n, eps[(size_t)p], geo[(size_t)p])) return SYBL_OK;
        rec_words = std::max(rec_words, (size_t)geo[(size_t)p].cap);
        table_words = std::max(table_words, geo[(size_t)p].table_words);
    }
    size_t bytes = rec_words * 4 + table_words * 4, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes + ((size_t)1 << 30) > free_b) return SYBL_OK;
    SYBL_HIP(hipMalloc((void **)&q->d_recs, rec_words * 4));
    // boff | wbase | wrap log in one allocation
    SYBL_HIP(hipMalloc((void **)&q->d_cursor, table_words * 4));
    q->eplan = eps[0];
    bind_part_pass(q, 0, geo[0], q->d_recs, q->d_cursor, q->eplan, q->pplan);
    q->part_packed = geo[0].packed;
    q->part_nf = geo[0].nf;
    q->part_ng = geo[0].ng;
    q->part_na = geo[0].na;
    for (int p = 1; p < n_pass; p++) {
        Query::PartPass pp;
        pp.E = eps[(size_t)p];
        pp.na = geo[(size_t)p].na;
        pp.packed = geo[(size_t)p].packed;
        bind_part_pass(q, p * kFastTemplatedA, geo[(size_t)p], q->d_recs, q->d_cursor, pp.E, pp.H);
        q->part_more.push_back(pp);
    }
    q->part_hist = true;
    return SYBL_OK;
}

// The planner: sybl_query_desc + table statistics -> ScanPlan (+ FastPlan / EmitPlan), work list and
// device buffers.  One method per step, in the order the reference builds a query
// (cmd_query.go:204-333: filters, groupings, aggregations, time / weight options).
struct Planner {
    Table *t;
    const sybl_query_desc *d;
    Query *q;
    Ctx *ctx;
    ScanPlan &P;
    std::vector<int> slot_col;          // table column index per slot
    std::vector<HostFilterFold> folds;  // per slot
    int64_t cells = 1;                  // group cells (product of key digits)
    int64_t n_cells = 1;                // cells x time buckets
    int F = 1, M = 0;                   // SUM / MAX fields per cell
    int64_t hist_stride = 0;            // bucket-array words per cell
    int64_t rows_scanned = 0, skipped = 0;

    Planner(Table *t_, const sybl_query_desc *d_, Query *q_) : t(t_), d(d_), q(q_), ctx(t_->ctx), P(q_->plan) {}

    // one slot per distinct referenced column
    int slot_of(const char *name, int *out) {
        Column *c = t->find(name);
        if (!c) return fail(SYBL_E_INVAL, "unknown column '%s'", name ? name : "(null)");
        int ci = t->col_ix[name];
        for (size_t s = 0; s < slot_col.size(); s++)
            if (slot_col[s] == ci) {
                *out = (int)s;
                return SYBL_OK;
            }
        if ((int)slot_col.size() >= kMaxSlots) return fail(SYBL_E_INVAL, "query references more than %d columns", kMaxSlots);
        slot_col.push_back(ci);
        folds.emplace_back();
        *out = (int)slot_col.size() - 1;
        return SYBL_OK;
    }

    int setup() {
        int rc;
        if (d->n_groups > SYBL_MAX_GROUPS) return fail(SYBL_E_INVAL, "too many group columns (%d > %d)", d->n_groups, SYBL_MAX_GROUPS);
        if (d->n_aggs > SYBL_MAX_AGGS) return fail(SYBL_E_INVAL, "too many aggregations (%d > %d)", d->n_aggs, SYBL_MAX_AGGS);
        if (d->n_filters > SYBL_MAX_FILTERS) return fail(SYBL_E_INVAL, "too many filters");
        if (d->op != SYBL_AGG_AVG && d->op != SYBL_AGG_HIST) return fail(SYBL_E_INVAL, "unknown op %d", d->op);
        rc = table_ensure_stats(t);
        if (rc) return rc;

        q->op = d->op;
        q->hist_bucket = d->hist_bucket;
        q->loghist = d->loghist != 0;
        // (a MultiHist's percentiles and stddev both come from its sub-histograms' buckets: they are always kept)
        q->want_percentiles = d->op == SYBL_AGG_HIST && (d->want_percentiles || q->loghist);
        q->order_by = d->order_by ? d->order_by : "";
        q->order_asc = d->order_asc != 0;
        q->limit = d->limit;
        q->printed_only = d->printed_only != 0;
        q->printed_level = d->printed_only;
        q->time_mode = d->time_bucket > 0 && d->time_col && d->time_col[0];
        q->time_bucket = q->time_mode ? d->time_bucket : 0;
        q->weighted = d->weight_col && d->weight_col[0];

        // ---- -str-replace (column_store_io.go:517-545): rewritten dictionaries of this query
        for (int i = 0; i < d->n_str_replace; i++) {
            const sybl_str_replace &sr = d->str_replace[i];
            Column *c = t->find(sr.col);
            if (!c || c->type != SYBL_STR_VAL) continue;  // (the reference only consults the map while unpacking str columns)
            auto R = std::make_unique<StrReplaced>();
            Re2Lite re;
            if (!sr.replaced) {
                std::string why;
                if (!re.compile(sr.pattern ? sr.pattern : "", &why))
                    return fail(SYBL_E_INVAL, "bad -str-replace pattern '%s': %s", sr.pattern ? sr.pattern : "", why.c_str());
            } else if (sr.n_replaced < (int64_t)c->dict.size()) {
                return fail(SYBL_E_INVAL, "-str-replace on '%s': %lld rewritten strings for a dictionary of %zu", c->name.c_str(),
                            (long long)sr.n_replaced, c->dict.size());
            }
            std::unordered_map<std::string, int32_t> seen;
            R->remap.resize(c->dict.size());
            for (size_t k = 0; k < c->dict.size(); k++) {
                std::string nv = sr.replaced ? std::string(sr.replaced[k] ? sr.replaced[k] : "") : re.replace_all(c->dict[k], sr.replace ? sr.replace : "");
                auto it = seen.find(nv);
                if (it == seen.end()) {
                    it = seen.emplace(nv, (int32_t)R->strs.size()).first;
                    R->strs.push_back(nv);
                }
                R->remap[k] = it->second;
            }
            q->replaced[t->col_ix[c->name]] = std::move(R);
        }

        memset(&P, 0, sizeof(P));
        P.time_slot = -1;
        P.weight_slot = -1;
        P.f_samples = -1;
        P.hist_mode = d->op == SYBL_AGG_HIST;
        P.weighted = q->weighted;

        return SYBL_OK;
    }

    int filters() {
        int rc;
        // ---- filters (filter.go:171-285), folded per column
        for (int i = 0; i < d->n_filters; i++) {
            const sybl_filter &f = d->filters[i];
            int s;
            if ((rc = slot_of(f.col, &s))) return rc;
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            HostFilterFold &ff = folds[(size_t)s];
            if (c->type == SYBL_INT_VAL) {
                int64_t v = f.int_value;
                switch (f.op) {
                case SYBL_OP_GT:  // field > v
                    ff.has_range = true;
                    if (v == INT64_MAX) q->never_matches = true; else ff.lo = std::max(ff.lo, v + 1);
                    break;
                case SYBL_OP_LT:
                    ff.has_range = true;
                    if (v == INT64_MIN) q->never_matches = true; else ff.hi = std::min(ff.hi, v - 1);
                    break;
                case SYBL_OP_EQ:
                    ff.has_range = true;
                    ff.lo = std::max(ff.lo, v);
                    ff.hi = std::min(ff.hi, v);
                    break;
                case SYBL_OP_NEQ:
                    if ((int)ff.neq.size() >= kMaxNeq) return fail(SYBL_E_INVAL, "more than %d neq filters on '%s'", kMaxNeq, f.col);
                    ff.neq.push_back(v);
                    break;
                default:
                    // IntFilter.Filter's default branch returns false for every row (filter.go:189-193)
                    q->never_matches = true;
                    ff.has_range = true;
                }
            } else if (c->type == SYBL_STR_VAL) {
                // eq/neq compare dictionary ids, re/nre go through a per-id match table
                // (the reference's RCache, filter.go:213-236); all become one bit per id.
                size_t n = c->dict.size();
                std::vector<uint8_t> m(n, 0);
                // -str-replace: the filters see the rewritten strings
                auto rp = q->replaced.find(slot_col[(size_t)s]);
                const StrReplaced *RW = rp == q->replaced.end() ? nullptr : rp->second.get();
                auto str_of = [&](size_t k) -> const std::string & { return RW ? RW->strs[(size_t)RW->remap[k]] : c->dict[k]; };
                if ((f.op == SYBL_OP_EQ || f.op == SYBL_OP_NEQ) && RW) {
                    const std::string want = f.str_value ? f.str_value : "";
                    for (size_t k = 0; k < n; k++) m[k] = (str_of(k) == want) == (f.op == SYBL_OP_EQ);
                } else if (f.op == SYBL_OP_EQ || f.op == SYBL_OP_NEQ) {
                    auto it = c->dict_ix.find(f.str_value ? f.str_value : "");
                    for (size_t k = 0; k < n; k++) m[k] = f.op == SYBL_OP_NEQ;
                    if (it != c->dict_ix.end()) m[(size_t)it->second] = f.op == SYBL_OP_EQ;
                } else if (f.op == SYBL_OP_RE || f.op == SYBL_OP_NRE) {
                    if (f.id_match) {
                        for (size_t k = 0; k < n; k++) {
                            bool hit = (int64_t)k < f.id_match_len && f.id_match[k];
                            m[k] = f.op == SYBL_OP_NRE ? !hit : hit;
                        }
                    } else {
                        // Go's regexp (RE2 syntax, linear-time matching): re2lite.h
                        Re2Lite re;
                        std::string why;
                        if (!re.compile(f.str_value ? f.str_value : "", &why))
                            return fail(SYBL_E_INVAL, "bad regex '%s': %s", f.str_value ? f.str_value : "", why.c_str());
                        for (size_t k = 0; k < n; k++) {
                            bool hit = re.search(str_of(k));
                            m[k] = f.op == SYBL_OP_NRE ? !hit : hit;
                        }
                    }
                } else {
                    q->never_matches = true;  // StrFilter default branch: ret stays false
                }
                if (!ff.has_mask) {
                    ff.mask = m;
                    ff.has_mask = true;
                } else {
                    for (size_t k = 0; k < n; k++) ff.mask[k] = ff.mask[k] && m[k];
                }
            } else {
                // SetFilter.Filter, filter.go:252-285; get_val_id of an unseen string yields an id no
                // member can have (table_column.go:27-48)
                if (f.op != SYBL_OP_IN && f.op != SYBL_OP_NIN) {
                    q->never_matches = true;  // default branch: ret stays false
                    ff.setp.emplace_back(-1, 1);
                    continue;
                }
                if ((int)ff.setp.size() >= kMaxNeq) return fail(SYBL_E_INVAL, "more than %d set filters on '%s'", kMaxNeq, f.col);
                auto it = c->dict_ix.find(f.str_value ? f.str_value : "");
                ff.setp.emplace_back(it == c->dict_ix.end() ? -1 : it->second, f.op == SYBL_OP_IN ? 1 : 0);
            }
        }
        return SYBL_OK;
    }

    int groups() {
        int rc;
        // ---- group columns (aggregate.go:125-143); direct-mapped on declared or exact bounds
        cells = 1;
        for (int g = 0; g < d->n_groups; g++) {
            int s;
            const size_t slots_before = slot_col.size();
            if ((rc = slot_of(d->groups[g], &s))) return rc;
            const bool fresh_slot = slot_col.size() > slots_before;  // (no other role has this column so far)
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            if (c->type == SYBL_SET_VAL) return fail(SYBL_E_INVAL, "cannot group by set column '%s' (cmd_query.go:254)", c->name.c_str());
            if (P.slot[s].flags & kSlotGroup) return fail(SYBL_E_INVAL, "column '%s' grouped twice", c->name.c_str());
            GroupInfo gi;
            gi.col = slot_col[(size_t)s];
            gi.type = c->type;
            gi.has_missing = c->has_missing;
            int64_t lo, hi;
            if (c->bounds_set) {
                lo = c->bound_lo;
                hi = c->bound_hi;
            } else if (c->type == SYBL_STR_VAL) {
                lo = 0;
                hi = (int64_t)c->dict.size() - 1;
            } else {
                lo = c->exact_min;
                hi = c->exact_max;
            }
            // (no value anywhere in the column: no key range.  A str column's range is its dictionary whatever THIS rank holds --
            // after sybl_table_agree that is the ranks' union, and a rank without a row of the column must still lay its partial
            // table out like the others: tests/test_gpu_cli_multirank.py::test_ranks_without_a_block)
            if (c->n_pop == 0 && !c->bounds_set && (c->type != SYBL_STR_VAL || c->dict.empty())) {
                lo = 0;
                hi = -1;
            }
            unsigned __int128 card = hi >= lo ? (unsigned __int128)((__int128)hi - (__int128)lo) + 1 : 0;
            // A missing key is written as MISSING_VALUE = 0xFFFFFFFFFFFFFFFF (aggregate.go:31,138), which
            // is also the 8-byte image of the int value -1: the reference folds both into ONE group.
            // When -1 is inside the key range the missing rows share its cell; otherwise they get an
            // extra digit of their own.
            gi.missing_digit = -1;
            gi.dict = false;
            {
                auto rp = q->replaced.find(gi.col);
                if (rp != q->replaced.end()) {
                    // the digit is the id of the rewritten string, looked up through a device map old id -> new id
                    StrReplaced *RW = rp->second.get();
                    gi.replaced = RW;
                    if (!RW->d_keys) {
                        uint32_t cap = 16;
                        while ((size_t)cap < 2 * RW->remap.size() + 2) cap <<= 1;
                        std::vector<int64_t> keys(cap, kDictEmpty);
                        std::vector<int32_t> ranks(cap, -1);
                        for (size_t k = 0; k < RW->remap.size(); k++) {
                            uint64_t z = (uint64_t)k + 0x9E3779B97F4A7C15ull;  // dict_hash (scan_generic.h)
                            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                            z = z ^ (z >> 31);
                            uint32_t hh = (uint32_t)(z >> 32) & (cap - 1);
                            while (keys[hh] != kDictEmpty) hh = (hh + 1) & (cap - 1);
                            keys[hh] = (int64_t)k;
                            ranks[hh] = RW->remap[k];
                        }
                        SYBL_HIP(hipMalloc((void **)&RW->d_keys, (size_t)cap * 8));
                        SYBL_HIP(hipMalloc((void **)&RW->d_ranks, (size_t)cap * 4));
                        SYBL_HIP(hipMemcpy(RW->d_keys, keys.data(), (size_t)cap * 8, hipMemcpyHostToDevice));
                        SYBL_HIP(hipMemcpy(RW->d_ranks, ranks.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
                        RW->mask = cap - 1;
                    }
                    lo = 0;
                    hi = (int64_t)RW->strs.size() - 1;
                    card = RW->strs.size();
                }
            }
            // sparse / wide key range: one digit per DISTINCT value instead of one per value of the range
            const bool hash_ok = !env("SYBL_NO_HASH");
            if (c->type == SYBL_INT_VAL && !env("SYBL_NO_GDICT") &&
                (c->gdict_blocks == -2 || group_key_wants_dict(card, cells))) {
                // (gdict_refused: the ranks found more distinct values between them than a dictionary holds -- sybl_table_agree)
                rc = c->gdict_refused ? SYBL_E_INVAL : column_build_gdict(t, c);
                if (rc == SYBL_OK) {
                    gi.dict = true;
                    card = c->gdict.size();
                    // Round 5: the ranks are laid out once, as a narrow derived column (Column::rank_col), and the group
                    // role moves to a slot of that column -- a plain direct-mapped key (gmin 0, one digit per distinct value)
                    // that every specialised row body takes; the dictionary is no longer probed per row by the plan
                    // interpreter (config 3 grouped through a dictionary digit: 28 ms per 1e9 rows in k_scan).  The table's
                    // column keeps the slot it has if the query also filters / aggregates it.
                    if (!env("SYBL_NO_RANKCOL") && !c->gdict.empty()) {
                        if ((rc = column_build_rank(t, c))) return rc;
                        gi.rank = true;
                        lo = 0;
                        hi = (int64_t)c->gdict.size() - 1;
                        const int sentinel = kRankColBase - gi.col;
                        if (fresh_slot) {
                            slot_col[(size_t)s] = sentinel;
                        } else {
                            if ((int)slot_col.size() >= kMaxSlots) return fail(SYBL_E_INVAL, "query references more than %d columns", kMaxSlots);
                            slot_col.push_back(sentinel);
                            folds.emplace_back();
                            s = (int)slot_col.size() - 1;
                        }
                    }
                } else if (rc == SYBL_E_INVAL && hash_ok && card < ((unsigned __int128)1 << 62)) {
                    // more distinct values than a dictionary holds: the digit is the value's offset in its range and the
                    // query goes through the hash table
                    set_error("%s", "");
                } else {
                    return rc;
                }
            }
            gi.value_card = (int64_t)card;
            if (gi.has_missing) {
                int64_t minus1 = -1;
                if (gi.dict) {
                    auto it = std::lower_bound(c->gdict.begin(), c->gdict.end(), (int64_t)-1);
                    minus1 = it != c->gdict.end() && *it == -1 ? (int64_t)(it - c->gdict.begin()) : -1;
                } else if (c->type == SYBL_INT_VAL && hi >= lo && lo <= -1 && hi >= -1) {
                    minus1 = -1 - lo;
                }
                if (minus1 >= 0) {
                    gi.missing_digit = (int64_t)minus1;
                } else {
                    gi.missing_digit = (int64_t)card;
                    card += 1;
                }
            }
            if (card == 0) card = 1;
            if (card * (unsigned __int128)cells > ((unsigned __int128)1 << 27)) {
                // The reference groups on arbitrary keys through a map (aggregate.go:186-200).  Key spaces that do not
                // direct-map go through the hash table (strategy 7) as long as the composite key fits 62 bits.
                if (!hash_ok || card * (unsigned __int128)cells >= ((unsigned __int128)1 << 62))
                    return fail(SYBL_E_INVAL,
                                "group-by on '%s' needs more than %s cells (value range [%lld,%lld])",
                                c->name.c_str(), hash_ok ? "2^62 hashed" : "2^27 direct-mapped (SYBL_NO_HASH)", (long long)lo,
                                (long long)hi);
                q->hash_mode = true;
            }
            gi.gmin = lo;
            gi.gcard = (int64_t)card;
            gi.slot = s;
            q->groups.push_back(gi);
            cells *= (int64_t)card;
        }
        if (env("SYBL_FORCE_HASH") && !q->groups.empty()) q->hash_mode = true;  // (tests: small key spaces too)
        // strides: first group column is the most significant digit (keeps canonical key order
        // equal to cell order)
        {
            int64_t stride = cells;
            for (size_t g = 0; g < q->groups.size(); g++) {
                stride /= q->groups[g].gcard;
                const int s = q->groups[g].slot;
                SlotDesc &sd = P.slot[s];
                sd.flags |= kSlotGroup;
                sd.gmin = q->groups[g].gmin;
                sd.gstride64 = stride;
                sd.gmissing64 = q->groups[g].missing_digit >= 0 ? q->groups[g].missing_digit * stride : -1;
                sd.gvalues64 = q->groups[g].value_card;
                // (the 32-bit copies serve the direct-mapped kernels; a hashed query never reads them)
                sd.gcard = (int32_t)std::min<int64_t>(q->groups[g].gcard, INT32_MAX);
                sd.gstride = (int32_t)std::min<int64_t>(stride, INT32_MAX);
                sd.gmissing = q->groups[g].missing_digit >= 0 ? (int32_t)std::min<int64_t>(q->groups[g].missing_digit * stride, INT32_MAX) : -1;
                sd.gvalues = (int32_t)std::min<int64_t>(q->groups[g].value_card, INT32_MAX);
                if (q->groups[g].replaced) {
                    sd.flags |= kSlotDict;
                    sd.dkeys = q->groups[g].replaced->d_keys;
                    sd.dranks = q->groups[g].replaced->d_ranks;
                    sd.dmask = q->groups[g].replaced->mask;
                } else if (q->groups[g].dict && !q->groups[g].rank) {
                    const Column *gc = t->cols[(size_t)q->groups[g].col].get();
                    sd.flags |= kSlotDict;
                    sd.dkeys = gc->d_gdict_keys;
                    sd.dranks = gc->d_gdict_ranks;
                    sd.dmask = gc->gdict_mask;
                }
            }
        }
        q->group_cells = cells;
        return SYBL_OK;
    }

    int time_series() {
        int rc;
        // ---- time series (aggregate.go:146-183)
        P.n_tb = 1;
        P.tb_stride = (int32_t)std::min<int64_t>(cells, INT32_MAX);
        P.tb_stride64 = cells;
        if (q->time_mode) {
            int s;
            if ((rc = slot_of(d->time_col, &s))) return rc;
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            if (c->type != SYBL_INT_VAL) return fail(SYBL_E_INVAL, "time column '%s' is not an int column", c->name.c_str());
            P.slot[s].flags |= kSlotTime;
            P.time_slot = s;
            P.time_bucket = d->time_bucket;
            P.inv_time_bucket = 1.0 / (double)d->time_bucket;
            int64_t lo = c->bounds_set ? c->bound_lo : c->exact_min, hi = c->bounds_set ? c->bound_hi : c->exact_max;
            if (c->n_pop == 0 && !c->bounds_set) lo = hi = 0;
            int64_t tlo = lo / d->time_bucket, thi = hi / d->time_bucket;  // truncating, like aggregate.go:174
            P.tb_min = tlo;
            int64_t ntb = thi - tlo + 1;
            if (ntb >= ((int64_t)1 << 31)) return fail(SYBL_E_INVAL, "more than 2^31 time buckets");
            if (!q->hash_mode && (unsigned __int128)ntb * (unsigned __int128)cells > ((unsigned __int128)1 << 27)) {
                // the reference groups arbitrary keys inside every time bucket (aggregate.go:146-200): [time bucket || key]
                // through the hash table when the product does not direct-map
                if (env("SYBL_NO_HASH") || q->groups.empty())
                    return fail(SYBL_E_INVAL, "time buckets x groups exceeds 2^27 cells");
                q->hash_mode = true;
            }
            if (q->hash_mode && (unsigned __int128)ntb * (unsigned __int128)cells >= ((unsigned __int128)1 << 62))
                return fail(SYBL_E_INVAL, "time buckets x group keys exceed the 2^62 composite keys of the hash table");
            P.n_tb = (int32_t)ntb;
            uint64_t amax = (uint64_t)std::max(llabs((long long)lo), llabs((long long)hi));
            P.tb_big_div = amax >= ((uint64_t)1 << 51);
        }
        n_cells = q->hash_mode ? 0 : cells * P.n_tb;
        if (q->hash_mode) {
            // slots: twice the keys the table can possibly hold (a row each / the whole key space), a power of two
            const unsigned __int128 space = (unsigned __int128)cells * (unsigned __int128)P.n_tb;
            int64_t want = 2 * (int64_t)std::min<unsigned __int128>((unsigned __int128)std::max<int64_t>(t->logical_rows, 1), space);
            if (const char *e = env("SYBL_HASH_SLOTS")) want = atoll(e);
            int64_t slots = 1 << 12;
            while (slots < want && slots < kHashMaxSlots) slots <<= 1;
            n_cells = slots;
            P.hash_mode = 1;
        }
        P.n_cells = (int32_t)n_cells;
        return SYBL_OK;
    }

    int weight() {
        int rc;
        // ---- weight column (aggregate.go:100-102)
        if (q->weighted) {
            int s;
            const size_t slots_before = slot_col.size();
            if ((rc = slot_of(d->weight_col, &s))) return rc;
            const bool fresh_slot = slot_col.size() > slots_before;  // (no other role has this column so far)
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            if (c->type != SYBL_INT_VAL) return fail(SYBL_E_INVAL, "weight column '%s' is not an int column", c->name.c_str());
            if (c->has_missing || c->d_valid) {
                // aggregate.go:68,100-102: a row without a weight aggregates with the weight of the last row of its block
                // that had one (1 before the first).  The weights in force are laid out once, as a dense column of this
                // query's own, and the weight role moves to a slot of that column (the table's column keeps the slot it
                // has if the query also filters / groups / aggregates it).
                // (one per column and table version, shared by the queries that weigh by it: Column::carried_weight)
                if (!c->carried_weight || c->carried_version != t->version) {
                    std::shared_ptr<Column> e(new Column(), [](Column *x) {
                        column_free(x);
                        delete x;
                    });
                    e->name = c->name;
                    e->type = SYBL_INT_VAL;
                    e->elem = 8;
                    if ((rc = table_reserve(t, e.get(), t->phys_rows))) return rc;
                    if ((rc = table_upload_blocks(t))) return rc;
                    hipError_t he = launch_weight_carry(c->d_data, c->elem, c->vbase, c->d_valid, t->d_blocks, (int)t->blocks.size(), (int64_t *)e->d_data,
                                                        t->ctx->stream);
                    if (he == hipSuccess) he = hipStreamSynchronize(t->ctx->stream);
                    if (he != hipSuccess) return hip_fail(he, "k_weight_carry");
                    e->exact_min = std::min<int64_t>(c->exact_min, 1);
                    e->exact_max = std::max<int64_t>(c->exact_max, 1);
                    e->n_pop = t->logical_rows;
                    e->stats_blocks = (int64_t)t->blocks.size();
                    c->carried_weight = e;
                    c->carried_version = t->version;
                }
                q->eff_weight = c->carried_weight;
                if (fresh_slot) {
                    slot_col[(size_t)s] = kEffWeightCol;  // (the slot slot_of just made for the weight alone: it becomes the dense column's)
                } else {
                    if ((int)slot_col.size() >= kMaxSlots) return fail(SYBL_E_INVAL, "query references more than %d columns", kMaxSlots);
                    slot_col.push_back(kEffWeightCol);
                    folds.emplace_back();
                    s = (int)slot_col.size() - 1;
                }
            }
            P.slot[s].flags |= kSlotWeight;
            P.weight_slot = s;
        }
        return SYBL_OK;
    }

    int aggregations() {
        int rc;
        // ---- aggregations (aggregate.go:246-261, hist_basic.go:72-151)
        F = 1;  // field 0: Result.Count
        if (q->weighted) P.f_samples = F++;
        M = 0;
        hist_stride = 0;
        for (int a = 0; a < d->n_aggs; a++) {
            int s;
            if ((rc = slot_of(d->aggs[a], &s))) return rc;
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            if (c->type != SYBL_INT_VAL) {
                // the reference silently ignores non-int aggregation columns (aggregate.go:247-248)
                return fail(SYBL_E_INVAL, "aggregation column '%s' is not an int column", c->name.c_str());
            }
            if (P.slot[s].flags & kSlotAgg) return fail(SYBL_E_INVAL, "column '%s' aggregated twice", c->name.c_str());
            AggInfo ai;
            ai.col = slot_col[(size_t)s];
            ai.name = c->name;
            memset(&ai.d, 0, sizeof(ai.d));
            AggDesc &A = ai.d;
            int64_t lo = c->bounds_set ? c->bound_lo : c->exact_min, hi = c->bounds_set ? c->bound_hi : c->exact_max;
            bool empty = c->n_pop == 0 && !c->bounds_set;
            int64_t imin = c->info_given ? c->info_min : (empty ? 0 : lo);
            int64_t imax = c->info_given ? c->info_max : (empty ? 0 : hi);
            A.info_min = imin;
            A.max10 = (int64_t)((uint64_t)imax * 10u);  // Go's wrapping int64 multiply (hist_basic.go:104)
            A.f_sum = F++;
            bool can_reject = c->has_missing || (!empty && (lo < A.info_min || hi > A.max10));
            A.f_cnt = (q->weighted || can_reject) ? F++ : -1;
            A.f_smp = q->weighted ? F++ : -1;
            A.f_pop = c->has_missing ? F++ : -1;
            A.f_sb = A.f_sb2 = A.f_out = -1;
            // BasicHist.Min/Max start at Info.Min/Info.Max in hist mode and at 0 in avg mode
            // (hist_basic.go:34-40,72-85) and accepted values are >= Info.Min, so the running
            // extrema only need tracking when the column bounds let a value beat the start value.
            {
                // (a MultiHist starts at Info.Min / Info.Max in avg mode too, hist_multi.go:31-32)
                const bool info_start = d->op == SYBL_AGG_HIST || q->loghist;
                bool need_max = info_start ? (empty ? false : hi > imax) : (empty ? false : hi > 0);
                bool need_min = info_start ? false : (empty ? false : lo < 0);
                if (c->bounds_set == false && empty) need_max = need_min = false;
                A.m_max = need_max ? M++ : -1;
                A.m_nmin = need_min ? M++ : -1;
            }
            ai.f_out = -1;
            ai.num_buckets = 0;
            ai.info_max = imax;
            if (d->op == SYBL_AGG_HIST && q->loghist) {
                // MultiHist.TrackPercentiles, hist_multi.go:223-257: sub-histograms over ranges that halve from Info.Max
                // downwards until one is at most NUM_BUCKETS wide, the last one from Info.Min to the left edge reached
                if (imax < imin) return fail(SYBL_E_INVAL, "IntInfo of '%s' has max < min", c->name.c_str());
                if ((unsigned __int128)((__int128)imax - (__int128)imin) >= ((unsigned __int128)1 << 62))
                    return fail(SYBL_E_INVAL, "IntInfo range of '%s' is too wide for -loghist", c->name.c_str());
                std::vector<std::pair<int64_t, int64_t>> ranges;
                {
                    int64_t width = imax - imin, right = imax;
                    int n = 0;
                    for (int64_t tw = width; tw > 1000; tw >>= 1) n++;
                    for (int k = 0; k < n; k++) {
                        width >>= 1;
                        ranges.emplace_back(right - width, right);
                        right -= width;
                    }
                    ranges.emplace_back(imin, right);
                }
                A.hmin = imin;
                A.bucket_size = 1;
                A.inv_bucket = 1.0;
                A.multi_off = (int32_t)q->h_multi.size();
                A.multi_n = (int32_t)ranges.size();
                int64_t words = 0;
                for (auto &rg : ranges) {
                    int64_t bs, nb, nv;
                    setup_buckets(rg.first, rg.second, d->hist_bucket, &bs, &nb, &nv);
                    if (bs <= 0 || nv <= 0) return fail(SYBL_E_INVAL, "bad bucket geometry for '%s'", c->name.c_str());
                    // outliers of the sub-histogram: values from mn + nv * bs up to mx, one exact counter each
                    const __int128 first = (__int128)rg.first + (__int128)nv * bs;
                    const int64_t n_ext = first <= rg.second ? (int64_t)((__int128)rg.second - first + 1) : 0;
                    MultiSub S;
                    memset(&S, 0, sizeof(S));
                    S.mn = rg.first;
                    S.mx = rg.second;
                    S.max10 = (int64_t)((uint64_t)rg.second * 10u);
                    S.bs = bs;
                    S.inv_bs = 1.0 / (double)bs;
                    S.nv = (int32_t)std::min<int64_t>(nv, INT32_MAX);
                    S.big_div = (rg.second - rg.first) >= ((int64_t)1 << 51);
                    S.off = words;
                    S.ext_off = words + nv;
                    S.ext_first = n_ext > 0 ? (int64_t)first : rg.second;
                    S.n_ext = n_ext;
                    words += nv + n_ext;
                    if (words > (1 << 20))
                        return fail(SYBL_E_INVAL, "-loghist on '%s' needs more than 2^20 bucket words per group (with -int-bucket the "
                                                  "sub-histograms' outliers span the whole range)", c->name.c_str());
                    q->h_multi.push_back(S);
                    sybl_subhist sh;
                    sh.info_min = rg.first;
                    sh.info_max = rg.second;
                    sh.bucket_size = bs;
                    sh.num_buckets = nb;
                    sh.n_values = nv;
                    sh.offset = S.off;
                    sh.ext_first = S.ext_first;
                    sh.n_ext = n_ext;
                    sh.ext_offset = S.ext_off;
                    ai.subs.push_back(sh);
                }
                A.n_values = (int32_t)words;
                A.hist_full = 1;
                P.hist_agg_off[a] = hist_stride;
                hist_stride += words;
            } else if (d->op == SYBL_AGG_HIST) {
                if (imax < imin) return fail(SYBL_E_INVAL, "IntInfo of '%s' has max < min", c->name.c_str());
                int64_t bs, nb, nv;
                setup_buckets(imin, imax, d->hist_bucket, &bs, &nb, &nv);
                if (bs <= 0 || nv <= 0 || nv > (1 << 20)) return fail(SYBL_E_INVAL, "bad bucket geometry for '%s'", c->name.c_str());
                A.hmin = imin;
                A.bucket_size = bs;
                A.inv_bucket = 1.0 / (double)bs;
                A.n_values = (int32_t)nv;
                ai.num_buckets = nb;
                // accepted values lie in [max(lo,imin), min(hi,max10)]
                int64_t vhi = empty ? imin : std::min(hi, A.max10), vlo = empty ? imin : std::max(lo, imin);
                unsigned __int128 span = vhi >= A.hmin ? (unsigned __int128)((__int128)vhi - (__int128)A.hmin) : 0;
                A.big_div = span >= ((unsigned __int128)1 << 51);
                bool can_outlie = span / (unsigned __int128)bs >= (unsigned __int128)nv || vlo < A.hmin;
                if (can_outlie) {
                    A.f_out = F;
                    ai.f_out = F;
                    F += 6;
                }
                if (q->want_percentiles) {
                    A.hist_full = 1;
                    P.hist_agg_off[a] = hist_stride;
                    hist_stride += nv;
                } else {
                    A.f_sb = F++;
                    A.f_sb2 = F++;
                }
            }
            P.slot[s].flags |= kSlotAgg;
            P.slot[s].agg_index = a;
            P.agg[a] = A;
            q->aggs.push_back(ai);
        }
        // outlier values are only ever shown next to bucket arrays (-json buckets, -encode-results)
        {
            bool any_out = false;
            for (auto &ai : q->aggs) any_out = any_out || ai.d.f_out >= 0;
            if (any_out && q->want_percentiles && !env("SYBL_NO_OUTLIER_LOG")) {
                q->out_cap = kOutLogDefaultCap;
                if (const char *e = env("SYBL_OUTLIER_LOG_CAP")) q->out_cap = std::max<int64_t>(1, atoll(e));
                q->out_cap = (q->out_cap + kOutStripes - 1) / kOutStripes * kOutStripes;  // (equal stripes)
                SYBL_HIP(hipMalloc((void **)&q->d_out_log, (size_t)q->out_cap * kOutLogWords * 8));
                SYBL_HIP(hipMalloc((void **)&q->d_out_stage, ((size_t)kOutStripes * kOutCursorWords + (size_t)q->out_cap * kOutLogWords) * 8));
                P.out_log = q->d_out_stage;
                P.out_cap = q->out_cap;
            }
        }
        P.n_aggs = d->n_aggs;
        P.n_sum_fields = F;
        P.n_max_fields = M;
        P.hist_stride = hist_stride;
        P.hist_off = kHeaderWords + (int64_t)F * n_cells;
        if ((unsigned __int128)n_cells * (unsigned __int128)hist_stride > ((unsigned __int128)1 << 31))
            return fail(SYBL_E_INVAL, "groups x buckets = %lld x %lld words does not fit the 16 GiB histogram budget",
                        (long long)n_cells, (long long)hist_stride);
        return SYBL_OK;
    }

    int finish_slots() {
        int rc;
        // ---- finish slots
        if (slot_col.empty()) {
            // count(*) with no referenced column still needs the row count: stream any column
            if (t->cols.empty()) return fail(SYBL_E_INVAL, "table has no columns");
            int pick = -1;
            for (size_t k = 0; k < t->cols.size(); k++)
                if (t->cols[k]->type != SYBL_SET_VAL) { pick = (int)k; break; }
            if (pick < 0) return fail(SYBL_E_INVAL, "table has no int/str column to drive the scan");
            slot_col.push_back(pick);
            folds.emplace_back();
        }
        P.n_slots = (int)slot_col.size();
        for (int s = 0; s < P.n_slots; s++) {
            Column *c = slot_column(t, q, slot_col[(size_t)s]);
            SlotDesc &sd = P.slot[s];
            sd.base = c->d_data;
            sd.valid = c->d_valid;
            sd.width = c->elem;
            sd.vbase = c->vbase;
            if (c->type == SYBL_SET_VAL) {
                if ((rc = column_upload_set(t, c))) return rc;
                sd.flags |= kSlotSet;
                sd.base = c->d_set_off;
                sd.set_vals = c->d_set_vals;
                sd.n_setp = (int)folds[(size_t)s].setp.size();
                for (int k = 0; k < sd.n_setp; k++) {
                    sd.set_id[k] = folds[(size_t)s].setp[(size_t)k].first;
                    sd.set_in[k] = folds[(size_t)s].setp[(size_t)k].second;
                }
            }
            HostFilterFold &ff = folds[(size_t)s];
            if (ff.has_range) {
                sd.flags |= kSlotRange;
                sd.lo = ff.lo;
                sd.hi = ff.hi;
            }
            if (!ff.neq.empty()) {
                sd.flags |= kSlotNeq;
                sd.n_neq = (int)ff.neq.size();
                for (size_t k = 0; k < ff.neq.size(); k++) sd.neq[k] = ff.neq[k];
            }
            if (ff.has_mask) {
                sd.flags |= kSlotIdMask;
                size_t nbits = ff.mask.size(), nw = (nbits + 31) / 32 + 1;
                std::vector<uint32_t> bits(nw, 0);
                for (size_t k = 0; k < nbits; k++)
                    if (ff.mask[k]) bits[k >> 5] |= 1u << (k & 31);
                uint32_t *dm = nullptr;
                SYBL_HIP(hipMalloc((void **)&dm, nw * 4));
                q->d_idmasks.push_back(dm);
                SYBL_HIP(hipMemcpy(dm, bits.data(), nw * 4, hipMemcpyHostToDevice));
                sd.idmask = dm;
                sd.idmask_bits = (int32_t)nbits;
            }
        }
        return SYBL_OK;
    }

    // ---- filter pre-pass.  The packed row bodies evaluate <= 4 filter columns (ranges, neq constants, dictionary-id masks);
    // a set-membership filter, or a fifth filter column, used to send the whole query to the plan-interpreting k_scan (28 ms
    // per 1e9 rows of config 3 against 2.7).  Now those filters run first, as a kernel of their own over only their columns
    // (k_prefilter: the generic row_prepare, filters only) that writes one bit per row, and the scan proper reads that bitmap
    // like one more validity word (FastPlan::xvalid, the NUL variants).  Tried here, tentatively: the slots' filter flags
    // move to q->preplan; prefilter_commit keeps the split only if the rest of the query then runs a packed body.
    std::vector<uint32_t> pre_saved;  // the moved slots' original flags (by slot), 0 = not moved
    int prefilter() {
        q->pre_n_slots = 0;
        pre_saved.assign((size_t)P.n_slots, 0);
        if (env("SYBL_NO_PREFILTER") || q->loghist || !t->compact_mode || q->never_matches || d->n_distincts > 0) return SYBL_OK;
        if (q->op == SYBL_AGG_HIST && q->want_percentiles) return SYBL_OK;  // (the partitioned histograms have no NUL variants)
        std::vector<int> move;
        int n_fast = 0;
        for (int s = 0; s < P.n_slots; s++) {
            const uint32_t fl = P.slot[s].flags;
            if (fl & kSlotSet) {
                move.push_back(s);
            } else if (fl & kSlotFilter) {
                if (n_fast == kFastMaxF) move.push_back(s);
                else n_fast++;
            }
        }
        if (move.empty()) return SYBL_OK;
        ScanPlan &R = q->preplan;
        memset(&R, 0, sizeof(R));
        R.time_slot = -1;
        R.weight_slot = -1;
        R.f_samples = -1;
        q->pre_fps.clear();
        q->pre_fp_nf.clear();
        int n_generic = 0;
        for (size_t k = 0; k < move.size(); k++) {
            const SlotDesc &src = P.slot[move[k]];
            const Column *c = slot_column(t, q, slot_col[(size_t)move[k]]);
            // plain filter columns of <= 4 stored bytes: the offset-domain pre-pass (k_prefilter_packed), four to a launch
            const bool packed_ok = !(src.flags & kSlotSet) && c->type != SYBL_SET_VAL && c->elem <= 4 && !env("SYBL_NO_PREFILTER_PACKED");
            if (packed_ok) {
                if (q->pre_fps.empty() || q->pre_fp_nf.back() == kFastMaxF) {
                    q->pre_fps.emplace_back();
                    memset(&q->pre_fps.back(), 0, sizeof(FastPlan));
                    q->pre_fp_nf.push_back(0);
                }
                FastPlan &FP = q->pre_fps.back();
                const int i = q->pre_fp_nf.back()++;
                FP.fcol[i] = (const int64_t *)src.base;
                FP.fwid[i] = src.width;
                FP.fbase[i] = src.vbase;
                FP.fvalid[i] = src.valid;
                // (rebased onto the stored offsets as fill_packed does)
                const __int128 umax = ((__int128)1 << (8 * src.width)) - 1;
                const __int128 L = (__int128)((src.flags & kSlotRange) ? src.lo : INT64_MIN) - src.vbase;
                const __int128 H = (__int128)((src.flags & kSlotRange) ? src.hi : INT64_MAX) - src.vbase;
                if (H < 0 || L > umax || L > H) {
                    FP.plo[i] = 1;
                    FP.phi[i] = 0;
                } else {
                    FP.plo[i] = (uint32_t)(L < 0 ? 0 : L);
                    FP.phi[i] = (uint32_t)(H > umax ? umax : H);
                }
                FP.npneq[i] = 0;
                if (src.flags & kSlotNeq)
                    for (int j = 0; j < src.n_neq; j++) {
                        const __int128 off = (__int128)src.neq[j] - src.vbase;
                        if (off >= 0 && off <= umax) FP.pneq[i][FP.npneq[i]++] = (uint32_t)off;
                    }
                if (src.flags & kSlotIdMask) {
                    FP.fmask[i] = src.idmask;
                    FP.fmask_bits[i] = src.idmask_bits;
                }
            } else {
                SlotDesc &dst = R.slot[n_generic++];
                dst = src;
                dst.flags &= (kSlotFilter | kSlotSet);
                dst.gmissing = -1;
                dst.gmissing64 = -1;
                dst.agg_index = -1;
            }
            pre_saved[(size_t)move[k]] = P.slot[move[k]].flags;
            P.slot[move[k]].flags &= ~(uint32_t)(kSlotFilter | kSlotSet);
        }
        R.n_slots = n_generic;
        q->pre_generic_slots = n_generic;
        q->pre_n_slots = (int)move.size();
        return SYBL_OK;
    }
    // (called behind select_fast_path and select_hash_fast: a direct-mapped query must have taken a packed LDS body, a
    // hashed one the packed hash body -- both have variants that read validity words, and with them the bitmap)
    void prefilter_commit() {
        if (!q->pre_n_slots) return;
        const bool ok = q->hash_mode ? (q->hash_fast && q->hash_packed) : (q->fast && (q->fast_packed || q->fast_packed_n) && q->fplan.hist_lds == 0);
        if (ok) {
            q->fplan.nul = 1;  // (the variants that read validity words read the bitmap)
            return;            // (the bitmap itself: device_copies)
        }
        // the rest of the query does not run a packed body: everything stays with the plan interpreter
        for (int s = 0; s < P.n_slots; s++)
            if (pre_saved[(size_t)s]) P.slot[s].flags = pre_saved[(size_t)s];
        q->pre_n_slots = 0;
        select_fast_path(t, q, slot_col);
        select_hash_fast(t, q, slot_col);
    }

    int strategy() {
        // ---- strategy: cell table in LDS when it fits (DESIGN.md "Strategies")
        q->n_wg = ctx->n_cus > 0 ? ctx->n_cus : 256;
        if (const char *e = env("SYBL_WG_PER_CU")) q->n_wg *= std::max(1, atoi(e));
        int64_t lds_words = (int64_t)(F + M) * n_cells;
        q->use_lds = lds_words * 8 <= kLdsBudgetBytes && !q->hash_mode;
        P.rep_shift = 0;
        if (q->use_lds) {
            int rs = 0;
            // (SYBL_REP_BUDGET_KB: tuning -- a smaller cell table lets SYBL_WG_PER_CU workgroups share a CU)
            int64_t rep_budget = kLdsBudgetBytes;
            if (const char *e = env("SYBL_REP_BUDGET_KB")) rep_budget = std::min<int64_t>(kLdsBudgetBytes, std::max<int64_t>(1, atoll(e)) * 1024);
            while (rs < 6 && (lds_words * 8 << (rs + 1)) <= rep_budget) rs++;
            P.rep_shift = rs;
            q->lds_bytes = (size_t)(lds_words * 8) << rs;
        }
        // (+ room for the in-place reduce-scatter of the bucket arrays over cell slices of equal size, rccl.cpp)
        q->n_sum_words = kHeaderWords + (int64_t)F * n_cells + (n_cells + (hist_stride > 0 ? kMaxScatterRanks : 0)) * hist_stride;
        q->n_max_words = std::max<int64_t>((int64_t)M * n_cells, 1);
        return SYBL_OK;
    }

    int work() {
        // ---- work: non-skipped blocks -> runs of physical rows -> an equal share of tiles per workgroup
        std::vector<Segment> runs;
        rows_scanned = 0;
        skipped = 0;
        for (size_t b = 0; b < t->blocks.size(); b++) {
            if (t->blocks[b].n == 0) continue;
            if (d->block_skip && !should_scan_block(t, d, (int64_t)b)) {
                skipped++;
                continue;
            }
            rows_scanned += t->blocks[b].n;
            const Segment &blk = t->blocks[b];
            if (!runs.empty() && runs.back().start + runs.back().n == blk.start) {
                runs.back().n += blk.n;
            } else {
                runs.push_back(blk);
            }
        }
        int64_t total_tiles = 0;
        for (auto &r : runs) total_tiles += (r.n + kTileRows - 1) / kTileRows;
        q->segs.clear();
        q->wg_seg_begin.assign((size_t)q->n_wg + 1, 0);
        {
            size_t ri = 0;
            int64_t tile_in_run = 0;  // tiles of runs[ri] already handed out
            for (int w = 0; w < q->n_wg; w++) {
                q->wg_seg_begin[(size_t)w] = (int32_t)q->segs.size();
                int64_t want = total_tiles * (w + 1) / q->n_wg - total_tiles * w / q->n_wg;
                while (want > 0 && ri < runs.size()) {
                    int64_t run_tiles = (runs[ri].n + kTileRows - 1) / kTileRows;
                    int64_t take = std::min(want, run_tiles - tile_in_run);
                    Segment sg;
                    sg.start = runs[ri].start + tile_in_run * kTileRows;
                    int64_t end = std::min(runs[ri].start + runs[ri].n, sg.start + take * kTileRows);
                    sg.n = end - sg.start;
                    q->segs.push_back(sg);
                    tile_in_run += take;
                    want -= take;
                    if (tile_in_run == run_tiles) {
                        ri++;
                        tile_in_run = 0;
                    }
                }
            }
            q->wg_seg_begin[(size_t)q->n_wg] = (int32_t)q->segs.size();
        }
        return SYBL_OK;
    }

    // FastPlan::cshift (scan_fast.h): in avg mode over compact storage, with every value populated, Result.Count shares an LDS
    // word with aggregation 0's sum of stored offsets.  A word belongs to one (cell, replica): it takes the rows of the
    // 1024 >> rep_shift lanes of that replica, each of which sees four rows of every tile of its workgroup -- that many rows, and
    // that many times the column's largest offset, must fit below and above bit `cshift`.
    void plan_count_packing() {
        FastPlan &FP = q->fplan;
        FP.cshift = 0;
        if (!q->fast || !q->fast_packed || q->fast_packed_n || q->part_hist || FP.nul || q->fast_na < 1 || env("SYBL_NO_CPACK")) return;
        if (q->fast_mode != kFastAvg && q->fast_mode != kFastAvgMax) return;
        int64_t lane_rows = 0;
        for (int w = 0; w < q->n_wg; w++) {
            int64_t mine = 0;
            for (int32_t si = q->wg_seg_begin[(size_t)w]; si < q->wg_seg_begin[(size_t)w + 1]; si++)
                mine += (q->segs[(size_t)si].n + kPackedTileRows - 1) / kPackedTileRows * kPackedRows;
            lane_rows = std::max(lane_rows, mine);
        }
        const unsigned __int128 slot_rows = (unsigned __int128)std::max<int64_t>(lane_rows, 1) * (unsigned __int128)(kWgThreads >> P.rep_shift);
        const Column *c = t->cols[(size_t)q->aggs[0].col].get();
        // (the stored offsets of the resident rows: exact extrema minus the storage base, never above what the width holds)
        unsigned __int128 umax = c->elem >= 4 ? 0xFFFFFFFFull : c->elem == 2 ? 0xFFFFull : 0xFFull;
        if (c->n_pop > 0 && c->exact_max >= c->vbase) umax = std::min<unsigned __int128>(umax, (unsigned __int128)((__int128)c->exact_max - (__int128)c->vbase));
        auto bits = [](unsigned __int128 x) {
            int b = 0;
            while (x) b++, x >>= 1;
            return b;
        };
        const int sum_bits = std::max(1, bits(umax * slot_rows)), count_bits = bits(slot_rows);
        if (sum_bits + count_bits <= 64 && sum_bits < 63) FP.cshift = sum_bits;
    }

    // -limit pushed into the scan (pushdown.hip).  Taken when the caller said the rows beyond the limit need nothing but their
    // Count (printed_only = 2), the query is one strategy 5 runs in a single pass over compact storage with ONE key column
    // and no filter, and the order is $COUNT descending.  The ranks of a job agree on the printed cells before pass 2: the
    // groups' counts are SUM-all-reduced between the passes (engine.cpp: scan -- sybl_query_scan is then a collective call);
    // every rank picks the same cells from them and the merge is round 5's limit-aware one (cell fields, Cumulative, the
    // printed rows' arrays).  Whether it is taken is itself agreed: what follows depends on a rank's rows, so the ranks ask
    // each other once per prepared query (pd_static / pd_agreed) and take it only if every one of them planned it.
    int plan_pushdown() {
        q->pushdown = false;
        // (what depends on the query alone, the same on every rank: whether the ranks have to ask each other at the first scan --
        // what follows depends on this rank's rows too, and a rank without any plans no partitioned histograms at all)
        q->pd_static = q->printed_level == 2 && !env("SYBL_NO_PUSHDOWN") && !env("SYBL_NO_PUSHDOWN_RANKS") && q->limit > 0 && q->order_by == "$COUNT" &&
                       !q->order_asc && d->n_groups == 1 && d->n_filters == 0 && d->op == SYBL_AGG_HIST && d->n_aggs >= 1 && d->n_aggs <= 2 &&
                       d->n_distincts == 0 && ctx->comm && ctx->comm_nranks > 1;
        q->pd_agreed = -1;
        if (q->printed_level != 2 || env("SYBL_NO_PUSHDOWN")) return SYBL_OK;
        if (!q->part_hist || !q->part_packed || !q->part_more.empty() || q->part_nf != 0 || q->part_ng != 1) return SYBL_OK;
        if (q->limit <= 0 || q->order_by != "$COUNT" || q->order_asc) return SYBL_OK;
        if (ctx->comm_nranks > 1 && env("SYBL_NO_PUSHDOWN_RANKS")) return SYBL_OK;  // (A/B: the limit-aware merge of round 5 instead)
        if (P.n_cells < 2048 || P.n_cells > 65536 || q->part_na < 1 || q->part_na > 2 || q->n_distinct) return SYBL_OK;
        const FastPlan &FP = q->eplan.fp;
        if (FP.gcard[0] != (uint32_t)P.n_cells || FP.gstride[0] != 1 || P.hist_stride > 8192) return SYBL_OK;
        for (int a = 0; a < q->part_na; a++)
            if (q->aggs[(size_t)a].d.f_out >= 0 || !q->aggs[(size_t)a].d.hist_full) return SYBL_OK;  // (an outlier is possible: the full path remembers it)
        PushdownPlan &D = q->dplan_pd;
        memset(&D, 0, sizeof(D));
        D.fp = FP;
        // (the emit plan's records carry v - h.Min + BucketSize -- plan_part_pass biased adoff --; the buckets here are plain)
        for (int a = 0; a < q->part_na; a++) D.fp.adoff[a] -= D.fp.bucket_size[a];
        D.n_cells = P.n_cells;
        D.n_aggs = q->part_na;
        D.n_wg = q->n_wg;
        D.limit = q->limit;
        D.hist_off = P.hist_off;
        D.hist_stride = P.hist_stride;
        const size_t words = ((size_t)P.n_cells + 1) / 2, bm = ((size_t)P.n_cells + 31) / 32;
        const size_t total = (size_t)q->n_wg * words + 2 * (size_t)P.n_cells + bm + (size_t)q->limit + 4;
        SYBL_HIP(hipMalloc((void **)&q->d_pd, total * 4));
        D.ws = q->d_pd;
        D.carry = D.ws + (size_t)q->n_wg * words;
        D.cnt = D.carry + P.n_cells;
        D.bitmap = D.cnt + P.n_cells;
        D.top_cells = (int32_t *)(D.bitmap + bm);
        D.n_top = D.top_cells + q->limit;
        q->pushdown = true;
        return SYBL_OK;
    }

    int window() {
        int rc;
        // ---- LDS-window strategy: a time-series table too large for LDS, scanned by workgroups whose
        // contiguous rows each span only a few time buckets (tables are digested in time order,
        // table_io.go:119-122 sorts by Timestamp).  Exact per-block extrema of the time column give
        // every workgroup its window.
        P.windowed = 0;
        P.lds_cells = (int32_t)n_cells;
        P.wg_cell_base = nullptr;
        if (q->hash_mode) {
            // LDS staging table of k_scan_hash: keys + every cell field per slot, as many slots (a power of two) as fit.
            // Bucket arrays cannot be staged (they live in the global table only).
            int64_t L = 0;
            if (hist_stride == 0 && !env("SYBL_NO_HASH_LDS")) {
                L = 1;
                while (2 * L * 8 * (1 + F + M) <= kLdsBudgetBytes) L <<= 1;
                if (L < 64) L = 0;
            }
            P.lds_cells = (int32_t)L;
            q->lds_bytes = (size_t)(L * 8 * (1 + F + M));
        }
        // (a hashed query keeps its staging table: the window below is a direct-mapped slice of [time bucket][cell])
        if (!q->use_lds && !q->hash_mode && q->time_mode && !env("SYBL_NO_WINDOW") && !t->blocks.empty()) {
            const Column *tc = slot_column(t, q, slot_col[(size_t)P.time_slot]);
            std::vector<int32_t> base((size_t)q->n_wg, 0);
            int64_t wmax = 1;
            bool ok = true;
            for (int w = 0; w < q->n_wg && ok; w++) {
                int64_t lo = INT64_MAX, hi = INT64_MIN;
                for (int32_t si = q->wg_seg_begin[(size_t)w]; si < q->wg_seg_begin[(size_t)w + 1]; si++) {
                    const Segment &sg = q->segs[(size_t)si];
                    // first block whose end is beyond the segment start
                    size_t b = (size_t)(std::upper_bound(t->blocks.begin(), t->blocks.end(), sg.start,
                                                         [](int64_t v, const Segment &blk) { return v < blk.start + blk.n; }) -
                                        t->blocks.begin());
                    for (; b < t->blocks.size() && t->blocks[b].start < sg.start + sg.n; b++) {
                        if (tc->blk_pop[b] == 0) continue;
                        lo = std::min(lo, tc->blk_min[b]);
                        hi = std::max(hi, tc->blk_max[b]);
                    }
                }
                if (hi < lo) continue;  // no populated time value: every row is dropped anyway
                int64_t tlo = lo / d->time_bucket - P.tb_min, thi = hi / d->time_bucket - P.tb_min;
                if (tlo < 0 || thi >= P.n_tb) {
                    ok = false;  // declared bounds narrower than the data: the kernel would count overflow
                    break;
                }
                base[(size_t)w] = (int32_t)(tlo * cells);
                wmax = std::max(wmax, thi - tlo + 1);
            }
            int64_t lds_cells = wmax * cells;
            if (ok && lds_cells * (F + M) * 8 <= kLdsBudgetBytes) {
                q->use_lds = true;
                P.windowed = 1;
                P.lds_cells = (int32_t)lds_cells;
                int rs = 0;
                int64_t words = lds_cells * (F + M);
                while (rs < 6 && (words * 8 << (rs + 1)) <= kLdsBudgetBytes) rs++;
                P.rep_shift = rs;
                q->lds_bytes = (size_t)(words * 8) << rs;
                SYBL_HIP(hipMalloc((void **)&q->d_wg_cell_base, base.size() * 4));
                SYBL_HIP(hipMemcpy(q->d_wg_cell_base, base.data(), base.size() * 4, hipMemcpyHostToDevice));
                P.wg_cell_base = q->d_wg_cell_base;
            }
        }
        select_fast_path(t, q, slot_col);
        select_hash_fast(t, q, slot_col);
        prefilter_commit();
        if ((rc = select_part_hist(t, q, slot_col, rows_scanned))) return rc;
        plan_count_packing();
        if ((rc = plan_pushdown())) return rc;
        q->stats.rows_scanned = rows_scanned;
        q->stats.blocks_skipped = skipped;
        q->stats.blocks_scanned = (int64_t)t->blocks.size() - skipped;
        int64_t width = 0, canon_width = 0;
        int64_t set_bytes = 0;
        for (int s = 0; s < P.n_slots; s++) {
            const Column *c = slot_column(t, q, slot_col[(size_t)s]);
            if (c->type == SYBL_SET_VAL) {
                width += 8;  // one CSR offset per row
                canon_width += 8;
                set_bytes += (int64_t)c->h_set_vals.size() * 4;
            } else {
                width += c->elem;
                canon_width += c->canon();
            }
        }
        q->stats.algorithmic_bytes = rows_scanned * width + set_bytes;
        q->stats.canonical_bytes = rows_scanned * canon_width + set_bytes;
        q->stats.n_cells = (int32_t)n_cells;
        q->stats.packed_kernel = q->part_hist ? q->part_packed : (q->fast && q->fast_packed);
        q->stats.strategy = q->pushdown ? 8 : q->part_hist ? 5 : (q->use_lds ? (P.windowed ? (q->fast ? 4 : 3) : (q->fast ? (q->fplan.hist_lds ? 6 : 2) : 0)) : (q->hash_mode ? 7 : 1));
        q->stats.lds_bytes = (int32_t)q->lds_bytes;
        q->stats.n_workgroups = q->n_wg;
        q->stats.replicas = 1 << P.rep_shift;
        q->stats.n_sum_fields = P.n_sum_fields;
        q->stats.n_max_fields = P.n_max_fields;
        return SYBL_OK;
    }

    int device_copies() {
        // ---- device-side copies
        size_t nseg = std::max<size_t>(q->segs.size(), 1);
        SYBL_HIP(hipMalloc((void **)&q->d_segs, nseg * sizeof(Segment)));
        if (!q->segs.empty())
            SYBL_HIP(hipMemcpy(q->d_segs, q->segs.data(), q->segs.size() * sizeof(Segment), hipMemcpyHostToDevice));
        SYBL_HIP(hipMalloc((void **)&q->d_wg_seg_begin, q->wg_seg_begin.size() * 4));
        SYBL_HIP(hipMemcpy(q->d_wg_seg_begin, q->wg_seg_begin.data(), q->wg_seg_begin.size() * 4, hipMemcpyHostToDevice));
        P.segs = q->d_segs;
        P.wg_seg_begin = q->d_wg_seg_begin;
        if (q->use_lds && !P.windowed) {
            SYBL_HIP(hipMalloc((void **)&q->d_ws_sum, (size_t)q->n_wg * F * n_cells * 8));
            SYBL_HIP(hipMalloc((void **)&q->d_ws_max, (size_t)q->n_wg * std::max<int64_t>((int64_t)M * n_cells, 1) * 8));
            P.ws_sum = q->d_ws_sum;
            P.ws_max = q->d_ws_max;
        }
        q->eplan.fp.segs = q->d_segs;
        q->eplan.fp.wg_seg_begin = q->d_wg_seg_begin;
        for (auto &pp : q->part_more) {
            pp.E.fp.segs = q->d_segs;
            pp.E.fp.wg_seg_begin = q->d_wg_seg_begin;
        }
        q->fplan.segs = q->d_segs;
        q->fplan.wg_seg_begin = q->d_wg_seg_begin;
        q->fplan.ws_sum = q->d_ws_sum;
        q->fplan.ws_max = q->d_ws_max;
        if (!q->h_multi.empty()) {
            SYBL_HIP(hipMalloc((void **)&q->d_multi, q->h_multi.size() * sizeof(MultiSub)));
            SYBL_HIP(hipMemcpy(q->d_multi, q->h_multi.data(), q->h_multi.size() * sizeof(MultiSub), hipMemcpyHostToDevice));
            P.multi = q->d_multi;
        }
        SYBL_HIP(hipMalloc((void **)&q->d_plan, sizeof(ScanPlan)));
        if (q->pre_n_slots) {
            const size_t words = (size_t)(t->phys_rows / 32 + 2);
            SYBL_HIP(hipMalloc((void **)&q->d_prebits, words * 4));
            SYBL_HIP(hipMemset(q->d_prebits, 0, words * 4));
            q->preplan.segs = q->d_segs;
            q->preplan.wg_seg_begin = q->d_wg_seg_begin;
            for (auto &fp : q->pre_fps) {
                fp.segs = q->d_segs;
                fp.wg_seg_begin = q->d_wg_seg_begin;
            }
            SYBL_HIP(hipMalloc((void **)&q->d_preplan, sizeof(ScanPlan)));
            SYBL_HIP(hipMemcpy(q->d_preplan, &q->preplan, sizeof(ScanPlan), hipMemcpyHostToDevice));
            q->fplan.xvalid = q->d_prebits;
        }
        for (auto &e : q->ev) SYBL_HIP(hipEventCreate(&e));
        q->plan_dirty = true;
        return SYBL_OK;
    }

    // ---- count distinct (aggregate.go:78-91,205-243): the sketches and the plan k_scan_distinct reads -- the scan's
    // own plan (filters, group key, time bucket: row_prepare) plus a slot per distinct column it does not reference
    int distinct() {
        q->n_distinct = 0;
        const int n = d->n_distincts;
        if (n <= 0) return SYBL_OK;
        if (n > kMaxDistinct) return fail(SYBL_E_INVAL, "too many distinct columns (%d > %d)", n, kMaxDistinct);
        // (a hashed group-by keeps a sketch per key it FOUND: sized, filled and merged once the key set is final --
        // engine.cpp: query_hash_distinct)
        if (!q->hash_mode && n_cells * (int64_t)kHllRegs > ((int64_t)8 << 30))
            return fail(SYBL_E_INVAL, "count distinct: %lld groups x 16 KB of sketch exceed 8 GiB", (long long)n_cells);
        std::vector<Column *> cols;
        int n_str = 0;
        for (int i = 0; i < n; i++) {
            Column *c = t->find(d->distincts[i]);
            if (!c) return fail(SYBL_E_INVAL, "unknown column '%s'", d->distincts[i] ? d->distincts[i] : "(null)");
            if (c->type == SYBL_SET_VAL) return fail(SYBL_E_INVAL, "count distinct over the set column '%s'", c->name.c_str());
            n_str += c->type == SYBL_STR_VAL;
            cols.push_back(c);
        }
        // only_ints_in_distinct (:84-91), the slow path over one str column (its hash follows from the dictionary id), or
        // the slow path over several columns (round 5: the buffer -- digits / strings / "\t" -- is assembled and hashed per row)
        ScanPlan &D = q->dplan;
        D = P;
        for (int i = 0; i < n; i++) {
            const int ci = t->col_ix[cols[(size_t)i]->name];
            int s = -1;
            for (size_t k = 0; k < slot_col.size(); k++)
                if (slot_col[k] == ci) s = (int)k;
            if (s < 0) {
                if (D.n_slots >= kMaxSlots) return fail(SYBL_E_INVAL, "query references more than %d columns", kMaxSlots);
                s = D.n_slots++;
                slot_col.push_back(ci);  // (after the scan's own slots: nothing below looks at them as scan slots)
                SlotDesc &sd = D.slot[s];
                memset(&sd, 0, sizeof(sd));
                sd.base = cols[(size_t)i]->d_data;
                sd.valid = cols[(size_t)i]->d_valid;
                sd.width = cols[(size_t)i]->elem;
                sd.vbase = cols[(size_t)i]->vbase;
                sd.gmissing = -1;
                sd.gmissing64 = -1;
                sd.agg_index = -1;
            }
            D.distinct_slot[i] = s;
        }
        D.n_distinct = n;
        D.hll_keys = nullptr;
        D.hll_nkeys = 0;
        q->hll_bytes = q->hash_mode ? 0 : n_cells * (int64_t)kHllRegs;
        if (!q->hash_mode) SYBL_HIP(hipMalloc((void **)&q->d_hll, (size_t)q->hll_bytes));
        D.hll = q->d_hll;
        D.hll_mixed = n_str > 0 && n > 1;
        for (int i = 0; i < 8; i++) {
            D.hll_chars[i] = nullptr;
            D.hll_stroff[i] = nullptr;
            D.hll_nids[i] = 0;
        }
        if (D.hll_mixed) {
            // the str columns' dictionaries on the device, as -str-replace left them: chars back to back + offsets
            std::string chars;
            std::vector<std::vector<int64_t>> offs((size_t)n);
            for (int i = 0; i < n; i++) {
                Column *c = cols[(size_t)i];
                if (c->type != SYBL_STR_VAL) continue;
                const StrReplaced *rep = nullptr;
                auto it = q->replaced.find(t->col_ix[c->name]);
                if (it != q->replaced.end()) rep = it->second.get();
                offs[(size_t)i].reserve(c->dict.size() + 1);
                for (size_t id = 0; id < c->dict.size(); id++) {
                    offs[(size_t)i].push_back((int64_t)chars.size());
                    chars += rep ? rep->strs[(size_t)rep->remap[id]] : c->dict[id];
                }
                offs[(size_t)i].push_back((int64_t)chars.size());
            }
            size_t n_off = 0;
            for (auto &o : offs) n_off += o.size();
            SYBL_HIP(hipMalloc((void **)&q->d_hll_chars, std::max<size_t>(chars.size(), 1)));
            SYBL_HIP(hipMalloc((void **)&q->d_hll_stroff, std::max<size_t>(n_off, 1) * 8));
            SYBL_HIP(hipMemcpy(q->d_hll_chars, chars.data(), chars.size(), hipMemcpyHostToDevice));
            size_t at = 0;
            for (int i = 0; i < n; i++) {
                if (offs[(size_t)i].empty()) continue;
                SYBL_HIP(hipMemcpy(q->d_hll_stroff + at, offs[(size_t)i].data(), offs[(size_t)i].size() * 8, hipMemcpyHostToDevice));
                D.hll_chars[i] = q->d_hll_chars;  // (the offsets are absolute)
                D.hll_stroff[i] = q->d_hll_stroff + at;
                D.hll_nids[i] = (int64_t)offs[(size_t)i].size() - 1;
                at += offs[(size_t)i].size();
            }
        } else if (n_str) {
            // aggregate.go:225-239: the string, then GROUP_DELIMITER; a row without the column hashes the delimiter alone.
            // (-str-replace rewrote the dictionary at load time in the reference: the rewritten string is what is hashed)
            Column *c = cols[0];
            const StrReplaced *rep = nullptr;
            auto it = q->replaced.find(t->col_ix[c->name]);
            if (it != q->replaced.end()) rep = it->second.get();
            std::vector<uint64_t> h(std::max<size_t>(c->dict.size(), 1));
            for (size_t id = 0; id < c->dict.size(); id++) {
                std::string v = rep ? rep->strs[(size_t)rep->remap[id]] : c->dict[id];
                v += "\t";
                h[id] = metro64_bytes((const uint8_t *)v.data(), v.size(), kHllSeed);
            }
            SYBL_HIP(hipMalloc((void **)&q->d_hll_idhash, h.size() * 8));
            SYBL_HIP(hipMemcpy(q->d_hll_idhash, h.data(), h.size() * 8, hipMemcpyHostToDevice));
            D.hll_idhash = q->d_hll_idhash;
            D.hll_ids = (int64_t)c->dict.size();
            D.hll_missing = metro64_bytes((const uint8_t *)"\t", 1, kHllSeed);
        }
        SYBL_HIP(hipMalloc((void **)&q->d_dplan, sizeof(ScanPlan)));
        SYBL_HIP(hipMemcpy(q->d_dplan, &D, sizeof(ScanPlan), hipMemcpyHostToDevice));
        q->n_distinct = n;
        return SYBL_OK;
    }

    int run() {
        int rc;
        if ((rc = setup())) return rc;
        if ((rc = filters())) return rc;
        if ((rc = groups())) return rc;
        if ((rc = time_series())) return rc;
        if ((rc = weight())) return rc;
        if ((rc = aggregations())) return rc;
        if ((rc = finish_slots())) return rc;
        if ((rc = prefilter())) return rc;
        if ((rc = strategy())) return rc;
        if ((rc = work())) return rc;
        if ((rc = window())) return rc;
        if ((rc = device_copies())) return rc;
        return distinct();
    }
};

int plan_query(Table *t, const sybl_query_desc *d, Query *q) {
    Planner p(t, d, q);
    return p.run();
}

}  // namespace sybl
