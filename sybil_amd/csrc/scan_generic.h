// scan_generic.h -- device code shared by the plan-interpreting scan kernels: k_scan<NC,USE_LDS> (kernels.hip: direct-mapped
// cell table in LDS or HBM) and k_scan_hash<NC> (hashgroup.hip: group-by through an open-addressing table).
//
// The per-row body of FilterAndAggRecords (reference src/lib/aggregate.go:96-263) + BasicHist.AddWeightedValue
// (hist_basic.go:101-151) in three pieces: tile loads (issue_tile / decode_tile), row_prepare (filters, group key,
// weight, time bucket) and row_accumulate (Count / Samples / per-aggregation fields of one cell).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"
#include "outlog.h"
#include "wg_header.h"

namespace sybl {

typedef long long ll2 __attribute__((ext_vector_type(2)));
// The plan lives in device memory and is read through the constant address space so every
// field access is a scalar (s_load) through the scalar cache: the plan is wave-uniform, far
// larger than the SGPR file, and must never be copied to scratch.
typedef const ScanPlan __attribute__((address_space(4))) CPlan;
typedef const SlotDesc __attribute__((address_space(4))) CSlot;
typedef const AggDesc __attribute__((address_space(4))) CAgg;

// ---------------------------------------------------------------- small helpers

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// floor(n / d) for 0 <= n, d > 0 without the ~100-instruction 64-bit divide:
// double reciprocal + one correction step (exact while n < 2^52).
__device__ __forceinline__ uint64_t udiv_fast(uint64_t n, uint64_t d, double inv_d, int big) {
    if (big) return n / d;
    uint64_t q = (uint64_t)((double)n * inv_d);
    int64_t r = (int64_t)(n - q * d);
    if (r < 0) {
        q -= 1;
    } else if ((uint64_t)r >= d) {
        q += 1;
    }
    return q;
}

// Go's truncating int64 division by a positive constant (aggregate.go:174, hist_basic.go:130)
__device__ __forceinline__ int64_t sdiv_trunc(int64_t x, int64_t d, double inv_d, int big) {
    uint64_t ux = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x;
    uint64_t q = udiv_fast(ux, (uint64_t)d, inv_d, big);
    return x < 0 ? -(int64_t)q : (int64_t)q;
}

__device__ __forceinline__ uint32_t dict_hash(int64_t x) { return (uint32_t)(splitmix64((uint64_t)x) >> 32); }

template <bool USE_LDS>
__device__ __forceinline__ void acc_add(int64_t *tab, int64_t idx, int64_t v) {
    if (USE_LDS) {
        __hip_atomic_fetch_add(&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_fetch_add(&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool USE_LDS>
__device__ __forceinline__ void acc_max(int64_t *tab, int64_t idx, int64_t v) {
    // a plain read first: after warm-up almost no value raises the extremum, and a stale
    // (smaller) read only costs one redundant atomic
    if (v > tab[idx]) {
        if (USE_LDS) {
            __hip_atomic_fetch_max(&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_max(&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ void gadd(int64_t *p, int64_t v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int64_t wave_sum(int64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- tiles

template <int NC>
struct Tile {
    ll2 v[NC];         // two consecutive rows per slot
    uint32_t pop[NC];  // 2 validity bits per slot (bit0 = row0, bit1 = row1)
};

// The loaded bits of a tile are kept raw and decoded (value = vbase + zero-extended raw) only when the
// tile is consumed.  The load instruction is the same 16-byte buffer load for every stored width -- its
// descriptor spans exactly the wave's 128 rows, so the lanes of a narrow column read nothing beyond
// them -- which keeps the issue branch-free and a tile's loads back to back (see fast_issue in
// scan_fast.h for what a width switch around the loads costs).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NC>
struct RawTile {
    u32x4 v[NC];
    uint32_t pw[NC];  // validity word of the two rows (all ones: fully populated column)
};

template <int NC>
__device__ __forceinline__ void issue_tile(CPlan &P, int64_t row, RawTile<NC> &t) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)row), hi = __builtin_amdgcn_readfirstlane((uint32_t)(row >> 32));
    const int64_t row0 = (int64_t)(((uint64_t)hi << 32) | lo);  // the wave's first row
    const uint32_t lane_row = (uint32_t)(row - row0);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        CSlot &s = P.slot[c];
        if (!(s.flags & kSlotSet)) {  // set columns: the CSR is walked per row (row_prepare)
            const int ws = s.width == 8 ? 3 : s.width >> 1;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)s.base + (row0 << ws)), 0,
                                                                                  (int)((64u * kRowsPerThread) << ws), 0x00020000);
            t.v[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((lane_row << ws) & ~3u), 0, 2);
        }
        t.pw[c] = s.valid ? s.valid[row >> 5] : 0xFFFFFFFFu;
    }
}

template <int NC>
__device__ __forceinline__ void decode_tile(CPlan &P, int64_t row, bool in_range, const RawTile<NC> &r, Tile<NC> &t) {
#pragma unroll
    for (int c = 0; c < NC; c++) {
        CSlot &s = P.slot[c];
        ll2 v = {0, 0};
        if (!(s.flags & kSlotSet)) {
            const u32x4 w = r.v[c];
            switch (s.width) {
            case 8:
                v.x = (long long)(((unsigned long long)w.y << 32) | w.x);
                v.y = (long long)(((unsigned long long)w.w << 32) | w.z);
                break;
            case 4:
                v.x = s.vbase + (long long)w.x;
                v.y = s.vbase + (long long)w.y;
                break;
            case 2:
                v.x = s.vbase + (long long)(w.x & 0xFFFFu);
                v.y = s.vbase + (long long)(w.x >> 16);
                break;
            default: {
                const uint32_t x = w.x >> ((uint32_t)(row & 2) * 8u);  // rows 4k+2, 4k+3 sit in the upper half
                v.x = s.vbase + (long long)(x & 0xFFu);
                v.y = s.vbase + (long long)((x >> 8) & 0xFFu);
                break;
            }
            }
        }
        t.v[c] = v;
        t.pop[c] = in_range ? (r.pw[c] >> (row & 31)) & 3u : 0u;
    }
}

// ---------------------------------------------------------------- one row

enum RowState : int {
    kRowFail = 0,      // a filter rejected the row
    kRowDropped = 1,   // matched, but a time-series row without a time value (aggregate.go:146-183)
    kRowOverflow = 2,  // matched, key / time bucket outside the declared bounds (reported by finalize)
    kRowOk = 3,        // matched: `key` is the composite group key (= the cell number when direct-mapped), `w` the weight
};

// filters (aggregate.go:105-116), group key (aggregate.go:125-143), weight (:100-102), time bucket (:146-183)
template <int NC>
__device__ __forceinline__ int row_prepare(CPlan &P, const Tile<NC> &t, int r, int64_t row0, uint64_t &key, int64_t &w) {
    bool pass = true;
    bool in_bounds = true;
    key = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        CSlot &s = P.slot[c];
        const int64_t x = r == 0 ? t.v[c].x : t.v[c].y;
        const bool pop = (t.pop[c] >> r) & 1u;
        if (s.flags & kSlotRange) pass = pass && pop && x >= s.lo && x <= s.hi;
        if (s.flags & kSlotNeq) {
            pass = pass && pop;
            for (int k = 0; k < s.n_neq; k++) pass = pass && x != s.neq[k];
        }
        if (s.flags & kSlotSet) {
            // SetFilter.Filter, filter.go:252-285: "in" = some member equals the id, "nin" = none;
            // a row without the set column fails both
            bool ok = pop;
            if (pop) {
                const int64_t *off = (const int64_t *)s.base;
                const int64_t lo = off[row0 + r], hi = off[row0 + r + 1];
                uint32_t hit = 0;
                for (int64_t m = lo; m < hi; m++) {
                    const int32_t id = s.set_vals[m];
                    for (int p = 0; p < s.n_setp; p++) hit |= (id == s.set_id[p] ? 1u : 0u) << p;
                }
                for (int p = 0; p < s.n_setp; p++) ok = ok && (((hit >> p) & 1u) == (uint32_t)s.set_in[p]);
            }
            pass = pass && ok;
        }
        if (s.flags & kSlotIdMask) {
            bool ok = false;
            if (pop && (uint64_t)x < (uint64_t)s.idmask_bits) ok = (s.idmask[x >> 5] >> (x & 31)) & 1u;
            pass = pass && ok;
        }
        if ((s.flags & kSlotGroup) && (s.flags & kSlotDict)) {
            // sparse key range: the digit is the value's rank among the column's distinct values
            if (pop) {
                uint32_t h = dict_hash(x) & s.dmask;
                int32_t rank = -1;
                for (uint32_t probe = 0; probe <= s.dmask; probe++) {
                    const int64_t kx = s.dkeys[h];
                    if (kx == x) {
                        rank = s.dranks[h];
                        break;
                    }
                    if (kx == kDictEmpty) break;
                    h = (h + 1) & s.dmask;
                }
                if (rank < 0) in_bounds = false;
                key += (uint64_t)(int64_t)rank * (uint64_t)s.gstride64;
            } else if (s.gmissing64 >= 0) {
                key += (uint64_t)s.gmissing64;
            } else {
                in_bounds = false;
            }
        } else if (s.flags & kSlotGroup) {
            if (pop) {
                uint64_t d = (uint64_t)x - (uint64_t)s.gmin;
                if (d >= (uint64_t)s.gvalues64) in_bounds = false;
                key += d * (uint64_t)s.gstride64;
            } else if (s.gmissing64 >= 0) {
                key += (uint64_t)s.gmissing64;
            } else {
                in_bounds = false;
            }
        }
    }
    if (!pass) return kRowFail;

    w = 1;
    if (P.weight_slot >= 0) {
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == P.weight_slot) w = r == 0 ? t.v[c].x : t.v[c].y;
    }
    // rows without a time value are dropped after they were counted as matched
    if (P.time_slot >= 0) {
        int64_t tv = 0;
        bool tpop = false;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == P.time_slot) {
                tv = r == 0 ? t.v[c].x : t.v[c].y;
                tpop = (t.pop[c] >> r) & 1u;
            }
        if (!tpop) return kRowDropped;
        int64_t tb = sdiv_trunc(tv, P.time_bucket, P.inv_time_bucket, P.tb_big_div) - P.tb_min;
        if ((uint64_t)tb >= (uint64_t)P.n_tb) in_bounds = false;
        key += (uint64_t)tb * (uint64_t)P.tb_stride64;
    }
    return in_bounds ? kRowOk : kRowOverflow;
}

// Count / Samples and the aggregations of one matched row (aggregate.go:202-261, hist_basic.go:101-151) into the cell
// table `sumtab` / `maxtab` laid out [field][ncell << rs]; cidx = (local cell << rs) + lane replica.  gcell is the
// cell's number in the global table: the bucket arrays of full-histogram aggregations live there ([gcell][hist_stride]).
// logkey: what the outlier log calls the group (the cell number; the composite key under hash group-by).
template <int NC, bool USE_LDS>
__device__ __forceinline__ void row_accumulate(CPlan &P, const Tile<NC> &t, int r, int64_t *sumtab, int64_t *maxtab, int64_t ncell,
                                               int rs, int64_t cidx, int64_t gcell, int64_t logkey, int64_t w, int64_t &overflow) {
    // w == 1 unless the query names a weight column: no 64-bit multiply per accumulated word then (wave-uniform branch)
    const bool weighted = P.weight_slot >= 0;
    auto times_w = [&](int64_t v) -> int64_t { return weighted ? (int64_t)((uint64_t)v * (uint64_t)w) : v; };
    acc_add<USE_LDS>(sumtab, cidx, w);  // Result.Count += weight (aggregate.go:203)
    if (P.f_samples >= 0) acc_add<USE_LDS>(sumtab, ((int64_t)P.f_samples * ncell << rs) + cidx, 1);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        CSlot &s = P.slot[c];
        if (!(s.flags & kSlotAgg)) continue;
        const int64_t x = r == 0 ? t.v[c].x : t.v[c].y;
        const bool pop = (t.pop[c] >> r) & 1u;
        CAgg &A = P.agg[s.agg_index];
        if (!pop) continue;
        if (A.f_pop >= 0) acc_add<USE_LDS>(sumtab, ((int64_t)A.f_pop * ncell << rs) + cidx, 1);
        if (x > A.max10 || x < A.info_min) continue;  // hist_basic.go:104
        acc_add<USE_LDS>(sumtab, ((int64_t)A.f_sum * ncell << rs) + cidx, times_w(x));
        if (A.f_cnt >= 0) acc_add<USE_LDS>(sumtab, ((int64_t)A.f_cnt * ncell << rs) + cidx, w);
        if (A.f_smp >= 0) acc_add<USE_LDS>(sumtab, ((int64_t)A.f_smp * ncell << rs) + cidx, 1);
        if (A.m_max >= 0) acc_max<USE_LDS>(maxtab, ((int64_t)A.m_max * ncell << rs) + cidx, x);
        if (A.m_nmin >= 0) acc_max<USE_LDS>(maxtab, ((int64_t)A.m_nmin * ncell << rs) + cidx, x == INT64_MIN ? INT64_MAX : -x);
        if (P.hist_mode && A.multi_n > 0) {
            // -loghist: the first sub-histogram whose range holds the value (hist_multi.go:84-89)
            int64_t *H = P.sum_out + P.hist_off + gcell * P.hist_stride + P.hist_agg_off[s.agg_index];
            for (int k = 0; k < A.multi_n; k++) {
                const MultiSub S = P.multi[A.multi_off + k];
                if (x < S.mn || x > S.mx) continue;
                if (x > S.max10) break;  // the sub-histogram's own gate (hist_basic.go:104)
                int64_t b = sdiv_trunc(x - S.mn, S.bs, S.inv_bs, S.big_div);
                if (b >= S.nv) {  // Outlier of the sub-histogram: remembered (one exact counter per value) and clipped
                    if (x - S.ext_first < S.n_ext) gadd(H + S.ext_off + (x - S.ext_first), 1);
                    else overflow += 1;
                    b = S.nv - 1;
                }
                gadd(H + S.off + b, w);
                break;
            }
        } else if (P.hist_mode) {
            // bucket_value := (value - h.Min) / BucketSize  (hist_basic.go:130)
            int64_t b = sdiv_trunc(x - A.hmin, A.bucket_size, A.inv_bucket, A.big_div);
            if (b >= A.n_values || b < 0) {
                // Outliers / Underliers (hist_basic.go:132-142): clipped into the edge bucket
                // AND remembered.  Their exact n, sum(o), sum(o^2) live in six extra cell
                // fields that the planner allocates only when the column bounds make an
                // outlier possible at all.
                if (A.f_out >= 0) {
                    const int64_t step = ncell << rs;
                    int64_t fi = ((int64_t)A.f_out * ncell << rs) + cidx;
                    unsigned __int128 sq = (unsigned __int128)((__int128)x * (__int128)x);
                    acc_add<USE_LDS>(sumtab, fi, 1);
                    acc_add<USE_LDS>(sumtab, fi + step, x);
                    acc_add<USE_LDS>(sumtab, fi + 2 * step, (int64_t)(uint64_t)(sq & 0xFFFFFFFFu));
                    acc_add<USE_LDS>(sumtab, fi + 3 * step, (int64_t)(uint64_t)((sq >> 32) & 0xFFFFFFFFu));
                    acc_add<USE_LDS>(sumtab, fi + 4 * step, (int64_t)(uint64_t)((sq >> 64) & 0xFFFFFFFFu));
                    acc_add<USE_LDS>(sumtab, fi + 5 * step, (int64_t)(uint64_t)(sq >> 96));
                    if (P.out_log) log_outlier(P.out_log, P.out_cap, logkey, s.agg_index, x);
                } else {
                    overflow += 1;  // declared bounds violated; reported by finalize
                }
                b = b < 0 ? 0 : A.n_values - 1;
            }
            if (A.hist_full) {
                gadd(P.sum_out + P.hist_off + gcell * P.hist_stride + P.hist_agg_off[s.agg_index] + b, w);
            } else {
                acc_add<USE_LDS>(sumtab, ((int64_t)A.f_sb * ncell << rs) + cidx, times_w(b));
                acc_add<USE_LDS>(sumtab, ((int64_t)A.f_sb2 * ncell << rs) + cidx, times_w((int64_t)((uint64_t)b * (uint64_t)b)));
            }
        }
    }
}

}  // namespace sybl
