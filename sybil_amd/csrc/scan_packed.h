// scan_packed.h -- the role-specialised scan over COMPACT storage (sybl_table_compact).
//
// Columns stored as 1 / 2 / 4-byte unsigned offsets from the column minimum make the scan of the
// common shape 2-4x lighter on HBM, which moves the bound from the memory system to instruction
// issue and LDS atomics.  k_scan_packed therefore works in the 32-bit OFFSET domain end to end:
//   * four consecutive rows per lane and tile: one 4 / 8 / 16-byte load per column (1 / 2 / 4-byte
//     values), fully coalesced across the wave, next tile in flight while this one is consumed;
//   * filters compare the raw offset against bounds the planner rebased (filter.go:171-195);
//   * the group digit is offset + (base - gmin), one 24-bit multiply-add per key column
//     (aggregate.go:125-143 as a direct-mapped cell index);
//   * the histogram bucket is (offset + (base - h.Min)) / BucketSize (hist_basic.go:130); the 64-bit
//     value is only rebuilt for the exact sum.
// Cell table, LDS layout, publication and fold are those of k_scan_fast (scan_fast.h), so the
// all-reduce and finalize do not care which kernel ran.
#pragma once
#include "scan_fast.h"

namespace sybl {

constexpr int kPackedRows = 4;                                // rows per lane and tile
constexpr int kPackedTileRows = kWgThreads * kPackedRows;

hipError_t launch_scan_packed(const FastPlan &P, int nf, int ng, int na, int mode, bool time, int n_wg, size_t lds_bytes,
                              hipStream_t st);
hipError_t launch_emit_packed(const EmitPlan &E, int nf, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_count_packed_nf0(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_packed_nf1(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_packed_nf2(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_packed_nf3(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_count_packed_nf4(const EmitPlan &E, int ng, int n_wg, hipStream_t st);
hipError_t launch_emit_packed_nf0(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_packed_nf1(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_packed_nf2(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_packed_nf3(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_emit_packed_nf4(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st);
hipError_t launch_scan_packed_nf0(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_packed_nf1(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_packed_nf2(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_packed_nf3(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st);
hipError_t launch_scan_packed_nf4(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st);

#ifdef __HIPCC__

typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

template <int N>
struct PackedTile {
    uint32_t u[N > 0 ? N : 1][kPackedRows];  // stored offsets (value - column base), zero-extended
    uint32_t pop[N > 0 ? N : 1];             // NUL kernels: validity bits of the four rows
};

template <int N>
struct PackedRaw {
    pu32x4 v[N > 0 ? N : 1];     // the loaded bytes of four rows: 1 / 2 / 4 dwords are meaningful
    uint32_t pw[N > 0 ? N : 1];  // NUL kernels: the validity word holding the four rows' bits
};

// Issues the load of four consecutive rows of one column.  The instruction is the same 16-byte
// buffer load for every stored width, so issuing needs no branch and nothing here waits for the
// data (the width only matters when the registers are decoded).  A narrower column would over-read
// into its neighbours' rows; the buffer descriptor is therefore set to exactly the wave's 256 rows
// (`col` = wave-uniform address of the wave's first row, `voff` = the lane's byte offset inside
// them), and the hardware range check drops the out-of-range dwords without touching memory --
// HBM traffic stays at the stored bytes.
constexpr uint32_t kBufferRsrcWord3 = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ void packed_issue(const uint8_t *col, int wshift, uint32_t voff, pu32x4 &raw) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)col, 0, (int)((64u * kPackedRows) << wshift), (int)kBufferRsrcWord3);
    raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 2);  // aux 2 = nt: streamed once
}

__device__ __forceinline__ void packed_decode(int width, const pu32x4 &raw, uint32_t (&u)[kPackedRows]) {
    if (width == 4) {
        u[0] = raw.x;
        u[1] = raw.y;
        u[2] = raw.z;
        u[3] = raw.w;
    } else if (width == 2) {
        u[0] = raw.x & 0xFFFFu;
        u[1] = raw.x >> 16;
        u[2] = raw.y & 0xFFFFu;
        u[3] = raw.y >> 16;
    } else {
        u[0] = raw.x & 0xFFu;
        u[1] = (raw.x >> 8) & 0xFFu;
        u[2] = (raw.x >> 16) & 0xFFu;
        u[3] = raw.x >> 24;
    }
}

// chunk-relative column addresses (wave-uniform)
template <int NF, int NG, int NA>
struct PackedBases {
    const uint8_t *f[NF > 0 ? NF : 1], *g[NG > 0 ? NG : 1], *a[NA > 0 ? NA : 1], *t;
    int64_t first;  // physical row of the chunk's first row (validity bitmaps are indexed by physical row)
};

// 1-byte columns: four rows are ONE dword.  Loaded with the 16-byte instruction they cost the address
// pipeline as much as a 4-byte column for a quarter of the data (loads-only: 4.0 TB/s against 6.1), and
// the instruction cannot depend on a run-time width (see packed_issue) -- so "every group column is one
// byte wide" (low-cardinality keys: the common case) is a compile-time property of the kernel (G1) and
// those columns are read with 4-byte loads.
__device__ __forceinline__ void packed_issue1(const uint8_t *col, uint32_t voff, pu32x4 &raw) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)col, 0, (int)(64u * kPackedRows), (int)kBufferRsrcWord3);
    raw.x = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, 0, 2);
}

// r: the lane's first row inside the chunk; the wave's first row is that of its first lane
template <int NF, int NG, int NA, bool TIME, bool G1, bool NUL, bool FRESH = false>
__device__ __forceinline__ void packed_issue_all(const FastPlan &P0, const PackedBases<NF, NG, NA> &B, uint32_t r, PackedRaw<NF> &f,
                                                 PackedRaw<NG> &g, PackedRaw<NA> &a, PackedRaw<1> &t) {
    const FastPlan &P = plan_fresh<FRESH>(P0);  // (FRESH: the plan words a stage uses are reloaded there, scan_fast.h)
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
    const uint32_t lane_row = r - r0;
    if (NUL) {
        // validity words of the lane's four rows (nullptr: every row populated)
        const int64_t w = (B.first + r) >> 5;
        if (TIME) t.pw[0] = P.tvalid ? P.tvalid[w] : 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < NF; c++) f.pw[c] = P.fvalid[c] ? P.fvalid[c][w] : 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < NG; c++) g.pw[c] = P.gvalid[c] ? P.gvalid[c][w] : 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < NA; c++) a.pw[c] = P.avalid[c] ? P.avalid[c][w] : 0xFFFFFFFFu;
    }
    auto issue = [&](const uint8_t *col, int width, pu32x4 &raw) {
        const int ws = width >> 1;  // width 1, 2, 4 -> shift 0, 1, 2
        packed_issue(col + ((size_t)r0 << ws), ws, lane_row << ws, raw);
    };
    if (TIME) issue(B.t, P.twid, t.v[0]);
#pragma unroll
    for (int c = 0; c < NF; c++) issue(B.f[c], P.fwid[c], f.v[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) {
        if (G1) {
            packed_issue1(B.g[c] + (size_t)r0, lane_row, g.v[c]);
        } else {
            issue(B.g[c], P.gwid[c], g.v[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NA; c++) issue(B.a[c], P.awid[c], a.v[c]);
}

// The same loads for the staged kernels (k_count_packed / k_emit_packed), which keep several tiles in flight:
// every lane ALWAYS issues them -- a tile (or the lanes of a wave) beyond the end of the chunk gets a
// descriptor that ends at row n, so the range check returns zeros without touching memory.  Loads issued
// under a condition make the compiler assume the younger ones may not exist, and it then waits for all but
// the newest whenever it needs the oldest (s_waitcnt vmcnt(1) instead of vmcnt(6)).
template <int NF, int NG, int NA>
__device__ __forceinline__ void packed_issue_always(const FastPlan &P, const PackedBases<NF, NG, NA> &B, uint32_t r, uint32_t n,
                                                    PackedRaw<NF> &f, PackedRaw<NG> &g, PackedRaw<NA> &a) {
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
    const uint32_t lane_row = r - r0;
    // wave-uniform; whole lanes (4 rows: at least one dword at any width -- the range check drops a dword that is
    // only partly inside), which never reaches past the padding every block ends with
    const uint32_t rows = r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
    auto issue = [&](const uint8_t *col, int width, pu32x4 &raw) {
        const int ws = width >> 1;  // width 1, 2, 4 -> shift 0, 1, 2
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(col + ((size_t)(rows ? r0 : 0u) << ws)), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
        raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);  // aux 2 = nt: streamed once
    };
#pragma unroll
    for (int c = 0; c < NF; c++) issue(B.f[c], P.fwid[c], f.v[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) issue(B.g[c], P.gwid[c], g.v[c]);
#pragma unroll
    for (int c = 0; c < NA; c++) issue(B.a[c], P.awid[c], a.v[c]);
}

// packed_issue_all for a ring of tiles in flight (k_scan_packed without NUL): like packed_issue_always every lane ALWAYS
// issues -- the descriptor ends at row n, so a tile (or the lanes of a wave) past the end of the chunk loads nothing --
// and the load count between a tile's loads and their use is the same on every path: the compiler's vmcnt is exact.
template <int NF, int NG, int NA, bool TIME, bool G1>
__device__ __forceinline__ void packed_issue_ring(const FastPlan &P, const PackedBases<NF, NG, NA> &B, uint32_t r, uint32_t n,
                                                  PackedRaw<NF> &f, PackedRaw<NG> &g, PackedRaw<NA> &a, PackedRaw<1> &t) {
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
    const uint32_t lane_row = r - r0;
    const uint32_t rows = r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
    const uint32_t base_row = rows ? r0 : 0u;
    auto issue = [&](const uint8_t *col, int width, pu32x4 &raw) {
        const int ws = width >> 1;  // width 1, 2, 4 -> shift 0, 1, 2
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(col + ((size_t)base_row << ws)), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
        raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);  // aux 2 = nt: streamed once
    };
    if (TIME) issue(B.t, P.twid, t.v[0]);
#pragma unroll
    for (int c = 0; c < NF; c++) issue(B.f[c], P.fwid[c], f.v[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) {
        if (G1) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(B.g[c] + (size_t)base_row), 0, (int)rows, (int)kBufferRsrcWord3);
            g.v[c].x = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_row, 0, 2);
        } else {
            issue(B.g[c], P.gwid[c], g.v[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NA; c++) issue(B.a[c], P.awid[c], a.v[c]);
}

// Tiles of loads a lane keeps in flight in k_scan_packed: ONE (the next tile is requested before the current one is
// consumed).  A ring of two or three tiles (RING > 1: the k_emit structure, every lane always issuing) was measured on one
// box against it (tools/bench_ring.py, round 3): config 2 (three columns) 0.177 ms with one tile, 0.180 with two, 0.182
// with three; the seven-column headline 2.81 / 2.92 / 2.92 ms; the time rollup 1.74 ms whichever -- the row bodies cover the
// latency of one tile, and the ring's clamped descriptors cost scalar work per load.  (Round 2 had found the same for
// queries of <= 3 columns.)  SYBL_PACKED_RING keeps the variants reachable for the BASELINE shapes.
constexpr int packed_depth(int) { return 1; }

template <int NF, int NG, int NA, bool TIME, bool G1, bool NUL, bool FRESH = false>
__device__ __forceinline__ void packed_decode_all(const FastPlan &P0, const PackedRaw<NF> &rf, const PackedRaw<NG> &rg,
                                                  const PackedRaw<NA> &ra, const PackedRaw<1> &rt, PackedTile<NF> &f,
                                                  PackedTile<NG> &g, PackedTile<NA> &a, PackedTile<1> &t, uint32_t bit0) {
    const FastPlan &P = plan_fresh<FRESH>(P0);
    if (NUL) {  // bit0 = (physical row of the lane's first row) & 31, a multiple of 4
        if (TIME) t.pop[0] = (rt.pw[0] >> bit0) & 0xFu;
#pragma unroll
        for (int c = 0; c < NF; c++) f.pop[c] = (rf.pw[c] >> bit0) & 0xFu;
#pragma unroll
        for (int c = 0; c < NG; c++) g.pop[c] = (rg.pw[c] >> bit0) & 0xFu;
#pragma unroll
        for (int c = 0; c < NA; c++) a.pop[c] = (ra.pw[c] >> bit0) & 0xFu;
    }
    if (TIME) packed_decode(P.twid, rt.v[0], t.u[0]);
#pragma unroll
    for (int c = 0; c < NF; c++) packed_decode(P.fwid[c], rf.v[c], f.u[c]);
#pragma unroll
    for (int c = 0; c < NG; c++) packed_decode(G1 ? 1 : P.gwid[c], rg.v[c], g.u[c]);
#pragma unroll
    for (int c = 0; c < NA; c++) packed_decode(P.awid[c], ra.v[c], a.u[c]);
}

// floor(n / d) for 32-bit n, d >= 1.  inv_lo is 1/d scaled down by (1 - 2^-40), so the product never
// exceeds the true quotient and is at most 1 short of it: one one-sided correction step is exact.
__device__ __forceinline__ uint32_t packed_udiv(uint32_t n, uint32_t d, double inv_lo) {
    uint32_t q = (uint32_t)((double)n * inv_lo);
    if (n - q * d >= d) q += 1;
    return q;
}

// (OUT: the plain body with the outlier test -- a value beyond the last bucket is clipped and remembered, hist_basic.go:132-135:
// what a fully populated tile of a NUL kernel runs when outliers are the only thing its plan asks of the NUL body)
// (MAX32: avg mode's maxima as 32-bit OFFSETS -- see the kFastAvgMax branch; the caller has checked !ext_general once per tile)
template <int NF, int NG, int NA, int MODE, bool TIME, bool NUL, bool FRESH = false, bool OUT = false, bool MAX32 = false>
__device__ __forceinline__ void packed_row(const FastPlan &P0, const PackedTile<NF> &f, const PackedTile<NG> &g,
                                           const PackedTile<NA> &a, const PackedTile<1> &t, const int r, bool pass, int64_t *lds,
                                           const FastLds &L, uint32_t &matched, uint32_t &overflow, const uint32_t xpop = 0xFu) {
    if (NUL) pass = pass & ((xpop >> r) & 1u);  // the filter pre-pass's verdict (FastPlan::xvalid; all ones without one)
    // no short-circuit anywhere: one predicate, one exec-masked region per row
#pragma unroll
    for (int c = 0; c < NF; c++) {
        const FastPlan &P = plan_fresh<FRESH>(P0);
        const uint32_t u = f.u[c][r];
        bool ok = (u >= P.plo[c]) & (u <= P.phi[c]);  // filter.go:171-195, folded to a range of offsets
        if (NUL) {
            if (P.fmask[c]) {
                // StrFilter eq / neq / re / nre, evaluated per dictionary id on the host (filter.go:199-250)
                const uint32_t id = u + (uint32_t)P.fbase[c];
                ok = id < (uint32_t)P.fmask_bits[c];
                if (ok) ok = (P.fmask[c][id >> 5] >> (id & 31)) & 1u;
            }
            for (int k = 0; k < P.npneq[c]; k++) ok = ok & (u != P.pneq[c][k]);  // int neq, rebased
            ok = ok & ((f.pop[c] >> r) & 1u);  // an unpopulated value fails every filter
        }
        pass = pass & ok;
    }
    uint32_t cell = 0;
    bool inb = true;
#pragma unroll
    for (int c = 0; c < NG; c++) {
        const FastPlan &P = plan_fresh<FRESH>(P0);
        const uint32_t d = g.u[c][r] + P.gdoff[c];  // value - gmin
        if (NUL) {
            // MISSING_VALUE key (aggregate.go:138): its own digit, or the digit of the value -1
            const bool p = (g.pop[c] >> r) & 1u;
            inb = inb & (p ? d < (uint32_t)P.gvalues[c] : P.gmissing[c] >= 0);
            cell += p ? __umul24(d, (uint32_t)P.gstride[c]) : (uint32_t)P.gmissing[c];
        } else {
            inb = inb & (d < P.gcard[c]);
            cell += __umul24(d, (uint32_t)P.gstride[c]);  // aggregate.go:125-143 as a direct-mapped index
        }
    }
    bool live = pass;
    const FastPlan &P = plan_fresh<FRESH>(P0);
    if (TIME) {
        // int(val) / TimeBucket (aggregate.go:174) for val >= 0, relative to the first bucket
        const uint32_t tb = packed_udiv(t.u[0][r] + P.tdoff, (uint32_t)P.time_bucket, P.pinv_time);
        if (NUL) {
            // no time value: the row was matched, then dropped (aggregate.go:147-153)
            const bool tp = (t.pop[0] >> r) & 1u;
            live = live & tp;
            inb = inb & (tb < (uint32_t)P.n_tb || !tp);
        } else {
            inb = inb & (tb < (uint32_t)P.n_tb);
        }
        cell += __umul24(tb, (uint32_t)P.tb_stride);
    }
    const uint32_t ncell = L.tab_cells;
    const uint32_t lcell = cell - L.cell_base;  // position inside this workgroup's LDS table
    inb = inb & (lcell < ncell);
    matched += pass ? 1u : 0u;                  // aggregate.go:117
    overflow += (live & !inb) ? 1u : 0u;
    if (!(live & inb)) return;
    // byte address of the cell's Count word; every other field of the cell is a wave-uniform byte
    // offset away (one VALU add per atomic)
    const uint32_t rs = (uint32_t)P.rep_shift;
    char *const cell_p = (char *)lds + (((lcell << rs) + L.rep) << 3);
    const uint32_t fstep = (ncell << rs) << 3;  // bytes between consecutive fields
    auto add = [&](uint32_t field, int64_t v) {
        __hip_atomic_fetch_add((int64_t *)(cell_p + field * fstep), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // avg modes with every value populated: Result.Count rides in the high bits of aggregation 0's sum word (FastPlan::cshift,
    // wave-uniform; 0 = it has a word of its own) -- one LDS atomic less per row
    const uint32_t cshift = ((MODE == kFastAvg || MODE == kFastAvgMax) && !NUL && !OUT && NA > 0) ? (uint32_t)P.cshift : 0u;
    if (!cshift) add(0, 1);  // Result.Count++ (aggregate.go:203)
#pragma unroll
    for (int c = 0; c < NA; c++) {
        const FastPlan &P = plan_fresh<FRESH>(P0);
        const uint32_t u = a.u[c][r];
        if (NUL) {
            if (!((a.pop[c] >> r) & 1u)) continue;  // no value: no hist for this row
            if (P.f_pop[c] >= 0) add((uint32_t)P.f_pop[c], 1);
            if (P.f_cnt[c] >= 0) {
                if (u < P.alo[c] || u > P.ahi[c]) continue;  // hist_basic.go:104, rebased
                add((uint32_t)P.f_cnt[c], 1);                // h.Count++
            }
        }
        const int64_t x = (int64_t)((uint64_t)P.abase[c] + u);
        if (c == 0 && cshift) add((uint32_t)P.f_sum[0], (int64_t)(((uint64_t)1 << cshift) + u));  // Count++ and sum of OFFSETS
        else add((uint32_t)P.f_sum[c], x);
        if (MODE == kFastAvgMax) {
            if (MAX32) {
                // BasicHist.Max (hist_basic.go:118-120) in the offset domain: max(v) = base + max(offset), so the lane's replica
                // of the MAX word holds the largest OFFSET in its low half -- one ds_max_u32 that returns nothing, where the
                // 64-bit form read the word, waited for it, compared and branched around a ds_max_i64 (two of those round trips
                // per row of config 2).  (A gate read of an unreplicated uint32 table in front of a rare atomic measured SLOWER
                // than the blind atomic: tools/micro/ldsrate.hip, profiles/r06_ldsrate.txt.)  A cell's maximum exists iff its
                // Count is not zero: fast_finish rebuilds the value.
                __hip_atomic_fetch_max((uint32_t *)(cell_p + ((uint32_t)P.n_sum_fields + (uint32_t)P.m_max[c]) * fstep), u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (!P.ext_general) {
                int64_t *m = (int64_t *)(cell_p + ((uint32_t)P.n_sum_fields + (uint32_t)P.m_max[c]) * fstep);
                if (x > *m) __hip_atomic_fetch_max(m, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                // (FastPlan::ext_general: a minimum is tracked as well -- negative values in avg mode --, or not every maximum)
                if (P.m_max[c] >= 0) {
                    int64_t *m = (int64_t *)(cell_p + ((uint32_t)P.n_sum_fields + (uint32_t)P.m_max[c]) * fstep);
                    if (x > *m) __hip_atomic_fetch_max(m, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (P.m_nmin[c] >= 0) {
                    int64_t *m = (int64_t *)(cell_p + ((uint32_t)P.n_sum_fields + (uint32_t)P.m_nmin[c]) * fstep);
                    const int64_t nx = x == INT64_MIN ? INT64_MAX : -x;
                    if (nx > *m) __hip_atomic_fetch_max(m, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (MODE == kFastMoments || MODE == kFastHist) {
            // bucket_value := (value - h.Min) / BucketSize, hist_basic.go:130; the planner guarantees
            // 0 <= value - h.Min < 2^32 and that no value reaches len(Values)
            uint32_t b = packed_udiv(u + P.adoff[c], P.bucket_size[c], P.pinv_bucket[c]);
            if ((NUL || OUT) && b >= (uint32_t)P.n_values[c]) {
                // Outlier (hist_basic.go:132-135; BucketSize = size / 1000 truncates, so the top of many a column's range
                // lies beyond the last bucket): clipped into the last bucket AND remembered as exact n, sum(o), sum(o^2)
                // in four 32-bit limbs (+ the value itself in the log when bucket arrays are kept); a cold path
                if (P.f_out[c] >= 0) {
                    const unsigned __int128 sq = (unsigned __int128)((__int128)x * (__int128)x);
                    const uint32_t fo = (uint32_t)P.f_out[c];
                    add(fo, 1);
                    add(fo + 1, x);
                    add(fo + 2, (int64_t)(uint64_t)(sq & 0xFFFFFFFFu));
                    add(fo + 3, (int64_t)(uint64_t)((sq >> 32) & 0xFFFFFFFFu));
                    add(fo + 4, (int64_t)(uint64_t)((sq >> 64) & 0xFFFFFFFFu));
                    add(fo + 5, (int64_t)(uint64_t)(sq >> 96));
                    if (P.out_log) log_outlier(P.out_log, P.out_cap, (int64_t)cell, c, x);
                } else {
                    overflow += 1;
                }
                b = (uint32_t)P.n_values[c] - 1;
            }
            if (MODE == kFastMoments) {
                add((uint32_t)P.f_sb[c], (int64_t)(uint64_t)b);
                add((uint32_t)P.f_sb2[c], (int64_t)(uint64_t)(uint32_t)__umul24(b, b));
            } else if (P.hist_lds) {
                __hip_atomic_fetch_add(L.hist32 + lcell * (uint32_t)P.hist_stride + (uint32_t)P.hist_agg_off[c] + b, 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_fetch_add(P.sum_out + P.hist_off + (int64_t)cell * P.hist_stride + P.hist_agg_off[c] + b,
                                       (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

constexpr int64_t kPackedChunkRows = (int64_t)1 << 28;  // rows addressed with one 32-bit byte offset (x4 bytes)

// SYBL_PACKED_LATE=0 at build time: k_scan_packed without late materialisation (same-box A/B builds)
#ifndef SYBL_PACKED_LATE
#define SYBL_PACKED_LATE 1
#endif
constexpr bool kPackedLate = SYBL_PACKED_LATE != 0;
#ifndef SYBL_PACKED_LATE_BRANCH
#define SYBL_PACKED_LATE_BRANCH 1
#endif
constexpr bool kPackedLateBranch = SYBL_PACKED_LATE_BRANCH != 0;
// (A late path inside the NUL variants was built and measured in round 6 and taken out again: in one body with the plain loop it
// cost the variant 268 B of scratch and its plain path 3.3 -> 4.05 ms per 1e9 rows of config 3, for 2.95 -> 2.79 ms at 0.1 %
// selectivity -- the NUL row body's own floor; profiles/r06_late_path_ab.txt.  As a kernel of its own it would be ~390 more
// instantiations of this template.)

#ifndef SYBL_PACKED_WAVES_PER_EU
#define SYBL_PACKED_WAVES_PER_EU 4
#endif
template <int NF, int NG, int NA, int MODE, bool TIME, bool G1, bool NUL, int RING = 0>
__global__ __launch_bounds__(kWgThreads, SYBL_PACKED_WAVES_PER_EU) void k_scan_packed(const FastPlan P) {
    extern __shared__ int64_t lds[];
    const uint32_t tid = threadIdx.x;
    // avg mode, every value populated (the NUL kernels mix row bodies: they keep 64-bit maxima): maxima as 32-bit offsets
    constexpr bool kMax32 = MODE == kFastAvgMax && !NUL;
    const FastLds L = fast_begin<MODE>(P, lds, kMax32 && !P.ext_general);

    uint32_t matched = 0, overflow = 0;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            PackedBases<NF, NG, NA> B;
#pragma unroll
            for (int c = 0; c < NF; c++) B.f[c] = (const uint8_t *)P.fcol[c] + first * P.fwid[c];
#pragma unroll
            for (int c = 0; c < NG; c++) B.g[c] = (const uint8_t *)P.gcol[c] + first * P.gwid[c];
#pragma unroll
            for (int c = 0; c < NA; c++) B.a[c] = (const uint8_t *)P.acol[c] + first * P.awid[c];
            B.t = TIME ? (const uint8_t *)P.tcol + first * P.twid : nullptr;
            B.first = first;

            PackedTile<NF> f;
            PackedTile<NG> g;
            PackedTile<NA> a;
            PackedTile<1> t;
            constexpr bool FRESH = RING == 4;  // (A/B: plan words reloaded per stage instead of spilled scalar registers)
            constexpr int D = NUL ? 1 : (RING > 0 && RING < 4 ? RING : packed_depth(NF + NG + NA + (TIME ? 1 : 0)));
            if (D > 1) {
                // a ring of D tiles of loads in flight, consumed oldest first; no exit inside a round (a tile past the end
                // loads nothing and its rows fail `k < left`)
                PackedRaw<NF> rf[D];
                PackedRaw<NG> rg[D];
                PackedRaw<NA> ra[D];
                PackedRaw<1> rt[D];
                const uint32_t n_tiles = (n + kPackedTileRows - 1) / kPackedTileRows;
                const uint32_t r_first = tid * kPackedRows;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    packed_issue_ring<NF, NG, NA, TIME, G1>(P, B, r_first + (uint32_t)d * kPackedTileRows, n, rf[d], rg[d], ra[d], rt[d]);
                    __builtin_amdgcn_sched_barrier(0);  // oldest tile first: the ring is consumed in this order
                }
                for (uint32_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const uint32_t r = r_first + (it0 + d) * kPackedTileRows;  // (< 2^28 + 2^14: no wrap)
                        packed_decode_all<NF, NG, NA, TIME, G1, false>(P, rf[d], rg[d], ra[d], rt[d], f, g, a, t, 0u);
                        packed_issue_ring<NF, NG, NA, TIME, G1>(P, B, r + (uint32_t)D * kPackedTileRows, n, rf[d], rg[d], ra[d], rt[d]);
                        const uint32_t left = r < n ? n - r : 0u;
                        if (kMax32 && !P.ext_general) {  // (wave-uniform, once per tile)
#pragma unroll
                            for (int k = 0; k < kPackedRows; k++)
                                packed_row<NF, NG, NA, MODE, TIME, false, false, false, true>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow);
                        } else {
#pragma unroll
                            for (int k = 0; k < kPackedRows; k++)
                                packed_row<NF, NG, NA, MODE, TIME, false>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow);
                        }
                    }
                }
                continue;
            }
            PackedRaw<NF> rf;
            PackedRaw<NG> rg;
            PackedRaw<NA> ra;
            PackedRaw<1> rt;
            // (NUL) the filter pre-pass's bitmap (FastPlan::xvalid): the lane's four bits for the tile being consumed, and
            // the word of the tile after it -- fetched a tile early, so that a wave none of whose 256 rows passed the
            // pre-pass issues no load at all for that tile
            uint32_t xpop = 0xFu, xw_n = 0xFFFFFFFFu;
            const bool xv = NUL && P.xvalid != nullptr;
            // The NUL row body costs twice the plain one (config 3 forced through it: 5.8 against 2.85 ms per 1e9 rows) --
            // validity bits, the MISSING digit, per-aggregation counts, the outlier test, each a branch per row and column.
            // Most tiles of most tables need none of it: a str group column, or a bitmap that is all ones where it is read.
            // `light` (wave-uniform, once): the plan itself asks for nothing the plain body lacks; a tile all of whose
            // values are populated (and pass the pre-pass) then runs the plain body.
            bool light = NUL, outliers = false;
            if (NUL) {
#pragma unroll
                for (int c = 0; c < NF; c++) light = light && P.fmask[c] == nullptr && P.npneq[c] == 0;
#pragma unroll
                for (int c = 0; c < NG; c++) light = light && (uint32_t)P.gvalues[c] == P.gcard[c];
#pragma unroll
                for (int c = 0; c < NA; c++) {
                    light = light && P.f_pop[c] < 0 && P.f_cnt[c] < 0;
                    outliers = outliers || P.f_out[c] >= 0;  // (BucketSize = size / 1000 truncates: an ordinary column's top values)
                }
            }
            if (kPackedLate && !NUL && NF > 0 && NG + NA + (TIME ? 1 : 0) > 0) {
                // Late materialisation (the reference's row loop leaves a row at its first failing filter, aggregate.go:105-116):
                // the filter columns run one tile ahead of the key / aggregation / time columns, the tile's predicate is
                // evaluated as soon as they arrive, and a WAVE none of whose 256 rows passes does not read the other columns
                // of that tile.  The load instructions are still issued -- a fixed count per iteration keeps the compiler's
                // vmcnt exact (packed_issue_always) -- but through a descriptor of zero records: the range check answers
                // without touching memory, so the skipped lines are never fetched (a wave's tile of 256 rows is whole 128-byte
                // lines at every stored width: 256, 512 or 1024 bytes).
                // Iteration t: decode keys / values of tile t and the filters of t + 1, issue keys / values of t + 1 (or
                // nothing) and the filters of t + 2, then the rows of t with the predicate bits kept from last time.
                const uint32_t r_first = tid * kPackedRows;
                const uint32_t n_tiles = (n + kPackedTileRows - 1) / kPackedTileRows;
                PackedTile<0> f0;
                auto issue_filters = [&](uint32_t r) {
                    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r), lane_row = r - r0;
                    const uint32_t rows = r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
                    const uint32_t base_row = rows ? r0 : 0u;
#pragma unroll
                    for (int c = 0; c < NF; c++) {
                        const int ws = P.fwid[c] >> 1;
                        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(B.f[c] + ((size_t)base_row << ws)), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
                        rf.v[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);
                    }
                };
                auto issue_rest = [&](uint32_t r, bool wanted) {
                    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r), lane_row = r - r0;
                    uint32_t rows = r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
                    rows = wanted ? rows : 0u;  // (wave-uniform: nothing of this tile is wanted -> a descriptor of zero records)
                    const uint32_t base_row = rows ? r0 : 0u;
                    auto issue = [&](const uint8_t *col, int width, pu32x4 &raw) {
                        const int ws = width >> 1;
                        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(col + ((size_t)base_row << ws)), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
                        raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);
                    };
                    if (TIME) issue(B.t, P.twid, rt.v[0]);
#pragma unroll
                    for (int c = 0; c < NG; c++) {
                        if (G1) {
                            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(B.g[c] + (size_t)base_row), 0, (int)rows, (int)kBufferRsrcWord3);
                            rg.v[c].x = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_row, 0, 2);
                        } else {
                            issue(B.g[c], P.gwid[c], rg.v[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NA; c++) issue(B.a[c], P.awid[c], ra.v[c]);
                };
                auto filter_bits = [&](uint32_t r) -> uint32_t {
                    // (decodes rf: the first use of the filter loads issued an iteration ago)
#pragma unroll
                    for (int c = 0; c < NF; c++) packed_decode(P.fwid[c], rf.v[c], f.u[c]);
                    const uint32_t left = r < n ? n - r : 0u;
                    uint32_t bits = 0;
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        bool pass = (uint32_t)k < left;
#pragma unroll
                        for (int c = 0; c < NF; c++) pass = pass & (f.u[c][k] >= P.plo[c]) & (f.u[c][k] <= P.phi[c]);
                        bits |= pass ? 1u << k : 0u;
                    }
                    return bits;
                };
                issue_filters(r_first);
                uint32_t bits = filter_bits(r_first);
                issue_rest(r_first, __builtin_amdgcn_ballot_w64(bits != 0) != 0);
                issue_filters(r_first + kPackedTileRows);
                for (uint32_t it = 0; it < n_tiles; it++) {
                    const uint32_t r = r_first + it * kPackedTileRows;  // (< 2^28 + 2^13: no wrap)
                    // keys / values / time of tile `it` (their loads were issued an iteration ago; zeros if nothing was wanted)
                    if (TIME) packed_decode(P.twid, rt.v[0], t.u[0]);
#pragma unroll
                    for (int c = 0; c < NG; c++) packed_decode(G1 ? 1 : P.gwid[c], rg.v[c], g.u[c]);
#pragma unroll
                    for (int c = 0; c < NA; c++) packed_decode(P.awid[c], ra.v[c], a.u[c]);
                    const uint32_t next_bits = filter_bits(r + kPackedTileRows);
                    if (kPackedLateBranch) {
                        if (__builtin_amdgcn_ballot_w64(next_bits != 0) != 0) issue_rest(r + kPackedTileRows, true);
                    } else {
                        issue_rest(r + kPackedTileRows, __builtin_amdgcn_ballot_w64(next_bits != 0) != 0);
                    }
                    issue_filters(r + 2u * kPackedTileRows);
                    if (kMax32 && !P.ext_general) {  // (wave-uniform, once per tile)
#pragma unroll
                        for (int k = 0; k < kPackedRows; k++)
                            packed_row<0, NG, NA, MODE, TIME, false, false, false, true>(P, f0, g, a, t, k, (bits >> k) & 1u, lds, L, matched, overflow);
                    } else {
#pragma unroll
                        for (int k = 0; k < kPackedRows; k++)
                            packed_row<0, NG, NA, MODE, TIME, false>(P, f0, g, a, t, k, (bits >> k) & 1u, lds, L, matched, overflow);
                    }
                    bits = next_bits;
                }
                continue;
            }
            uint32_t r = tid * kPackedRows;
            if (r < n) {
                packed_issue_all<NF, NG, NA, TIME, G1, NUL, FRESH>(P, B, r, rf, rg, ra, rt);
                if (xv) {
                    xpop = (P.xvalid[(first + r) >> 5] >> ((uint32_t)(first + r) & 31u)) & 0xFu;
                    if (r + kPackedTileRows < n) xw_n = P.xvalid[(first + r + kPackedTileRows) >> 5];
                }
                packed_decode_all<NF, NG, NA, TIME, G1, NUL, FRESH>(P, rf, rg, ra, rt, f, g, a, t, (uint32_t)(first + r) & 31u);
            }
            for (; r < n; r += kPackedTileRows) {
                // the next tile's loads are in flight while this one is consumed; they are decoded
                // (the first use of the loaded registers) only after the rows below
                const uint32_t rn = r + kPackedTileRows;
                const uint32_t xpop_n = xv ? (xw_n >> ((uint32_t)(first + rn) & 31u)) & 0xFu : 0xFu;
                const bool more = rn < n && (!xv || __builtin_amdgcn_ballot_w64(xpop_n != 0) != 0);  // (wave-uniform)
                if (more) packed_issue_all<NF, NG, NA, TIME, G1, NUL, FRESH>(P, B, rn, rf, rg, ra, rt);
                if (xv && rn + kPackedTileRows < n) xw_n = P.xvalid[(first + rn + kPackedTileRows) >> 5];
                const uint32_t left = n - r;
                bool plain_tile = false;
                if (NUL && light) {
                    uint32_t allpop = xpop;
                    if (TIME) allpop &= t.pop[0];
#pragma unroll
                    for (int c = 0; c < NF; c++) allpop &= f.pop[c];
#pragma unroll
                    for (int c = 0; c < NG; c++) allpop &= g.pop[c];
#pragma unroll
                    for (int c = 0; c < NA; c++) allpop &= a.pop[c];
                    plain_tile = __builtin_amdgcn_ballot_w64(allpop != 0xFu) == 0;  // (wave-uniform)
                }
                if (plain_tile && !outliers) {
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++)
                        packed_row<NF, NG, NA, MODE, TIME, false, FRESH>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow);
                } else if (plain_tile) {
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++)
                        packed_row<NF, NG, NA, MODE, TIME, false, FRESH, true>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow);
                } else if (kMax32 && !P.ext_general) {  // (wave-uniform, once per tile; kMax32: not a NUL kernel)
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++)
                        packed_row<NF, NG, NA, MODE, TIME, false, FRESH, false, true>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow);
                } else {
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++)
                        packed_row<NF, NG, NA, MODE, TIME, NUL, FRESH>(P, f, g, a, t, k, (uint32_t)k < left, lds, L, matched, overflow, xpop);
                }
                if (more) packed_decode_all<NF, NG, NA, TIME, G1, NUL, FRESH>(P, rf, rg, ra, rt, f, g, a, t, (uint32_t)(first + rn) & 31u);
                xpop = xpop_n;
            }
        }
    }
    fast_finish(P, lds, L, matched, overflow, kMax32 && !P.ext_general);
}

// k_emit over compact storage (strategy 5, see k_emit in scan_fast.h): the same records, staged and
// flushed the same way; rows are loaded and filtered in the offset domain like k_scan_packed.
// One column's four rows for this lane with a descriptor that ends at row n -- or holds nothing at all (`wanted` false,
// wave-uniform): the range check then returns zeros without touching memory.  What the late paths below issue.
__device__ __forceinline__ void packed_issue_clamped(const uint8_t *col, int width, uint32_t r, uint32_t n, bool wanted, pu32x4 &raw) {
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
    const uint32_t lane_row = r - r0;
    const uint32_t rows = wanted && r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
    const int ws = width >> 1;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(col + ((size_t)(rows ? r0 : 0u) << ws)), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
    raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);
}

// LATE (round 6, a kernel of its own -- two loops in one body wreck the other's register allocation): the filter
// columns run a tile ahead; the key and value columns of a tile are only requested when some row of the wave passed.  Chosen
// by the planner's selectivity estimate (FastPlan::late): at 0.1 % most waves never touch them.
template <int NF, int NG, int NA, bool LATE = false>
__global__ __launch_bounds__(kWgThreads, 4) void k_emit_packed(const EmitPlan E) {
    extern __shared__ uint32_t elds[];
    const FastPlan &P = E.fp;
    const uint32_t tid = threadIdx.x;
    const EmitLds S = emit_begin(E, elds);

    uint32_t matched = 0, overflow = 0;
    constexpr uint32_t kTile = kPackedTileRows;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            PackedBases<NF, NG, NA> B;
#pragma unroll
            for (int c = 0; c < NF; c++) B.f[c] = (const uint8_t *)P.fcol[c] + first * P.fwid[c];
#pragma unroll
            for (int c = 0; c < NG; c++) B.g[c] = (const uint8_t *)P.gcol[c] + first * P.gwid[c];
#pragma unroll
            for (int c = 0; c < NA; c++) B.a[c] = (const uint8_t *)P.acol[c] + first * P.awid[c];
            B.t = nullptr;
            if constexpr (LATE) {
                const uint32_t n_tiles = (n + kTile - 1) / kTile;
                const uint32_t r_first = tid * kPackedRows;
                PackedRaw<NF> rfn;
                PackedRaw<NG> rgl;
                PackedRaw<NA> ral;
#pragma unroll
                for (int c = 0; c < NF; c++) packed_issue_clamped(B.f[c], P.fwid[c], r_first, n, true, rfn.v[c]);
                for (uint32_t it = 0; it < n_tiles; it++) {
                    const uint32_t r = r_first + it * kTile;
                    uint32_t fu[NF > 0 ? NF : 1][kPackedRows];
#pragma unroll
                    for (int c = 0; c < NF; c++) packed_decode(P.fwid[c], rfn.v[c], fu[c]);
#pragma unroll
                    for (int c = 0; c < NF; c++) packed_issue_clamped(B.f[c], P.fwid[c], r + kTile, n, true, rfn.v[c]);
                    const uint32_t left = r < n ? n - r : 0u;
                    bool pass[kPackedRows], any = false;
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        pass[k] = (uint32_t)k < left;
#pragma unroll
                        for (int c = 0; c < NF; c++) pass[k] = pass[k] & (fu[c][k] >= P.plo[c]) & (fu[c][k] <= P.phi[c]);
                        any = any | pass[k];
                        matched += pass[k] ? 1u : 0u;
                    }
                    if (!__builtin_amdgcn_ballot_w64(any)) continue;  // (wave-uniform: no row of these 256 passed)
#pragma unroll
                    for (int c = 0; c < NG; c++) packed_issue_clamped(B.g[c], P.gwid[c], r, n, true, rgl.v[c]);
#pragma unroll
                    for (int c = 0; c < NA; c++) packed_issue_clamped(B.a[c], P.awid[c], r, n, true, ral.v[c]);
                    uint32_t gu[NG > 0 ? NG : 1][kPackedRows], au[NA > 0 ? NA : 1][kPackedRows];
#pragma unroll
                    for (int c = 0; c < NG; c++) packed_decode(P.gwid[c], rgl.v[c], gu[c]);
#pragma unroll
                    for (int c = 0; c < NA; c++) packed_decode(P.awid[c], ral.v[c], au[c]);
                    uint32_t bin[kPackedRows * NA], rec[kPackedRows * NA];
                    bool act[kPackedRows * NA];
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        uint32_t cell = 0;
                        bool inb = true;
#pragma unroll
                        for (int c = 0; c < NG; c++) {
                            const uint32_t d = gu[c][k] + P.gdoff[c];
                            inb = inb & (d < P.gcard[c]);
                            cell += __umul24(d, (uint32_t)P.gstride[c]);
                        }
                        overflow += (pass[k] & !inb) ? 1u : 0u;
#pragma unroll
                        for (int c = 0; c < NA; c++) {
                            const uint32_t n32 = au[c][k] + P.adoff[c];  // value - h.Min
                            const uint32_t pair = cell * (uint32_t)NA + (uint32_t)c;
                            const int at = k * NA + c;
                            bin[at] = emit_bin(S, pair);
                            rec[at] = emit_record(pair, n32);
                            act[at] = pass[k] & inb;
                        }
                    }
                    emit_push_all<kPackedRows * NA, kEmitQueue>(E, S, bin, rec, act);
                }
                continue;
            }
            // D tiles of loads in flight per lane (narrow queries move few bytes per tile)
            constexpr int D = emit_depth(NF + NG + NA);
            // With one aggregation a tile is only four records per lane, and a push is a chain of ~7 dependent LDS round
            // trips (slot, wr, write + publish, queue, chunk, copy) that four waves per SIMD do not hide: two tiles are
            // pushed together (eight records per round trip; the drain then has four store instructions).
            constexpr int T = (NA == 1 && D % 2 == 0) ? 2 : 1;
            constexpr uint32_t Q = T == 2 ? kEmitQueueMax : kEmitQueue;
            PackedRaw<NF> rf[D];
            PackedRaw<NG> rg[D];
            PackedRaw<NA> ra[D];
            PackedRaw<1> rt;
            PackedTile<NF> f;
            PackedTile<NG> g;
            PackedTile<NA> a;
            PackedTile<1> t;
            const uint32_t n_tiles = (n + kTile - 1) / kTile;
            const uint32_t r_first = tid * kPackedRows;
#pragma unroll
            for (int d = 0; d < D; d++) {
                packed_issue_always<NF, NG, NA>(P, B, r_first + (uint32_t)d * kTile, n, rf[d], rg[d], ra[d]);
                if (d % T == T - 1) emit_pad_stores<Q>(S);  // (the ring as the steady state has it: see emit_pad_stores)
                __builtin_amdgcn_sched_barrier(0);  // oldest tile first: the ring is consumed in this order
            }
            for (uint32_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
              for (int d0 = 0; d0 < D; d0 += T) {
                // (no early exit inside a round: a tile past the end loads nothing -- the descriptor's range check -- and
                // pushes nothing; a conditional exit here is one more path the compiler's load / store counting must
                // take the minimum over)
                uint32_t bin[kPackedRows * NA * T], rec[kPackedRows * NA * T];
                bool act[kPackedRows * NA * T];
#pragma unroll
                for (int dd = 0; dd < T; dd++) {
                    const int d = d0 + dd;
                    const uint32_t r = r_first + (it0 + d) * kTile;   // (< 2^28 + 2^12: no wrap)
                    packed_decode_all<NF, NG, NA, false, false, false>(P, rf[d], rg[d], ra[d], rt, f, g, a, t, 0u);
                    packed_issue_always<NF, NG, NA>(P, B, r + (uint32_t)D * kTile, n, rf[d], rg[d], ra[d]);
                    const uint32_t left = r < n ? n - r : 0u;
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        bool pass = (uint32_t)k < left;
#pragma unroll
                        for (int c = 0; c < NF; c++) {
                            const uint32_t u = f.u[c][k];
                            pass = pass & (u >= P.plo[c]) & (u <= P.phi[c]);
                        }
                        uint32_t cell = 0;
                        bool inb = true;
#pragma unroll
                        for (int c = 0; c < NG; c++) {
                            const uint32_t d = g.u[c][k] + P.gdoff[c];
                            inb = inb & (d < P.gcard[c]);
                            cell += __umul24(d, (uint32_t)P.gstride[c]);
                        }
                        matched += pass ? 1u : 0u;
                        overflow += (pass & !inb) ? 1u : 0u;
#pragma unroll
                        for (int c = 0; c < NA; c++) {
                            const uint32_t n32 = a.u[c][k] + P.adoff[c];  // value - h.Min
                            const uint32_t pair = cell * (uint32_t)NA + (uint32_t)c;
                            const int at = (dd * kPackedRows + k) * NA + c;
                            bin[at] = emit_bin(S, pair);
                            rec[at] = emit_record(pair, n32);
                            act[at] = pass & inb;
                        }
                    }
                }
                emit_push_all<kPackedRows * NA * T, Q>(E, S, bin, rec, act);
              }
            }
        }
    }
    emit_finish(E, S, matched, overflow);
}

// k_count over compact storage: the counting pass of k_emit_packed (see k_count in scan_fast.h).
template <int NF, int NG, bool LATE = false>
__global__ __launch_bounds__(kWgThreads, 4) void k_count_packed(const EmitPlan E) {
    extern __shared__ uint32_t elds[];
    const FastPlan &P = E.fp;
    const uint32_t tid = threadIdx.x;
    uint32_t *mine = count_begin(E, elds);
    const uint32_t na = (uint32_t)E.n_aggs, ss = (uint32_t)E.sub_shift;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            PackedBases<NF, NG, 0> B;
#pragma unroll
            for (int c = 0; c < NF; c++) B.f[c] = (const uint8_t *)P.fcol[c] + first * P.fwid[c];
#pragma unroll
            for (int c = 0; c < NG; c++) B.g[c] = (const uint8_t *)P.gcol[c] + first * P.gwid[c];
            B.a[0] = nullptr;
            B.t = nullptr;
            if constexpr (LATE) {
                // (see k_emit_packed<.., LATE>: the same rows pass, so the regions it fills are the ones counted here)
                const uint32_t n_tiles = (n + kPackedTileRows - 1) / kPackedTileRows;
                const uint32_t r_first = tid * kPackedRows;
                PackedRaw<NF> rfn;
                PackedRaw<NG> rgl;
#pragma unroll
                for (int c = 0; c < NF; c++) packed_issue_clamped(B.f[c], P.fwid[c], r_first, n, true, rfn.v[c]);
                for (uint32_t it = 0; it < n_tiles; it++) {
                    const uint32_t r = r_first + it * kPackedTileRows;
                    uint32_t fu[NF > 0 ? NF : 1][kPackedRows];
#pragma unroll
                    for (int c = 0; c < NF; c++) packed_decode(P.fwid[c], rfn.v[c], fu[c]);
#pragma unroll
                    for (int c = 0; c < NF; c++) packed_issue_clamped(B.f[c], P.fwid[c], r + kPackedTileRows, n, true, rfn.v[c]);
                    const uint32_t left = r < n ? n - r : 0u;
                    bool pass[kPackedRows], any = false;
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        pass[k] = (uint32_t)k < left;
#pragma unroll
                        for (int c = 0; c < NF; c++) pass[k] = pass[k] & (fu[c][k] >= P.plo[c]) & (fu[c][k] <= P.phi[c]);
                        any = any | pass[k];
                    }
                    if (!__builtin_amdgcn_ballot_w64(any)) continue;
#pragma unroll
                    for (int c = 0; c < NG; c++) packed_issue_clamped(B.g[c], P.gwid[c], r, n, true, rgl.v[c]);
                    uint32_t gu[NG > 0 ? NG : 1][kPackedRows];
#pragma unroll
                    for (int c = 0; c < NG; c++) packed_decode(P.gwid[c], rgl.v[c], gu[c]);
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        uint32_t cell = 0;
                        bool ok = pass[k];
#pragma unroll
                        for (int c = 0; c < NG; c++) {
                            const uint32_t d = gu[c][k] + P.gdoff[c];
                            ok = ok & (d < P.gcard[c]);
                            cell += __umul24(d, (uint32_t)P.gstride[c]);
                        }
                        if (ok) __hip_atomic_fetch_add(mine + (((cell * na) >> kPartCellBits) << ss), na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                continue;
            }
            constexpr int D = count_depth(NF + NG);
            PackedRaw<NF> rf[D];
            PackedRaw<NG> rg[D];
            PackedRaw<0> ra;
            PackedRaw<1> rt;
            PackedTile<NF> f;
            PackedTile<NG> g;
            PackedTile<0> a;
            PackedTile<1> t;
            const uint32_t n_tiles = (n + kPackedTileRows - 1) / kPackedTileRows;
            const uint32_t r_first = tid * kPackedRows;
#pragma unroll
            for (int d = 0; d < D; d++) {
                packed_issue_always<NF, NG, 0>(P, B, r_first + (uint32_t)d * kPackedTileRows, n, rf[d], rg[d], ra);
                __builtin_amdgcn_sched_barrier(0);  // oldest tile first: the ring is consumed in this order
            }
            for (uint32_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
              for (int d = 0; d < D; d++) {
                if (it0 + d >= n_tiles) break;
                const uint32_t r = r_first + (it0 + d) * kPackedTileRows;
                packed_decode_all<NF, NG, 0, false, false, false>(P, rf[d], rg[d], ra, rt, f, g, a, t, 0u);
                packed_issue_always<NF, NG, 0>(P, B, r + (uint32_t)D * kPackedTileRows, n, rf[d], rg[d], ra);
                const uint32_t left = r < n ? n - r : 0u;
#pragma unroll
                for (int k = 0; k < kPackedRows; k++) {
                    bool pass = (uint32_t)k < left;
#pragma unroll
                    for (int c = 0; c < NF; c++) {
                        const uint32_t u = f.u[c][k];
                        pass = pass & (u >= P.plo[c]) & (u <= P.phi[c]);
                    }
                    uint32_t cell = 0;
#pragma unroll
                    for (int c = 0; c < NG; c++) {
                        const uint32_t d = g.u[c][k] + P.gdoff[c];
                        pass = pass & (d < P.gcard[c]);
                        cell += __umul24(d, (uint32_t)P.gstride[c]);
                    }
                    if (pass) __hip_atomic_fetch_add(mine + (((cell * na) >> kPartCellBits) << ss), na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
            }
        }
    }
    count_finish(E, elds);
}

// k_count_key: the counting pass for its commonest shape -- ONE key column, no filter, one bin per partition (sub_shift 0:
// config 4) -- with every lane's 16-byte load FULL of keys (eight 2-byte keys, sixteen 1-byte, four 4-byte) instead of the
// four rows k_count_packed shares with k_emit_packed's row -> lane mapping (a 2-byte key column then brings 8 useful
// bytes per 16-byte instruction: half the load instructions and half the tile overhead for the same 2 GB).  With one bin
// per partition the count of a (workgroup, bin) does not depend on WHICH lane saw a row, only on the workgroup's row range,
// which is the plan's: k_emit_packed's regions come out the same.  (SYBL_NO_COUNT16=1: k_count_packed.)
template <int W>
__global__ __launch_bounds__(kWgThreads, 4) void k_count_key(const EmitPlan E) {
    extern __shared__ uint32_t elds[];
    const FastPlan &P = E.fp;
    const uint32_t tid = threadIdx.x;
    uint32_t *bins = count_begin(E, elds);  // (sub_shift 0: the counters themselves)
    const uint32_t na = (uint32_t)E.n_aggs;
    constexpr uint32_t R = 16u / (uint32_t)W;       // keys per lane and tile
    constexpr uint32_t kTile = kWgThreads * R;       // rows per workgroup and tile
    constexpr int D = 4;                             // tiles of loads in flight
    const uint32_t gdoff = P.gdoff[0], gcard = P.gcard[0], gstride = (uint32_t)P.gstride[0];
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            const uint8_t *col = (const uint8_t *)P.gcol[0] + first * W;
            // one descriptor for the chunk: a lane (or a dword of it) past row n reads zeros without touching memory; the
            // rows it would stand for are masked by `left` below
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)col, 0, (int)(((n * (uint32_t)W) + 3u) & ~3u), (int)kBufferRsrcWord3);
            pu32x4 raw[D];
            const uint32_t r_first = tid * R;
            const uint32_t n_tiles = (n + kTile - 1) / kTile;
#pragma unroll
            for (int d = 0; d < D; d++) raw[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((r_first + (uint32_t)d * kTile) * (uint32_t)W), 0, 2);
            for (uint32_t it0 = 0; it0 < n_tiles; it0 += D) {
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const uint32_t r = r_first + (it0 + d) * kTile;  // (< 2^28 + 2^16: no wrap)
                    const pu32x4 v = raw[d];
                    raw[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((r + (uint32_t)D * kTile) * (uint32_t)W), 0, 2);
                    const uint32_t left = r < n ? n - r : 0u;
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t k = 0; k < R; k++) {
                        uint32_t u;
                        if (W == 4) u = w4[k];
                        else if (W == 2) u = (w4[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu;
                        else u = (w4[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu;
                        const uint32_t dgt = u + gdoff;
                        if (k < left && dgt < gcard)
                            __hip_atomic_fetch_add(bins + ((__umul24(dgt, gstride) * na) >> kPartCellBits), na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
    }
    count_finish(E, elds);
}

template <int NF>
static hipError_t count_packed_launch_nf(const EmitPlan &E, int ng, int n_wg, hipStream_t st) {
    const size_t lds = count_lds_bytes(E);
    if (NF == 0 && ng == 1 && E.sub_shift == 0 && !env("SYBL_NO_COUNT16")) {
        switch (E.fp.gwid[0]) {
        case 1: hipLaunchKernelGGL((k_count_key<1>), dim3(n_wg), dim3(kWgThreads), lds, st, E); return hipGetLastError();
        case 2: hipLaunchKernelGGL((k_count_key<2>), dim3(n_wg), dim3(kWgThreads), lds, st, E); return hipGetLastError();
        case 4: hipLaunchKernelGGL((k_count_key<4>), dim3(n_wg), dim3(kWgThreads), lds, st, E); return hipGetLastError();
        default: break;
        }
    }
    if constexpr (NF > 0) {
        if (E.fp.late) {  // (the planner's estimate: few rows pass -- filters first, keys for the waves with a match)
            switch (ng) {
            case 0: hipLaunchKernelGGL((k_count_packed<NF, 0, true>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
            case 1: hipLaunchKernelGGL((k_count_packed<NF, 1, true>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
            case 2: hipLaunchKernelGGL((k_count_packed<NF, 2, true>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
            default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    switch (ng) {
    case 0: hipLaunchKernelGGL((k_count_packed<NF, 0>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    case 1: hipLaunchKernelGGL((k_count_packed<NF, 1>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    case 2: hipLaunchKernelGGL((k_count_packed<NF, 2>), dim3(n_wg), dim3(kWgThreads), lds, st, E); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int NF>
static hipError_t emit_packed_launch_nf(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    const size_t lds = emit_lds_bytes(E);
#define SYBL_EMITP_CASE(G, A)                                                                              \
    case (G)*3 + (A): {                                                                                    \
        auto k = k_emit_packed<NF, G, A>;                                                                  \
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(kWgThreads), lds, st, E);                                   \
        return hipGetLastError();                                                                          \
    }
#define SYBL_EMITL_CASE(G, A)                                                                              \
    case (G)*3 + (A): {                                                                                    \
        auto k = k_emit_packed<NF, G, A, true>;                                                            \
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(kWgThreads), lds, st, E);                                   \
        return hipGetLastError();                                                                          \
    }
    if constexpr (NF > 0) {
        if (E.fp.late) {
            switch (ng * 3 + na) {
                SYBL_EMITL_CASE(0, 1)
                SYBL_EMITL_CASE(0, 2)
                SYBL_EMITL_CASE(1, 1)
                SYBL_EMITL_CASE(1, 2)
                SYBL_EMITL_CASE(2, 1)
                SYBL_EMITL_CASE(2, 2)
            default: return hipErrorInvalidValue;
            }
        }
    }
#undef SYBL_EMITL_CASE
    switch (ng * 3 + na) {
        SYBL_EMITP_CASE(0, 1)
        SYBL_EMITP_CASE(0, 2)
        SYBL_EMITP_CASE(1, 1)
        SYBL_EMITP_CASE(1, 2)
        SYBL_EMITP_CASE(2, 1)
        SYBL_EMITP_CASE(2, 2)
    default: return hipErrorInvalidValue;
    }
#undef SYBL_EMITP_CASE
}

template <int NF, int NG, int NA, int MODE, bool TIME, bool G1, bool NUL>
static hipError_t packed_launch_k1(const FastPlan &P, int n_wg, size_t lds_bytes, hipStream_t st) {
    // SYBL_PACKED_RING=1..3 (tuning): the BASELINE shapes with an explicit ring depth, for same-box A/B runs
    constexpr bool kAb = !NUL && ((NF == 3 && NG == 2 && NA == 2 && MODE == kFastMoments && !TIME && G1) ||
                                  (NF == 0 && NG == 1 && NA == 2 && MODE == kFastAvgMax && !TIME && G1) ||
                                  (NF == 0 && NG == 1 && NA == 1 && MODE == kFastAvg && TIME && !G1));
    if (kAb) {
        if (const char *e = env("SYBL_PACKED_RING")) {
            const int d = atoi(e);
            auto launch = [&](auto kern) {
                hipError_t e2 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e2 != hipSuccess) return e2;
                hipLaunchKernelGGL(kern, dim3(n_wg), dim3(kWgThreads), lds_bytes, st, P);
                return hipGetLastError();
            };
            if (d == 1) return launch(k_scan_packed<NF, NG, NA, MODE, TIME, G1, NUL, kAb ? 1 : 0>);
            if (d == 2) return launch(k_scan_packed<NF, NG, NA, MODE, TIME, G1, NUL, kAb ? 2 : 0>);
            if (d == 3) return launch(k_scan_packed<NF, NG, NA, MODE, TIME, G1, NUL, kAb ? 3 : 0>);
            if (d == 4) return launch(k_scan_packed<NF, NG, NA, MODE, TIME, G1, NUL, kAb ? 4 : 0>);
        }
    }
    auto k = k_scan_packed<NF, NG, NA, MODE, TIME, G1, NUL>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(kWgThreads), lds_bytes, st, P);
    return hipGetLastError();
}

template <int NF, int NG, int NA, int MODE, bool TIME>
static hipError_t packed_launch_k(const FastPlan &P, int n_wg, size_t lds_bytes, hipStream_t st) {
    bool g1 = NG > 0;  // every group column is stored in one byte
    for (int c = 0; c < NG; c++) g1 = g1 && P.gwid[c] == 1;
    // NUL: missing rows / str ids / the reject gate (validity bitmaps are loaded next to the columns)
    if (P.nul) return packed_launch_k1<NF, NG, NA, MODE, TIME, false, true>(P, n_wg, lds_bytes, st);
    if (NG > 0 && g1) return packed_launch_k1<NF, NG, NA, MODE, TIME, (NG > 0), false>(P, n_wg, lds_bytes, st);
    return packed_launch_k1<NF, NG, NA, MODE, TIME, false, false>(P, n_wg, lds_bytes, st);
}

template <int NF, int NG, int NA>
static hipError_t packed_launch_mode(const FastPlan &P, int mode, bool time, int n_wg, size_t lds, hipStream_t st) {
#define SYBL_PACKED_MODE(M) \
    return time ? packed_launch_k<NF, NG, NA, M, true>(P, n_wg, lds, st) : packed_launch_k<NF, NG, NA, M, false>(P, n_wg, lds, st)
    if (NA == 0) SYBL_PACKED_MODE(kFastAvg);
    switch (mode) {
    case kFastAvg: SYBL_PACKED_MODE(kFastAvg);
    case kFastAvgMax: SYBL_PACKED_MODE(kFastAvgMax);
    case kFastMoments: SYBL_PACKED_MODE(kFastMoments);
    case kFastHist: SYBL_PACKED_MODE(kFastHist);
    default: return hipErrorInvalidValue;
    }
#undef SYBL_PACKED_MODE
}

template <int NF>
static hipError_t packed_launch_nf(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st) {
    switch (ng * 3 + na) {
    case 0: return packed_launch_mode<NF, 0, 0>(P, mode, time, n_wg, lds, st);
    case 1: return packed_launch_mode<NF, 0, 1>(P, mode, time, n_wg, lds, st);
    case 2: return packed_launch_mode<NF, 0, 2>(P, mode, time, n_wg, lds, st);
    case 3: return packed_launch_mode<NF, 1, 0>(P, mode, time, n_wg, lds, st);
    case 4: return packed_launch_mode<NF, 1, 1>(P, mode, time, n_wg, lds, st);
    case 5: return packed_launch_mode<NF, 1, 2>(P, mode, time, n_wg, lds, st);
    case 6: return packed_launch_mode<NF, 2, 0>(P, mode, time, n_wg, lds, st);
    case 7: return packed_launch_mode<NF, 2, 1>(P, mode, time, n_wg, lds, st);
    case 8: return packed_launch_mode<NF, 2, 2>(P, mode, time, n_wg, lds, st);
    default: return hipErrorInvalidValue;
    }
}

#endif  // __HIPCC__

}  // namespace sybl
