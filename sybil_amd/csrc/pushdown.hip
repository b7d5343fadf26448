// pushdown.hip -- -limit pushed into the scan for a PRINTER's histogram query sorted by $COUNT (round 6).
//
// `sybil query -group k -int v -op hist` prints the first FLAGS.LIMIT rows of the sort order and the TOTAL row
// (printer.go:154-158,291-308; the sort: aggregate.go:497-525), but the reference -- and this engine's partitioned
// histograms, strategy 5 -- build EVERY group's bucket array first (hist_basic.go:101-151): config 4 writes 4.7 GB of records,
// reads them back and fills a 525 MB table to print 100 of 65 536 rows.  When the sort key is the group's Count the printed
// rows are known before any bucket is touched:
//   pass 1  k_pd_count   the key column alone (2 GB): a Count per group in 15-bit LDS counters, two to a word, the 16th bit a
//                        guard that absorbs the wrap (the lane that sees it takes 32768 out and adds it to a device-side
//                        carry word: rare); one 128 KB table per workgroup, folded by k_pd_fold into Result.Count of every cell
//           k_pd_select  one workgroup: the `limit` largest counts, ties by cell number -- exactly the stable sort over the
//                        canonical key order that finalize does on the host -- as a bitmap of printed cells
//   pass 2  k_pd_scan    key + value columns once (6 GB): every row adds to Cumulative's buckets in LDS and to a per-lane sum;
//                        a row whose group is printed (~limit / groups of them) adds to that group's bucket array, sum and
//                        maximum in HBM with device-scope atomics -- in the same places strategy 5 leaves them, so snapshot and
//                        finalize read them as they always do.
// No records, no bucket table: ~8 GB moved instead of ~16.  The rows beyond the limit carry their Count and nothing else
// (sybl_query_desc.printed_only = 2 says so); both printers' output is byte for byte that of the full path.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine.h"
#include "scan_generic.h"
#include "scan_packed.h"

namespace sybl {

constexpr uint32_t kPdGuard = 0x8000u;  // bit 15 of a 16-bit counter field: set by the add that takes the field past 32767

// ---------------------------------------------------------------- pass 1: a Count per group
template <int W>
__global__ __launch_bounds__(kWgThreads, 4) void k_pd_count(const PushdownPlan D) {
    extern __shared__ uint32_t plds[];
    const FastPlan &P = D.fp;
    const uint32_t tid = threadIdx.x;
    const uint32_t words = ((uint32_t)D.n_cells + 1u) >> 1;
    for (uint32_t i = tid; i < words; i += kWgThreads) plds[i] = 0;
    __syncthreads();
    constexpr uint32_t R = 16u / (uint32_t)W;   // keys per lane and tile: every lane's 16-byte load full of keys (k_count_key)
    constexpr uint32_t kTile = kWgThreads * R;
    constexpr int DEPTH = 4;
    const uint32_t gdoff = P.gdoff[0], gcard = P.gcard[0];
    uint32_t overflow = 0;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            const uint8_t *col = (const uint8_t *)P.gcol[0] + first * W;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)col, 0, (int)(((n * (uint32_t)W) + 3u) & ~3u), (int)kBufferRsrcWord3);
            pu32x4 raw[DEPTH];
            const uint32_t r_first = tid * R;
            const uint32_t n_tiles = (n + kTile - 1) / kTile;
#pragma unroll
            for (int d = 0; d < DEPTH; d++) raw[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((r_first + (uint32_t)d * kTile) * (uint32_t)W), 0, 2);
            for (uint32_t it0 = 0; it0 < n_tiles; it0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
                    const uint32_t r = r_first + (it0 + d) * kTile;
                    const pu32x4 v = raw[d];
                    raw[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((r + (uint32_t)DEPTH * kTile) * (uint32_t)W), 0, 2);
                    const uint32_t left = r < n ? n - r : 0u;
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                    // the tile's adds first, every lane's R of them in flight together, their returns looked at afterwards (a
                    // check between two adds made each wait for the one before it)
                    uint32_t cells[R], olds[R];
#pragma unroll
                    for (uint32_t k = 0; k < R; k++) {
                        uint32_t u;
                        if (W == 4) u = w4[k];
                        else if (W == 2) u = (w4[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu;
                        else u = (w4[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu;
                        const uint32_t cell = u + gdoff;
                        const bool in = k < left && cell < gcard;
                        overflow += (k < left && cell >= gcard) ? 1u : 0u;
                        cells[k] = in ? cell : 0xFFFFFFFFu;
                        olds[k] = 0;
                        if (in) olds[k] = __hip_atomic_fetch_add(plds + (cell >> 1), 1u << ((cell & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#pragma unroll
                    for (uint32_t k = 0; k < R; k++) {
                        const uint32_t cell = cells[k], sh = (cell & 1u) << 4;
                        if (cell != 0xFFFFFFFFu && ((olds[k] >> sh) & 0x7FFFu) == 0x7FFFu) {
                            // this add took the field to 32768: the guard bit holds it (nothing carried into the neighbour) until
                            // the 32768 are taken out here and remembered device-side.  (Another full wrap of the field inside
                            // that window would need 32767 more adds to this cell before this lane gets here.)
                            __hip_atomic_fetch_sub(plds + (cell >> 1), kPdGuard << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(D.carry + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    uint32_t *ws = D.ws + (size_t)blockIdx.x * words;
    for (uint32_t i = tid; i < words; i += kWgThreads) ws[i] = plds[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) overflow += __shfl_xor(overflow, o, 64);
    if ((tid & 63u) == 0 && overflow) __hip_atomic_fetch_add(D.sum_out + kHdrOverflow, (int64_t)overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the workgroups' tables -> Result.Count of every cell (int64, field 0 of the cell table) and the uint32 copy the select reads
__global__ __launch_bounds__(256) void k_pd_fold(const PushdownPlan D) {
    const uint32_t words = ((uint32_t)D.n_cells + 1u) >> 1;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint32_t lo = 0, hi = 0;
    for (int g = 0; g < D.n_wg; g++) {
        const uint32_t x = D.ws[(size_t)g * words + w];
        lo += x & 0xFFFFu;
        hi += x >> 16;
    }
    const uint32_t c0 = 2u * w, c1 = 2u * w + 1u;
    lo += D.carry[c0] * 32768u;
    D.cnt[c0] = lo;
    D.sum_out[kHeaderWords + c0] = (int64_t)lo;
    if (c1 < (uint32_t)D.n_cells) {
        hi += D.carry[c1] * 32768u;
        D.cnt[c1] = hi;
        D.sum_out[kHeaderWords + c1] = (int64_t)hi;
    }
}

// ---------------------------------------------------------------- the printed cells
// One workgroup.  The `limit` first cells of SortResults' order (aggregate.go:497-525 as finalize runs it: a stable sort by
// Count, descending, over the live cells in cell order): every cell whose count exceeds the limit-th largest count T, then the
// cells with count == T in cell order until `limit` are taken.
__global__ __launch_bounds__(1024) void k_pd_select(const PushdownPlan D) {
    __shared__ uint32_t red[1024 / 64];
    __shared__ uint32_t s_total;
    const uint32_t tid = threadIdx.x, n = (uint32_t)D.n_cells, L = (uint32_t)D.limit;
    auto block_sum = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63u) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            uint32_t s = 0;
            for (int k = 0; k < 16; k++) s += red[k];
            s_total = s;
        }
        __syncthreads();
        return s_total;
    };
    // the counts of this thread's cells, once, in registers (n <= 65536: at most 64 per thread); every probe of the search
    // below is then 64 compares and one block sum, and the search runs over [1, largest count] only
    constexpr uint32_t kPer = 64;
    uint32_t mine[kPer];
    uint32_t top = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t i = tid + k * 1024u;
        mine[k] = i < n ? D.cnt[i] : 0u;
        top = mine[k] > top ? mine[k] : top;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m = __shfl_xor(top, o, 64);
        top = m > top ? m : top;
    }
    if ((tid & 63u) == 0) red[tid >> 6] = top;
    __syncthreads();
    top = 0;
    for (int k = 0; k < 16; k++) top = red[k] > top ? red[k] : top;
    __syncthreads();
    auto count_ge = [&](uint32_t t) -> uint32_t {  // live cells with count >= t (t >= 1)
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) c += mine[k] >= t ? 1u : 0u;
        return block_sum(c);
    };
    // T = the largest t >= 1 with count_ge(t) >= L; fewer than L live cells: T = 1 (all of them are printed)
    uint32_t T = 1;
    if (top >= 1 && count_ge(1) >= L) {
        uint32_t lo = 1, hi = top;  // invariant: count_ge(lo) >= L
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            if (count_ge(mid) >= L) lo = mid;
            else hi = mid - 1;
        }
        T = lo;
    }
    const uint32_t above = T == 0xFFFFFFFFu ? 0u : count_ge(T + 1);
    const uint32_t take_eq = L > above ? L - above : 0u;  // cells with count == T, in cell order
    for (uint32_t i = tid; i < (n + 31u) / 32u; i += 1024) D.bitmap[i] = 0;
    __syncthreads();
    // cells in cell order, a contiguous range per thread, so that the ties' ranks follow from one exclusive scan
    const uint32_t per = (n + 1023u) / 1024u, c_lo = tid * per, c_hi = min(n, c_lo + per);
    uint32_t mine_eq = 0;
    for (uint32_t c = c_lo; c < c_hi; c++) mine_eq += D.cnt[c] == T ? 1u : 0u;
    __shared__ uint32_t scan[1024];
    scan[tid] = mine_eq;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? scan[tid - o] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    uint32_t rank = scan[tid] - mine_eq;
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint32_t x = D.cnt[c];
        bool pick = x > T;
        if (x == T && x > 0) {
            pick = rank < take_eq;
            rank++;
        }
        if (pick) {
            atomicOr(D.bitmap + (c >> 5), 1u << (c & 31u));
            const uint32_t at = atomicAdd((uint32_t *)D.n_top, 1u);
            if (at < L) D.top_cells[at] = (int32_t)c;
        }
    }
}

// the printed cells' bucket arrays start from zero (the rest of the table is never read in this mode)
__global__ __launch_bounds__(256) void k_pd_clear(const PushdownPlan D) {
    const uint32_t n_top = min((uint32_t)*D.n_top, (uint32_t)D.limit);
    if (blockIdx.x >= n_top) return;
    int64_t *h = D.sum_out + D.hist_off + (int64_t)D.top_cells[blockIdx.x] * D.hist_stride;
    for (int64_t w = threadIdx.x; w < D.hist_stride; w += blockDim.x) h[w] = 0;
}

// ---------------------------------------------------------------- pass 2: Cumulative, and the printed groups
template <int NA>
__global__ __launch_bounds__(kWgThreads, 4) void k_pd_scan(const PushdownPlan D) {
    extern __shared__ uint32_t plds[];
    const FastPlan &P = D.fp;
    const uint32_t tid = threadIdx.x;
    const uint32_t bm_words = ((uint32_t)D.n_cells + 31u) >> 5;
    uint32_t *bitmap = plds, *cum = plds + bm_words;  // [hist_stride]: Cumulative's buckets, the aggregations' arrays back to back
    for (uint32_t i = tid; i < bm_words; i += kWgThreads) bitmap[i] = D.bitmap[i];
    for (uint32_t i = tid; i < (uint32_t)D.hist_stride; i += kWgThreads) cum[i] = 0;
    __syncthreads();
    const uint32_t gdoff = P.gdoff[0], gcard = P.gcard[0];
    int64_t *F = D.sum_out + kHeaderWords, *H = D.sum_out + D.hist_off;
    unsigned long long sum[NA];
    long long vmax[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) sum[a] = 0, vmax[a] = INT64_MIN;
    uint32_t matched = 0;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            const uint32_t r_first = tid * kPackedRows;
            const uint32_t n_tiles = (n + kPackedTileRows - 1) / kPackedTileRows;
            pu32x4 rg, ra[NA];
            auto issue = [&](uint32_t r) {
                const uint32_t r0 = __builtin_amdgcn_readfirstlane(r), lane_row = r - r0;
                const uint32_t rows = r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
                auto ld = [&](const void *col, int width, pu32x4 &raw) {
                    const int ws = width >> 1;
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)((const uint8_t *)col + (size_t)(first + (rows ? r0 : 0u)) * (size_t)width), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
                    raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);
                };
                ld(P.gcol[0], P.gwid[0], rg);
#pragma unroll
                for (int a = 0; a < NA; a++) ld(P.acol[a], P.awid[a], ra[a]);
            };
            issue(r_first);
            for (uint32_t it = 0; it < n_tiles; it++) {
                const uint32_t r = r_first + it * kPackedTileRows;
                uint32_t g[kPackedRows], av[NA][kPackedRows];
                packed_decode(P.gwid[0], rg, g);
#pragma unroll
                for (int a = 0; a < NA; a++) packed_decode(P.awid[a], ra[a], av[a]);
                issue(r + kPackedTileRows);  // (past the chunk: a descriptor of zero records)
                const uint32_t left = r < n ? n - r : 0u;
#pragma unroll
                for (int k = 0; k < kPackedRows; k++) {
                    if ((uint32_t)k >= left) continue;
                    const uint32_t cell = g[k] + gdoff;
                    if (cell >= gcard) continue;  // (counted as overflow by pass 1)
                    matched += 1;
                    const bool printed = (bitmap[cell >> 5] >> (cell & 31u)) & 1u;
#pragma unroll
                    for (int a = 0; a < NA; a++) {
                        const uint32_t u = av[a][k];
                        const int64_t x = (int64_t)((uint64_t)P.abase[a] + u);
                        // bucket_value := (value - h.Min) / BucketSize (hist_basic.go:130); no value reaches len(Values) (planner)
                        const uint32_t b = packed_udiv(u + P.adoff[a], P.bucket_size[a], P.pinv_bucket[a]);
                        __hip_atomic_fetch_add(cum + (uint32_t)P.hist_agg_off[a] + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        sum[a] += (unsigned long long)x;
                        vmax[a] = x > vmax[a] ? x : vmax[a];
                        if (printed) {
                            __hip_atomic_fetch_add(H + (int64_t)cell * D.hist_stride + P.hist_agg_off[a] + b, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(F + (int64_t)P.f_sum[a] * D.n_cells + cell, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (P.m_max[a] >= 0)
                                __hip_atomic_fetch_max(D.max_out + (int64_t)P.m_max[a] * D.n_cells + cell, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)D.hist_stride; i += kWgThreads)
        if (cum[i]) __hip_atomic_fetch_add(D.total + i, (int64_t)cum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        matched += __shfl_xor(matched, o, 64);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            sum[a] += __shfl_xor(sum[a], o, 64);
            const long long m = __shfl_xor(vmax[a], o, 64);
            vmax[a] = m > vmax[a] ? m : vmax[a];
        }
    }
    // (per WORKGROUP, not per wave: thousands of atomics on one device-scope word are serialised by the memory side -- tens of
    // microseconds at the end of the launch, scan_generic.h: wg_header_add)
    {
        __shared__ unsigned long long wsum[1 + NA];
        __shared__ long long wmax[NA];
        if (tid < 1u + (uint32_t)NA) wsum[tid] = 0;
        if (tid < (uint32_t)NA) wmax[tid] = INT64_MIN;
        __syncthreads();
        if ((tid & 63u) == 0) {
            if (matched) __hip_atomic_fetch_add(&wsum[0], (unsigned long long)matched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int a = 0; a < NA; a++) {
                __hip_atomic_fetch_add(&wsum[1 + a], (unsigned long long)sum[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&wmax[a], (long long)vmax[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if (tid == 0 && wsum[0]) __hip_atomic_fetch_add(D.sum_out + kHdrMatched, (int64_t)wsum[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid >= 1u && tid < 1u + (uint32_t)NA)
            __hip_atomic_fetch_add(D.sum_out + kHdrPdSum + (tid - 1u), (int64_t)wsum[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid >= 64u && tid < 64u + (uint32_t)NA)
            __hip_atomic_fetch_max(D.sum_out + kHdrPdMax + (tid - 64u), (int64_t)wmax[tid - 64u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the whole sequence on one stream; the caller has zeroed the header, the sum fields, carry, n_top and d_total and filled the
// MAX section with INT64_MIN
// pass 1: every group's count (D.cnt; the cells' Count fields)
hipError_t launch_pushdown_count(const PushdownPlan &D, hipStream_t st) {
    const uint32_t words = ((uint32_t)D.n_cells + 1u) >> 1;
    const size_t lds1 = (size_t)words * 4;
    hipError_t e;
    switch (D.fp.gwid[0]) {
    case 1:
        if ((e = hipFuncSetAttribute((const void *)k_pd_count<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_count<1>, dim3(D.n_wg), dim3(kWgThreads), lds1, st, D);
        break;
    case 2:
        if ((e = hipFuncSetAttribute((const void *)k_pd_count<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_count<2>, dim3(D.n_wg), dim3(kWgThreads), lds1, st, D);
        break;
    case 4:
        if ((e = hipFuncSetAttribute((const void *)k_pd_count<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_count<4>, dim3(D.n_wg), dim3(kWgThreads), lds1, st, D);
        break;
    default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_pd_fold, dim3((words + 255) / 256), dim3(256), 0, st, D);
    return hipGetLastError();
}

// the printed cells from D.cnt (across ranks: the all-reduced counts), then pass 2
hipError_t launch_pushdown_scan(const PushdownPlan &D, hipStream_t st) {
    hipError_t e;
    hipLaunchKernelGGL(k_pd_select, dim3(1), dim3(1024), 0, st, D);
    hipLaunchKernelGGL(k_pd_clear, dim3((unsigned)std::max(D.limit, 1)), dim3(256), 0, st, D);
    const size_t lds2 = ((((size_t)D.n_cells + 31) >> 5) + (size_t)D.hist_stride) * 4;
    if (D.n_aggs == 1) {
        if ((e = hipFuncSetAttribute((const void *)k_pd_scan<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_scan<1>, dim3(D.n_wg), dim3(kWgThreads), lds2, st, D);
    } else if (D.n_aggs == 2) {
        if ((e = hipFuncSetAttribute((const void *)k_pd_scan<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_scan<2>, dim3(D.n_wg), dim3(kWgThreads), lds2, st, D);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace sybl
