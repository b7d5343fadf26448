// kernels.hip -- hand-written CDNA4 (gfx950) kernels for sybil's scan hot path.
//
// k_scan<NC,USE_LDS> is the per-row body of FilterAndAggRecords (reference
// src/lib/aggregate.go:96-263) + BasicHist.AddWeightedValue (hist_basic.go:101-151),
// restructured for the machine:
//   * columnar: each referenced column ("slot") is streamed once with 16-byte
//     non-temporal loads, 64 lanes x 16 B = 1 KiB per wave instruction; the next tile
//     is requested before the current one is consumed (register double buffer);
//   * filters are folded to an inclusive range / neq list / dictionary bitmask per
//     slot and evaluated in registers; the matched count is a per-lane register
//     reduced once per wave at the end;
//   * group-by is a direct-mapped cell index; the cell table (Count, per-agg
//     sum / sum(b) / sum(b^2) / extrema) lives in LDS, replicated across lanes at low
//     cardinality to defeat same-address serialisation, and is written out ONCE per
//     workgroup; k_fold then reduces the per-workgroup tables;
//   * when the cell table does not fit in LDS the same body accumulates with
//     device-scope atomics straight into HBM (USE_LDS = false);
//   * no MFMA anywhere: the path is HBM-bound integer work.
//
// Everything is integer-exact, so results do not depend on the order rows are
// visited in -- which is what lets 256 workgroups x N GPUs own disjoint row ranges
// and be merged with a SUM/MAX all-reduce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "plan.h"
#include "gob_bins.h"
#include "scan_fast.h"
#include "scan_generic.h"

namespace sybl {

// ---------------------------------------------------------------- the scan
// (tile loads, the row body and the small helpers live in scan_generic.h, shared with k_scan_hash)

template <int NC, bool USE_LDS>
__device__ __forceinline__ void process_tile(CPlan &P, const Tile<NC> &t, int64_t row0, int nvalid, int64_t *sumtab,
                                             int64_t *maxtab, int rep, int32_t cell_base, int64_t &matched,
                                             int64_t &overflow) {
    const int rs = USE_LDS ? P.rep_shift : 0;
#pragma unroll
    for (int r = 0; r < kRowsPerThread; r++) {
        if (r >= nvalid) break;
        uint64_t key;
        int64_t w;
        const int st = row_prepare<NC>(P, t, r, row0, key, w);
        if (st == kRowFail) continue;
        matched += 1;  // aggregate.go:117
        if (st == kRowDropped) continue;
        if (st == kRowOverflow) {
            overflow += 1;
            continue;
        }
        const int32_t cell = (int32_t)key;
        // the table this lane accumulates into: the whole cell table, or (LDS window) the
        // lds_cells cells starting at the workgroup's base
        const int64_t ncell = USE_LDS ? P.lds_cells : P.n_cells;
        const int32_t lcell = cell - cell_base;
        if ((uint32_t)lcell >= (uint32_t)ncell) {
            overflow += 1;
            continue;
        }
        // field f of cell: ((f * ncell + lcell) << rep_shift) + rep
        const int64_t cidx = ((int64_t)lcell << rs) + rep;
        row_accumulate<NC, USE_LDS>(P, t, r, sumtab, maxtab, ncell, rs, cidx, (int64_t)cell, (int64_t)cell, w, overflow);
    }
}

// T threads per workgroup: at 1024 the wide bodies (NC >= 5) sat at the 128-VGPR cap and spilled inside the row loop.
template <int NC, bool USE_LDS, int T>
__global__ __launch_bounds__(T) void k_scan(CPlan *Pp) {
    CPlan &P = *Pp;
    extern __shared__ int64_t lds[];
    const int tid = threadIdx.x;
    const int64_t tab_cells = USE_LDS ? P.lds_cells : P.n_cells;
    const int64_t words_sum = (int64_t)P.n_sum_fields * tab_cells;
    const int64_t words_max = (int64_t)P.n_max_fields * tab_cells;
    const int32_t cell_base = (USE_LDS && P.windowed) ? P.wg_cell_base[blockIdx.x] : 0;
    int64_t *sumtab, *maxtab;
    int rep = 0;
    if (USE_LDS) {
        const int R = 1 << P.rep_shift;
        sumtab = lds;
        maxtab = lds + words_sum * R;
        for (int64_t i = tid; i < words_sum * R; i += T) sumtab[i] = 0;
        for (int64_t i = tid; i < words_max * R; i += T) maxtab[i] = INT64_MIN;
        rep = tid & (R - 1);
        __syncthreads();
    } else {
        sumtab = P.sum_out + kHeaderWords;
        maxtab = P.max_out;
    }

    int64_t matched = 0, overflow = 0;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        int64_t row = seg.start + (int64_t)tid * kRowsPerThread;
        Tile<NC> cur;
        RawTile<NC> raw;
        if (row < end) issue_tile<NC>(P, row, raw);
        decode_tile<NC>(P, row, row < end, raw, cur);
        for (int64_t base = seg.start; base < end; base += (T * kRowsPerThread)) {
            // prefetch: the next tile's raw bits are in flight under the LDS work and are decoded (their
            // first use) only after it
            const int64_t nrow = row + (T * kRowsPerThread);
            if (nrow < end) issue_tile<NC>(P, nrow, raw);
            int64_t left = end - row;
            int nvalid = left >= kRowsPerThread ? kRowsPerThread : (left > 0 ? (int)left : 0);
            process_tile<NC, USE_LDS>(P, cur, row, nvalid, sumtab, maxtab, rep, cell_base, matched, overflow);
            decode_tile<NC>(P, nrow, nrow < end, raw, cur);
            row = nrow;
        }
    }

    matched = wave_sum(matched);
    overflow = wave_sum(overflow);
    {
        const int slot[2] = {kHdrMatched, kHdrOverflow};
        const int64_t v[2] = {(int64_t)matched, (int64_t)overflow};
        wg_header_add<2>(P.sum_out, slot, v);  // (one atomic per workgroup and counter: scan_generic.h)
    }

    if (USE_LDS) {
        // fold the lane replicas and publish this workgroup's table (plain stores, no atomics)
        __syncthreads();
        const int R = 1 << P.rep_shift;
        if (P.windowed) {
            // flush the window into the global table: only touched cells, device-scope atomics
            int64_t *gs = P.sum_out + kHeaderWords;
            for (int64_t i = tid; i < words_sum; i += T) {
                int64_t a = 0;
                for (int k = 0; k < R; k++) a += sumtab[(i << P.rep_shift) + k];
                if (a != 0) {
                    const int64_t f = i / tab_cells, c = i - f * tab_cells;
                    gadd(gs + f * P.n_cells + cell_base + c, a);
                }
            }
            for (int64_t i = tid; i < words_max; i += T) {
                int64_t a = INT64_MIN;
                for (int k = 0; k < R; k++) {
                    int64_t b = maxtab[(i << P.rep_shift) + k];
                    a = b > a ? b : a;
                }
                if (a != INT64_MIN) {
                    const int64_t f = i / tab_cells, c = i - f * tab_cells;
                    __hip_atomic_fetch_max(P.max_out + f * P.n_cells + cell_base + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        int64_t *ws = P.ws_sum + (int64_t)blockIdx.x * words_sum;
        for (int64_t i = tid; i < words_sum; i += T) {
            int64_t a = 0;
            for (int k = 0; k < R; k++) a += sumtab[(i << P.rep_shift) + k];
            ws[i] = a;
        }
        int64_t *wm = P.ws_max + (int64_t)blockIdx.x * words_max;
        for (int64_t i = tid; i < words_max; i += T) {
            int64_t a = INT64_MIN;
            for (int k = 0; k < R; k++) {
                int64_t b = maxtab[(i << P.rep_shift) + k];
                a = b > a ? b : a;
            }
            wm[i] = a;
        }
    }
}

// out[i] = reduce over workgroups of ws[w][i]: SUM for the first nb_sum blocks, MAX for the rest.
// A block owns 64 consecutive words; its 16 waves stride over the workgroup tables (coalesced
// 512-byte reads, 16 independent loads per lane) and meet in LDS.
__global__ __launch_bounds__(1024) void k_fold(const int64_t *__restrict__ ws_sum, int64_t *__restrict__ out_sum,
                                               int64_t words_sum, const int64_t *__restrict__ ws_max,
                                               int64_t *__restrict__ out_max, int64_t words_max, int n_wg, int nb_sum) {
    __shared__ int64_t part[16][64];
    const bool is_max = (int)blockIdx.x >= nb_sum;
    const int64_t *ws = is_max ? ws_max : ws_sum;
    int64_t *out = is_max ? out_max : out_sum;
    const int64_t words = is_max ? words_max : words_sum;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int64_t i = (int64_t)(is_max ? blockIdx.x - nb_sum : blockIdx.x) * 64 + lane;
    int64_t a = is_max ? INT64_MIN : 0;
    if (i < words) {
#pragma unroll 4
        for (int w = sub; w < n_wg; w += 16) {
            const int64_t b = ws[(int64_t)w * words + i];
            a = is_max ? (b > a ? b : a) : a + b;
        }
    }
    part[sub][lane] = a;
    __syncthreads();
    if (sub == 0 && i < words) {
#pragma unroll
        for (int s = 1; s < 16; s++) {
            const int64_t b = part[s][lane];
            a = is_max ? (b > a ? b : a) : a + b;
        }
        out[i] = a;
    }
}

__global__ __launch_bounds__(256) void k_fill64(int64_t *p, int64_t n, int64_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ---------------------------------------------------------------- SYBL_VERIFY_COPIES
// k_copy_digest: an order-free digest of n 32-bit words as they lie in HBM -- the sum of splitmix64(word + (index << 32)) --
// compared with the same sum over the host bytes a host -> device copy was made from (table.cpp: host_to_device).  One
// small kernel and an 8-byte read-back per copy, only under SYBL_VERIFY_COPIES=1: it exists because of ONE unexplained
// event (round 5, DESIGN.md section 5: a freshly appended key column held ~1 KB of its neighbour's bytes) -- a repeat fails
// at the copy that went wrong, not at a group count much later.
__global__ __launch_bounds__(256) void k_copy_digest(const uint32_t *__restrict__ p, int64_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += splitmix64((uint64_t)p[i] + ((uint64_t)i << 32));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) __hip_atomic_fetch_add(out, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_copy_digest(const void *p, int64_t n_words, unsigned long long *out, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
    const unsigned nb = (unsigned)std::min<int64_t>(1024, (n_words + 255) / 256);
    hipLaunchKernelGGL(k_copy_digest, dim3(nb), dim3(256), 0, st, (const uint32_t *)p, n_words, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- synthetic columns
// Bit-identical to oracle/sybil_oracle.c:orc_synth_fill (the generator is ours, not the
// reference's: SURVEY.md 8d).
__global__ __launch_bounds__(256) void k_synth(int64_t *out, int64_t n, int64_t row0, int64_t total_rows, int kind,
                                               int64_t a, int64_t b, uint64_t col_seed) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < n; k += stride) {
        const uint64_t i = (uint64_t)(row0 + k);
        int64_t v;
        if (kind == 1) {  // TIME
            v = a + (int64_t)((i * (uint64_t)b) / (uint64_t)total_rows);
        } else if (kind == 2) {  // BELL
            uint64_t h = splitmix64(col_seed ^ i);
            uint64_t s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) s += (((h >> (16 * j)) & 0xFFFFu) * (uint64_t)b) >> 16;
            v = a + (int64_t)s;
        } else {  // UNIFORM
            uint64_t h = splitmix64(col_seed ^ i);
            v = a + (int64_t)__umul64hi(h, (uint64_t)b);
        }
        out[k] = v;
    }
}

// one stored value of a column, decoded (one-time kernels: statistics, distinct values, repacking)
__device__ __forceinline__ int64_t load_val(const void *base, int width, int64_t vbase, int64_t row) {
    switch (width) {
    case 8: return ((const int64_t *)base)[row];
    case 4: return vbase + (int64_t)((const uint32_t *)base)[row];
    case 2: return vbase + (int64_t)((const uint16_t *)base)[row];
    default: return vbase + (int64_t)((const uint8_t *)base)[row];
    }
}

// ---------------------------------------------------------------- per-block extrema
// One workgroup per block of rows: min/max over populated values.  Feeds block skipping
// (table_block_io.go:110-182) and the direct-mapped group layout.
__global__ __launch_bounds__(256) void k_block_minmax(const void *__restrict__ col, int width, int64_t vbase,
                                                      const uint32_t *__restrict__ valid,
                                                      const Segment *__restrict__ blocks, int64_t *__restrict__ out_min,
                                                      int64_t *__restrict__ out_max, int64_t *__restrict__ out_pop) {
    const Segment b = blocks[blockIdx.x];
    int64_t mn = INT64_MAX, mx = INT64_MIN, pc = 0;
    for (int64_t i = threadIdx.x; i < b.n; i += blockDim.x) {
        const int64_t row = b.start + i;
        if (valid && !((valid[row >> 5] >> (row & 31)) & 1u)) continue;
        const int64_t v = load_val(col, width, vbase, row);
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        pc++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int64_t m2 = __shfl_xor(mn, o, 64), x2 = __shfl_xor(mx, o, 64);
        mn = m2 < mn ? m2 : mn;
        mx = x2 > mx ? x2 : mx;
        pc += __shfl_xor(pc, o, 64);
    }
    __shared__ int64_t smn[4], smx[4], spc[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        smn[wave] = mn;
        smx[wave] = mx;
        spc[wave] = pc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            mn = smn[k] < mn ? smn[k] : mn;
            mx = smx[k] > mx ? smx[k] : mx;
            pc += spc[k];
        }
        out_min[blockIdx.x] = mn;
        out_max[blockIdx.x] = mx;
        out_pop[blockIdx.x] = pc;
    }
}

// ---------------------------------------------------------------- a sparse key column as ranks in its group dictionary
// out[row] = rank of the row's value among the column's sorted distinct values (Column::gdict, through its device map), at
// the narrowest width that holds the dictionary's size; `miss` (a digit no cell has) for a value the dictionary lacks --
// only a host-supplied dictionary can lack one -- so that the scan reports the row as outside the declared key space, as
// the probing kernels do.  Built once per (dictionary, table version): a group-by on the column then direct-maps on the
// derived column and runs the specialised row bodies instead of probing the dictionary per row in the plan interpreter.
__global__ __launch_bounds__(256) void k_rank_column(const void *__restrict__ col, int width, int64_t vbase, const uint32_t *__restrict__ valid,
                                                     const int64_t *__restrict__ dkeys, const int32_t *__restrict__ dranks, uint32_t dmask, int64_t n,
                                                     void *__restrict__ out, int ow, uint32_t miss) {
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
        uint32_t rank = 0;
        if (!valid || ((valid[row >> 5] >> (row & 31)) & 1u)) {
            const int64_t x = load_val(col, width, vbase, row);
            rank = miss;
            uint32_t h = dict_hash(x) & dmask;
            for (uint32_t probe = 0; probe <= dmask; probe++) {
                const int64_t kx = dkeys[h];
                if (kx == x) {
                    rank = (uint32_t)dranks[h];
                    break;
                }
                if (kx == kDictEmpty) break;
                h = (h + 1) & dmask;
            }
        }
        if (ow == 1) ((uint8_t *)out)[row] = (uint8_t)rank;
        else if (ow == 2) ((uint16_t *)out)[row] = (uint16_t)rank;
        else ((uint32_t *)out)[row] = rank;
    }
}
hipError_t launch_rank_column(const void *col, int width, int64_t vbase, const uint32_t *valid, const int64_t *dkeys, const int32_t *dranks, uint32_t dmask,
                              int64_t n, void *out, int ow, uint32_t miss, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_rank_column, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, col, width, vbase, valid, dkeys, dranks, dmask, n, out,
                       ow, miss);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the weight a row aggregates with
// aggregate.go:68,100-102: `weight` is declared once per FilterAndAggRecords call -- per block -- as 1, and a row sets it
// only when its weight column is populated: a row WITHOUT a weight aggregates with the weight of the last row before it
// in its block that had one (whether or not that row passed the filters), or with 1.  One workgroup per block: every
// thread finds the last populated row of its contiguous share, the shares' carries are resolved through LDS, and the
// second walk writes the weight in force at every row -- a dense int64 column the scan kernels read like any fully
// populated weight column.
__global__ __launch_bounds__(256) void k_weight_carry(const void *__restrict__ col, int width, int64_t vbase, const uint32_t *__restrict__ valid,
                                                      const Segment *__restrict__ blocks, int64_t *__restrict__ out) {
    const Segment b = blocks[blockIdx.x];
    __shared__ int64_t last_of[256];
    const int64_t per = (b.n + 255) / 256;
    const int64_t i0 = (int64_t)threadIdx.x * per < b.n ? (int64_t)threadIdx.x * per : b.n, i1 = i0 + per < b.n ? i0 + per : b.n;
    int64_t last = -1;
    for (int64_t i = i0; i < i1; i++) {
        const int64_t row = b.start + i;
        if (!valid || ((valid[row >> 5] >> (row & 31)) & 1u)) last = i;
    }
    last_of[threadIdx.x] = last;
    __syncthreads();
    int64_t carry = -1;
    for (int k = (int)threadIdx.x - 1; k >= 0 && carry < 0; k--) carry = last_of[k];
    int64_t w = carry >= 0 ? load_val(col, width, vbase, b.start + carry) : 1;
    for (int64_t i = i0; i < i1; i++) {
        const int64_t row = b.start + i;
        if (!valid || ((valid[row >> 5] >> (row & 31)) & 1u)) w = load_val(col, width, vbase, row);
        out[row] = w;
    }
}
hipError_t launch_weight_carry(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *d_blocks, int n_blocks, int64_t *out, hipStream_t st) {
    if (n_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_weight_carry, dim3((unsigned)n_blocks), dim3(256), 0, st, col, width, vbase, valid, d_blocks, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- distinct values of a column
// Inserts every populated value into an open-addressing set (capacity = mask + 1, empty =
// kDictEmpty).  Most probes hit an existing key with a plain load; only first sightings CAS.
// *n_distinct counts insertions; the host gives up when it exceeds the dictionary budget.
__global__ __launch_bounds__(256) void k_distinct(const void *__restrict__ col, int width, int64_t vbase,
                                                  const uint32_t *__restrict__ valid,
                                                  const Segment *__restrict__ blocks, int64_t *keys, uint32_t mask,
                                                  unsigned long long *n_distinct, unsigned long long limit) {
    const Segment b = blocks[blockIdx.x];
    for (int64_t i = threadIdx.x; i < b.n; i += blockDim.x) {
        const int64_t row = b.start + i;
        if (valid && !((valid[row >> 5] >> (row & 31)) & 1u)) continue;
        const int64_t x = load_val(col, width, vbase, row);
        if (x == kDictEmpty) continue;  // the sentinel itself cannot be stored; reported by the host
        uint32_t h = dict_hash(x) & mask;
        for (uint32_t probe = 0; probe <= mask; probe++) {
            long long cur = __hip_atomic_load((long long *)keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == x) break;
            if (cur == kDictEmpty) {
                long long expected = kDictEmpty;
                if (__hip_atomic_compare_exchange_strong((long long *)keys + h, &expected, (long long)x, __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_fetch_add(n_distinct, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (expected == x) break;
            }
            h = (h + 1) & mask;
        }
        if (__hip_atomic_load(n_distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > limit) return;
    }
}

hipError_t launch_distinct(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *blocks, int n_blocks,
                           int64_t *keys, uint32_t mask, unsigned long long *n_distinct, unsigned long long limit, hipStream_t st) {
    if (n_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_distinct, dim3(n_blocks), dim3(256), 0, st, col, width, vbase, valid, blocks, keys, mask, n_distinct, limit);
    return hipGetLastError();
}

// ---------------------------------------------------------------- compact storage
// dst = the same values at another width / base (sybl_table_compact and its inverse).  Rows are
// independent; unpopulated rows carry don't-care bits either way.
__global__ __launch_bounds__(256) void k_repack(const void *__restrict__ src, int sw, int64_t sbase, void *__restrict__ dst, int dw,
                                                int64_t dbase, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t u = (uint64_t)load_val(src, sw, sbase, i) - (uint64_t)dbase;
        switch (dw) {
        case 8: ((int64_t *)dst)[i] = (int64_t)u; break;
        case 4: ((uint32_t *)dst)[i] = (uint32_t)u; break;
        case 2: ((uint16_t *)dst)[i] = (uint16_t)u; break;
        default: ((uint8_t *)dst)[i] = (uint8_t)u; break;
        }
    }
}

hipError_t launch_repack(const void *src, int sw, int64_t sbase, void *dst, int dw, int64_t dbase, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_repack, dim3((unsigned)blocks), dim3(256), 0, st, src, sw, sbase, dst, dw, dbase, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- column decode (TableBlock load)
// unpackIntCol / unpackStrCol, column_store_io.go:493-609,690-780, as kernels: the gob stream is
// parsed on the host (varints are a byte-serial format), the compact decoded form -- per-bin
// value + delta-encoded ascending record ids, or a delta-encoded value array -- crosses PCIe,
// and the un-delta + scatter into the dense column happens here.

__device__ __forceinline__ uint32_t block_inclusive_scan_256(uint32_t v, uint32_t *wave_tot /*[4]*/, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return v + base;
}

// One workgroup per bin: record ids are ascending and (optionally) delta-encoded
// (delta_encode_col, column_store_io.go:21-30); every listed row gets the bin's value.  R: how the ids crossed PCIe --
// uint16 when the block has at most 65536 rows (every id and every delta then fits), else uint32.  T: the column's
// stored type -- canonical (int64 values / 32-bit dictionary ids, vbase 0) or, when the block goes straight into compact
// storage, the column's narrow unsigned offset from vbase.
template <typename T, typename R>
__device__ __forceinline__ void decode_bins_body(const R *__restrict__ recs, const int64_t *__restrict__ bin_off,
                                                 const int64_t *__restrict__ bin_val, int delta_encoded, int64_t vbase, T *__restrict__ col,
                                                 uint32_t *__restrict__ valid, uint32_t nrows, uint32_t bin, uint32_t *wave_tot /*[4]*/,
                                                 unsigned long long *chk_flags = nullptr) {
    // (R = 64-bit values: the GPU varint walk's columns, a [first, one-past-last) pair per bucket -- plan.h: DecodeBinsJob)
    constexpr bool kTok = sizeof(R) == 8;
    const int64_t b0 = kTok ? bin_off[2 * (size_t)bin] : bin_off[bin], b1 = kTok ? bin_off[2 * (size_t)bin + 1] : bin_off[bin + 1];
    bool stray = false;
    const T value = (T)((uint64_t)bin_val[bin] - (uint64_t)vbase);
    uint32_t carry = 0;
    for (int64_t base = b0; base < b1; base += 256) {
        const int64_t i = base + threadIdx.x;
        uint32_t d = i < b1 ? (uint32_t)recs[i] : 0u;
        if (kTok && i < b1) stray = stray || (uint64_t)recs[i] >= (uint64_t)nrows;  // (a delta or an id that large cannot be in the block)
        uint32_t r = d;
        if (delta_encoded) {
            uint32_t total;
            r = carry + block_inclusive_scan_256(d, wave_tot, total);
            carry += total;
        }
        if (i < b1 && r < nrows) {
            col[r] = value;
            if (valid) atomicOr(&valid[r >> 5], 1u << (r & 31));
        }
        if (kTok && i < b1 && r >= nrows) stray = true;
    }
    if (kTok && chk_flags && __any(stray) && (threadIdx.x & 63) == 0) atomicOr(chk_flags, (unsigned long long)kGobOutOfBounds);
}

template <typename T, typename R>
__global__ __launch_bounds__(256) void k_decode_bins(const R *__restrict__ recs, const int64_t *__restrict__ bin_off,
                                                     const int64_t *__restrict__ bin_val, int delta_encoded, int64_t vbase,
                                                     T *__restrict__ col, uint32_t *__restrict__ valid, uint32_t nrows) {
    __shared__ uint32_t wave_tot[4];
    decode_bins_body<T, R>(recs, bin_off, bin_val, delta_encoded, vbase, col, valid, nrows, blockIdx.x, wave_tot);
}

// Every bucket-encoded column of one block in ONE launch (blockIdx.y = column, blockIdx.x = bin; a column with fewer
// bins than the widest leaves its surplus workgroups idle): a block's decode was seven launches of a few microseconds
// of work each, and launching them -- not the work -- was a third of the loading thread's time per block.
template <typename R>
__device__ __forceinline__ void decode_bins_job(const DecodeBinsJob &J, uint32_t nrows, uint32_t bin, uint32_t *wave_tot) {
    switch (J.out_w) {
    case 1: decode_bins_body<uint8_t, R>((const R *)J.recs, J.bin_off, J.bin_val, J.delta, J.vbase, (uint8_t *)J.col, J.valid, nrows, bin, wave_tot, J.chk_flags); break;
    case 2: decode_bins_body<uint16_t, R>((const R *)J.recs, J.bin_off, J.bin_val, J.delta, J.vbase, (uint16_t *)J.col, J.valid, nrows, bin, wave_tot, J.chk_flags); break;
    case 4: decode_bins_body<uint32_t, R>((const R *)J.recs, J.bin_off, J.bin_val, J.delta, J.vbase, (uint32_t *)J.col, J.valid, nrows, bin, wave_tot, J.chk_flags); break;
    default: decode_bins_body<int64_t, R>((const R *)J.recs, J.bin_off, J.bin_val, J.delta, J.vbase, (int64_t *)J.col, J.valid, nrows, bin, wave_tot, J.chk_flags); break;
    }
}
__global__ __launch_bounds__(256) void k_decode_bins_multi(const DecodeBinsBatch B) {
    __shared__ uint32_t wave_tot[4];
    const DecodeBinsJob &J = B.job[blockIdx.y];
    if ((int32_t)blockIdx.x >= J.n_bins) return;
    if (J.rec_w == 2) decode_bins_job<uint16_t>(J, J.nrows, blockIdx.x, wave_tot);
    else if (J.rec_w == 8) decode_bins_job<unsigned long long>(J, J.nrows, blockIdx.x, wave_tot);
    else decode_bins_job<uint32_t>(J, J.nrows, blockIdx.x, wave_tot);
}

// Value-encoded int column: Values[r] is a delta from Values[r-1] (column_store_io.go:109-113,748-777).
// A block (<= 65536 rows in the reference) is split into segments of 1024 x kDeltaPerThread values, one workgroup each:
// the workgroup first sums the deltas AHEAD of its segment (a plain reduction: up to 56 values per lane), then scans its
// own 8192 values in one round -- lane totals through a wave scan, wave totals through LDS.  (Round 2 ran one workgroup
// per block: eight dependent rounds of two barriers each, 114 us per block and 78 % of the GPU time of a table load;
// the extra reads of the reduction are a few hundred KB per block.)  V: int32 when every stored value / delta of the
// block fits (the worker checked), else int64.  O: the column's stored type (see above).
constexpr int kDeltaPerThread = 8;
constexpr int64_t kDeltaSegment = 1024 * kDeltaPerThread;
template <typename V, typename O>
__device__ __forceinline__ void decode_delta_body(const V *__restrict__ deltas, int64_t n, int value_encoded, int64_t vbase, O *__restrict__ col,
                                                  uint32_t segment, int64_t *wave_tot /*[16]*/, unsigned long long *chk_flags = nullptr,
                                                  int64_t chk_min = 0, int64_t chk_max = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)segment * kDeltaSegment;
    int64_t carry = 0;  // the sum of everything ahead of this segment
    if (value_encoded && base > 0) {
        int64_t t = 0;
        for (int64_t i = threadIdx.x; i < base; i += 1024) t += (int64_t)deltas[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) wave_tot[wave] = t;
        __syncthreads();
        for (int w = 0; w < 16; w++) carry += wave_tot[w];
        __syncthreads();
    }
    const int64_t i0 = base + (int64_t)threadIdx.x * kDeltaPerThread;
    int64_t v[kDeltaPerThread];
#pragma unroll
    for (int j = 0; j < kDeltaPerThread; j++) v[j] = i0 + j < n ? (int64_t)deltas[i0 + j] : 0;
    if (value_encoded) {
#pragma unroll
        for (int j = 1; j < kDeltaPerThread; j++) v[j] += v[j - 1];  // the lane's own running sum
        int64_t t = v[kDeltaPerThread - 1];                           // inclusive scan of the lanes' totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int64_t u = __shfl_up(t, o, 64);
            if (lane >= o) t += u;
        }
        if (lane == 63) wave_tot[wave] = t;
        __syncthreads();
        int64_t pre = 0;
        for (int w = 0; w < 16; w++)
            if (w < wave) pre += wave_tot[w];
        const int64_t before = t - v[kDeltaPerThread - 1] + pre + carry;  // everything ahead of this lane's values
#pragma unroll
        for (int j = 0; j < kDeltaPerThread; j++) v[j] += before;
    }
#pragma unroll
    for (int j = 0; j < kDeltaPerThread; j++)
        if (i0 + j < n) col[i0 + j] = (O)((uint64_t)v[j] - (uint64_t)vbase);
    if (chk_flags) {
        // (the loader's GPU varint walk, gobgpu.hip: the block was placed by its info.db bounds before anyone had seen a value)
        bool out = false;
#pragma unroll
        for (int j = 0; j < kDeltaPerThread; j++) out = out || (i0 + j < n && (v[j] < chk_min || v[j] > chk_max));
        if (__any(out) && lane == 0) atomicOr(chk_flags, (unsigned long long)kGobOutOfBounds);
    }
}

template <typename V, typename O>
__global__ __launch_bounds__(1024) void k_decode_delta(const V *__restrict__ deltas, int64_t n, int value_encoded, int64_t vbase,
                                                       O *__restrict__ col) {
    __shared__ int64_t wave_tot[16];
    decode_delta_body<V, O>(deltas, n, value_encoded, vbase, col, blockIdx.x, wave_tot);
}

// every value-encoded int column of one block in one launch (see k_decode_bins_multi)
template <typename V>
__device__ __forceinline__ void decode_delta_job(const DecodeDeltaJob &J, uint32_t segment, int64_t *wave_tot) {
    switch (J.out_w) {
    case 1: decode_delta_body<V, uint8_t>((const V *)J.deltas, J.n, J.venc, J.vbase, (uint8_t *)J.col, segment, wave_tot, J.chk_flags, J.chk_min, J.chk_max); break;
    case 2: decode_delta_body<V, uint16_t>((const V *)J.deltas, J.n, J.venc, J.vbase, (uint16_t *)J.col, segment, wave_tot, J.chk_flags, J.chk_min, J.chk_max); break;
    case 4: decode_delta_body<V, uint32_t>((const V *)J.deltas, J.n, J.venc, J.vbase, (uint32_t *)J.col, segment, wave_tot, J.chk_flags, J.chk_min, J.chk_max); break;
    default: decode_delta_body<V, int64_t>((const V *)J.deltas, J.n, J.venc, J.vbase, (int64_t *)J.col, segment, wave_tot, J.chk_flags, J.chk_min, J.chk_max); break;
    }
}
// (blockIdx.y beyond the delta jobs: the bucket parsers of the GPU varint walk's bucket-encoded columns, gob_bins.h -- one
// workgroup each; like the delta jobs they wait for the walk only, so they share its successor launch)
__global__ __launch_bounds__(1024) void k_decode_delta_multi(const DecodeDeltaBatch B, const GobBinsBatch G) {
    __shared__ int64_t wave_tot[16];
    if ((int32_t)blockIdx.y >= B.n) {
        if (blockIdx.x == 0) gob_bins_body(G.job[blockIdx.y - B.n]);
        return;
    }
    const DecodeDeltaJob &J = B.job[blockIdx.y];
    if ((int64_t)blockIdx.x * kDeltaSegment >= J.n) return;
    if (J.val_w == 4) decode_delta_job<int32_t>(J, blockIdx.x, wave_tot);
    else decode_delta_job<int64_t>(J, blockIdx.x, wave_tot);
}

// Per-row block-local dictionary ids -> table-global ids (non-bucket str columns)
// (local and col may be the same array -- sybl_table_set_dict remaps resident ids in place -- so neither is restrict)
template <typename L>
__global__ __launch_bounds__(256) void k_remap_ids(const L *local, const int32_t *__restrict__ lut, int32_t n_lut, int64_t n,
                                                   int32_t *col) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t id = (int32_t)local[i];
    col[i] = (id >= 0 && id < n_lut) ? lut[id] : 0;
}

template <typename T>
static void decode_bins_t(const void *recs, int rec_width, const int64_t *bin_off, const int64_t *bin_val, int n_bins, int de, int64_t vbase,
                          void *col, uint32_t *valid, uint32_t nrows, hipStream_t st) {
    if (rec_width == 2) {
        hipLaunchKernelGGL((k_decode_bins<T, uint16_t>), dim3(n_bins), dim3(256), 0, st, (const uint16_t *)recs, bin_off, bin_val, de, vbase, (T *)col,
                           valid, nrows);
    } else {
        hipLaunchKernelGGL((k_decode_bins<T, uint32_t>), dim3(n_bins), dim3(256), 0, st, (const uint32_t *)recs, bin_off, bin_val, de, vbase, (T *)col,
                           valid, nrows);
    }
}

// rec_width: 2 (uint16 ids) or 4 (uint32 ids); out_width / vbase: the stored form of the destination (1, 2, 4 or 8 bytes;
// canonical int64: 8 / 0, canonical str ids: 4 / 0)
hipError_t launch_decode_bins(const void *recs, int rec_width, const int64_t *bin_off, const int64_t *bin_val, int n_bins,
                              bool delta_encoded, void *col, int out_width, int64_t vbase, uint32_t *valid, uint32_t nrows, hipStream_t st) {
    if (n_bins <= 0) return hipSuccess;
    const int de = delta_encoded ? 1 : 0;
    switch (out_width) {
    case 1: decode_bins_t<uint8_t>(recs, rec_width, bin_off, bin_val, n_bins, de, vbase, col, valid, nrows, st); break;
    case 2: decode_bins_t<uint16_t>(recs, rec_width, bin_off, bin_val, n_bins, de, vbase, col, valid, nrows, st); break;
    case 4: decode_bins_t<uint32_t>(recs, rec_width, bin_off, bin_val, n_bins, de, vbase, col, valid, nrows, st); break;
    default: decode_bins_t<int64_t>(recs, rec_width, bin_off, bin_val, n_bins, de, vbase, col, valid, nrows, st); break;
    }
    return hipGetLastError();
}

template <typename V>
static void decode_delta_t(const void *deltas, int64_t n, int ve, void *col, int out_width, int64_t vbase, hipStream_t st) {
    const dim3 grid((unsigned)((n + kDeltaSegment - 1) / kDeltaSegment));
    switch (out_width) {
    case 1: hipLaunchKernelGGL((k_decode_delta<V, uint8_t>), grid, dim3(1024), 0, st, (const V *)deltas, n, ve, vbase, (uint8_t *)col); break;
    case 2: hipLaunchKernelGGL((k_decode_delta<V, uint16_t>), grid, dim3(1024), 0, st, (const V *)deltas, n, ve, vbase, (uint16_t *)col); break;
    case 4: hipLaunchKernelGGL((k_decode_delta<V, uint32_t>), grid, dim3(1024), 0, st, (const V *)deltas, n, ve, vbase, (uint32_t *)col); break;
    default: hipLaunchKernelGGL((k_decode_delta<V, int64_t>), grid, dim3(1024), 0, st, (const V *)deltas, n, ve, vbase, (int64_t *)col); break;
    }
}

// val_width: 4 (int32) or 8 (int64)
hipError_t launch_decode_delta(const void *deltas, int val_width, int64_t n, bool value_encoded, void *col, int out_width, int64_t vbase,
                               hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (val_width == 4) decode_delta_t<int32_t>(deltas, n, value_encoded ? 1 : 0, col, out_width, vbase, st);
    else decode_delta_t<int64_t>(deltas, n, value_encoded ? 1 : 0, col, out_width, vbase, st);
    return hipGetLastError();
}

hipError_t launch_decode_bins_multi(const DecodeBinsBatch &B, hipStream_t st) {
    int max_bins = 0;
    for (int i = 0; i < B.n; i++) max_bins = B.job[i].n_bins > max_bins ? B.job[i].n_bins : max_bins;
    if (B.n <= 0 || max_bins <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode_bins_multi, dim3((unsigned)max_bins, (unsigned)B.n), dim3(256), 0, st, B);
    return hipGetLastError();
}

hipError_t launch_decode_delta_multi(const DecodeDeltaBatch &B, const GobBinsBatch &G, hipStream_t st) {
    int64_t max_n = 0;
    for (int i = 0; i < B.n; i++) max_n = B.job[i].n > max_n ? B.job[i].n : max_n;
    const int n_delta = max_n > 0 ? B.n : 0, n_gob = G.n > 0 ? G.n : 0;
    if (n_delta <= 0 && n_gob <= 0) return hipSuccess;
    if (n_delta == 0) {  // (no delta job with values: only the bucket parsers run)
        DecodeDeltaBatch none;
        none.n = 0;
        hipLaunchKernelGGL(k_decode_delta_multi, dim3(1u, (unsigned)n_gob), dim3(1024), 0, st, none, G);
        return hipGetLastError();
    }
    const unsigned segments = (unsigned)((max_n + kDeltaSegment - 1) / kDeltaSegment);
    hipLaunchKernelGGL(k_decode_delta_multi, dim3(segments, (unsigned)(B.n + n_gob)), dim3(1024), 0, st, B, G);
    return hipGetLastError();
}

// local_width: 2 (uint16 block-local ids) or 4 (int32)
hipError_t launch_remap_ids(const void *local, int local_width, const int32_t *lut, int32_t n_lut, int64_t n, int32_t *col, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (local_width == 2) {
        hipLaunchKernelGGL((k_remap_ids<uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t *)local, lut, n_lut, n, col);
    } else {
        hipLaunchKernelGGL((k_remap_ids<int32_t>), dim3(nb), dim3(256), 0, st, (const int32_t *)local, lut, n_lut, n, col);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- partitioned histograms
// k_part_hist: one workgroup owns one partition = kPartCells (cell, agg) pairs.  Their bucket arrays and exact sums of
// v - h.Min live in LDS (the bucket divide is done here, not in k_emit); every record costs two LDS atomics (three when a
// maximum is tracked); the results are written with plain stores (each pair has exactly one owner), so the
// [cell][agg][bucket] table, Count and sum(v) come out deterministic and atomics-free in HBM.
//
// The bucket counters are 16 bits wide, two to an LDS word (64 pairs x 1002 buckets x 2 B = 125 KB: what lets a
// partition be 64 pairs and k_emit stage two chunks per bin, scan_fast.h).  A counter that wraps is not lost: the
// atomic add returns the word as it was, so the one lane whose add wrapped a field sees it (field == 0xFFFF before the
// add) and logs the event; k_part_fix adds the logged 65536s to the table afterwards.  A wrap of the low field also
// carries into the high field of the same word: the same lane logs a -1 for that bucket, and a +65536 if the carry in
// turn wrapped it (old word == 0xFFFFFFFF).  Every wrap of every field is seen by exactly one add, so
//   true count = field + 65536 x (logged wraps) - (logged carries in)
// holds exactly whatever the order.  At most 3 entries per 65536 records: the log is sized for that.
constexpr int kPartSumRep = 8;   // replicas of the per-pair value sums (lanes of a wave hit only 64 pairs)
constexpr int kPartUnroll = 4;   // 16-byte record loads per lane in flight, twice (current + next)
constexpr uint32_t kWrapPlus = 0u, kWrapMinusOne = 1u;  // log entry kinds: +65536 / -1

__device__ __noinline__ void part_log_wrap(const PartHistPlan &P, uint32_t pair, uint32_t field, uint32_t old, uint32_t n_fields) {
    // (rare path) `field` was 0xFFFF before this lane's add; old >> 16 = an even field's neighbour then.  Field f holds
    // bucket f - 1; field 0 is where padding records are counted (k_part_hist): its own wraps mean nothing, its carries do
    uint32_t kinds[3], buckets[3], n = 0;
    if (field > 0) buckets[n] = field - 1, kinds[n++] = kWrapPlus;
    if ((field & 1u) == 0 && field + 1 < n_fields) {
        buckets[n] = field, kinds[n++] = kWrapMinusOne;                         // the carry into the high field
        if ((old >> 16) == 0xFFFFu) buckets[n] = field, kinds[n++] = kWrapPlus;  // ... which wrapped it
    }
    if (!n) return;
    const uint32_t at = __hip_atomic_fetch_add(P.wrap_log, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (at + n > P.wrap_cap) {
        __hip_atomic_fetch_add(P.sum_out + kHdrPartOverflow, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (uint32_t k = 0; k < n; k++) {
        P.wrap_log[2 + 2 * (at + k)] = pair;
        P.wrap_log[3 + 2 * (at + k)] = buckets[k] | kinds[k] << 16;
    }
}

// (rare path) a value beyond the last bucket: remembered exactly -- n, sum(o), sum(o^2) in four 32-bit limbs, as the GEN row
// body does (scan_fast.h: fast_accumulate) -- per pair in LDS (round 3: six device-scope atomics per outlier straight into
// the cell's fields; 1 % outliers of 1e9 values were 6e7 of them) and added to the cell's outlier fields once per item; the
// value itself goes to the log when one is kept
struct PartOutLds {
    uint32_t *n;               // [kPartCells]
    unsigned long long *sum;   // [kPartCells]
    unsigned long long *sq;    // [kPartCells][4]
};
__device__ __noinline__ void part_outlier(const PartHistPlan &P, const PartOutLds O, uint32_t local, uint32_t pair, uint32_t n32) {
    const uint32_t na = (uint32_t)P.n_aggs, cell = pair / na, a = pair % na;
    const int64_t x = (int64_t)((uint64_t)P.hmin[a] + (uint64_t)n32);
    if (P.f_out[a] < 0) {  // declared bounds violated: reported by finalize
        gadd(P.sum_out + kHdrOverflow, 1);
        return;
    }
    const unsigned __int128 sq = (unsigned __int128)((__int128)x * (__int128)x);
    __hip_atomic_fetch_add(O.n + local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(O.sum + local, (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(O.sq + local * 4u, (unsigned long long)(uint64_t)(sq & 0xFFFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(O.sq + local * 4u + 1, (unsigned long long)(uint64_t)((sq >> 32) & 0xFFFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(O.sq + local * 4u + 2, (unsigned long long)(uint64_t)((sq >> 64) & 0xFFFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(O.sq + local * 4u + 3, (unsigned long long)(uint64_t)(sq >> 96), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (P.out_log) log_outlier(P.out_log, P.out_cap, (int64_t)cell, P.agg0 + (int)a, x);
}

// Round 4: the kernel is PERSISTENT -- one workgroup per compute unit claims (partition, share) items from a counter
// until none is left -- and its waves claim the regions of an item from an LDS cursor instead of owning every 16th.
// The phase trace of the one-workgroup-per-item version (SYBL_PARTHIST_TRACE, config 4: 1024 workgroups over 256 CUs)
// showed why it ran at 75 % occupancy: per item 18 us of launch + zeroing + region table, 267 us of walk in which the
// waves of a workgroup finished up to 98 us apart (54 us of idle wave time on average: older waves win the issue
// arbitration), 28 us for 64 serial count reductions, 20 us of write-out and 9 us until the next workgroup started.
// The epilogue is one pass now: the wave that owns a pair reads its counters once, stores the buckets and sums them.
template <int NA, bool TRACK_MAX, bool OUT>
__global__ __launch_bounds__(kWgThreads) void k_part_hist(const PartHistPlan P) {
    extern __shared__ uint32_t plds[];
    const uint32_t tid = threadIdx.x;
    // A record is (local pair) << 26 | (v - h.Min + BucketSize): its quotient by BucketSize is the bucket + 1, the FIELD
    // the record counts in.  Field 0 of every pair counts nothing real: the all-zero word -- a region's padding, and what
    // the range-checked loads return past a region's end -- lands in field 0 of pair 0 and adds 0 to its sum, so the walk
    // needs no "is this lane's record real" test at all (round 3 spent ~5 of its ~30 vector instructions per record on
    // it: bounds compare, sentinel compare, two address selects towards per-lane scratch words).
    const uint32_t nv = (uint32_t)P.nv_max, nw = (nv + 2u) >> 1;  // buckets; words per pair (nv + 1 fields, two to a word)
    const uint32_t split = (uint32_t)P.split, n_items = (uint32_t)P.n_parts * split;
    const uint32_t n_reg_max = ((uint32_t)P.n_wg + split - 1u) / split;
    uint32_t *hist = plds;                                       // [kPartCells][nw] two 16-bit counters per word
    unsigned long long *sum = (unsigned long long *)(hist + kPartCells * nw);               // [kPartCells][kPartSumRep]
    long long *vmax = (long long *)(sum + kPartSumRep * kPartCells);                        // [kPartCells]
    uint2 *regions = (uint2 *)(vmax + kPartCells);                                          // [n_reg_max] {first chunk, pieces}
    const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *ctl = (uint32_t *)(regions + n_reg_max);  // [1] region cursor, [2], [3] the item claimed / the next one
    PartOutLds O;                                       // (OUT) the pairs' outlier sums, behind the cursors
    O.sum = (unsigned long long *)(ctl + 4);
    O.sq = O.sum + kPartCells;
    O.n = (uint32_t *)(O.sq + 4 * kPartCells);
    // (diagnostic, SYBL_PARTHIST_TRACE: the 100 MHz wall clock at the phase boundaries of every item; word 0 claimed, 1 tables
    // zeroed and regions read, 2 records walked, 4 table written, 5 the compute unit, 16.. when each wave finished its walk)
    uint32_t item = 0;
    auto stamp = [&](int k) {
        if (P.trace && tid == 0) P.trace[(size_t)item * kPartTraceWords + k] = wall_clock64();
    };
    unsigned long long *my_sum = sum + (tid & (kPartSumRep - 1));  // [pair][replica]: the replicas of a pair in different banks
    // pair0 is a multiple of kPartCells, so with two aggregations a pair's aggregation is the low bit of `local`
    // (NA, and whether any maximum is tracked, are template parameters: as run-time values they cost a chain of
    // selects per record in a kernel whose vector ALUs are ~60 % busy)
    constexpr bool two = NA == 2;
    const uint32_t bs0 = (uint32_t)P.bucket_size[0], bs1 = (uint32_t)P.bucket_size[two ? 1 : 0];
    const uint32_t nv0 = (uint32_t)P.n_values[0], nv1 = (uint32_t)P.n_values[two ? 1 : 0];
    const double inv0 = P.pinv_bucket[0], inv1 = P.pinv_bucket[two ? 1 : 0];
    const uint32_t *recs = P.recs;
    const uint32_t nb1 = ((uint32_t)P.n_parts << P.sub_shift) + 1u;
    const uint32_t total_pairs = (uint32_t)P.n_cells * (uint32_t)NA;
    int64_t *F = P.sum_out + kHeaderWords;
    typedef unsigned int rec4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t kBatch = 64u * kPartUnroll;  // pieces per wave and batch

    // (an item is claimed one item ahead -- behind the previous item's region table, its device-scope round trip hidden
    // under that item's walk -- into ctl[2 | 3] by turns)
    if (tid == 0) ctl[2] = __hip_atomic_fetch_add(P.wrap_log + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t turn = 0;; turn ^= 1u) {
        // ---- the tables start from zero
        if (tid == 0) ctl[1] = kWgThreads / 64;  // (every wave starts with the region of its own number)
        {
            typedef unsigned int z4 __attribute__((ext_vector_type(4)));
            const z4 zero = {0u, 0u, 0u, 0u};
            for (uint32_t i = tid; i < kPartCells * nw / 4u; i += kWgThreads) ((z4 *)hist)[i] = zero;  // (kPartCells x nw words: a multiple of 4)
        }
        if (tid < kPartSumRep * kPartCells) sum[tid] = 0;
        if (tid < kPartCells) vmax[tid] = INT64_MIN;
        if (OUT && tid < 6 * kPartCells) O.sum[tid] = 0;  // (sum | sq | n: 5 x 8 + 4 bytes per pair, contiguous)
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(ctl[2u + turn]);
        if (item >= n_items) break;
        stamp(0);
        // (wave-uniform by construction; said so explicitly: the divide goes through the vector ALU, and with these in
        // vector registers every region loop and record load below became a divergent loop / a waterfall loop)
        const uint32_t part = __builtin_amdgcn_readfirstlane(item / split), sub = __builtin_amdgcn_readfirstlane(item % split);
        const uint32_t pair0 = part * kPartCells;
        const uint32_t b_lo = part << P.sub_shift, b_hi = (part + 1u) << P.sub_shift;
        const uint32_t n_reg = __builtin_amdgcn_readfirstlane((uint32_t)P.n_wg > sub ? ((uint32_t)P.n_wg - sub + split - 1u) / split : 0u);
        // The item's records: one region per scanning workgroup w (the partition's sub-bins are neighbours in w's
        // output: chunks boff[w][part << ss] .. boff[w][(part + 1) << ss] behind wbase[w]).  `split` workgroups share a
        // partition by taking every split-th region (boff is [workgroup][bin]: a strided read, once per item).
        for (uint32_t k = tid; k < n_reg; k += kWgThreads) {
            const uint32_t w = sub + k * split;
            const uint32_t *bo = P.boff + (size_t)w * nb1;
            const uint32_t lo = bo[b_lo], hi = bo[b_hi];
            regions[k] = make_uint2(P.wbase[w] + lo, (hi - lo) * (kEmitChunk / 4u));
        }
        uint32_t claimed = 0;  // (stored behind the walk: nobody waits for the round trip at a barrier)
        if (tid == kWgThreads - 1) claimed = __hip_atomic_fetch_add(P.wrap_log + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stamp(1);

        auto field_of = [&](uint32_t rec, uint32_t &local, uint32_t &val) -> uint32_t {
            val = rec & ((1u << kRecValueBits) - 1);  // v - h.Min + BucketSize
            local = rec >> kRecValueBits;
            const uint32_t a = two ? (local & 1u) : 0u;
            const uint32_t bs = two && a ? bs1 : bs0;
            const double inv = two && a ? inv1 : inv0;
            // floor(val / BucketSize) = bucket + 1, hist_basic.go:130-150: the estimate is never above the quotient and at most
            // one short of it (scan_packed.h: packed_udiv); field <= 2^10 and BucketSize < 2^24 (planner): a 24-bit product
            uint32_t f = (uint32_t)((double)val * inv);
            if (val - __umul24(f, bs) >= bs) f += 1;
            return f;
        };
        // One record: returns the word as it was before the add, shifted so that the field's own half is the low one (an
        // even field's neighbour is then the high half).  No branch per record, and the sixteen adds of a batch are in
        // flight together: their results are looked at once.
        auto add_record = [&](uint32_t rec) -> uint32_t {
            uint32_t local, val;
            uint32_t f = field_of(rec, local, val);
            if (OUT) {
                // Outlier (hist_basic.go:132-135): clipped into the last bucket and remembered (part_outlier)
                const uint32_t nva = two && (local & 1u) ? nv1 : nv0;
                if (f > nva) {
                    part_outlier(P, O, local, pair0 + local, val - (two && (local & 1u) ? bs1 : bs0));
                    f = nva;
                }
            }
            const uint32_t sh = (f & 1u) << 4;
            const uint32_t old = __hip_atomic_fetch_add(hist + __umul24(local, nw) + (f >> 1), 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(my_sum + local * kPartSumRep, (unsigned long long)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (TRACK_MAX && rec != kRecSentinel) {
                const uint32_t a = two ? (local & 1u) : 0u;
                if (P.m_max[a] >= 0) {
                    const long long v = (long long)((unsigned long long)P.hmin[a] + (unsigned long long)(val - (two && a ? bs1 : bs0)));
                    if (v > vmax[local]) __hip_atomic_fetch_max(vmax + local, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            return old >> sh;
        };
        auto wrapped = [](uint32_t seen) { return (seen & 0xFFFFu) == 0xFFFFu; };
        auto log_wrap = [&](uint32_t rec, uint32_t seen) {  // (rare)
            uint32_t local, val;
            uint32_t f = field_of(rec, local, val);
            const uint32_t nva = two && (local & 1u) ? nv1 : nv0;
            if (OUT) f = f > nva ? nva : f;  // (an outlier was counted in the last bucket)
            // (the aggregation's own field count: the high half next to an even last field belongs to nobody -- a carry
            // into it is neither logged nor, below, counted)
            part_log_wrap(P, pair0 + local, f, (f & 1u) ? 0u : seen, nva + 1u);
        };

        // ---- the walk: every wave takes regions from the cursor (its first one is its own number), 16-byte pieces (four
        // records) per lane and kPartUnroll loads per lane in flight twice over (current + next batch) -- raw buffer loads
        // whose descriptor is the region, so nothing depends on a lane's bounds when the loads are issued.
        uint32_t r = wave, i0 = 0, n4 = 0, c0 = 0;      // region, first piece of the next batch, pieces, first chunk (wave-uniform)
        auto open_region = [&]() {
            // skips empty regions; n4 == 0 afterwards: no region left
            n4 = 0;
            while (r < n_reg) {
                const uint2 g = regions[r];
                const uint32_t pieces = __builtin_amdgcn_readfirstlane(g.y);
                if (pieces) {
                    c0 = __builtin_amdgcn_readfirstlane(g.x);
                    n4 = pieces;
                    i0 = 0;
                    return;
                }
                uint32_t nxt_r = 0;
                if (lane == 0) nxt_r = __hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                r = __builtin_amdgcn_readfirstlane(nxt_r);
            }
        };
        auto issue = [&](rec4 (&d)[kPartUnroll], uint32_t &first, uint32_t &pieces) {
            const uint64_t at = (uint64_t)(recs + (size_t)c0 * kEmitChunk);  // (wave-uniform: kept in scalar registers)
            const uint32_t at_lo = __builtin_amdgcn_readfirstlane((uint32_t)at), at_hi = __builtin_amdgcn_readfirstlane((uint32_t)(at >> 32));
            const uint64_t at_s = (uint64_t)at_lo | (uint64_t)at_hi << 32;  // (the builtin returns int: through uint32_t, no sign extension)
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void *)at_s, 0, (int)__builtin_amdgcn_readfirstlane(n4 * 16u), (int)0x00020000);
#pragma unroll
            for (int u = 0; u < kPartUnroll; u++) d[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((i0 + (uint32_t)u * 64u + lane) * 16u), 0, 2);
            first = i0;
            pieces = n4;
            if (n4) {
                i0 += kBatch;
                if (i0 >= n4) {
                    uint32_t nxt_r = 0;
                    if (lane == 0) nxt_r = __hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    r = __builtin_amdgcn_readfirstlane(nxt_r);
                    open_region();
                }
            }
        };
        rec4 cur[kPartUnroll], nxt[kPartUnroll];
        uint32_t cur_first, cur_n, nxt_first, nxt_n;
        open_region();
        issue(cur, cur_first, cur_n);
        while (cur_n) {
            issue(nxt, nxt_first, nxt_n);
            uint32_t seen[kPartUnroll][4];
            if (cur_first + kBatch <= cur_n || P.tail_mode == 0) {
#pragma unroll
                for (int u = 0; u < kPartUnroll; u++) {
                    seen[u][0] = add_record(cur[u].x);
                    seen[u][1] = add_record(cur[u].y);
                    seen[u][2] = add_record(cur[u].z);
                    seen[u][3] = add_record(cur[u].w);
                }
            } else {
                // a region's last batch: the lanes past its end hold zeros, which would all add to ONE word (field 0 of pair
                // 0) -- same-address LDS atomics of a wave are carried out one after the other
#pragma unroll
                for (int u = 0; u < kPartUnroll; u++) {
                    seen[u][0] = seen[u][1] = seen[u][2] = seen[u][3] = 0u;
                    if (cur_first + (uint32_t)u * 64u + lane < cur_n) {
                        seen[u][0] = add_record(cur[u].x);
                        seen[u][1] = add_record(cur[u].y);
                        seen[u][2] = add_record(cur[u].z);
                        seen[u][3] = add_record(cur[u].w);
                    }
                }
            }
            bool any = false;
#pragma unroll
            for (int u = 0; u < kPartUnroll; u++)
                any = any || wrapped(seen[u][0]) || wrapped(seen[u][1]) || wrapped(seen[u][2]) || wrapped(seen[u][3]);
            if (any) {
#pragma unroll
                for (int u = 0; u < kPartUnroll; u++) {
                    if (wrapped(seen[u][0])) log_wrap(cur[u].x, seen[u][0]);
                    if (wrapped(seen[u][1])) log_wrap(cur[u].y, seen[u][1]);
                    if (wrapped(seen[u][2])) log_wrap(cur[u].z, seen[u][2]);
                    if (wrapped(seen[u][3])) log_wrap(cur[u].w, seen[u][3]);
                }
            }
#pragma unroll
            for (int u = 0; u < kPartUnroll; u++) cur[u] = nxt[u];
            cur_first = nxt_first;
            cur_n = nxt_n;
        }
        if (P.trace && lane == 0) P.trace[(size_t)item * kPartTraceWords + 16 + wave] = wall_clock64();
        if (tid == kWgThreads - 1) ctl[3u - turn] = claimed;
        __syncthreads();
        stamp(2);

        // ---- the epilogue: the wave that owns a pair (every 16th) reads the pair's counters once -- lanes own consecutive
        // buckets, so a store instruction writes 512 contiguous bytes -- stores them and sums them up: a pair's count is the
        // sum of its buckets as the fields hold them (k_part_fix adds what the wrap log says); sum(v) = sum(v - h.Min) +
        // count x h.Min.
        for (uint32_t l = wave; l < (uint32_t)kPartCells; l += kWgThreads / 64) {
            const uint32_t pair = pair0 + l;
            if (pair >= total_pairs) break;
            const uint32_t cell = pair / (uint32_t)NA, a = pair % (uint32_t)NA;
            const uint32_t nva = (uint32_t)P.n_values[a];
            int64_t *h = P.sum_out + P.hist_off + (int64_t)cell * P.hist_stride + P.hist_agg_off[a];
            const uint32_t *hl = hist + l * nw;
            uint32_t n = 0;
            for (uint32_t b = lane; b < nva; b += 64u) {
                const uint32_t x = (hl[(b + 1u) >> 1] >> (((b + 1u) & 1u) << 4)) & 0xFFFFu;  // bucket b is field b + 1
                n += x;
                if (split == 1) h[b] = (int64_t)x;          // sole owner of the pair: plain stores, nothing to zero beforehand
                else if (x) gadd(h + b, (int64_t)x);         // `split` workgroups share the pair: into the zeroed table
            }
            unsigned long long vs = lane < (uint32_t)kPartSumRep ? sum[l * kPartSumRep + lane] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                n += __shfl_xor(n, o, 64);
                vs += __shfl_xor(vs, o, 64);
            }
            // every aggregation accepts every row here (planner: no rejects / missing values), so
            // Result.Count of the cell is the count of any of its aggregations
            // (the records carry v - h.Min + BucketSize)
            const int64_t vsum = (int64_t)(vs + (unsigned long long)n * ((unsigned long long)P.hmin[a] - (unsigned long long)P.bucket_size[a]));
            if (OUT && lane == 0 && P.f_out[a] >= 0 && O.n[l] != 0) {
                // (the cell's outlier fields start from zero with every scan and are added to: shares of a pair, and passes)
                int64_t *Fo = F + (int64_t)P.f_out[a] * P.n_cells + cell;
                gadd(Fo, (int64_t)O.n[l]);
                gadd(Fo + P.n_cells, (int64_t)O.sum[l]);
                for (int k = 0; k < 4; k++) gadd(Fo + (int64_t)(2 + k) * P.n_cells, (int64_t)O.sq[l * 4u + (uint32_t)k]);
            }
            if (lane == 0) {
                if (split == 1) {
                    if (a == 0 && !P.no_count) F[cell] = (int64_t)n;
                    F[(int64_t)P.f_sum[a] * P.n_cells + cell] = vsum;
                    if (P.m_max[a] >= 0) P.max_out[(int64_t)P.m_max[a] * P.n_cells + cell] = vmax[l];
                } else if (n) {
                    if (a == 0 && !P.no_count) gadd(F + cell, (int64_t)n);
                    gadd(F + (int64_t)P.f_sum[a] * P.n_cells + cell, vsum);
                    if (P.m_max[a] >= 0)
                        __hip_atomic_fetch_max(P.max_out + (int64_t)P.m_max[a] * P.n_cells + cell, (int64_t)vmax[l], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        stamp(4);
        if (P.trace && tid == 0) {
            uint32_t hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            P.trace[(size_t)item * kPartTraceWords + 5] = (unsigned long long)hwid | (unsigned long long)xcc << 32;
        }
        __syncthreads();  // (the tables are zeroed for the next item)
    }
}

// k_part_fix: the wraps k_part_hist logged (see there), applied to the finished table: +65536 (or -1) on the bucket,
// and what follows from it for the pair's count -- Result.Count of the cell (taken from aggregation 0) and sum(v) =
// sum(v - h.Min) + count x h.Min.  Runs behind k_part_hist on the same stream; the log is empty unless some
// (group, bucket) of one partition pass holds 65536 or more values.
__global__ __launch_bounds__(256) void k_part_fix(const PartHistPlan P) {
    const uint32_t n = min(P.wrap_log[0], P.wrap_cap);
    int64_t *F = P.sum_out + kHeaderWords;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pair = P.wrap_log[2 + 2 * i], e = P.wrap_log[3 + 2 * i];
        const uint32_t bucket = e & 0xFFFFu, cell = pair / (uint32_t)P.n_aggs, a = pair % (uint32_t)P.n_aggs;
        const int64_t delta = (e >> 16) == kWrapPlus ? 65536 : -1;
        gadd(P.sum_out + P.hist_off + (int64_t)cell * P.hist_stride + P.hist_agg_off[a] + bucket, delta);
        if (a == 0 && !P.no_count) gadd(F + cell, delta);
        // (k_part_hist's sum of the records' value parts is exact; what was off is the count its h.Min - BucketSize went in with)
        gadd(F + (int64_t)P.f_sum[a] * P.n_cells + cell, delta * (P.hmin[a] - P.bucket_size[a]));
    }
}

hipError_t launch_part_fix(const PartHistPlan &P, hipStream_t st) {
    hipLaunchKernelGGL(k_part_fix, dim3(64), dim3(256), 0, st, P);
    return hipGetLastError();
}

template <int NA, bool TRACK_MAX>
static hipError_t part_hist_launch(const PartHistPlan &P, size_t lds, hipStream_t st) {
    bool out = false;  // a column whose bounds let a value land beyond the last bucket
    for (int a = 0; a < NA; a++) out = out || P.f_out[a] >= 0;
    auto k = out ? k_part_hist<NA, TRACK_MAX, true> : k_part_hist<NA, TRACK_MAX, false>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // persistent: one workgroup per compute unit (its ~135 KB of LDS would not let two share one anyway)
    const int items = P.n_parts * P.split;
    hipLaunchKernelGGL(k, dim3(std::min(items, std::max(P.n_cus, 1))), dim3(kWgThreads), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_part_hist(const PartHistPlan &P, hipStream_t st) {
    if (P.n_parts <= 0) return hipSuccess;
    const uint32_t n_reg = ((uint32_t)P.n_wg + (uint32_t)P.split - 1u) / (uint32_t)P.split;
    size_t lds = (size_t)kPartCells * ((P.nv_max + 2) / 2) * 4 + (size_t)kPartCells * (kPartSumRep + 1) * 8 + (size_t)n_reg * 8 + 16 +  // + item / region cursors
                 (size_t)kPartCells * 48;  // + the pairs' outlier sums (OUT)
    const bool track_max = P.m_max[0] >= 0 || (P.n_aggs > 1 && P.m_max[1] >= 0);
    if (P.n_aggs == 1) return track_max ? part_hist_launch<1, true>(P, lds, st) : part_hist_launch<1, false>(P, lds, st);
    if (P.n_aggs == 2) return track_max ? part_hist_launch<2, true>(P, lds, st) : part_hist_launch<2, false>(P, lds, st);
    return hipErrorInvalidValue;
}

// k_outlog_gather: the staging stripes of the outlier log (outlog.h) closed up into the query's dense log, stripe after
// stripe, and the number of records appended into the header.  A stripe's cursor may stand beyond its share (the lanes
// that found it filling up went on to the next stripe: outlog.h); its second word is non-zero when some record found no
// place in kOutSpill stripes.  The header then says "more than the capacity" (the printers treat the values as
// unavailable, as when the whole log overflowed).
__global__ __launch_bounds__(256) void k_outlog_gather(const int64_t *__restrict__ stage, int64_t cap, int64_t *__restrict__ log, int64_t *header) {
    const int64_t per = cap / kOutStripes;
    const int s = blockIdx.x;
    int64_t before = 0, kept = 0, dropped = 0, mine = 0;
    for (int k = 0; k < kOutStripes; k++) {
        const int64_t c = stage[(size_t)k * kOutCursorWords];
        const int64_t n = c < per ? c : per;
        dropped += stage[(size_t)k * kOutCursorWords + 1];
        kept += n;
        before += k < s ? n : 0;
        mine = k == s ? n : mine;
    }
    const int64_t *src = stage + (size_t)kOutStripes * kOutCursorWords + (size_t)s * (size_t)per * kOutLogWords;
    int64_t *dst = log + before * kOutLogWords;
    for (int64_t i = threadIdx.x; i < mine * kOutLogWords; i += blockDim.x) dst[i] = src[i];
    if (s == 0 && threadIdx.x == 0) header[kHdrOutLog] = dropped != 0 ? cap + 1 : kept;
}
hipError_t launch_outlog_gather(const int64_t *stage, int64_t cap, int64_t *log, int64_t *header, hipStream_t st) {
    hipLaunchKernelGGL(k_outlog_gather, dim3(kOutStripes), dim3(256), 0, st, stage, cap, log, header);
    return hipGetLastError();
}

// Bucket tables on their way through a reduce-scatter (rccl.cpp): int64 counters narrowed to int32 when the ranks'
// rows together stay below 2^31 (half the bytes over xGMI), and the reduced slice widened again in place.
__global__ __launch_bounds__(256) void k_pack32(const int64_t *__restrict__ src, int32_t *__restrict__ dst, int64_t n) {
    // (src = d_sum + hist_off is only 8-byte aligned -- hist_off may be odd, d_sum may be caller-bound: two 8-byte loads)
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
        if (i + 1 < n) {
            const int64_t v0 = src[i], v1 = src[i + 1];
            *(int2 *)(dst + i) = make_int2((int32_t)v0, (int32_t)v1);
        } else {
            dst[i] = (int32_t)src[i];
        }
    }
}
__global__ __launch_bounds__(256) void k_unpack32(const int32_t *__restrict__ src, int64_t *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int64_t)src[i];
}
hipError_t launch_pack32(const int64_t *src, int32_t *dst, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack32, dim3(2048), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_unpack32(const int32_t *src, int64_t *dst, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack32, dim3(2048), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- histogram summaries (finalize)
// With tens of thousands of groups the [cell][agg][bucket] table is hundreds of MB: copying it to
// the host and walking it there costs more than the scan.  These kernels derive what the result
// rows need -- GetPercentiles (hist_basic.go:153-183), the bucket moments of GetStdDev
// (:192-219) and the Cumulative bucket arrays (aggregate.go:422-438) -- where the table lives.
// One wave per (cell, aggregation); lanes own consecutive buckets, so the bucket array is read with
// coalesced 512-byte wave loads (a thread per pair walking its own 8 KB row ran at 0.33 TB/s).
// GetPercentiles in the reference walks the buckets k in order with the running count c(k),
//   p(k) = clamp(100 c(k) / Count, 0, 100);  for ip in [p(k-1), p(k)]: out[ip] = k * BucketSize + Min (ip < 100);
//   if p(k) < 100: out[p(k)] = k     -- overwritten by the next bucket's range, which starts at p(k)
// so slot ip ends up with the value of the FIRST bucket whose p(k) exceeds ip: bucket k owns the slots
// [p(k-1), p(k)) and those ranges are disjoint -- every lane writes its own.  What is left when the
// last bucket stops short of 100 (Count larger than the bucket total) is the last iteration's
// out[p] = k on slot p(n-1), and untouched zeros above it.
__global__ __launch_bounds__(256) void k_hist_summary(const HistSummaryPlan S) {
    const int64_t pair = S.cell0 * S.n_aggs + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= S.cell1 * S.n_aggs) return;
    const uint32_t lane = threadIdx.x & 63;
    const int64_t cell = pair / S.n_aggs;
    const int a = (int)(pair - cell * S.n_aggs);
    const int64_t *v = S.H + cell * S.hist_stride + S.agg_off[a];
    const int64_t count = S.F[(int64_t)S.f_cnt[a] * S.n_cells + cell];
    int64_t *out = S.pct + pair * 100;
    const int64_t bs = S.bucket_size[a], hmin = S.hmin[a], nvals = S.n_values[a];
    int64_t sb = 0, sb2 = 0, carry = 0, carry_p = 0;
    // (loading the whole bucket array into registers first -- sixteen wave loads in flight -- measured slower: 0.33 ms against
    // 0.28 for config 4's 65 536 arrays; the next chunk is requested while this one is scanned instead)
    int64_t x_next = lane < nvals ? v[lane] : 0;
    for (int64_t k0 = 0; k0 < nvals; k0 += 64) {
        const int64_t k = k0 + lane;
        const int64_t x = x_next;
        x_next = k + 64 < nvals ? v[k + 64] : 0;
        sb += k * x;
        sb2 += k * k * x;
        if (count == 0) continue;
        // inclusive scan of x over the wave
        int64_t c = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(c, o, 64);
            if ((int)lane >= o) c += y;
        }
        c += carry;
        int64_t p = (100 * c) / count;
        p = p < 0 ? 0 : (p > 100 ? 100 : p);
        int64_t pp = __shfl_up(p, 1, 64);
        if (lane == 0) pp = carry_p;
        if (k < nvals) {
            const int64_t val = k * bs + hmin;
            for (int64_t ip = pp; ip < p; ip++) out[ip] = val;  // (p <= 100: ip < 100)
            if (k == nvals - 1 && p < 100) out[p] = k;
        }
        carry = __shfl(c, 63, 64);
        carry_p = __shfl(p, 63, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sb += __shfl_xor(sb, o, 64);
        sb2 += __shfl_xor(sb2, o, 64);
    }
    if (lane == 0) {
        S.mom[pair * 2] = sb;
        S.mom[pair * 2 + 1] = sb2;
    }
}

// total[w] = sum over cells of H[cell][w]: a block owns a range of cells, threads stride over the
// words of a cell (coalesced), one atomic per word and block into the zeroed total
__global__ __launch_bounds__(256) void k_hist_total(const int64_t *__restrict__ H, int64_t hist_stride, int64_t cell0, int64_t cell1,
                                                    int64_t cells_per_block, int64_t *__restrict__ total) {
    const int64_t c0 = cell0 + (int64_t)blockIdx.x * cells_per_block;
    const int64_t c1 = c0 + cells_per_block < cell1 ? c0 + cells_per_block : cell1;
    for (int64_t w = threadIdx.x; w < hist_stride; w += blockDim.x) {
        int64_t acc = 0;
        int64_t cell = c0;
        for (; cell + 8 <= c1; cell += 8) {  // eight independent loads in flight per lane
            int64_t x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = __builtin_nontemporal_load(H + (cell + u) * hist_stride + w);
#pragma unroll
            for (int u = 0; u < 8; u++) acc += x[u];
        }
        for (; cell < c1; cell++) acc += H[cell * hist_stride + w];
        if (acc) gadd(total + w, acc);
    }
}

// out[i][w] = H[cells[i]][w]: the bucket arrays of the rows that will be printed.  After a reduce-scatter a rank
// holds only the cells [cell0, cell1): the other rows are zero and the ranks' buffers are summed.
__global__ __launch_bounds__(256) void k_hist_gather(const int64_t *__restrict__ H, int64_t hist_stride,
                                                     const int64_t *__restrict__ cells, int64_t cell0, int64_t cell1,
                                                     int64_t *__restrict__ out) {
    const int64_t cell = cells[blockIdx.x];
    const bool mine = cell >= cell0 && cell < cell1;
    for (int64_t w = threadIdx.x; w < hist_stride; w += blockDim.x)
        out[(int64_t)blockIdx.x * hist_stride + w] = mine ? H[cell * hist_stride + w] : 0;
}

hipError_t launch_hist_summary(const HistSummaryPlan &S, int64_t *total, hipStream_t st) {
    const int64_t pairs = (S.cell1 - S.cell0) * S.n_aggs;
    if (pairs <= 0) return hipSuccess;
    const int64_t cpb = 128;
    if (total)  // (nullptr: k_part_hist summed the Cumulative buckets itself)
        hipLaunchKernelGGL(k_hist_total, dim3((unsigned)((S.cell1 - S.cell0 + cpb - 1) / cpb)), dim3(256), 0, st, S.H, S.hist_stride, S.cell0,
                       S.cell1, cpb, total);
    hipLaunchKernelGGL(k_hist_summary, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, S);
    return hipGetLastError();
}

// Cumulative's buckets alone (a printer's query: result.cpp query_snapshot, top_only)
hipError_t launch_hist_total(const int64_t *H, int64_t hist_stride, int64_t cell0, int64_t cell1, int64_t *total, hipStream_t st) {
    if (cell1 <= cell0 || hist_stride <= 0) return hipSuccess;
    const int64_t cpb = 128;
    hipLaunchKernelGGL(k_hist_total, dim3((unsigned)((cell1 - cell0 + cpb - 1) / cpb)), dim3(256), 0, st, H, hist_stride, cell0, cell1, cpb, total);
    return hipGetLastError();
}

hipError_t launch_hist_gather(const int64_t *H, int64_t hist_stride, const int64_t *d_cells, int64_t n, int64_t cell0, int64_t cell1,
                              int64_t *out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hist_gather, dim3((unsigned)n), dim3(256), 0, st, H, hist_stride, d_cells, cell0, cell1, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- launchers (host)

template <int NC>
static hipError_t launch_scan_nc(const ScanPlan *d_plan, int n_wg, bool use_lds, size_t lds_bytes, hipStream_t st) {
    int T = 1024;
#ifdef SYBL_THREADS_AB  // (tools/bench_threads.py: 768 / 512 threads measured 1.1x / 1.5x SLOWER, profiles/r06_wg_threads_ab.txt)
    if (const char *e = env("SYBL_SCAN_THREADS")) T = atoi(e);
    if (T != 512 && T != 768) T = 1024;
#endif
    if (use_lds) {
#ifdef SYBL_THREADS_AB
        auto kfn = T == 512 ? k_scan<NC, true, 512> : T == 768 ? k_scan<NC, true, 768> : k_scan<NC, true, 1024>;
#else
        auto kfn = k_scan<NC, true, 1024>;
#endif
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), lds_bytes, st, (CPlan *)d_plan);
    } else {
#ifdef SYBL_THREADS_AB
        auto kfn = T == 512 ? k_scan<NC, false, 512> : T == 768 ? k_scan<NC, false, 768> : k_scan<NC, false, 1024>;
#else
        auto kfn = k_scan<NC, false, 1024>;
#endif
        hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), 0, st, (CPlan *)d_plan);
    }
    return hipGetLastError();
}

// k_prefilter: the filter pre-pass (planner.cpp: Planner::prefilter).  The plan holds only filter slots -- set members,
// the filter columns beyond the packed bodies' four --; every row's verdict (the generic row_prepare: filter.go:171-285)
// becomes one bit of a bitmap indexed by physical row.  A lane holds two consecutive rows, sixteen lanes a word: their bits
// are ORed together with four butterfly steps and the first lane of the sixteen stores the word (runs of rows start on
// 32-row boundaries, so words never straddle workgroups).
template <int NC>
__global__ __launch_bounds__(kWgThreads) void k_prefilter(CPlan *Pp, uint32_t *bits) {
    CPlan &P = *Pp;
    const int tid = threadIdx.x;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        int64_t row = seg.start + (int64_t)tid * kRowsPerThread;
        Tile<NC> cur;
        RawTile<NC> raw;
        if (row < end) issue_tile<NC>(P, row, raw);
        decode_tile<NC>(P, row, row < end, raw, cur);
        for (int64_t base = seg.start; base < end; base += kTileRows) {
            const int64_t nrow = row + kTileRows;
            if (nrow < end) issue_tile<NC>(P, nrow, raw);
            uint32_t v = 0;
#pragma unroll
            for (int r = 0; r < kRowsPerThread; r++) {
                uint64_t key;
                int64_t w;
                const bool pass = row + r < end && row_prepare<NC>(P, cur, r, row, key, w) != kRowFail;
                v |= (pass ? 1u : 0u) << (((uint32_t)row + (uint32_t)r) & 31u);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v |= __shfl_xor(v, o, 64);
            if ((tid & 15) == 0 && row < end) bits[row >> 5] = v;
            decode_tile<NC>(P, nrow, nrow < end, raw, cur);
            row = nrow;
        }
    }
}

hipError_t launch_prefilter(const ScanPlan *d_plan, int n_slots, uint32_t *bits, int n_wg, hipStream_t st) {
#define SYBL_PF(N) case N: hipLaunchKernelGGL((k_prefilter<N>), dim3(n_wg), dim3(kWgThreads), 0, st, (CPlan *)d_plan, bits); break;
    switch (n_slots) {
        SYBL_PF(1) SYBL_PF(2) SYBL_PF(3) SYBL_PF(4) SYBL_PF(5) SYBL_PF(6) SYBL_PF(7) SYBL_PF(8)
    default: return hipErrorInvalidValue;
    }
#undef SYBL_PF
    return hipGetLastError();
}

hipError_t launch_scan(const ScanPlan *d_plan, int n_slots, int n_wg, bool use_lds, size_t lds_bytes, hipStream_t st) {
    switch (n_slots) {
    case 1: return launch_scan_nc<1>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 2: return launch_scan_nc<2>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 3: return launch_scan_nc<3>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 4: return launch_scan_nc<4>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 5: return launch_scan_nc<5>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 6: return launch_scan_nc<6>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 7: return launch_scan_nc<7>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 8: return launch_scan_nc<8>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 9: return launch_scan_nc<9>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 10: return launch_scan_nc<10>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 11: return launch_scan_nc<11>(d_plan, n_wg, use_lds, lds_bytes, st);
    case 12: return launch_scan_nc<12>(d_plan, n_wg, use_lds, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fold(const int64_t *ws_sum, int64_t *out_sum, int64_t words_sum, const int64_t *ws_max, int64_t *out_max,
                       int64_t words_max, int n_wg, hipStream_t st) {
    const int nb_sum = (int)((words_sum + 63) / 64), nb_max = (int)((words_max + 63) / 64);
    if (nb_sum + nb_max <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fold, dim3((unsigned)(nb_sum + nb_max)), dim3(1024), 0, st, ws_sum, out_sum, words_sum, ws_max, out_max,
                       words_max, n_wg, nb_sum);
    return hipGetLastError();
}

hipError_t launch_fill64(int64_t *p, int64_t n, int64_t v, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill64, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
    return hipGetLastError();
}

hipError_t launch_synth(int64_t *out, int64_t n, int64_t row0, int64_t total_rows, int kind, int64_t a, int64_t b,
                        uint64_t col_seed, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, st, out, n, row0, total_rows, kind, a, b, col_seed);
    return hipGetLastError();
}

hipError_t launch_block_minmax(const void *col, int width, int64_t vbase, const uint32_t *valid, const Segment *blocks, int n_blocks,
                               int64_t *out_min, int64_t *out_max, int64_t *out_pop, hipStream_t st) {
    if (n_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_block_minmax, dim3(n_blocks), dim3(256), 0, st, col, width, vbase, valid, blocks, out_min, out_max, out_pop);
    return hipGetLastError();
}

}  // namespace sybl
