// distinct.hip -- count distinct (reference: aggregate.go:205-243, query_spec.go:87,100,180-188; hll.h).
//
// k_scan_distinct<NC> walks the same rows as the query's scan kernel with the same filters, group key and time bucket
// (row_prepare, scan_generic.h) and, instead of accumulating fields, hashes the row's distinct value and raises one
// register of the cell's LogLog-Beta sketch.  It is a pass of its own over the filter / key / distinct columns, so the
// query's scan keeps whatever strategy it has; the sketches ([cell][16384] bytes) live in HBM.
#include <hip/hip_runtime.h>

#include "hll.h"
#include "scan_generic.h"

namespace sybl {

// registers[reg] = max(registers[reg], rank) on the byte's 32-bit word.  A register only ever grows and after the first
// few thousand rows of a group almost no row raises one: the plain read settles nearly every row without an atomic.
__device__ __forceinline__ void hll_raise(uint8_t *regs, uint64_t hash) {
    uint32_t reg, rank;
    hll_place(hash, reg, rank);
    uint32_t *word = (uint32_t *)regs + (reg >> 2);
    const uint32_t sh = (reg & 3u) * 8u;
    uint32_t cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (((cur >> sh) & 0xFFu) < rank) {
        const uint32_t want = (cur & ~(0xFFu << sh)) | (rank << sh);
        if (__hip_atomic_compare_exchange_strong(word, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}

// the slow path over several columns, one of them a str column: aggregate.go:224-239
template <int NC>
__device__ __noinline__ uint64_t distinct_hash_mixed(CPlan &P, const Tile<NC> &t, int r) {
    Metro64Stream S;
    S.init(kHllSeed);
    for (int i = 0; i < P.n_distinct; i++) {
        bool pop = false;
        int64_t v = 0;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == P.distinct_slot[i] && ((t.pop[c] >> r) & 1u)) {
                pop = true;
                v = r == 0 ? t.v[c].x : t.v[c].y;
            }
        if (pop) {
            const char *chars = P.hll_chars[i];
            if (!chars) {
                S.put_decimal(v);
            } else if ((uint64_t)v < (uint64_t)P.hll_nids[i]) {
                const int64_t e = P.hll_stroff[i][v + 1];
                for (int64_t k = P.hll_stroff[i][v]; k < e; k++) S.put((uint8_t)chars[k]);
            }
        }
        S.put((uint8_t)'\t');
    }
    return S.finish();
}

template <int NC>
__device__ __forceinline__ uint64_t distinct_hash(CPlan &P, const Tile<NC> &t, int r) {
    if (P.hll_mixed) return distinct_hash_mixed<NC>(P, t, r);
    if (P.hll_idhash) {
        // one str column: the string's hash by dictionary id
        int64_t id = -1;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == P.distinct_slot[0] && ((t.pop[c] >> r) & 1u)) id = r == 0 ? t.v[c].x : t.v[c].y;
        return (uint64_t)id < (uint64_t)P.hll_ids ? P.hll_idhash[id] : P.hll_missing;
    }
    // int columns: 8 little-endian bytes each, MISSING_VALUE (all ones) for a row without the column (aggregate.go:210-219)
    uint64_t w[kMaxDistinct];
#pragma unroll
    for (int i = 0; i < kMaxDistinct; i++) {
        w[i] = ~(uint64_t)0;
        if (i < P.n_distinct) {
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (c == P.distinct_slot[i] && ((t.pop[c] >> r) & 1u)) w[i] = (uint64_t)(r == 0 ? t.v[c].x : t.v[c].y);
        }
    }
    return metro64_words(w, P.n_distinct, kHllSeed);
}

template <int NC>
__device__ __forceinline__ void distinct_row(CPlan &P, const Tile<NC> &t, int r, int64_t row0) {
    uint64_t key;
    int64_t w;
    // rows the scan drops or reports (no time value; key outside the declared bounds) own no Result here either
    if (row_prepare<NC>(P, t, r, row0, key, w) != kRowOk) return;
    if (P.hll_keys) {
        // hashed group-by: the row's Result is the one of its composite key (aggregate.go:186-200: map[string]*Result) -- its
        // place in the sorted key list (a key the table could not take has none)
        int64_t lo = 0, hi = P.hll_nkeys;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (P.hll_keys[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= P.hll_nkeys || P.hll_keys[lo] != key) return;
        key = (uint64_t)lo;
    } else if (key >= (uint64_t)P.n_cells) {
        return;
    }
    hll_raise(P.hll + key * (uint64_t)kHllRegs, distinct_hash<NC>(P, t, r));
}

template <int NC, int T>  // (T threads per workgroup: kernels.hip, k_scan)
__global__ __launch_bounds__(T) void k_scan_distinct(CPlan *Pp) {
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
        for (int64_t base = seg.start; base < end; base += (T * kRowsPerThread)) {
            const int64_t nrow = row + (T * kRowsPerThread);
            if (nrow < end) issue_tile<NC>(P, nrow, raw);
            const int64_t left = end - row;
            const int nvalid = left >= kRowsPerThread ? kRowsPerThread : (left > 0 ? (int)left : 0);
            if (nvalid > 0) distinct_row<NC>(P, cur, 0, row);
            if (nvalid > 1) distinct_row<NC>(P, cur, 1, row);
            decode_tile<NC>(P, nrow, nrow < end, raw, cur);
            row = nrow;
        }
    }
}

template <int NC>
static hipError_t launch_nc(const ScanPlan *d_plan, int n_wg, hipStream_t st) {
    int T = 1024;
#ifdef SYBL_THREADS_AB  // (kernels.hip: launch_scan_nc)
    if (const char *e = env("SYBL_SCAN_THREADS")) T = atoi(e);
    if (T != 512 && T != 768) T = 1024;
    auto kfn = T == 512 ? k_scan_distinct<NC, 512> : T == 768 ? k_scan_distinct<NC, 768> : k_scan_distinct<NC, 1024>;
#else
    auto kfn = k_scan_distinct<NC, 1024>;
#endif
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), 0, st, (CPlan *)d_plan);
    return hipGetLastError();
}

hipError_t launch_scan_distinct(const ScanPlan *d_plan, int n_slots, int n_wg, hipStream_t st) {
    switch (n_slots) {
    case 1: return launch_nc<1>(d_plan, n_wg, st);
    case 2: return launch_nc<2>(d_plan, n_wg, st);
    case 3: return launch_nc<3>(d_plan, n_wg, st);
    case 4: return launch_nc<4>(d_plan, n_wg, st);
    case 5: return launch_nc<5>(d_plan, n_wg, st);
    case 6: return launch_nc<6>(d_plan, n_wg, st);
    case 7: return launch_nc<7>(d_plan, n_wg, st);
    case 8: return launch_nc<8>(d_plan, n_wg, st);
    case 9: return launch_nc<9>(d_plan, n_wg, st);
    case 10: return launch_nc<10>(d_plan, n_wg, st);
    case 11: return launch_nc<11>(d_plan, n_wg, st);
    case 12: return launch_nc<12>(d_plan, n_wg, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sybl
