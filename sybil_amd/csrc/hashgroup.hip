// hashgroup.hip -- hash group-by (strategy 7): what happens after the scan.
//
// The reference groups on arbitrary keys through a Go map per block and merges the maps (aggregate.go:186-200,
// query_spec.go:107-193).  Key spaces that do not direct-map (more than 2^27 cells) are scanned by k_scan into an
// open-addressing table: slot -> composite key (ScanPlan::hash_keys) with the usual [field][slot] accumulators
// behind it.  Which slot a key lands in depends on insertion order, so the table itself can neither be all-reduced
// nor walked in key order.  query_hash_compact turns it into the CANONICAL form everything downstream uses:
//     keys[i]           the live composite keys in ascending order (= ascending group-key order: the first
//                       group column is the most significant digit)
//     dense_sum / max   [header][field][i] / [field][i]: the accumulators of key i
// and query_hash_install_union re-lays a rank's dense arrays out over the sorted union of every rank's keys, so
// that the partial tables of all ranks line up and one SUM (+ one MAX) all-reduce merges them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "engine.h"

namespace sybl {

// live slots -> (key, slot) pairs in arbitrary order
__global__ __launch_bounds__(256) void k_hash_collect(const uint64_t *__restrict__ slot_keys, int64_t n_slots, uint64_t *__restrict__ keys,
                                                      uint32_t *__restrict__ slots, unsigned long long *__restrict__ count) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const uint64_t k = slot_keys[s];
    if (k == kHashEmpty) return;
    const unsigned long long i = atomicAdd(count, 1ull);
    keys[i] = k;
    slots[i] = (uint32_t)s;
}

// dst[f][pos(i)] = src[f][slot[i]] for every field; pos(i) = i, or the place of keys[i] in `target` (a sorted superset)
__global__ __launch_bounds__(256) void k_hash_gather(const int64_t *__restrict__ src, int64_t src_cells, int n_fields,
                                                     const uint64_t *__restrict__ keys, const uint32_t *__restrict__ slots, int64_t n,
                                                     const uint64_t *__restrict__ target, int64_t n_target, int64_t *__restrict__ dst,
                                                     int64_t dst_cells, unsigned long long *__restrict__ missing) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t pos = i;
    if (target) {
        const uint64_t k = keys[i];
        int64_t lo = 0, hi = n_target;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (target[mid] < k) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= n_target || target[lo] != k) {  // the union must contain every local key
            atomicAdd(missing, 1ull);
            return;
        }
        pos = lo;
    }
    const int64_t s = slots ? (int64_t)slots[i] : i;
    for (int f = 0; f < n_fields; f++) dst[(int64_t)f * dst_cells + pos] = src[(int64_t)f * src_cells + s];
}

static int ensure(void **p, size_t bytes) {
    if (*p) return SYBL_OK;
    SYBL_HIP(hipMalloc(p, std::max<size_t>(bytes, 16)));
    return SYBL_OK;
}

int query_hash_compact(Query *q) {
    const ScanPlan &P = q->plan;
    hipStream_t st = q->ctx->stream;
    const int64_t slots = P.n_cells;
    const int F = P.n_sum_fields, M = P.n_max_fields;
    int rc;
    q->hash_cap = slots;
    if ((rc = ensure((void **)&q->d_dense_keys, (size_t)slots * 8 * 2))) return rc;  // unsorted | sorted
    if ((rc = ensure((void **)&q->d_dense_slots, (size_t)slots * 4 * 2))) return rc;
    if ((rc = ensure((void **)&q->d_hash_count, 16))) return rc;
    uint64_t *keys_in = q->d_dense_keys + slots, *keys_out = q->d_dense_keys;
    uint32_t *slots_in = q->d_dense_slots + slots, *slots_out = q->d_dense_slots;
    SYBL_HIP(hipMemsetAsync(q->d_hash_count, 0, 16, st));
    hipLaunchKernelGGL(k_hash_collect, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, P.hash_keys, slots, keys_in, slots_in,
                       (unsigned long long *)q->d_hash_count);
    uint64_t n_live = 0;
    SYBL_HIP(hipMemcpyAsync(&n_live, q->d_hash_count, 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    q->hash_live = (int64_t)n_live;
    if (n_live > 0) {
        size_t need = 0;
        SYBL_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, slots_in, slots_out, (int)n_live, 0, 64, st));
        if (need > q->sort_tmp_bytes) {
            if (q->d_sort_tmp) SYBL_HIP(hipFree(q->d_sort_tmp));
            q->d_sort_tmp = nullptr;
            SYBL_HIP(hipMalloc(&q->d_sort_tmp, need));
            q->sort_tmp_bytes = need;
        }
        need = q->sort_tmp_bytes;
        SYBL_HIP(hipcub::DeviceRadixSort::SortPairs(q->d_sort_tmp, need, keys_in, keys_out, slots_in, slots_out, (int)n_live, 0, 64, st));
    }
    // dense accumulators in key order: [header][F][n_live], [M][n_live]
    if ((rc = ensure((void **)&q->d_dense_sum, (size_t)(kHeaderWords + (int64_t)F * slots) * 8))) return rc;
    if ((rc = ensure((void **)&q->d_dense_max, (size_t)std::max<int64_t>((int64_t)M * slots, 1) * 8))) return rc;
    SYBL_HIP(hipMemcpyAsync(q->d_dense_sum, q->d_sum, (size_t)kHeaderWords * 8, hipMemcpyDeviceToDevice, st));
    if (n_live > 0) {
        const unsigned nb = (unsigned)((n_live + 255) / 256);
        hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, q->d_sum + kHeaderWords, slots, F, keys_out, slots_out, (int64_t)n_live,
                           (const uint64_t *)nullptr, (int64_t)0, q->d_dense_sum + kHeaderWords, (int64_t)n_live, (unsigned long long *)nullptr);
        if (M > 0)
            hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, q->d_max, slots, M, keys_out, slots_out, (int64_t)n_live,
                               (const uint64_t *)nullptr, (int64_t)0, q->d_dense_max, (int64_t)n_live, (unsigned long long *)nullptr);
    }
    q->h_dense_keys.resize((size_t)n_live);
    if (n_live > 0) SYBL_HIP(hipMemcpyAsync(q->h_dense_keys.data(), keys_out, (size_t)n_live * 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hash compaction");
    q->hash_compacted = true;
    return SYBL_OK;
}

// Multi-rank: every rank installs the sorted union of the ranks' key lists; the dense arrays are re-laid out over it
// (zeros / INT64_MIN where this rank has no row of a key).
int query_hash_install_union(Query *q, const uint64_t *keys, int64_t n) {
    const ScanPlan &P = q->plan;
    hipStream_t st = q->ctx->stream;
    if (!q->hash_compacted) return fail(SYBL_E_STATE, "hash union before the scan");
    if (n < q->hash_live || n > P.n_cells) return fail(SYBL_E_INVAL, "the union holds %lld keys, this rank %lld, the table %lld", (long long)n,
                                                       (long long)q->hash_live, (long long)P.n_cells);
    for (int64_t i = 1; i < n; i++)
        if (keys[i - 1] >= keys[i]) return fail(SYBL_E_INVAL, "union keys must be strictly ascending");
    const int F = P.n_sum_fields, M = P.n_max_fields;
    const int64_t live = q->hash_live, slots = P.n_cells;
    // the local dense arrays move aside (into the unsorted halves / fresh buffers), the union layout takes their place
    uint64_t *d_union = q->d_dense_keys + slots;  // (the unsorted half is free after the sort)
    SYBL_HIP(hipMemcpyAsync(d_union, keys, (size_t)n * 8, hipMemcpyHostToDevice, st));
    int64_t *old_sum = nullptr, *old_max = nullptr;
    SYBL_HIP(hipMalloc((void **)&old_sum, (size_t)std::max<int64_t>((int64_t)F * live, 1) * 8));
    SYBL_HIP(hipMalloc((void **)&old_max, (size_t)std::max<int64_t>((int64_t)M * live, 1) * 8));
    SYBL_HIP(hipMemcpyAsync(old_sum, q->d_dense_sum + kHeaderWords, (size_t)F * live * 8, hipMemcpyDeviceToDevice, st));
    if (M > 0) SYBL_HIP(hipMemcpyAsync(old_max, q->d_dense_max, (size_t)M * live * 8, hipMemcpyDeviceToDevice, st));
    SYBL_HIP(hipMemsetAsync(q->d_dense_sum + kHeaderWords, 0, (size_t)F * n * 8, st));
    hipError_t e = M > 0 ? launch_fill64(q->d_dense_max, (int64_t)M * n, INT64_MIN, st) : hipSuccess;
    if (e != hipSuccess) return hip_fail(e, "k_fill64");
    SYBL_HIP(hipMemsetAsync(q->d_hash_count, 0, 16, st));
    if (live > 0) {
        const unsigned nb = (unsigned)((live + 255) / 256);
        hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, old_sum, live, F, q->d_dense_keys, (const uint32_t *)nullptr, live, d_union, n,
                           q->d_dense_sum + kHeaderWords, n, (unsigned long long *)q->d_hash_count);
        if (M > 0)
            hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, old_max, live, M, q->d_dense_keys, (const uint32_t *)nullptr, live, d_union, n,
                               q->d_dense_max, n, (unsigned long long *)q->d_hash_count);
    }
    uint64_t missing = 0;
    SYBL_HIP(hipMemcpyAsync(&missing, q->d_hash_count, 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipMemcpyAsync(q->d_dense_keys, d_union, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    SYBL_HIP(hipStreamSynchronize(st));
    (void)hipFree(old_sum);
    (void)hipFree(old_max);
    if (missing) return fail(SYBL_E_INVAL, "the union lacks %llu of this rank's keys", (unsigned long long)missing);
    q->h_dense_keys.assign(keys, keys + n);
    q->hash_live = n;
    return SYBL_OK;
}

}  // namespace sybl
