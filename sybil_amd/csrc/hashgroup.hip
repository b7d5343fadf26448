// hashgroup.hip -- hash group-by (strategy 7): group keys that do not direct-map.
//
// The reference groups on arbitrary keys through a Go map per block and merges the maps (aggregate.go:186-200,
// query_spec.go:107-193).  The direct-mapped kernels need one cell per POSSIBLE key; when the key space is too wide for
// that (more than 2^27 cells, more than 2^22 distinct values in a key column) the composite key of a row -- the sum of
// digit x stride over the group columns, below 2^62 -- is looked up in an open-addressing table instead:
//
//   k_scan_hash<NC>   the generic row body (scan_generic.h) with two table levels:
//                       * an LDS staging table per workgroup (keys + every cell field, linear probing, bounded probe
//                         count, filled to 3/4): keys that repeat inside a workgroup's rows are aggregated with LDS
//                         atomics and reach HBM once per workgroup, when the table is flushed at the end;
//                       * the global table in HBM (P.hash_keys + the usual [field][slot] accumulators): rows whose key
//                         found no room in LDS, and the flush, find-or-claim a slot with one CAS per first sighting
//                         and accumulate with device-scope atomics.
//   query_hash_compact   slot order depends on insertion order, so the table itself can neither be all-reduced nor
//                        walked in key order.  The live slots are collected, radix-sorted by key (hipcub) and the
//                        accumulators gathered into the CANONICAL form everything downstream uses:
//                            keys[i]          the live composite keys, ascending (= ascending group-key order: the
//                                             first group column is the most significant digit)
//                            dense_sum        [header][field][i], then [i][hist_stride] bucket arrays
//                            dense_max        [field][i]
//   query_hash_install_union   multi-GPU: the dense arrays re-laid out over the sorted union of every rank's keys, so
//                        that the partial tables of all ranks line up and one SUM (+ one MAX) all-reduce merges them
//                        (rccl.cpp: query_hash_allreduce; hosts with their own collective runtime: sybl_query_hash_keys /
//                        sybl_query_hash_install_union).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <type_traits>

#include "engine.h"
#include "hash_table.h"
#include "scan_generic.h"

namespace sybl {

// (probe limits and hash_find_or_insert: hash_table.h; k_scan_hash_fast: hashfast.hip; k_scan_hash_packed: hashpacked.hip)

// one row: LDS staging table first, the global table when the key finds no room there
template <int NC>
__device__ __forceinline__ void hash_row(CPlan &P, const Tile<NC> &cur, const int r, int64_t row, uint64_t *lkeys, int64_t *lsum, int64_t *lmax,
                                         uint32_t *l_used, int64_t L, int64_t &matched, int64_t &overflow, int64_t &full) {
    uint64_t key;
    int64_t w;
    const int st = row_prepare<NC>(P, cur, r, row, key, w);
    if (st == kRowFail) return;
    matched += 1;
    if (st == kRowDropped) return;
    if (st == kRowOverflow) {
        overflow += 1;
        return;
    }
    int32_t ls = -1;
    if (L > 0) {
        const uint32_t lmask = (uint32_t)L - 1u, l_limit = (uint32_t)(L - (L >> 2));
        // (the low half of the hash: independent of the slot the key gets in the global table)
        uint32_t h = (uint32_t)splitmix64(key) & lmask;
        for (int probe = 0; probe < kHashLdsProbes; probe++) {
            uint64_t k = __hip_atomic_load(lkeys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == kHashEmpty) {
                // the key is not staged (no deletions: it would sit before the first free slot of its probe
                // sequence); claim the slot unless the table is full enough
                if (__hip_atomic_load(l_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= l_limit) break;
                unsigned long long expect = kHashEmpty;
                if (__hip_atomic_compare_exchange_strong((unsigned long long *)lkeys + h, &expect, (unsigned long long)key, __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    __hip_atomic_fetch_add(l_used, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    k = key;
                } else {
                    k = expect;
                }
            }
            if (k == key) {
                ls = (int32_t)h;
                break;
            }
            h = (h + 1) & lmask;
        }
    }
    if (ls >= 0) {
        row_accumulate<NC, true>(P, cur, r, lsum, lmax, L, 0, (int64_t)ls, (int64_t)-1, (int64_t)key, w, overflow);
    } else {
        const int32_t g = hash_find_or_insert(P.hash_keys, (uint32_t)P.n_cells - 1u, key, P.sum_out);
        if (g < 0) {  // more distinct keys than the table holds: reported by finalize
            full += 1;
            return;
        }
        row_accumulate<NC, false>(P, cur, r, P.sum_out + kHeaderWords, P.max_out, (int64_t)P.n_cells, 0, (int64_t)g, (int64_t)g, (int64_t)key, w, overflow);
    }
}

template <int NC, int T>  // (T threads per workgroup: kernels.hip, k_scan)
__global__ __launch_bounds__(T) void k_scan_hash(CPlan *Pp) {
    CPlan &P = *Pp;
    extern __shared__ int64_t lds[];
    __shared__ uint32_t l_used;
    const int tid = threadIdx.x;
    const int64_t L = P.lds_cells;  // LDS staging slots (a power of two); 0: every row goes to the global table
    const int F = P.n_sum_fields, M = P.n_max_fields;
    uint64_t *lkeys = (uint64_t *)lds;
    int64_t *lsum = lds + L, *lmax = lsum + (int64_t)F * L;
    for (int64_t i = tid; i < L; i += T) lkeys[i] = kHashEmpty;
    for (int64_t i = tid; i < (int64_t)F * L; i += T) lsum[i] = 0;
    for (int64_t i = tid; i < (int64_t)M * L; i += T) lmax[i] = INT64_MIN;
    if (tid == 0) l_used = 0;
    __syncthreads();
    int64_t *gsum = P.sum_out + kHeaderWords, *gmax = P.max_out;
    const uint32_t gmask = (uint32_t)P.n_cells - 1u;

    int64_t matched = 0, overflow = 0, full = 0;
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
            static_assert(kRowsPerThread == 2, "two rows per lane and tile");
            if (left > 0) hash_row<NC>(P, cur, 0, row, lkeys, lsum, lmax, &l_used, L, matched, overflow, full);
            if (left > 1) hash_row<NC>(P, cur, 1, row, lkeys, lsum, lmax, &l_used, L, matched, overflow, full);
            decode_tile<NC>(P, nrow, nrow < end, raw, cur);
            row = nrow;
        }
    }

    // flush the staging table: one find-or-claim per staged key, one atomic per non-zero field
    if (L > 0) {
        __syncthreads();
        for (int64_t i = tid; i < L; i += T) {
            const uint64_t k = lkeys[i];
            if (k == kHashEmpty) continue;
            const int32_t g = hash_find_or_insert(P.hash_keys, gmask, k, P.sum_out);
            if (g < 0) {
                full += 1;
                continue;
            }
            for (int f = 0; f < F; f++) {
                const int64_t v = lsum[(int64_t)f * L + i];
                if (v != 0) gadd(gsum + (int64_t)f * P.n_cells + g, v);
            }
            for (int m = 0; m < M; m++) {
                const int64_t v = lmax[(int64_t)m * L + i];
                if (v != INT64_MIN) __hip_atomic_fetch_max(gmax + (int64_t)m * P.n_cells + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    matched = wave_sum(matched);
    overflow = wave_sum(overflow);
    full = wave_sum(full);
    {
        const int slot[3] = {kHdrMatched, kHdrOverflow, kHdrHashFull};
        const int64_t v[3] = {(int64_t)matched, (int64_t)overflow, (int64_t)full};
        wg_header_add<3>(P.sum_out, slot, v);  // (one atomic per workgroup and counter: scan_generic.h)
    }
}

template <int NC>
static hipError_t launch_scan_hash_nc(const ScanPlan *d_plan, int n_wg, size_t lds_bytes, hipStream_t st) {
    int T = 1024;
#ifdef SYBL_THREADS_AB  // (kernels.hip: launch_scan_nc)
    if (const char *e = env("SYBL_SCAN_THREADS")) T = atoi(e);
    if (T != 512 && T != 768) T = 1024;
    auto kfn = T == 512 ? k_scan_hash<NC, 512> : T == 768 ? k_scan_hash<NC, 768> : k_scan_hash<NC, 1024>;
#else
    auto kfn = k_scan_hash<NC, 1024>;
#endif
    hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds_bytes, 16));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), lds_bytes, st, (CPlan *)d_plan);
    return hipGetLastError();
}

hipError_t launch_scan_hash(const ScanPlan *d_plan, int n_slots, int n_wg, size_t lds_bytes, hipStream_t st) {
    switch (n_slots) {
    case 1: return launch_scan_hash_nc<1>(d_plan, n_wg, lds_bytes, st);
    case 2: return launch_scan_hash_nc<2>(d_plan, n_wg, lds_bytes, st);
    case 3: return launch_scan_hash_nc<3>(d_plan, n_wg, lds_bytes, st);
    case 4: return launch_scan_hash_nc<4>(d_plan, n_wg, lds_bytes, st);
    case 5: return launch_scan_hash_nc<5>(d_plan, n_wg, lds_bytes, st);
    case 6: return launch_scan_hash_nc<6>(d_plan, n_wg, lds_bytes, st);
    case 7: return launch_scan_hash_nc<7>(d_plan, n_wg, lds_bytes, st);
    case 8: return launch_scan_hash_nc<8>(d_plan, n_wg, lds_bytes, st);
    case 9: return launch_scan_hash_nc<9>(d_plan, n_wg, lds_bytes, st);
    case 10: return launch_scan_hash_nc<10>(d_plan, n_wg, lds_bytes, st);
    case 11: return launch_scan_hash_nc<11>(d_plan, n_wg, lds_bytes, st);
    case 12: return launch_scan_hash_nc<12>(d_plan, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------- compaction

// number of live slots (one atomic per wave)
__global__ __launch_bounds__(256) void k_hash_count(const uint64_t *__restrict__ slot_keys, int64_t n_slots, unsigned long long *__restrict__ count) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t n = 0;
    for (; s < n_slots; s += stride) n += slot_keys[s] != kHashEmpty;
    n = wave_sum(n);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, (unsigned long long)n);
}

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

// where keys[i] sits in `target` (sorted, a superset): lower bound; -1 when it is absent
__device__ __forceinline__ int64_t key_position(const uint64_t *__restrict__ target, int64_t n_target, uint64_t k) {
    int64_t lo = 0, hi = n_target;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (target[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo < n_target && target[lo] == k ? lo : -1;
}

// dst[f][pos(i)] = src[f][slot(i)] for every field; slot(i) = slots[i] (or i), pos(i) = i, or the place of keys[i] in `target`
__global__ __launch_bounds__(256) void k_hash_gather(const int64_t *__restrict__ src, int64_t src_cells, int n_fields,
                                                     const uint64_t *__restrict__ keys, const uint32_t *__restrict__ slots, int64_t n,
                                                     const uint64_t *__restrict__ target, int64_t n_target, int64_t *__restrict__ dst,
                                                     int64_t dst_cells, unsigned long long *__restrict__ missing) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t pos = i;
    if (target) {
        pos = key_position(target, n_target, keys[i]);
        if (pos < 0) {  // the union must contain every local key
            atomicAdd(missing, 1ull);
            return;
        }
    }
    const int64_t s = slots ? (int64_t)slots[i] : i;
    for (int f = 0; f < n_fields; f++) dst[(int64_t)f * dst_cells + pos] = src[(int64_t)f * src_cells + s];
}

// the bucket arrays: dst[pos(i)][w] = src[slot(i)][w]; one block per key, threads stride over the words (coalesced)
__global__ __launch_bounds__(256) void k_hash_gather_hist(const int64_t *__restrict__ src, int64_t stride, const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ slots, const uint64_t *__restrict__ target,
                                                          int64_t n_target, int64_t *__restrict__ dst) {
    const int64_t i = blockIdx.x;
    int64_t pos = i;
    if (target) {
        pos = key_position(target, n_target, keys[i]);
        if (pos < 0) return;  // (counted by k_hash_gather)
    }
    const int64_t s = slots ? (int64_t)slots[i] : i;
    for (int64_t w = threadIdx.x; w < stride; w += blockDim.x) dst[pos * stride + w] = src[s * stride + w];
}

template <typename T>
static int grow(T **p, int64_t *cap, int64_t need) {
    if (*p && *cap >= need) return SYBL_OK;
    if (*p) SYBL_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const int64_t want = std::max<int64_t>(need + need / 8, 64);
    SYBL_HIP(hipMalloc((void **)p, (size_t)want * sizeof(T)));
    *cap = want;
    return SYBL_OK;
}

static int sort_tmp(Query *q, size_t need) {
    if (need <= q->sort_tmp_bytes && q->d_sort_tmp) return SYBL_OK;
    if (q->d_sort_tmp) SYBL_HIP(hipFree(q->d_sort_tmp));
    q->d_sort_tmp = nullptr;
    q->sort_tmp_bytes = 0;
    SYBL_HIP(hipMalloc(&q->d_sort_tmp, std::max<size_t>(need, 256)));
    q->sort_tmp_bytes = std::max<size_t>(need, 256);
    return SYBL_OK;
}

static int key_bits(const Query *q) {
    // composite keys are below group cells x time buckets (< 2^62: planner)
    const unsigned __int128 space = (unsigned __int128)q->group_cells * (unsigned __int128)std::max(q->plan.n_tb, 1);
    int bits = 1;
    while (bits < 63 && ((unsigned __int128)1 << bits) < space) bits++;
    return std::min(bits + 1, 64);
}

int64_t hash_dense_sum_words(const Query *q, int64_t n) { return kHeaderWords + ((int64_t)q->plan.n_sum_fields + q->plan.hist_stride) * n; }
int64_t hash_dense_max_words(const Query *q, int64_t n) { return std::max<int64_t>((int64_t)q->plan.n_max_fields * n, 1); }

void query_hash_free(Query *q) {
    if (q->d_hash_keys) (void)hipFree(q->d_hash_keys);
    if (q->d_dense_keys) (void)hipFree(q->d_dense_keys);
    if (q->d_pair_keys) (void)hipFree(q->d_pair_keys);
    if (q->d_dense_slots) (void)hipFree(q->d_dense_slots);
    if (q->d_pair_slots) (void)hipFree(q->d_pair_slots);
    if (q->d_dense_sum) (void)hipFree(q->d_dense_sum);
    if (q->d_dense_max) (void)hipFree(q->d_dense_max);
    if (q->d_sort_tmp) (void)hipFree(q->d_sort_tmp);
    if (q->d_hash_count) (void)hipFree(q->d_hash_count);
}

// before every scan: every slot free (the accumulators are cleared by the scan driver like any global-atomic table)
int query_hash_reset(Query *q) {
    hipStream_t st = q->ctx->stream;
    const int64_t slots = q->plan.n_cells;
    if (!q->d_hash_keys) {
        SYBL_HIP(hipMalloc((void **)&q->d_hash_keys, (size_t)slots * 8));
        q->plan.hash_keys = q->d_hash_keys;
        q->plan_dirty = true;
    }
    SYBL_HIP(hipMemsetAsync(q->d_hash_keys, 0xFF, (size_t)slots * 8, st));
    q->hash_compacted = false;
    q->hash_live = 0;
    return SYBL_OK;
}

int query_hash_compact(Query *q) {
    if (q->hash_compacted) return SYBL_OK;
    const ScanPlan &P = q->plan;
    hipStream_t st = q->ctx->stream;
    const int64_t slots = P.n_cells;
    const int F = P.n_sum_fields, M = P.n_max_fields;
    int rc;
    if (!q->d_hash_count) SYBL_HIP(hipMalloc((void **)&q->d_hash_count, 16));
    SYBL_HIP(hipMemsetAsync(q->d_hash_count, 0, 16, st));
    {
        const unsigned nb = (unsigned)std::min<int64_t>((slots + 255) / 256, 4096);
        hipLaunchKernelGGL(k_hash_count, dim3(nb), dim3(256), 0, st, (const uint64_t *)q->d_hash_keys, slots, (unsigned long long *)q->d_hash_count);
    }
    uint64_t n_live = 0;
    SYBL_HIP(hipMemcpyAsync(&n_live, q->d_hash_count, 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    const int64_t n = (int64_t)n_live;
    if ((rc = grow(&q->d_pair_keys, &q->pair_cap, n))) return rc;
    if ((rc = grow(&q->d_pair_slots, &q->pair_slots_cap, n))) return rc;
    if ((rc = grow(&q->d_dense_keys, &q->dense_keys_cap, n))) return rc;
    if ((rc = grow(&q->d_dense_slots, &q->dense_slots_cap, n))) return rc;
    if ((rc = grow(&q->d_dense_sum, &q->dense_sum_cap, hash_dense_sum_words(q, n)))) return rc;
    if ((rc = grow(&q->d_dense_max, &q->dense_max_cap, hash_dense_max_words(q, n)))) return rc;
    SYBL_HIP(hipMemcpyAsync(q->d_dense_sum, q->d_sum, (size_t)kHeaderWords * 8, hipMemcpyDeviceToDevice, st));
    if (n > 0) {
        SYBL_HIP(hipMemsetAsync(q->d_hash_count, 0, 16, st));
        hipLaunchKernelGGL(k_hash_collect, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, (const uint64_t *)q->d_hash_keys, slots,
                           q->d_pair_keys, q->d_pair_slots, (unsigned long long *)q->d_hash_count);
        size_t need = 0;
        const int end_bit = key_bits(q);
        SYBL_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, q->d_pair_keys, q->d_dense_keys, q->d_pair_slots, q->d_dense_slots, (int)n, 0,
                                                    end_bit, st));
        if ((rc = sort_tmp(q, need))) return rc;
        need = q->sort_tmp_bytes;
        SYBL_HIP(hipcub::DeviceRadixSort::SortPairs(q->d_sort_tmp, need, q->d_pair_keys, q->d_dense_keys, q->d_pair_slots, q->d_dense_slots, (int)n, 0,
                                                    end_bit, st));
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, (const int64_t *)(q->d_sum + kHeaderWords), slots, F,
                           (const uint64_t *)q->d_dense_keys, (const uint32_t *)q->d_dense_slots, n, (const uint64_t *)nullptr, (int64_t)0,
                           q->d_dense_sum + kHeaderWords, n, (unsigned long long *)nullptr);
        if (M > 0)
            hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, (const int64_t *)q->d_max, slots, M, (const uint64_t *)q->d_dense_keys,
                               (const uint32_t *)q->d_dense_slots, n, (const uint64_t *)nullptr, (int64_t)0, q->d_dense_max, n,
                               (unsigned long long *)nullptr);
        if (P.hist_stride > 0)
            hipLaunchKernelGGL(k_hash_gather_hist, dim3((unsigned)n), dim3(256), 0, st, (const int64_t *)(q->d_sum + P.hist_off), P.hist_stride,
                               (const uint64_t *)q->d_dense_keys, (const uint32_t *)q->d_dense_slots, (const uint64_t *)nullptr, (int64_t)0,
                               q->d_dense_sum + kHeaderWords + (int64_t)F * n);
    }
    {
        int rc = query_host_keys(q, n);
        if (rc) return rc;
    }
    if (n > 0) SYBL_HIP(hipMemcpyAsync(q->h_dense_keys, q->d_dense_keys, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hash compaction");
    q->hash_live = n;
    q->hash_compacted = true;
    return SYBL_OK;
}

// Multi-rank: every rank installs the sorted union of the ranks' key lists; the dense arrays are re-laid out over it
// (zeros / INT64_MIN where this rank has no row of a key).  d_union: n strictly ascending keys in device memory.
int query_hash_install_union_device(Query *q, const uint64_t *d_union, int64_t n) {
    const ScanPlan &P = q->plan;
    hipStream_t st = q->ctx->stream;
    if (!q->hash_compacted) return fail(SYBL_E_STATE, "hash union before the scan was compacted");
    const int64_t live = q->hash_live;
    if (n < live) return fail(SYBL_E_INVAL, "the union holds %lld keys, this rank alone %lld", (long long)n, (long long)live);
    const int F = P.n_sum_fields, M = P.n_max_fields;
    int64_t *new_sum = nullptr, *new_max = nullptr;
    uint64_t *new_keys = nullptr;
    const int64_t sum_words = hash_dense_sum_words(q, n), max_words = hash_dense_max_words(q, n);
    const int64_t sum_cap = sum_words + sum_words / 8, max_cap = max_words + max_words / 8, key_cap = std::max<int64_t>(n + n / 8, 64);
    DevOwner own_sum, own_max, own_keys;  // (freed on every error return below)
    SYBL_HIP(hipMalloc(&own_sum.p, (size_t)sum_cap * 8));
    SYBL_HIP(hipMalloc(&own_max.p, (size_t)max_cap * 8));
    SYBL_HIP(hipMalloc(&own_keys.p, (size_t)key_cap * 8));
    new_sum = (int64_t *)own_sum.p;
    new_max = (int64_t *)own_max.p;
    new_keys = (uint64_t *)own_keys.p;
    SYBL_HIP(hipMemcpyAsync(new_keys, d_union, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    SYBL_HIP(hipMemsetAsync(new_sum, 0, (size_t)sum_words * 8, st));
    SYBL_HIP(hipMemcpyAsync(new_sum, q->d_dense_sum, (size_t)kHeaderWords * 8, hipMemcpyDeviceToDevice, st));
    hipError_t e = launch_fill64(new_max, max_words, INT64_MIN, st);
    if (e != hipSuccess) return hip_fail(e, "k_fill64");
    if (!q->d_hash_count) SYBL_HIP(hipMalloc((void **)&q->d_hash_count, 16));
    SYBL_HIP(hipMemsetAsync(q->d_hash_count, 0, 16, st));
    if (live > 0) {
        const unsigned nb = (unsigned)((live + 255) / 256);
        hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, (const int64_t *)(q->d_dense_sum + kHeaderWords), live, F,
                           (const uint64_t *)q->d_dense_keys, (const uint32_t *)nullptr, live, (const uint64_t *)new_keys, n, new_sum + kHeaderWords, n,
                           (unsigned long long *)q->d_hash_count);
        if (M > 0)
            hipLaunchKernelGGL(k_hash_gather, dim3(nb), dim3(256), 0, st, (const int64_t *)q->d_dense_max, live, M, (const uint64_t *)q->d_dense_keys,
                               (const uint32_t *)nullptr, live, (const uint64_t *)new_keys, n, new_max, n, (unsigned long long *)q->d_hash_count);
        if (P.hist_stride > 0)
            hipLaunchKernelGGL(k_hash_gather_hist, dim3((unsigned)live), dim3(256), 0, st,
                               (const int64_t *)(q->d_dense_sum + kHeaderWords + (int64_t)F * live), P.hist_stride, (const uint64_t *)q->d_dense_keys,
                               (const uint32_t *)nullptr, (const uint64_t *)new_keys, n, new_sum + kHeaderWords + (int64_t)F * n);
    }
    uint64_t missing = 0;
    SYBL_HIP(hipMemcpyAsync(&missing, q->d_hash_count, 8, hipMemcpyDeviceToHost, st));
    {
        int rc = query_host_keys(q, n);
        if (rc) return rc;
    }
    if (n > 0) SYBL_HIP(hipMemcpyAsync(q->h_dense_keys, new_keys, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    e = hipGetLastError();
    if (e != hipSuccess || missing) {
        if (e != hipSuccess) return hip_fail(e, "hash union");
        return fail(SYBL_E_INVAL, "the union lacks %llu of this rank's keys", (unsigned long long)missing);
    }
    for (int64_t i = 1; i < n; i++)
        if (q->h_dense_keys[(size_t)i - 1] >= q->h_dense_keys[(size_t)i]) return fail(SYBL_E_INVAL, "union keys must be strictly ascending");
    (void)hipFree(q->d_dense_sum);
    (void)hipFree(q->d_dense_max);
    (void)hipFree(q->d_dense_keys);
    q->d_dense_sum = own_sum.release<int64_t>();
    q->dense_sum_cap = sum_cap;
    q->d_dense_max = own_max.release<int64_t>();
    q->dense_max_cap = max_cap;
    q->d_dense_keys = own_keys.release<uint64_t>();
    q->dense_keys_cap = key_cap;
    q->hash_live = n;
    if (q->n_distinct) q->distinct_pending = true;  // (count distinct: the sketches follow the key list -- engine.cpp: query_hash_distinct)
    return SYBL_OK;
}

int query_hash_install_union(Query *q, const uint64_t *keys, int64_t n) {
    if (n < 0 || (n > 0 && !keys)) return fail(SYBL_E_INVAL, "bad key list");
    uint64_t *d = nullptr;
    SYBL_HIP(hipMalloc((void **)&d, (size_t)std::max<int64_t>(n, 1) * 8));
    hipError_t e = n > 0 ? hipMemcpyAsync(d, keys, (size_t)n * 8, hipMemcpyHostToDevice, q->ctx->stream) : hipSuccess;
    int rc = e != hipSuccess ? hip_fail(e, "hipMemcpyAsync") : query_hash_install_union_device(q, d, n);
    (void)hipStreamSynchronize(q->ctx->stream);
    (void)hipFree(d);
    return rc;
}

// sorted union of R key lists of `per` entries each (padded with kHashEmpty) -> *out (device, owned by the caller), *n_out
int hash_union_of_lists(Query *q, const uint64_t *d_lists, int64_t total, uint64_t **out, int64_t *n_out) {
    hipStream_t st = q->ctx->stream;
    *out = nullptr;
    *n_out = 0;
    if (total <= 0) return SYBL_OK;
    if (total >= ((int64_t)1 << 31)) return fail(SYBL_E_INVAL, "too many keys to merge across ranks (%lld)", (long long)total);
    uint64_t *sorted = nullptr, *uniq = nullptr;
    SYBL_HIP(hipMalloc((void **)&sorted, (size_t)total * 8));
    SYBL_HIP(hipMalloc((void **)&uniq, (size_t)total * 8));
    if (!q->d_hash_count) SYBL_HIP(hipMalloc((void **)&q->d_hash_count, 16));
    size_t need = 0, need2 = 0;
    int rc = SYBL_OK;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, need, d_lists, sorted, (int)total, 0, 64, st);
    if (e == hipSuccess) e = hipcub::DeviceSelect::Unique(nullptr, need2, sorted, uniq, (int *)q->d_hash_count, (int)total, st);
    if (e == hipSuccess) rc = sort_tmp(q, std::max(need, need2));
    if (e == hipSuccess && !rc) {
        need = q->sort_tmp_bytes;
        e = hipcub::DeviceRadixSort::SortKeys(q->d_sort_tmp, need, d_lists, sorted, (int)total, 0, 64, st);
        need2 = q->sort_tmp_bytes;
        if (e == hipSuccess) e = hipcub::DeviceSelect::Unique(q->d_sort_tmp, need2, sorted, uniq, (int *)q->d_hash_count, (int)total, st);
    }
    int n_sel = 0;
    uint64_t last = 0;
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(&n_sel, q->d_hash_count, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
    if (e == hipSuccess && !rc && n_sel > 0) e = hipMemcpy(&last, uniq + (n_sel - 1), 8, hipMemcpyDeviceToHost);
    (void)hipFree(sorted);
    if (e != hipSuccess || rc) {
        (void)hipFree(uniq);
        return rc ? rc : hip_fail(e, "hash union sort");
    }
    if (n_sel > 0 && last == kHashEmpty) n_sel--;  // the padding
    *out = uniq;
    *n_out = n_sel;
    return SYBL_OK;
}

}  // namespace sybl
