// hashfast.hip -- k_scan_hash_fast: hash group-by behind the role-specialised 64-bit row body (see hashgroup.hip for the
// table levels and the canonical form downstream).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine.h"
#include "hash_table.h"
#include "scan_fast.h"
#include "scan_generic.h"

namespace sybl {

// ---------------------------------------------------------------- the role-specialised form
// k_scan_hash_fast: the same two table levels behind the row body of k_scan_fast<GEN> (scan_fast.h) -- the plan by value
// in scalar registers, roles known at compile time, raw 16-byte buffer loads of any stored width decoded when the tile is
// consumed -- for the queries select_fast_path would take if their keys direct-mapped: range / id-mask / neq filters,
// up to four group columns, up to two aggregations, optional time and weight columns.  Round 2 measured the
// plan-interpreting k_scan_hash at 26 ms per 1e9 rows on config 3's 1024 groups (every row staged in LDS: the
// interpretation, not the table, was the cost).  nf / ng are run-time counts (a wave-uniform `break` per column): one
// instantiation per (aggregations, mode, time) instead of one per column-count combination -- times two: MG = 2 or
// kFastMaxG group columns compiled in (the tiles of every column a kernel could take are register arrays; with four
// group columns next to three filters and two aggregations the 128 registers of a 1024-thread workgroup spilled).
// T threads per workgroup: at 1024 the body is capped at 128 VGPRs and 20 of its 48 instantiations reserve scratch
// (profiles/r05_kernel_resources.txt); at 768 / 512 threads none does (95-163 VGPRs) -- and config 3 through this kernel takes
// 19.0 / 26.8 ms instead of 15.6 (profiles/r06_wg_threads_ab.txt, -DSYBL_THREADS_AB): what the body needs is waves in flight
// (LDS round trips, scalar loads of the plan), not registers.  The spill is the cheaper evil; 1024 it stays.
constexpr int kHashFastThreads = 1024;
template <int NA, int MODE, bool TIME, int MG, int T>
__global__ __launch_bounds__(T) void k_scan_hash_fast(const FastPlan P, uint64_t *hash_keys, const int nf, const int ng, const int L_,
                                                               const int F, const int M) {
    extern __shared__ int64_t lds[];
    __shared__ uint32_t l_used;
    const uint32_t tid = threadIdx.x;
    const uint32_t L = (uint32_t)L_;  // LDS staging slots (a power of two); 0: every row goes to the global table
    uint64_t *lkeys = (uint64_t *)lds;
    int64_t *lsum = lds + L, *lmax = lsum + (size_t)F * L;
    for (uint32_t i = tid; i < L; i += T) lkeys[i] = kHashEmpty;
    for (uint32_t i = tid; i < (uint32_t)F * L; i += T) lsum[i] = 0;
    for (uint32_t i = tid; i < (uint32_t)M * L; i += T) lmax[i] = INT64_MIN;
    if (tid == 0) l_used = 0;
    __syncthreads();
    int64_t *gsum = P.sum_out + kHeaderWords, *gmax = P.max_out;
    const uint32_t gmask = (uint32_t)P.n_cells - 1u;
    const uint32_t lmask = L - 1u, l_limit = L - (L >> 2);

    uint32_t matched = 0, overflow = 0, full = 0;
    auto one_row = [&](const FastTile<kFastMaxF> &f, const FastTile<MG> &g, const FastTile<NA> &a, const FastTile<1> &t,
                       const FastTile<1> &w, const int r) {
        uint64_t key;
        const int st = fast_prepare<kFastMaxF, MG, TIME, true, true>(P, f, g, t, r, nf, ng, key, matched);
        if (st == 0) return;
        if (st == 2) {
            overflow += 1;
            return;
        }
        int32_t ls = -1;
        if (L > 0) {
            // (the low half of the hash: independent of the slot the key gets in the global table)
            uint32_t h = (uint32_t)splitmix64(key) & lmask;
            for (int probe = 0; probe < kHashLdsProbes; probe++) {
                uint64_t k = __hip_atomic_load(lkeys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (k == kHashEmpty) {
                    if (__hip_atomic_load(&l_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= l_limit) break;
                    unsigned long long expect = kHashEmpty;
                    if (__hip_atomic_compare_exchange_strong((unsigned long long *)lkeys + h, &expect, (unsigned long long)key, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        __hip_atomic_fetch_add(&l_used, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            fast_accumulate<NA, MODE, true, true>(P, a, w, r, lsum, lmax, (uint64_t)L, 0u, (uint64_t)(uint32_t)ls, (int64_t)-1, 0u, (int64_t)key, nullptr,
                                                  overflow);
        } else {
            const int32_t gs = hash_find_or_insert(hash_keys, gmask, key, P.sum_out);
            if (gs < 0) {  // more distinct keys than the table holds: reported by finalize
                full += 1;
                return;
            }
            fast_accumulate<NA, MODE, true, false>(P, a, w, r, gsum, gmax, (uint64_t)(uint32_t)P.n_cells, 0u, (uint64_t)(uint32_t)gs, (int64_t)gs, 0u,
                                                   (int64_t)key, nullptr, overflow);
        }
    };

    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        const int64_t end = seg.start + seg.n;
        int64_t row = seg.start + (int64_t)tid * kRowsPerThread;
        FastTile<kFastMaxF> f0;
        FastTile<MG> g0;
        FastTile<NA> a0;
        FastTile<1> t0, w0;
        FastRaw<kFastMaxF> rf;
        FastRaw<MG> rg;
        FastRaw<NA> ra;
        FastRaw<1> rt, rw;
        auto issue = [&](int64_t at) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)at), hi = __builtin_amdgcn_readfirstlane((uint32_t)(at >> 32));
            const int64_t row0 = (int64_t)(((uint64_t)hi << 32) | lo);
            const uint32_t lane_row = (uint32_t)(at - row0);
            const FastPlan &Q = plan_fresh<true>(P);
            if (Q.wcol) fast_issue(Q.wcol, Q.wwid, nullptr, row0, lane_row, rw.v[0], rw.pw[0]);
            if (TIME) fast_issue(Q.tcol, Q.twid, Q.tvalid, row0, lane_row, rt.v[0], rt.pw[0]);
#pragma unroll
            for (int c = 0; c < kFastMaxF; c++)
                if (c < nf) fast_issue(Q.fcol[c], Q.fwid[c], Q.fvalid[c], row0, lane_row, rf.v[c], rf.pw[c]);
#pragma unroll
            for (int c = 0; c < MG; c++)
                if (c < ng) fast_issue(Q.gcol[c], Q.gwid[c], Q.gvalid[c], row0, lane_row, rg.v[c], rg.pw[c]);
#pragma unroll
            for (int c = 0; c < NA; c++) fast_issue(Q.acol[c], Q.awid[c], Q.avalid[c], row0, lane_row, ra.v[c], ra.pw[c]);
        };
        auto decode = [&](int64_t at) {
            const FastPlan &Q = plan_fresh<true>(P);
            if (Q.wcol) fast_decode(Q.wwid, Q.wbase, rw.v[0], rw.pw[0], at, w0.v[0], w0.pop[0]);
            if (TIME) fast_decode(Q.twid, Q.tbase, rt.v[0], rt.pw[0], at, t0.v[0], t0.pop[0]);
#pragma unroll
            for (int c = 0; c < kFastMaxF; c++)
                if (c < nf) fast_decode(Q.fwid[c], Q.fbase[c], rf.v[c], rf.pw[c], at, f0.v[c], f0.pop[c]);
#pragma unroll
            for (int c = 0; c < MG; c++)
                if (c < ng) fast_decode(Q.gwid[c], Q.gbase[c], rg.v[c], rg.pw[c], at, g0.v[c], g0.pop[c]);
#pragma unroll
            for (int c = 0; c < NA; c++) fast_decode(Q.awid[c], Q.abase[c], ra.v[c], ra.pw[c], at, a0.v[c], a0.pop[c]);
        };
        if (row < end) {
            issue(row);
            decode(row);
        }
        for (; row < end; row += (T * kRowsPerThread)) {
            const int64_t nrow = row + (T * kRowsPerThread);
            if (nrow < end) issue(nrow);
            one_row(f0, g0, a0, t0, w0, 0);
            if (row + 1 < end) one_row(f0, g0, a0, t0, w0, 1);
            if (nrow < end) decode(nrow);
        }
    }

    // flush the staging table: one find-or-claim per staged key, one atomic per non-zero field
    if (L > 0) {
        __syncthreads();
        for (uint32_t i = tid; i < L; i += T) {
            const uint64_t k = lkeys[i];
            if (k == kHashEmpty) continue;
            const int32_t gs = hash_find_or_insert(hash_keys, gmask, k, P.sum_out);
            if (gs < 0) {
                full += 1;
                continue;
            }
            for (int fi = 0; fi < F; fi++) {
                const int64_t v = lsum[(size_t)fi * L + i];
                if (v != 0) gadd(gsum + (int64_t)fi * P.n_cells + gs, v);
            }
            for (int m = 0; m < M; m++) {
                const int64_t v = lmax[(size_t)m * L + i];
                if (v != INT64_MIN) __hip_atomic_fetch_max(gmax + (int64_t)m * P.n_cells + gs, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    int64_t m64 = wave_sum((int64_t)matched), o64 = wave_sum((int64_t)overflow), f64 = wave_sum((int64_t)full);
    {
        const int slot[3] = {kHdrMatched, kHdrOverflow, kHdrHashFull};
        const int64_t v[3] = {m64, o64, f64};
        wg_header_add<3>(P.sum_out, slot, v);  // (one atomic per workgroup and counter: scan_generic.h)
    }
}

template <int NA, int MODE, bool TIME>
static hipError_t hash_fast_launch(const FastPlan &P, uint64_t *keys, int nf, int ng, int L, int F, int M, int n_wg, size_t lds_bytes, hipStream_t st) {
    int T = kHashFastThreads;
    auto kfn = ng <= 2 ? k_scan_hash_fast<NA, MODE, TIME, 2, kHashFastThreads> : k_scan_hash_fast<NA, MODE, TIME, kFastMaxG, kHashFastThreads>;
#ifdef SYBL_THREADS_AB
    if (const char *e = env("SYBL_HASH_FAST_THREADS")) T = atoi(e);
    if (T == 768) kfn = ng <= 2 ? k_scan_hash_fast<NA, MODE, TIME, 2, 768> : k_scan_hash_fast<NA, MODE, TIME, kFastMaxG, 768>;
    if (T == 512) kfn = ng <= 2 ? k_scan_hash_fast<NA, MODE, TIME, 2, 512> : k_scan_hash_fast<NA, MODE, TIME, kFastMaxG, 512>;
    if (T != 768 && T != 512) T = kHashFastThreads;
#endif
    hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds_bytes, 16));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), lds_bytes, st, P, keys, nf, ng, L, F, M);
    return hipGetLastError();
}

template <int NA, int MODE>
static hipError_t hash_fast_time(const FastPlan &P, uint64_t *keys, int nf, int ng, bool time, int L, int F, int M, int n_wg, size_t lds, hipStream_t st) {
    return time ? hash_fast_launch<NA, MODE, true>(P, keys, nf, ng, L, F, M, n_wg, lds, st) : hash_fast_launch<NA, MODE, false>(P, keys, nf, ng, L, F, M, n_wg, lds, st);
}

template <int NA>
static hipError_t hash_fast_mode(const FastPlan &P, uint64_t *keys, int nf, int ng, int mode, bool time, int L, int F, int M, int n_wg, size_t lds, hipStream_t st) {
    if (NA == 0) return hash_fast_time<0, kFastAvg>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st);
    switch (mode) {
    case kFastAvg: return hash_fast_time<NA, kFastAvg>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st);
    case kFastAvgMax: return hash_fast_time<NA, kFastAvgMax>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st);
    case kFastMoments: return hash_fast_time<NA, kFastMoments>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st);
    case kFastHist: return hash_fast_time<NA, kFastHist>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_hash_fast(const FastPlan &P, uint64_t *keys, int nf, int ng, int na, int mode, bool time, int L, int F, int M, int n_wg,
                                 size_t lds_bytes, hipStream_t st) {
    switch (na) {
    case 0: return hash_fast_mode<0>(P, keys, nf, ng, mode, time, L, F, M, n_wg, lds_bytes, st);
    case 1: return hash_fast_mode<1>(P, keys, nf, ng, mode, time, L, F, M, n_wg, lds_bytes, st);
    case 2: return hash_fast_mode<2>(P, keys, nf, ng, mode, time, L, F, M, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sybl
