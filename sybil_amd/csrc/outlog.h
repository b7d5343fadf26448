// outlog.h -- appending to the outlier log (plan.h) from any scan kernel's row body.
// Reference: BasicHist.AddWeightedValue remembers the values it clips (hist_basic.go:132-142).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"

namespace sybl {

#ifdef __HIPCC__
// one record of the outlier log (plan.h), called by whichever lanes of a wave hold an outlier at this point of the row
// body.  Round 3 took every record's place from ONE cursor: 10^7 outliers were 2^20 same-address device-scope atomics
// (~45 ms) before the log was full.  Now (a) the lanes that are here together reserve their places with one atomic
// (ballot of the active lanes, the first one adds their number, the others take their rank's offset), and (b) the wave
// appends to one of kOutStripes staging stripes, each behind a cursor on a line of its own, chosen by workgroup and wave
// number -- with ~1 % outliers only one or two lanes of a wave meet here, so it is the stripes that spread the
// contention.  Round 5: a stripe that is full (noticed with a load) or that fills up under the reservation passes the
// lanes it has no room for on to the next stripe, up to kOutSpill of them -- skewed outliers (one block, one time range)
// no longer lose values while the log as a whole has room; a record that found no place after that raises a FLAG in its
// home stripe's second word (a plain store, once: k_outlog_gather only has to know THAT values were dropped), and every
// later outlier of that stripe's waves leaves at the first load -- 1e7 outliers against a 2^20-record log cost what they
// did with the one-load test (counting the dropped ones with an atomic each: 8.1 -> 24.5 ms on config 4 with
// -hist-bucket 990, profiles/r05_wide_aggs.txt before / after).
constexpr int kOutSpill = 8;
__device__ __forceinline__ void log_outlier(int64_t *stage, int64_t cap, int64_t where, int agg, int64_t value) {
    const uint32_t home = ((blockIdx.x * 16u + (threadIdx.x >> 6)) * 0x9E3779B1u) >> 26;  // (wave-uniform; kOutStripes = 64)
    static_assert(kOutStripes == 64, "six hash bits pick the stripe");
    const int64_t per = cap / kOutStripes;
    int64_t *dropped = stage + (size_t)home * kOutCursorWords + 1;
    if (__hip_atomic_load(dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;  // (this neighbourhood is full)
    bool pending = true;
#pragma unroll 1
    for (int t = 0; t < kOutSpill; t++) {
        const uint32_t stripe = (home + (uint32_t)t) & (kOutStripes - 1);
        int64_t *cursor = stage + (size_t)stripe * kOutCursorWords;
        if (__hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per) continue;
        const unsigned long long here = __builtin_amdgcn_ballot_w64(true);  // the lanes still looking for a place together
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
        uint32_t lo = 0, hi = 0;
        if (rank == 0) {
            const int64_t b = __hip_atomic_fetch_add(cursor, (int64_t)__builtin_popcountll(here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = (uint32_t)(uint64_t)b;
            hi = (uint32_t)((uint64_t)b >> 32);
        }
        lo = __builtin_amdgcn_readfirstlane(lo);  // (the first active lane is the one of rank 0)
        hi = __builtin_amdgcn_readfirstlane(hi);
        const int64_t i = (int64_t)((uint64_t)lo | (uint64_t)hi << 32) + (int64_t)rank;
        if (i < per) {
            int64_t *rec = stage + (size_t)kOutStripes * kOutCursorWords + ((size_t)stripe * (size_t)per + (size_t)i) * kOutLogWords;
            rec[0] = where;
            rec[1] = agg;
            rec[2] = value;
            pending = false;
            break;
        }
    }
    if (pending) __hip_atomic_store(dropped, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif  // __HIPCC__

}  // namespace sybl
