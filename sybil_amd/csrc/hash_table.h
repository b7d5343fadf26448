// hash_table.h -- the open-addressing table every hash group-by kernel shares (k_scan_hash in hashgroup.hip,
// k_scan_hash_fast in hashfast.hip, k_scan_hash_packed in hashpacked.hip): probe limits and find-or-claim.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"
#include "scan_generic.h"

namespace sybl {

constexpr int kHashLdsProbes = 8;  // probes a row spends on the LDS staging table before it goes to HBM

// find the key's slot or claim a free one (linear probing; a claimed slot never changes hands); -1: the table is full.
// The probe count is bounded: with twice as many slots as keys a probe sequence is a handful of slots long, and a key that
// finds neither itself nor a free slot within kHashMaxProbes means the table holds (nearly) as many keys as slots -- the
// query is going to fail with SYBL_E_NOMEM.  The first row that gives up says so in the header, and every later row that
// would have to insert a key gives up at once instead of walking a full table (up to 2^27 device-scope loads per row).
// Default sizing keeps the load factor at or below 1/2.  When the slot count is capped (kHashMaxSlots) or set through
// SYBL_HASH_SLOTS the load factor can approach 1: linear probing's unsuccessful search then walks ~(1 + 1/(1-a)^2)/2
// slots (a = 0.95: 200, 0.98: 1250, 0.99: 5000), so the bound below reports a table as full (SYBL_E_NOMEM) from about
// 98-99 % occupancy on, a little before its last free slot is taken.
constexpr uint32_t kHashMaxProbes = 4096;
__device__ __forceinline__ int32_t hash_find_or_insert(uint64_t *keys, uint32_t mask, uint64_t key, int64_t *hdr) {
    uint32_t h = (uint32_t)(splitmix64(key) >> 32) & mask;
    const uint32_t limit = mask + 1u < kHashMaxProbes ? mask + 1u : kHashMaxProbes;
    for (uint32_t probe = 0; probe < limit; probe++) {
        uint64_t k = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == kHashEmpty) {
            if (__hip_atomic_load(hdr + kHdrHashFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;  // (lost already)
            unsigned long long expect = kHashEmpty;
            if (__hip_atomic_compare_exchange_strong((unsigned long long *)keys + h, &expect, (unsigned long long)key, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                k = key;
            else
                k = expect;
        }
        if (k == key) return (int32_t)h;
        h = (h + 1) & mask;
        // a long walk: has another row already found the table full?
        if ((probe & 31u) == 31u && __hip_atomic_load(hdr + kHdrHashFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;
    }
    __hip_atomic_fetch_add(hdr + kHdrHashFull, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

}  // namespace sybl
