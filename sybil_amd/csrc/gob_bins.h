// gob_bins.h -- device code shared by gobgpu.hip and kernels.hip: the buckets of a bucket-encoded int column file from the
// values the varint walk found in its `Bins` region (round 6, SYBL_LOADER_GPU_VARINT).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "plan.h"

namespace sybl {

// ---------------------------------------------------------------------------------------------------------------------------
// The buckets of a bucket-encoded int column, from the values of its `Bins` region.  On the wire (encoding/gob, a slice of
// struct{Value int64; Records []uint32}) a bucket is
//     [1][Value]   -- left out when Value is 0 --   [1 or 2][n][n records]   [0]
// (field numbers as deltas, the struct's end as 0), and every one of those is a value the walk found.  Buckets are told apart
// by the zeros: a record is never 0 but for row 0, which is the FIRST record of the one bucket that holds it (ids ascend; with
// DeltaEncodedIDs the later ones are differences >= 1) -- then that bucket's zero-delimited piece stops right behind its count
// ("head only": [1][V][1][n] or [2][n] and nothing else) and the next piece is the rest of its records.  Whether a piece is a
// bucket's start or such a tail follows from the piece before it (tail <=> the one before is a start and head-only), a two-state
// recurrence that one workgroup scans over the <= 8192 zeros of a file.  Every start then knows its rank among the starts (the
// bucket's index), its value, and where its records lie -- n consecutive values, the zero for row 0 among them, which
// k_decode_bins reads where they are.  Checked on the way: the pieces have the lengths their counts announce, there are as
// many buckets as the slice said, they hold as many records as the block's info.db said, the values lie within its bounds.
// One workgroup of 1024 threads per file; runs in the launch that undoes the value-encoded columns' deltas (kernels.hip:
// k_decode_delta_multi -- both wait for the walk only, and a block's chain of kernels on its stream is what bounds a load).
__device__ __forceinline__ void gob_bins_body(const GobBinsJob &J) {
    constexpr int kPer = (kGobMaxBins + 8 + 1023) / 1024;  // zeros per thread, at most
    __shared__ uint32_t wave_fn[16], wave_cnt[16];
    __shared__ unsigned long long wave_recs[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long *tok = J.tok;
    const uint32_t n_tok = (uint32_t)std::min<unsigned long long>(J.state[kGobStateFound], J.tok_cap);
    const uint32_t n_zero = (uint32_t)std::min<unsigned long long>(J.state[kGobStateZeros], J.zpos_cap);
    const uint32_t n_bins = J.n_bins;
    uint32_t bad = 0;
    // piece j = [S, z): z = zpos[j], S = zpos[j - 1] + 1.  As a bucket's start: header length, announced count, head-only?
    uint32_t S[kPer], Z[kPer], hdr[kPer], cnt[kPer];
    bool head[kPer], fine[kPer];
    const uint32_t per = (n_zero + 1023u) / 1024u;  // (consecutive pieces per thread: one, for a file of up to 1024 buckets)
    const uint32_t j0 = tid * per;
    // f: how "this piece is a tail" follows for the NEXT piece -- bit 0: if this one is a start, bit 1: if it is a tail (never)
    uint32_t fn = 0x2u | 0x0u;  // identity: (start -> start: bit 0 = 0, tail -> tail: bit 1 = 1)
    auto compose = [](uint32_t a, uint32_t b) {  // a, then b
        const uint32_t o0 = (a & 1u) ? (b >> 1) & 1u : b & 1u, o1 = (a & 2u) ? (b >> 1) & 1u : b & 1u;
        return o0 | (o1 << 1);
    };
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t j = j0 + e;
        S[e] = Z[e] = hdr[e] = cnt[e] = 0;
        head[e] = fine[e] = false;
        if ((uint32_t)e >= per || j >= n_zero) continue;
        Z[e] = J.zpos[j];
        S[e] = j > 0 ? J.zpos[j - 1] + 1u : 0u;
        if (Z[e] >= n_tok || S[e] > Z[e]) {
            bad |= kGobBadBins;
            continue;
        }
        const uint32_t len = Z[e] - S[e];
        const unsigned long long a = len >= 1 ? tok[S[e]] : 0ull;
        // [1][V][1][n]..., or [2][n]... when Value is 0
        uint32_t h = 0;
        if (a == 1ull && len >= 4 && tok[S[e] + 2] == 1ull) h = 4;
        else if (a == 2ull && len >= 2) h = 2;
        if (h) {
            const unsigned long long n = tok[S[e] + h - 1];
            hdr[e] = h;
            cnt[e] = (uint32_t)std::min<unsigned long long>(n, 0xFFFFFFFFull);
            head[e] = len == h && n >= 1;
            fine[e] = head[e] || (unsigned long long)len == (unsigned long long)h + n;
        }
        // (as a start: the next piece is a tail iff this one is head-only; as a tail: the next one is a start)
        fn = compose(fn, head[e] ? 0x1u : 0x0u);
    }
    // inclusive scan of the composition over the threads; the state a thread's first piece is met in
    uint32_t incl = fn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t prev = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl = compose(prev, incl);
    }
    if (lane == 63) wave_fn[wave] = incl;
    __syncthreads();
    uint32_t before = 0x2u;
    for (uint32_t q = 0; q < wave; q++) before = compose(before, wave_fn[q]);
    uint32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0x2u;
    excl = compose(before, excl);
    bool tail = (excl & 1u) != 0;  // (the first piece of the region is a start: state 0 goes in)
    // the pieces' kinds; the starts counted (a bucket's index) and, for the total, their records
    bool is_tail[kPer];
    uint32_t starts = 0;
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        is_tail[e] = tail;
        if ((uint32_t)e < per && j0 + e < n_zero) {
            starts += tail ? 0u : 1u;
            tail = !tail && head[e];
        }
    }
    uint32_t sincl = starts;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t prev = __shfl_up(sincl, o, 64);
        if ((int)lane >= o) sincl += prev;
    }
    if (lane == 63) wave_cnt[wave] = sincl;
    __syncthreads();
    uint32_t k = sincl - starts;
    for (uint32_t q = 0; q < wave; q++) k += wave_cnt[q];
    uint32_t found = 0;
    for (uint32_t q = 0; q < 16; q++) found += wave_cnt[q];
    unsigned long long recs = 0;
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t j = j0 + e;
        if ((uint32_t)e >= per || j >= n_zero || Z[e] >= n_tok || S[e] > Z[e]) continue;
        if (is_tail[e]) {
            // the rest of the records of the bucket whose head stands before the zero at S - 1: its count is the value before that
            const bool in_use = k >= 1 && k - 1 < n_bins;
            if (in_use && (S[e] < 2 || (unsigned long long)(Z[e] - S[e]) + 1ull != tok[S[e] - 2])) bad |= kGobBadBins;
            continue;
        }
        if (k == n_bins) {
            // behind the last bucket the struct ends: [0], or [2][VERSION][0] (SavedIntColumn's field behind Bins and the -- empty
            // -- Values), and with it the region; anything else the host parser shall look at
            const uint32_t len = Z[e] - S[e];
            if (Z[e] + 1u != n_tok || !(len == 0 || (len == 2 && tok[S[e]] == 2ull))) bad |= kGobBadBins;
        }
        if (k < n_bins) {
            if (!fine[e]) bad |= kGobBadBins;
            const unsigned long long u = hdr[e] == 4 ? tok[S[e] + 1] : 0ull;
            const long long v = (long long)((u >> 1) ^ (0ull - (u & 1ull)));
            if (v < J.chk_min || v > J.chk_max) bad |= kGobOutOfBounds;
            J.bin_val[k] = v;
            const unsigned long long first = (unsigned long long)S[e] + hdr[e], last = first + (fine[e] ? cnt[e] : 0u);
            // (a head-only bucket's records run over the zero into the next piece; its tail checks that they end on a zero)
            const bool within = last <= (unsigned long long)n_tok;
            if (!within) bad |= kGobBadBins;
            J.bin_rng[2 * (size_t)k] = (long long)first;
            J.bin_rng[2 * (size_t)k + 1] = (long long)(within ? last : first);
            recs += within && fine[e] ? cnt[e] : 0u;
        }
        k++;
    }
    // buckets the region does not hold (a damaged file): empty
    for (uint32_t q = found + tid; q < n_bins; q += 1024) {
        J.bin_val[q] = 0;
        J.bin_rng[2 * (size_t)q] = J.bin_rng[2 * (size_t)q + 1] = 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        recs += ((unsigned long long)__shfl_xor((uint32_t)(recs >> 32), o, 64) << 32) | __shfl_xor((uint32_t)recs, o, 64);
        bad |= __shfl_xor(bad, o, 64);
    }
    if (lane == 0) wave_recs[wave] = recs;
    __syncthreads();
    if (tid == 0) {
        unsigned long long all = 0;
        for (int q = 0; q < 16; q++) all += wave_recs[q];
        J.state[kGobStateRecs] = all;
        if (found < n_bins + 1u || all != (unsigned long long)J.n_recs) bad |= kGobBadBins;  // (+ 1: the piece the struct ends with)
    }
    if (lane == 0 && bad) atomicOr(&J.state[kGobStateFlags], (unsigned long long)bad);
}

}  // namespace sybl
