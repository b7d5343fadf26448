// hashpacked.hip -- k_scan_hash_packed: hash group-by (and direct-mapped 3-4 group columns) behind the offset-domain row
// body of k_scan_packed (see hashgroup.hip for the table levels and the canonical form downstream).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine.h"
#include "hash_table.h"
#include "scan_generic.h"
#include "scan_packed.h"

namespace sybl {

// ---------------------------------------------------------------- ... over compact storage, in the offset domain
// k_scan_hash_packed: the row body of k_scan_packed (scan_packed.h: stored 1 / 2 / 4-byte offsets, four rows per lane,
// 32-bit compares and multiplies) in front of the same two table levels, for hashed queries over compact tables whose
// composite key fits 32 bits.  Measured on config 3's 1024 groups forced through the table (1e9 rows, profiles/
// r03_hash_variants.txt): k_scan_hash 26.4 ms (7.1 G vector instructions), k_scan_hash_fast 24.3 ms (8.8 G: the 64-bit
// row body of the GEN kernels alone is 4.4 G), k_scan_packed direct-mapped 2.9 ms (1.0 G).  The staging table is probed
// with a multiplicative 32-bit hash; misses and the flush go through hash_find_or_insert like every other writer of the
// global table.  nf / ng / time are run-time (wave-uniform) so that one instantiation per (aggregations, mode, NUL)
// serves every column count; bucket arrays (hist mode) never stage in LDS and stay with k_scan_hash_fast.
// T threads per workgroup (hashfast.hip: 1024 threads cap the body at 128 VGPRs, and 30 of its 60 instantiations spilled
// inside the row loop).
// SYBL_HASH_PACKED_LATE=0 at build time: without late materialisation (same-box A/B builds)
#ifndef SYBL_HASH_PACKED_LATE
#define SYBL_HASH_PACKED_LATE 1
#endif
constexpr bool kHashPackedLate = SYBL_HASH_PACKED_LATE != 0;

// LATE: the kernel IS the late path (filter columns a tile ahead).  A kernel of its own, not a branch of the plain one: with both
// loops in one body the register allocator served neither -- the plain loop ran at 7.97 ms where it takes 4.62 alone (config 3
// through the table), the late one at 4.85 (profiles/r06_late_path_ab.txt).
template <int NA, int MODE, bool NUL, bool HASH, int T, bool LATE = false>
__global__ __launch_bounds__(T) void k_scan_hash_packed(const FastPlan P, uint64_t *hash_keys, const int nf, const int ng, const int time,
                                                                 const int L_, const int F, const int M) {
    extern __shared__ int64_t lds[];
    __shared__ uint32_t l_used;
    const uint32_t tid = threadIdx.x;
    const uint32_t L = (uint32_t)L_;
    uint64_t *lkeys = (uint64_t *)lds;
    int64_t *lsum = lds + L, *lmax = lsum + (size_t)F * L;
    FastLds DL = {};  // !HASH: the direct-mapped cell table of the k_scan_fast family (replicas, LDS window)
    if (HASH) {
        for (uint32_t i = tid; i < L; i += T) lkeys[i] = kHashEmpty;
        for (uint32_t i = tid; i < (uint32_t)F * L; i += T) lsum[i] = 0;
        for (uint32_t i = tid; i < (uint32_t)M * L; i += T) lmax[i] = INT64_MIN;
        if (tid == 0) l_used = 0;
        __syncthreads();
    } else {
        DL = fast_begin<MODE, T>(P, lds);
    }
    int64_t *gsum = P.sum_out + kHeaderWords, *gmax = P.max_out;
    const uint32_t gmask = (uint32_t)P.n_cells - 1u;
    const uint32_t lmask = L - 1u, l_limit = L - (L >> 2), lshift = L > 1 ? 32u - (uint32_t)__builtin_ctz(L) : 31u;

    uint32_t matched = 0, overflow = 0, full = 0;
    constexpr int MF = kFastMaxF, MG = HASH ? 2 : kFastMaxG;  // (hashed: the planner takes at most two group columns here)
    auto accumulate = [&](auto lds_tag, int64_t *tab, int64_t *maxtab, const uint32_t ncell, const uint32_t slot, const PackedTile<NA> &a, const int r,
                          const int64_t logkey) {
        constexpr bool LDS = decltype(lds_tag)::value;
        fast_add64<LDS>(tab, slot, 1);  // Result.Count++ (aggregate.go:203)
#pragma unroll
        for (int c = 0; c < NA; c++) {
            const uint32_t u = a.u[c][r];
            if (NUL) {
                if (!((a.pop[c] >> r) & 1u)) continue;  // no value: no hist for this row
                if (P.f_pop[c] >= 0) fast_add64<LDS>(tab, (uint64_t)((uint32_t)P.f_pop[c] * ncell + slot), 1);
                if (P.f_cnt[c] >= 0) {
                    if (u < P.alo[c] || u > P.ahi[c]) continue;  // hist_basic.go:104, rebased
                    fast_add64<LDS>(tab, (uint64_t)((uint32_t)P.f_cnt[c] * ncell + slot), 1);
                }
            }
            const int64_t x = (int64_t)((uint64_t)P.abase[c] + u);
            fast_add64<LDS>(tab, (uint64_t)(uint32_t)P.f_sum[c] * ncell + slot, x);
            if (MODE == kFastAvgMax) {
                if (!P.ext_general) {
                    fast_max64<LDS>(maxtab, (uint64_t)(uint32_t)P.m_max[c] * ncell + slot, x);
                } else {
                    if (P.m_max[c] >= 0) fast_max64<LDS>(maxtab, (uint64_t)(uint32_t)P.m_max[c] * ncell + slot, x);
                    if (P.m_nmin[c] >= 0) fast_max64<LDS>(maxtab, (uint64_t)(uint32_t)P.m_nmin[c] * ncell + slot, x == INT64_MIN ? INT64_MAX : -x);
                }
            }
            if (MODE == kFastMoments) {
                uint32_t b = packed_udiv(u + P.adoff[c], P.bucket_size[c], P.pinv_bucket[c]);
                if (NUL && b >= (uint32_t)P.n_values[c]) {
                    // Outlier (hist_basic.go:132-135): as in packed_row (scan_packed.h)
                    if (P.f_out[c] >= 0) {
                        const unsigned __int128 sq = (unsigned __int128)((__int128)x * (__int128)x);
                        const uint64_t fo = (uint64_t)(uint32_t)P.f_out[c] * ncell + slot;
                        fast_add64<LDS>(tab, fo, 1);
                        fast_add64<LDS>(tab, fo + ncell, x);
                        fast_add64<LDS>(tab, fo + 2 * (uint64_t)ncell, (int64_t)(uint64_t)(sq & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fo + 3 * (uint64_t)ncell, (int64_t)(uint64_t)((sq >> 32) & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fo + 4 * (uint64_t)ncell, (int64_t)(uint64_t)((sq >> 64) & 0xFFFFFFFFu));
                        fast_add64<LDS>(tab, fo + 5 * (uint64_t)ncell, (int64_t)(uint64_t)(sq >> 96));
                        if (P.out_log) log_outlier(P.out_log, P.out_cap, logkey, c, x);
                    } else {
                        overflow += 1;
                    }
                    b = (uint32_t)P.n_values[c] - 1;
                }
                fast_add64<LDS>(tab, (uint64_t)(uint32_t)P.f_sb[c] * ncell + slot, (int64_t)(uint64_t)b);
                fast_add64<LDS>(tab, (uint64_t)(uint32_t)P.f_sb2[c] * ncell + slot, (int64_t)(uint64_t)(uint32_t)__umul24(b, b));
            }
        }
    };
    // (nfe: filter columns still to be evaluated for the row -- nf, or 0 when `pass` already is their verdict: the late path)
    auto one_row = [&](const PackedTile<MF> &f, const PackedTile<MG> &g, const PackedTile<NA> &a, const PackedTile<1> &t, const int r, bool pass,
                       const uint32_t xpop, const int nfe) {
        if (NUL) pass = pass & ((xpop >> r) & 1u);  // the filter pre-pass's verdict (FastPlan::xvalid)
        // (packed_row's filters and key, scan_packed.h: one predicate, no short-circuit)
#pragma unroll
        for (int c = 0; c < MF; c++) {
            if (c >= nfe) break;
            const uint32_t u = f.u[c][r];
            bool ok = (u >= P.plo[c]) & (u <= P.phi[c]);
            if (NUL) {
                if (P.fmask[c]) {
                    const uint32_t id = u + (uint32_t)P.fbase[c];
                    ok = id < (uint32_t)P.fmask_bits[c];
                    if (ok) ok = (P.fmask[c][id >> 5] >> (id & 31)) & 1u;
                }
                for (int k = 0; k < P.npneq[c]; k++) ok = ok & (u != P.pneq[c][k]);
                ok = ok & ((f.pop[c] >> r) & 1u);
            }
            pass = pass & ok;
        }
        uint32_t key = 0;
        bool inb = true;
#pragma unroll
        for (int c = 0; c < MG; c++) {
            if (c >= ng) break;
            const uint32_t d = g.u[c][r] + P.gdoff[c];
            if (NUL) {
                const bool p = (g.pop[c] >> r) & 1u;
                inb = inb & (p ? d < (uint32_t)P.gvalues64[c] : P.gmissing64[c] >= 0);
                key += p ? d * (uint32_t)P.gstride64[c] : (uint32_t)P.gmissing64[c];
            } else {
                inb = inb & (d < (uint32_t)P.gvalues64[c]);
                key += d * (uint32_t)P.gstride64[c];
            }
        }
        bool live = pass;
        if (time) {
            const uint32_t tb = packed_udiv(t.u[0][r] + P.tdoff, (uint32_t)P.time_bucket, P.pinv_time);
            if (NUL) {
                const bool tp = (t.pop[0] >> r) & 1u;
                live = live & tp;
                inb = inb & (tb < (uint32_t)P.n_tb || !tp);
            } else {
                inb = inb & (tb < (uint32_t)P.n_tb);
            }
            key += tb * (uint32_t)P.tb_stride64;
        }
        matched += pass ? 1u : 0u;
        if (!HASH) {
            // direct-mapped: the key IS the cell (aggregate.go:125-143), inside this workgroup's LDS table / window
            const uint32_t lcell = key - DL.cell_base;
            inb = inb & (lcell < DL.tab_cells);
            overflow += (live & !inb) ? 1u : 0u;
            if (!(live & inb)) return;
            const uint32_t rs = (uint32_t)P.rep_shift;
            accumulate(std::true_type{}, lds, lds + DL.max_base, DL.tab_cells << rs, (lcell << rs) + DL.rep, a, r, (int64_t)key);
            return;
        }
        overflow += (live & !inb) ? 1u : 0u;
        if (!(live & inb)) return;
        int32_t ls = -1;
        if (L > 0) {
            uint32_t h = (key * 0x9E3779B1u) >> lshift;
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
                if (k == (uint64_t)key) {
                    ls = (int32_t)h;
                    break;
                }
                h = (h + 1) & lmask;
            }
        }
        if (ls >= 0) {
            accumulate(std::true_type{}, lsum, lmax, L, (uint32_t)ls, a, r, (int64_t)key);
        } else {
            const int32_t gs = hash_find_or_insert(hash_keys, gmask, (uint64_t)key, P.sum_out);
            if (gs < 0) {
                full += 1;
                return;
            }
            accumulate(std::false_type{}, gsum, gmax, (uint32_t)P.n_cells, (uint32_t)gs, a, r, (int64_t)key);
        }
    };

    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            PackedRaw<MF> rf;
            PackedRaw<MG> rg;
            PackedRaw<NA> ra;
            PackedRaw<1> rt;
            PackedTile<MF> f;
            PackedTile<MG> g;
            PackedTile<NA> a;
            PackedTile<1> t;
            // (the filter pre-pass's bitmap: as in k_scan_packed -- the word of the tile after next is fetched a tile early and
            // a wave with no passing row in a tile loads nothing of it)
            uint32_t xpop = 0xFu, xw_n = 0xFFFFFFFFu;
            const bool xv = NUL && P.xvalid != nullptr;
            auto issue = [&](uint32_t r) {
                const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
                const uint32_t lane_row = r - r0;
                auto ld = [&](const void *col, int width, pu32x4 &raw) {
                    const int ws = width >> 1;
                    packed_issue((const uint8_t *)col + (size_t)(first + r0) * (size_t)width, ws, lane_row << ws, raw);
                };
                if (NUL) {
                    const int64_t wd = (first + r) >> 5;
                    if (time) rt.pw[0] = P.tvalid ? P.tvalid[wd] : 0xFFFFFFFFu;
#pragma unroll
                    for (int c = 0; c < MF; c++)
                        if (c < nf) rf.pw[c] = P.fvalid[c] ? P.fvalid[c][wd] : 0xFFFFFFFFu;
#pragma unroll
                    for (int c = 0; c < MG; c++)
                        if (c < ng) rg.pw[c] = P.gvalid[c] ? P.gvalid[c][wd] : 0xFFFFFFFFu;
#pragma unroll
                    for (int c = 0; c < NA; c++) ra.pw[c] = P.avalid[c] ? P.avalid[c][wd] : 0xFFFFFFFFu;
                }
                if (time) ld(P.tcol, P.twid, rt.v[0]);
#pragma unroll
                for (int c = 0; c < MF; c++)
                    if (c < nf) ld(P.fcol[c], P.fwid[c], rf.v[c]);
#pragma unroll
                for (int c = 0; c < MG; c++)
                    if (c < ng) ld(P.gcol[c], P.gwid[c], rg.v[c]);
#pragma unroll
                for (int c = 0; c < NA; c++) ld(P.acol[c], P.awid[c], ra.v[c]);
            };
            auto decode = [&](uint32_t r) {
                const uint32_t bit0 = (uint32_t)(first + r) & 31u;
                if (NUL) {
                    if (time) t.pop[0] = (rt.pw[0] >> bit0) & 0xFu;
#pragma unroll
                    for (int c = 0; c < MF; c++)
                        if (c < nf) f.pop[c] = (rf.pw[c] >> bit0) & 0xFu;
#pragma unroll
                    for (int c = 0; c < MG; c++)
                        if (c < ng) g.pop[c] = (rg.pw[c] >> bit0) & 0xFu;
#pragma unroll
                    for (int c = 0; c < NA; c++) a.pop[c] = (ra.pw[c] >> bit0) & 0xFu;
                }
                if (time) packed_decode(P.twid, rt.v[0], t.u[0]);
#pragma unroll
                for (int c = 0; c < MF; c++)
                    if (c < nf) packed_decode(P.fwid[c], rf.v[c], f.u[c]);
#pragma unroll
                for (int c = 0; c < MG; c++)
                    if (c < ng) packed_decode(P.gwid[c], rg.v[c], g.u[c]);
#pragma unroll
                for (int c = 0; c < NA; c++) packed_decode(P.awid[c], ra.v[c], a.u[c]);
            };
            if (kHashPackedLate && LATE && !NUL) {
                // Late materialisation (the reference's row loop leaves a row at its first failing filter, aggregate.go:105-116),
                // as in k_scan_packed (scan_packed.h): the filter columns run one tile ahead of the key / aggregation / time
                // columns, and a WAVE none of whose 256 rows passes does not read the other columns of that tile -- the loads
                // are still issued, through a descriptor of zero records, so the count in flight is the same on every path.
                // Iteration t: decode keys / values of tile t and the filters of t + 1, issue keys / values of t + 1 (or
                // nothing) and the filters of t + 2, then the rows of t with the predicate bits kept from last time.
                const uint32_t r_first = tid * kPackedRows;
                const uint32_t n_tiles = (n + (uint32_t)(T * kPackedRows) - 1) / (uint32_t)(T * kPackedRows);
                auto ldn = [&](const void *col, int width, pu32x4 &raw, uint32_t r0, uint32_t lane_row, uint32_t rows) {
                    const int ws = width >> 1;
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)((const uint8_t *)col + (size_t)(first + (rows ? r0 : 0u)) * (size_t)width), 0, (int)(rows << ws), (int)kBufferRsrcWord3);
                    raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_row << ws), 0, 2);
                };
                auto rows_of = [&](uint32_t r0) -> uint32_t {
                    return r0 < n ? (n - r0 < 64u * kPackedRows ? (n - r0 + kPackedRows - 1) & ~(uint32_t)(kPackedRows - 1) : 64u * kPackedRows) : 0u;
                };
                auto issue_filters = [&](uint32_t r) {
                    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r), lane_row = r - r0, rows = rows_of(r0);
#pragma unroll
                    for (int c = 0; c < MF; c++)
                        if (c < nf) ldn(P.fcol[c], P.fwid[c], rf.v[c], r0, lane_row, rows);
                };
                auto issue_rest = [&](uint32_t r, bool wanted) {
                    const uint32_t r0 = __builtin_amdgcn_readfirstlane(r), lane_row = r - r0, rows = wanted ? rows_of(r0) : 0u;
                    if (time) ldn(P.tcol, P.twid, rt.v[0], r0, lane_row, rows);
#pragma unroll
                    for (int c = 0; c < MG; c++)
                        if (c < ng) ldn(P.gcol[c], P.gwid[c], rg.v[c], r0, lane_row, rows);
#pragma unroll
                    for (int c = 0; c < NA; c++) ldn(P.acol[c], P.awid[c], ra.v[c], r0, lane_row, rows);
                };
                auto filter_bits = [&](uint32_t r) -> uint32_t {
#pragma unroll
                    for (int c = 0; c < MF; c++)
                        if (c < nf) packed_decode(P.fwid[c], rf.v[c], f.u[c]);
                    const uint32_t left = r < n ? n - r : 0u;
                    uint32_t bits = 0;
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) {
                        bool pass = (uint32_t)k < left;
#pragma unroll
                        for (int c = 0; c < MF; c++)
                            if (c < nf) pass = pass & (f.u[c][k] >= P.plo[c]) & (f.u[c][k] <= P.phi[c]);
                        bits |= pass ? 1u << k : 0u;
                    }
                    return bits;
                };
                issue_filters(r_first);
                uint32_t bits = filter_bits(r_first);
                issue_rest(r_first, __builtin_amdgcn_ballot_w64(bits != 0) != 0);
                issue_filters(r_first + (uint32_t)(T * kPackedRows));
                for (uint32_t it = 0; it < n_tiles; it++) {
                    const uint32_t r = r_first + it * (uint32_t)(T * kPackedRows);
                    if (time) packed_decode(P.twid, rt.v[0], t.u[0]);
#pragma unroll
                    for (int c = 0; c < MG; c++)
                        if (c < ng) packed_decode(P.gwid[c], rg.v[c], g.u[c]);
#pragma unroll
                    for (int c = 0; c < NA; c++) packed_decode(P.awid[c], ra.v[c], a.u[c]);
                    const uint32_t next_bits = filter_bits(r + (uint32_t)(T * kPackedRows));
                    issue_rest(r + (uint32_t)(T * kPackedRows), __builtin_amdgcn_ballot_w64(next_bits != 0) != 0);
                    issue_filters(r + 2u * (uint32_t)(T * kPackedRows));
#pragma unroll
                    for (int k = 0; k < kPackedRows; k++) one_row(f, g, a, t, k, (bits >> k) & 1u, 0xFu, 0);
                    bits = next_bits;
                }
                continue;
            }
            uint32_t r = tid * kPackedRows;
            if (r < n) {
                issue(r);
                if (xv) {
                    xpop = (P.xvalid[(first + r) >> 5] >> ((uint32_t)(first + r) & 31u)) & 0xFu;
                    if (r + (uint32_t)(T * kPackedRows) < n) xw_n = P.xvalid[(first + r + (uint32_t)(T * kPackedRows)) >> 5];
                }
                decode(r);
            }
            for (; r < n; r += (uint32_t)(T * kPackedRows)) {
                const uint32_t rn = r + (uint32_t)(T * kPackedRows);
                const uint32_t xpop_n = xv ? (xw_n >> ((uint32_t)(first + rn) & 31u)) & 0xFu : 0xFu;
                const bool more = rn < n && (!xv || __builtin_amdgcn_ballot_w64(xpop_n != 0) != 0);  // (wave-uniform)
                if (more) issue(rn);
                if (xv && rn + (uint32_t)(T * kPackedRows) < n) xw_n = P.xvalid[(first + rn + (uint32_t)(T * kPackedRows)) >> 5];
                const uint32_t left = n - r;
#pragma unroll
                for (int k = 0; k < kPackedRows; k++) one_row(f, g, a, t, k, (uint32_t)k < left, xpop, nf);
                if (more) decode(rn);
                xpop = xpop_n;
            }
        }
    }

    if (!HASH) {
        fast_finish<T>(P, lds, DL, matched, overflow);
        return;
    }
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

template <int NA, int MODE, bool NUL, bool HASH = true>
static hipError_t hash_packed_launch(const FastPlan &P, uint64_t *keys, int nf, int ng, int time, int L, int F, int M, int n_wg, size_t lds_bytes,
                                     hipStream_t st) {
    int T = 1024;
    auto kfn = k_scan_hash_packed<NA, MODE, NUL, HASH, 1024>;
    if constexpr (kHashPackedLate && !NUL) {
        if (nf > 0 && P.late) kfn = k_scan_hash_packed<NA, MODE, NUL, HASH, 1024, true>;
    }
#ifdef SYBL_THREADS_AB  // (hashfast.hip: fewer threads, no spills -- and slower)
    if (const char *e = env("SYBL_HASH_PACKED_THREADS")) T = atoi(e);
    if (T == 768) kfn = k_scan_hash_packed<NA, MODE, NUL, HASH, 768>;
    else if (T == 512) kfn = k_scan_hash_packed<NA, MODE, NUL, HASH, 512>;
    else T = 1024;
#endif
    hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds_bytes, 16));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(T), lds_bytes, st, P, keys, nf, ng, time, L, F, M);
    return hipGetLastError();
}

template <int NA>
static hipError_t hash_packed_mode(const FastPlan &P, uint64_t *keys, int nf, int ng, int mode, int time, int L, int F, int M, int n_wg, size_t lds,
                                   hipStream_t st) {
    const bool nul = P.nul != 0;
#define SYBL_HP(MODE_) \
    return nul ? hash_packed_launch<NA, MODE_, true>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st) \
               : hash_packed_launch<NA, MODE_, false>(P, keys, nf, ng, time, L, F, M, n_wg, lds, st)
    if (NA == 0) SYBL_HP(kFastAvg);
    switch (mode) {
    case kFastAvg: SYBL_HP(kFastAvg);
    case kFastAvgMax: SYBL_HP(kFastAvgMax);
    case kFastMoments: SYBL_HP(kFastMoments);
    default: return hipErrorInvalidValue;
    }
#undef SYBL_HP
}

// The same row body direct-mapped (HASH = false): what k_scan_packed does, for the column counts it is not instantiated
// for -- three or four group columns, three or four aggregation columns -- with run-time filter / group column counts.
// (plan_fresh, scan_fast.h, was tried on this kernel: config 3 through the table 4.8 -> 5.6 ms.  Its 230-500 spilled scalar
// registers cost less than the scalar loads and their waits between the LDS atomics; the same on k_scan_packed: 2.72 ->
// 3.32 ms, SYBL_PACKED_RING=4.)
template <int NA>
static hipError_t packed_n_mode(const FastPlan &P, int nf, int ng, int mode, int time, int n_wg, size_t lds, hipStream_t st) {
    const bool nul = P.nul != 0;
#define SYBL_PN(MODE_) \
    return nul ? hash_packed_launch<NA, MODE_, true, false>(P, nullptr, nf, ng, time, 0, 0, 0, n_wg, lds, st) \
               : hash_packed_launch<NA, MODE_, false, false>(P, nullptr, nf, ng, time, 0, 0, 0, n_wg, lds, st)
    if (NA == 0) SYBL_PN(kFastAvg);
    switch (mode) {
    case kFastAvg: SYBL_PN(kFastAvg);
    case kFastAvgMax: SYBL_PN(kFastAvgMax);
    case kFastMoments: SYBL_PN(kFastMoments);
    default: return hipErrorInvalidValue;
    }
#undef SYBL_PN
}

// k_prefilter_packed: the filter pre-pass (planner.cpp: Planner::prefilter) for plain filter columns over compact storage
// -- ranges, neq constants, dictionary-id masks, validity: what packed_row<NUL> evaluates, four rows per lane in the offset
// domain -- instead of the generic two-rows-per-lane tile (k_prefilter, kernels.hip: 2.4 ms per 1e9 rows of one 2-byte
// column).  Up to kFastMaxF columns per launch; AND: a later launch of the same pre-pass narrows the bitmap of an earlier
// one.  A lane's four rows are a nibble of the word eight lanes share: three butterfly steps, the first lane stores.
template <bool AND>
__global__ __launch_bounds__(kWgThreads) void k_prefilter_packed(const FastPlan P, const int nf, uint32_t *bits) {
    const uint32_t tid = threadIdx.x;
    constexpr int MF = kFastMaxF;
    const int s0 = P.wg_seg_begin[blockIdx.x], s1 = P.wg_seg_begin[blockIdx.x + 1];
    for (int si = s0; si < s1; si++) {
        const Segment seg = P.segs[si];
        for (int64_t c0 = 0; c0 < seg.n; c0 += kPackedChunkRows) {
            const int64_t first = seg.start + c0;
            const uint32_t n = (uint32_t)(seg.n - c0 < kPackedChunkRows ? seg.n - c0 : kPackedChunkRows);
            PackedRaw<MF> rf;
            PackedTile<MF> f;
            auto issue = [&](uint32_t r) {
                const uint32_t r0 = __builtin_amdgcn_readfirstlane(r);
                const uint32_t lane_row = r - r0;
                const int64_t wd = (first + r) >> 5;
#pragma unroll
                for (int c = 0; c < MF; c++)
                    if (c < nf) {
                        rf.pw[c] = P.fvalid[c] ? P.fvalid[c][wd] : 0xFFFFFFFFu;
                        const int ws = P.fwid[c] >> 1;
                        packed_issue((const uint8_t *)P.fcol[c] + (size_t)(first + r0) * (size_t)P.fwid[c], ws, lane_row << ws, rf.v[c]);
                    }
            };
            auto decode = [&](uint32_t r) {
                const uint32_t bit0 = (uint32_t)(first + r) & 31u;
#pragma unroll
                for (int c = 0; c < MF; c++)
                    if (c < nf) {
                        f.pop[c] = (rf.pw[c] >> bit0) & 0xFu;
                        packed_decode(P.fwid[c], rf.v[c], f.u[c]);
                    }
            };
            uint32_t r = tid * kPackedRows;
            if (r < n) {
                issue(r);
                decode(r);
            }
            for (; r < n; r += kPackedTileRows) {
                const uint32_t rn = r + kPackedTileRows;
                const bool more = rn < n;
                if (more) issue(rn);
                const uint32_t left = n - r;
                uint32_t nib = 0;
#pragma unroll
                for (int k = 0; k < kPackedRows; k++) {
                    bool pass = (uint32_t)k < left;
#pragma unroll
                    for (int c = 0; c < MF; c++) {
                        if (c >= nf) break;
                        const uint32_t u = f.u[c][k];
                        bool ok = (u >= P.plo[c]) & (u <= P.phi[c]);
                        if (P.fmask[c]) {
                            const uint32_t id = u + (uint32_t)P.fbase[c];
                            ok = id < (uint32_t)P.fmask_bits[c];
                            if (ok) ok = (P.fmask[c][id >> 5] >> (id & 31)) & 1u;
                        }
                        for (int j = 0; j < P.npneq[c]; j++) ok = ok & (u != P.pneq[c][j]);
                        ok = ok & ((f.pop[c] >> k) & 1u);
                        pass = pass & ok;
                    }
                    nib |= pass ? 1u << k : 0u;
                }
                uint32_t v = nib << ((uint32_t)(first + r) & 31u);
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) v |= __shfl_xor(v, o, 64);
                if ((tid & 7u) == 0) {
                    uint32_t *w = bits + ((first + r) >> 5);
                    *w = AND ? (*w & v) : v;
                }
                if (more) decode(rn);
            }
        }
    }
}

hipError_t launch_prefilter_packed(const FastPlan &P, int nf, uint32_t *bits, bool and_into, int n_wg, hipStream_t st) {
    if (and_into) hipLaunchKernelGGL(k_prefilter_packed<true>, dim3(n_wg), dim3(kWgThreads), 0, st, P, nf, bits);
    else hipLaunchKernelGGL(k_prefilter_packed<false>, dim3(n_wg), dim3(kWgThreads), 0, st, P, nf, bits);
    return hipGetLastError();
}

hipError_t launch_scan_packed_n(const FastPlan &P, int nf, int ng, int na, int mode, bool time, int n_wg, size_t lds_bytes, hipStream_t st) {
    switch (na) {
    case 0: return packed_n_mode<0>(P, nf, ng, mode, time ? 1 : 0, n_wg, lds_bytes, st);
    case 1: return packed_n_mode<1>(P, nf, ng, mode, time ? 1 : 0, n_wg, lds_bytes, st);
    case 2: return packed_n_mode<2>(P, nf, ng, mode, time ? 1 : 0, n_wg, lds_bytes, st);
    case 3: return packed_n_mode<3>(P, nf, ng, mode, time ? 1 : 0, n_wg, lds_bytes, st);
    case 4: return packed_n_mode<4>(P, nf, ng, mode, time ? 1 : 0, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_hash_packed(const FastPlan &P, uint64_t *keys, int nf, int ng, int na, int mode, bool time, int L, int F, int M, int n_wg,
                                   size_t lds_bytes, hipStream_t st) {
    switch (na) {
    case 0: return hash_packed_mode<0>(P, keys, nf, ng, mode, time ? 1 : 0, L, F, M, n_wg, lds_bytes, st);
    case 1: return hash_packed_mode<1>(P, keys, nf, ng, mode, time ? 1 : 0, L, F, M, n_wg, lds_bytes, st);
    case 2: return hash_packed_mode<2>(P, keys, nf, ng, mode, time ? 1 : 0, L, F, M, n_wg, lds_bytes, st);
    case 3: return hash_packed_mode<3>(P, keys, nf, ng, mode, time ? 1 : 0, L, F, M, n_wg, lds_bytes, st);
    case 4: return hash_packed_mode<4>(P, keys, nf, ng, mode, time ? 1 : 0, L, F, M, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sybl
