// agree.cpp -- what the ranks of a multi-GPU job settle BEFORE the first query: sybl_table_agree.
//
// Every rank opened its own contiguous range of block directories (sybl_table_open with rank / nranks), so each holds its
// own column extrema, its own first-seen str / set dictionaries and its own distinct values of a sparse group key.  The
// merge of the partial group tables (rccl.cpp) is one SUM (+ MAX) all-reduce only because every rank lays its table out
// alike; this file makes that so, over the ctx's own communicator -- a host needs no collective runtime of its own (the Go
// host has none; rounds 2-5 had this protocol in Python over torch.distributed only: sybil_amd/dist.py).
//
// Reference: sybil's only distributed mechanism merges whole gob results by their translated string keys
// (node_aggregator.go:147-177, aggregate.go:414-467: CombineResults by GroupByKey), so it needs no agreement -- and
// ships every group of every node to one host.  Here keys are digits of a direct-mapped cell, and a digit must mean the
// same value on every GPU.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "engine.h"
#include "rccl_lazy.h"

using namespace sybl;

namespace {

int nccl_fail2(ncclResult_t r, const char *what) { return fail(SYBL_E_NODEVICE, "RCCL error: %s in %s", ncclGetErrorString(r), what); }

// a small device buffer reduced in place from / to host words (one blocking round trip: this is set-up, not the step)
int allreduce_host(Ctx *ctx, int64_t *h, size_t n, ncclRedOp_t op, const char *what) {
    if (n == 0) return SYBL_OK;
    DevOwner own;
    SYBL_HIP(hipMalloc(&own.p, n * 8));
    SYBL_HIP(hipMemcpyAsync(own.p, h, n * 8, hipMemcpyHostToDevice, ctx->stream));
    ncclResult_t nr = ncclAllReduce(own.p, own.p, n, ncclInt64, op, (ncclComm_t)ctx->comm, ctx->stream);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(h, own.p, n * 8, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (nr != ncclSuccess) return nccl_fail2(nr, what);
    if (e != hipSuccess) return hip_fail(e, what);
    return SYBL_OK;
}

// every rank's byte string on every rank: the lengths are all-gathered, then the strings padded to the longest
int allgather_bytes(Ctx *ctx, const std::string &mine, std::vector<std::string> *all, const char *what) {
    const int R = ctx->comm_nranks, me = ctx->comm_rank;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    std::vector<int64_t> len((size_t)R, 0);
    {
        DevOwner own;
        SYBL_HIP(hipMalloc(&own.p, (size_t)R * 8));
        int64_t *d = (int64_t *)own.p;
        const int64_t n = (int64_t)mine.size();
        SYBL_HIP(hipMemcpyAsync(d + me, &n, 8, hipMemcpyHostToDevice, st));
        ncclResult_t nr = ncclAllGather(d + me, d, 1, ncclInt64, comm, st);
        hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(len.data(), d, (size_t)R * 8, hipMemcpyDeviceToHost, st) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (nr != ncclSuccess) return nccl_fail2(nr, what);
        if (e != hipSuccess) return hip_fail(e, what);
    }
    int64_t per = 0;
    for (int r = 0; r < R; r++) per = std::max(per, len[(size_t)r]);
    all->assign((size_t)R, std::string());
    if (per == 0) return SYBL_OK;
    per = (per + 7) / 8 * 8;
    DevOwner own;
    SYBL_HIP(hipMalloc(&own.p, (size_t)per * (size_t)R));
    char *d = (char *)own.p;
    if (!mine.empty()) SYBL_HIP(hipMemcpyAsync(d + (size_t)me * per, mine.data(), mine.size(), hipMemcpyHostToDevice, st));
    ncclResult_t nr = ncclAllGather(d + (size_t)me * per, d, (size_t)per, ncclUint8, comm, st);
    std::vector<char> h((size_t)per * (size_t)R);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(h.data(), d, h.size(), hipMemcpyDeviceToHost, st) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (nr != ncclSuccess) return nccl_fail2(nr, what);
    if (e != hipSuccess) return hip_fail(e, what);
    for (int r = 0; r < R; r++) (*all)[(size_t)r].assign(h.data() + (size_t)r * per, (size_t)len[(size_t)r]);
    return SYBL_OK;
}

// Do all ranks hold the same 62-bit word?  One MAX all-reduce of (x, -x).
int same_everywhere(Ctx *ctx, uint64_t x, bool *same, const char *what) {
    int64_t w[2] = {(int64_t)(x >> 2), -(int64_t)(x >> 2)};
    int rc = allreduce_host(ctx, w, 2, ncclMax, what);
    if (rc) return rc;
    *same = w[0] == -w[1];
    return SYBL_OK;
}

uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

void put_u32(std::string &s, uint32_t v) { s.append((const char *)&v, 4); }

}  // namespace

namespace sybl {

// The planner's test for "this int key goes through a dictionary of its distinct values" (planner.cpp: groups()), shared
// with sybl_table_agree so that the ranks settle a union dictionary for exactly the keys the planner will want one for.
bool group_key_wants_dict(unsigned __int128 card, int64_t cells) {
    return card > ((unsigned __int128)1 << 22) || card * (unsigned __int128)cells > ((unsigned __int128)1 << 27);
}

// First collective of a query (sybl_query_allreduce): every rank must hold the same partial-table layout, or the
// all-reduce would add unrelated words or hang -- the error is raised on EVERY rank instead.  Hashed queries settle their
// sizes by the key union; what must already agree there is the field layout.
int query_check_layout(Query *q) {
    Ctx *ctx = q->ctx;
    if (ctx->comm_nranks <= 1 || q->layout_checked) return SYBL_OK;
    uint64_t h = 0xCBF29CE484222325ull;
    const int64_t words[8] = {q->hash_mode ? -1 : q->n_sum_words, q->hash_mode ? -1 : q->n_max_words, q->plan.n_sum_fields, q->plan.n_max_fields,
                              q->plan.hist_stride, q->hash_mode ? 1 : 0, (int64_t)q->n_distinct, q->hash_mode ? -1 : (int64_t)q->plan.n_cells};
    h = fnv(h, words, sizeof(words));
    bool same = false;
    int rc = same_everywhere(ctx, h, &same, "layout check");
    if (rc) return rc;
    if (!same)
        return fail(SYBL_E_STATE,
                    "partial tables differ across ranks (this rank: %lld SUM words, %lld MAX words, %lld cells): every rank must declare "
                    "the same bounds and dictionaries first -- sybl_table_agree",
                    (long long)q->n_sum_words, (long long)q->n_max_words, (long long)q->plan.n_cells);
    q->layout_checked = true;
    return SYBL_OK;
}

}  // namespace sybl

extern "C" {

int sybl_comm_info(const sybl_ctx *ctx, int32_t *rank, int32_t *nranks) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return fail(SYBL_E_INVAL, "ctx is NULL");
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm_nranks;
    return SYBL_OK;
}

int sybl_table_agree(sybl_table *t, const char *const *group_cols, int32_t n_group_cols) {
    SYBL_API_GUARD(t);
    if (!t || n_group_cols < 0 || (n_group_cols > 0 && !group_cols)) return fail(SYBL_E_INVAL, "sybl_table_agree: bad argument");
    Ctx *ctx = t->ctx;
    SYBL_HIP(hipSetDevice(ctx->device));
    const bool multi = ctx->comm && ctx->comm_nranks > 1;
    int rc = table_ensure_stats(t);
    if (rc) return rc;
    std::vector<Column *> cols;
    for (auto &c : t->cols) cols.push_back(c.get());
    std::sort(cols.begin(), cols.end(), [](const Column *a, const Column *b) { return a->name < b->name; });

    // ---- 0. the same columns, of the same types, on every rank (a rank whose blocks lack a column still declares it:
    // sybl_table_open adds every column the table's info.db names)
    if (multi) {
        uint64_t h = 0xCBF29CE484222325ull;
        for (Column *c : cols) {
            h = fnv(h, c->name.data(), c->name.size() + 1);
            h = fnv(h, &c->type, sizeof(c->type));
        }
        for (int g = 0; g < n_group_cols; g++) h = fnv(h, group_cols[g], strlen(group_cols[g] ? group_cols[g] : "") + 1);
        bool same = false;
        if ((rc = same_everywhere(ctx, h, &same, "schema check"))) return rc;
        if (!same) return fail(SYBL_E_STATE, "sybl_table_agree: the ranks hold different columns (or name different group columns) for table '%s'", t->name.c_str());
    }

    // ---- 1. bounds and has_missing: one MAX all-reduce; minima travel as their bitwise complement (~x = -x - 1 reverses
    // the order and, unlike -x, cannot overflow at INT64_MIN); a flag says whether any rank holds a value at all
    if (multi) {
        const size_t k = cols.size();
        std::vector<int64_t> w(4 * k, INT64_MIN);
        for (size_t i = 0; i < k; i++) {
            Column *c = cols[i];
            // (a str column's ids are about to be renumbered by the union dictionary: no bounds for it, as dist.py had it)
            const bool rows = c->n_pop > 0 && c->exact_min <= c->exact_max;
            const bool has = c->type == SYBL_INT_VAL && (c->bounds_set || rows);
            if (has) {  // (bounds declared earlier only ever widen: a refresh may have brought rows outside them)
                const int64_t lo = !rows ? c->bound_lo : c->bounds_set ? std::min(c->bound_lo, c->exact_min) : c->exact_min;
                const int64_t hi = !rows ? c->bound_hi : c->bounds_set ? std::max(c->bound_hi, c->exact_max) : c->exact_max;
                w[i] = ~lo;
                w[k + i] = hi;
            }
            w[2 * k + i] = c->has_missing ? 1 : 0;
            w[3 * k + i] = has ? 1 : 0;
        }
        if ((rc = allreduce_host(ctx, w.data(), w.size(), ncclMax, "bounds"))) return rc;
        for (size_t i = 0; i < k; i++) {
            Column *c = cols[i];
            if (w[3 * k + i] > 0) {
                c->bounds_set = true;
                c->bound_lo = ~w[i];
                c->bound_hi = w[k + i];
            }
            // (the MISSING key digit / populated-count field must exist on every rank or on none)
            if (w[2 * k + i] > 0) c->has_missing = true;
        }
        t->version++;
    }

    // ---- 2. str / set dictionaries: every rank installs the SORTED union, so an id -- a str group cell, a bit of a
    // per-id filter mask -- means the same string everywhere.  One rank sorts its own: the order of equal-count groups
    // in the output (stable over cell order, aggregate.go:497-525) then does not depend on how many GPUs ran the query.
    for (Column *c : cols) {
        if (c->type == SYBL_INT_VAL) continue;
        std::vector<std::string> uni;
        if (multi) {
            std::string blob;
            for (auto &s : c->dict) {
                put_u32(blob, (uint32_t)s.size());
                blob += s;
            }
            std::vector<std::string> all;
            if ((rc = allgather_bytes(ctx, blob, &all, "dictionaries"))) return rc;
            for (auto &b : all) {
                size_t at = 0;
                while (at + 4 <= b.size()) {
                    uint32_t n;
                    memcpy(&n, b.data() + at, 4);
                    at += 4;
                    if (at + n > b.size()) return fail(SYBL_E_STATE, "sybl_table_agree: damaged dictionary exchange for '%s'", c->name.c_str());
                    uni.emplace_back(b.data() + at, n);
                    at += n;
                }
            }
        } else {
            uni = c->dict;
        }
        std::sort(uni.begin(), uni.end());
        uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
        if (uni == c->dict) continue;  // (already the sorted union: nothing to renumber)
        std::vector<const char *> ptr;
        for (auto &s : uni) ptr.push_back(s.c_str());
        if ((rc = sybl_table_set_dict(t, c->name.c_str(), ptr.data(), (int64_t)ptr.size()))) return rc;
    }

    // ---- 3. sparse int group keys: where the planner will group through a dictionary of distinct values, the digit is the
    // value's rank in the UNION of the ranks' values.  Same walk over the keys as planner.cpp: groups().
    int64_t cells = 1;
    for (int g = 0; g < n_group_cols; g++) {
        Column *c = t->find(group_cols[g]);
        if (!c) return fail(SYBL_E_INVAL, "sybl_table_agree: unknown group column '%s'", group_cols[g] ? group_cols[g] : "(null)");
        if (c->type == SYBL_SET_VAL) return fail(SYBL_E_INVAL, "cannot group by set column '%s' (cmd_query.go:254)", c->name.c_str());
        if ((rc = table_ensure_stats(t))) return rc;
        int64_t lo, hi;
        if (c->bounds_set) {
            lo = c->bound_lo, hi = c->bound_hi;
        } else if (c->type == SYBL_STR_VAL) {
            lo = 0, hi = (int64_t)c->dict.size() - 1;
        } else {
            lo = c->exact_min, hi = c->exact_max;
        }
        if (c->n_pop == 0 && !c->bounds_set && (c->type != SYBL_STR_VAL || c->dict.empty())) lo = 0, hi = -1;
        unsigned __int128 card = hi >= lo ? (unsigned __int128)((__int128)hi - (__int128)lo) + 1 : 0;
        bool dict = false;
        if (multi && c->type == SYBL_INT_VAL && !env("SYBL_NO_GDICT") && (c->gdict_blocks == -2 || group_key_wants_dict(card, cells))) {
            // every rank's distinct values (a rank with more than a dictionary holds says so: then nobody installs one and
            // the planner of every rank takes the hash table)
            c->gdict_refused = false;
            if (c->gdict_blocks == -2) c->gdict_blocks = -1;  // (a union installed earlier: rebuilt from this rank's rows)
            int brc = column_build_gdict(t, c);
            if (brc && brc != SYBL_E_INVAL) return brc;
            if (brc) set_error("%s", "");
            std::string blob;
            if (!brc) blob.assign((const char *)c->gdict.data(), c->gdict.size() * 8);
            int64_t flag[1] = {brc ? 1 : 0};
            if ((rc = allreduce_host(ctx, flag, 1, ncclMax, "group dictionary sizes"))) return rc;
            std::vector<int64_t> uni;
            bool too_many = flag[0] > 0;
            if (!too_many) {
                std::vector<std::string> all;
                if ((rc = allgather_bytes(ctx, blob, &all, "group dictionaries"))) return rc;
                for (auto &b : all) {
                    const size_t n = b.size() / 8, at = uni.size();
                    uni.resize(at + n);
                    if (n) memcpy(uni.data() + at, b.data(), n * 8);
                }
                std::sort(uni.begin(), uni.end());
                uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
                too_many = (int64_t)uni.size() > kDictMaxDistinct;
            }
            if (too_many) {
                c->gdict_refused = true;
                c->gdict_blocks = -1;
                t->version++;
            } else {
                if ((rc = sybl_table_set_group_dict(t, c->name.c_str(), uni.data(), (int64_t)uni.size()))) return rc;
                dict = true;
                card = uni.size();
            }
        }
        if (c->has_missing) {
            bool shares = false;  // (a missing key is the 8-byte image of -1: aggregate.go:31,138)
            if (dict) shares = std::binary_search(c->gdict.begin(), c->gdict.end(), (int64_t)-1);
            else if (c->type == SYBL_INT_VAL && hi >= lo && lo <= -1 && hi >= -1) shares = true;
            if (!shares) card += 1;
        }
        if (card == 0) card = 1;
        // (beyond 2^27 cells the query is hashed -- and the planner narrows every further int key through a dictionary too)
        if (card * (unsigned __int128)cells >= ((unsigned __int128)1 << 62)) break;  // (the planner refuses such a key space)
        cells *= (int64_t)card;
    }
    // (installing a dictionary returned its str column to int32 ids: a compact table packs it again)
    if (t->compact_mode && (rc = sybl_table_compact(t))) return rc;
    return SYBL_OK;
}

}  // extern "C"
