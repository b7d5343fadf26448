// table.cpp -- HBM-resident tables: column arrays, validity bitmaps, dictionaries, set-column
// CSR, block statistics, the block writer and the table half of the C ABI (include/sybilgpu.h).
//
// Reference mapping (src/lib/): Table/TableBlock/Record slabs (table.go, table_block.go,
// record_slab.go:5-122) become dense per-column device arrays; LoadBlockFromDir's product
// (table_block_io.go:225-310) arrives through sybl_table_append_block or loader.cpp.
#include <string.h>

#include <algorithm>

#include "engine.h"

namespace sybl {

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Every copy of caller / heap memory into HBM on the append path (engine.h).  The bytes cross in pieces of the ctx's pinned
// staging buffer -- the runtime's own staging of pageable hipMemcpyAsync sources is out of the picture --, each piece complete
// before the buffer is reused.
int host_to_device(Ctx *ctx, void *dst, const void *src, size_t bytes, const char *what) {
    if (bytes == 0) return SYBL_OK;
    constexpr size_t kPiece = (size_t)4 << 20;
    hipStream_t st = ctx->stream;
    if (!ctx->h2d_stage) {
        SYBL_HIP(hipHostMalloc((void **)&ctx->h2d_stage, kPiece, hipHostMallocDefault));
        ctx->h2d_stage_bytes = kPiece;
    }
    for (size_t at = 0; at < bytes; at += kPiece) {
        const size_t n = std::min(kPiece, bytes - at);
        memcpy(ctx->h2d_stage, (const char *)src + at, n);
        SYBL_HIP(hipMemcpyAsync((char *)dst + at, ctx->h2d_stage, n, hipMemcpyHostToDevice, st));
        SYBL_HIP(hipStreamSynchronize(st));
    }
    if (!env("SYBL_VERIFY_COPIES")) return SYBL_OK;
    if (bytes % 4) return fail(SYBL_E_INVAL, "host_to_device(%s): %zu bytes are not whole words", what, bytes);
    if (!ctx->d_copy_digest) SYBL_HIP(hipMalloc((void **)&ctx->d_copy_digest, 8));
    // (test hook: one word of the destination is overwritten behind the copy -- what the guard is there to notice)
    if (env("SYBL_VERIFY_COPIES_FAULT")) SYBL_HIP(hipMemsetAsync((char *)dst + (bytes / 8) * 4, 0x5A, 4, st));
    SYBL_HIP(hipMemsetAsync(ctx->d_copy_digest, 0, 8, st));
    hipError_t e = launch_copy_digest(dst, (int64_t)(bytes / 4), ctx->d_copy_digest, st);
    if (e != hipSuccess) return hip_fail(e, "k_copy_digest");
    unsigned long long got = 0, want = 0;
    SYBL_HIP(hipMemcpyAsync(&got, ctx->d_copy_digest, 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    const uint32_t *w = (const uint32_t *)src;
    for (size_t i = 0; i < bytes / 4; i++) {
        uint64_t z = (uint64_t)w[i] + ((uint64_t)i << 32) + 0x9E3779B97F4A7C15ull;  // splitmix64 (scan_generic.h)
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        want += z ^ (z >> 31);
    }
    if (got != want)
        return fail(SYBL_E_NODEVICE, "SYBL_VERIFY_COPIES: %zu bytes of %s differ in HBM from the host bytes they were copied from (digest %016llx, expected %016llx)",
                    bytes, what, got, want);
    return SYBL_OK;
}

// ------------------------------------------------------------------ device arrays

int table_reserve(Table *t, Column *c, int64_t phys_rows) {
    // + one tile of slack: the scan's 16-byte loads and its prefetch may touch up to one
    // row past a segment end (never used, but it must be mapped)
    int64_t need = phys_rows + kTileRows;
    if (need <= c->cap_rows) return SYBL_OK;
    int64_t cap = std::max<int64_t>(need, c->cap_rows + c->cap_rows / 2);
    if (t->reserve_hint_rows + kTileRows > cap) {
        // (a hint that does not fit in memory is only a hint)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)(t->reserve_hint_rows + kTileRows) * (size_t)c->elem < free_b / 2)
            cap = t->reserve_hint_rows + kTileRows;
    }
    void *nd = nullptr;
    int rc0 = load_sync_all(t->ctx);  // (other blocks of a multi-stream load may still be writing the old array)
    if (rc0) return rc0;
    SYBL_HIP(hipMalloc(&nd, (size_t)cap * c->elem));
    if (c->d_data) {
        SYBL_HIP(hipMemcpyAsync(nd, c->d_data, (size_t)t->phys_rows * c->elem, hipMemcpyDeviceToDevice, t->ctx->stream));
        SYBL_HIP(hipStreamSynchronize(t->ctx->stream));
        SYBL_HIP(hipFree(c->d_data));
    }
    c->d_data = nd;
    c->cap_rows = cap;
    return SYBL_OK;
}

int valid_reserve(Table *t, Column *c, int64_t phys_rows) {
    int64_t words = round_up(phys_rows + kTileRows, 32) / 32 + 1;
    if (c->d_valid && words <= c->valid_cap_words) return SYBL_OK;
    int64_t cap = std::max<int64_t>(words, c->valid_cap_words * 2);
    cap = std::max<int64_t>(cap, round_up(t->reserve_hint_rows + kTileRows, 32) / 32 + 1);
    uint32_t *nd = nullptr;
    int rc0 = load_sync_all(t->ctx);
    if (rc0) return rc0;
    SYBL_HIP(hipMalloc((void **)&nd, (size_t)cap * 4));
    int64_t old_words = round_up(t->phys_rows, 32) / 32;
    if (c->d_valid) {
        SYBL_HIP(hipMemcpyAsync(nd, c->d_valid, (size_t)old_words * 4, hipMemcpyDeviceToDevice, t->ctx->stream));
    } else {
        // every earlier row of this column was populated
        SYBL_HIP(hipMemsetAsync(nd, 0xFF, (size_t)old_words * 4, t->ctx->stream));
    }
    SYBL_HIP(hipMemsetAsync(nd + old_words, 0, (size_t)(cap - old_words) * 4, t->ctx->stream));
    SYBL_HIP(hipStreamSynchronize(t->ctx->stream));
    if (c->d_valid) SYBL_HIP(hipFree(c->d_valid));
    c->d_valid = nd;
    c->valid_cap_words = cap;
    if (c->rank_col) c->rank_col->d_valid = nd;  // (the derived rank column BORROWS this bitmap: never left pointing at the freed one)
    return SYBL_OK;
}

// The device copy of the block segments (start, n) that k_block_minmax / k_distinct walk.  Uploaded by whoever is about to
// launch one of them: a reclaim or a dropped tail moves blocks without any column's statistics becoming pending.
int table_upload_blocks(Table *t) {
    const int64_t nb = (int64_t)t->blocks.size();
    if (nb == 0) return SYBL_OK;
    if (t->d_blocks_n < nb) {
        if (t->d_blocks) SYBL_HIP(hipFree(t->d_blocks));
        t->d_blocks = nullptr;
        t->d_blocks_n = 0;
        SYBL_HIP(hipMalloc((void **)&t->d_blocks, (size_t)nb * sizeof(Segment)));
        t->d_blocks_n = nb;
    }
    static_assert(sizeof(Segment) % 4 == 0, "whole words");
    {
        int rc = host_to_device(t->ctx, t->d_blocks, t->blocks.data(), (size_t)nb * sizeof(Segment), "block segments");
        if (rc) return rc;
    }
    return SYBL_OK;
}

int table_ensure_stats(Table *t) {
    int64_t nb = (int64_t)t->blocks.size();
    bool pending = false;
    for (auto &c : t->cols)
        if (c->type != SYBL_SET_VAL && c->stats_blocks < nb) pending = true;
    if (!pending) return SYBL_OK;
    hipStream_t st = t->ctx->stream;
    int urc = table_upload_blocks(t);
    if (urc) return urc;
    int64_t *d_out = nullptr;
    SYBL_HIP(hipMalloc((void **)&d_out, (size_t)nb * 3 * sizeof(int64_t)));
    std::vector<int64_t> h((size_t)nb * 3);
    for (auto &cp : t->cols) {
        Column *c = cp.get();
        if (c->type == SYBL_SET_VAL || c->stats_blocks >= nb) continue;
        int64_t b0 = c->stats_blocks, n = nb - b0;
        hipError_t e = launch_block_minmax(c->d_data, c->elem, c->vbase, c->d_valid, t->d_blocks + b0, (int)n, d_out,
                                           d_out + nb, d_out + 2 * nb, st);
        if (e != hipSuccess) {
            hipFree(d_out);
            return hip_fail(e, "k_block_minmax");
        }
        SYBL_HIP(hipMemcpyAsync(h.data(), d_out, (size_t)nb * 3 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        SYBL_HIP(hipStreamSynchronize(st));
        c->blk_min.resize((size_t)nb);
        c->blk_max.resize((size_t)nb);
        c->blk_pop.resize((size_t)nb);
        for (int64_t k = 0; k < n; k++) {
            c->blk_min[(size_t)(b0 + k)] = h[(size_t)k];
            c->blk_max[(size_t)(b0 + k)] = h[(size_t)(nb + k)];
            c->blk_pop[(size_t)(b0 + k)] = h[(size_t)(2 * nb + k)];
            if (h[(size_t)(2 * nb + k)] > 0) {
                c->exact_min = std::min(c->exact_min, h[(size_t)k]);
                c->exact_max = std::max(c->exact_max, h[(size_t)(nb + k)]);
            }
            c->n_pop += h[(size_t)(2 * nb + k)];
            if (h[(size_t)(2 * nb + k)] < t->blocks[(size_t)(b0 + k)].n) c->has_missing = true;
        }
        c->stats_blocks = nb;
    }
    SYBL_HIP(hipFree(d_out));
    return SYBL_OK;
}

// Re-encodes a column at another stored width / value base (out of place, then swaps the arrays).
int column_repack(Table *t, Column *c, int width, int64_t vbase) {
    if (c->type == SYBL_SET_VAL) return SYBL_OK;
    if (width == c->elem && vbase == c->vbase) return SYBL_OK;
    if (c->d_data) {
        int rc0 = load_sync_all(t->ctx);
        if (rc0) return rc0;
        hipStream_t st = t->ctx->stream;
        void *nd = nullptr;
        SYBL_HIP(hipMalloc(&nd, (size_t)c->cap_rows * (size_t)width));
        hipError_t e = launch_repack(c->d_data, c->elem, c->vbase, nd, width, vbase, t->phys_rows, st);
        if (e != hipSuccess) {
            hipFree(nd);
            return hip_fail(e, "k_repack");
        }
        SYBL_HIP(hipStreamSynchronize(st));
        SYBL_HIP(hipFree(c->d_data));
        c->d_data = nd;
    }
    c->elem = width;
    c->vbase = vbase;
    return SYBL_OK;
}

void column_free(Column *c) {
    if (c->rank_col) {
        c->rank_col->d_valid = nullptr;  // (borrowed from c)
        column_free(c->rank_col.get());
        c->rank_col.reset();
    }
    c->carried_weight.reset();  // (queries that weigh by it keep their share)
    if (c->d_data) hipFree(c->d_data);
    if (c->d_valid) hipFree(c->d_valid);
    if (c->d_set_off) hipFree(c->d_set_off);
    if (c->d_set_vals) hipFree(c->d_set_vals);
    if (c->d_gdict_keys) hipFree(c->d_gdict_keys);
    if (c->d_gdict_ranks) hipFree(c->d_gdict_ranks);
    if (c->d_stage) hipFree(c->d_stage);
}

int32_t dict_intern(Column *c, const std::string &s) {
    auto it = c->dict_ix.find(s);
    if (it != c->dict_ix.end()) return it->second;
    int32_t id = (int32_t)c->dict.size();
    c->dict.push_back(s);
    c->dict_ix.emplace(s, id);
    return id;
}


// ------------------------------------------------------------------ block writer
// Shared by sybl_table_append_block (host-decoded columns) and the native loader (loader.cpp):
// a block is staged column by column at the next 32-row boundary and only becomes visible to
// queries at block_commit, so a block that fails half way is simply not added -- the reference
// likewise skips a block whose files do not decode (table_query.go:134-139).

int block_begin(Table *t, int64_t nrows, BlockWriter *w) {
    if (nrows < 0) return fail(SYBL_E_INVAL, "negative row count");
    w->t = t;
    w->staged.clear();
    w->direct.clear();
    w->serial = false;
    w->nrows = nrows;
    w->start = round_up(t->phys_rows, 32);
    w->new_phys = w->start + nrows;
    return SYBL_OK;
}

// the validity words of the block: all ones / zeroed for the writer to set (valid == nullptr: nothing left to set)
static int block_valid(BlockWriter &w, Column *c, bool all_populated, uint32_t **valid) {
    Table *t = w.t;
    hipStream_t st = t->ctx->stream;
    int rc;
    *valid = nullptr;
    if (w.nrows == 0) return SYBL_OK;
    if (!all_populated || c->d_valid) {
        if ((rc = valid_reserve(t, c, w.new_phys))) return rc;
        *valid = c->d_valid + w.start / 32;
        size_t words = (size_t)(round_up(w.nrows, 32) / 32);
        if (all_populated) {
            SYBL_HIP(hipMemsetAsync(*valid, 0xFF, words * 4, st));
            *valid = nullptr;  // nothing left to set
        } else {
            SYBL_HIP(hipMemsetAsync(*valid, 0, words * 4, st));
            c->has_missing = true;
        }
    }
    return SYBL_OK;
}

// Reserves storage for the column in this block.  all_populated = every row of the block has a
// value.  Returns device pointers at the block start (valid == nullptr: no bitmap needed).
int block_col_device(BlockWriter &w, Column *c, bool all_populated, void **col, uint32_t **valid) {
    Table *t = w.t;
    hipStream_t st = t->ctx->stream;
    int rc;
    if (c->type != SYBL_SET_VAL && t->compact_mode) {
        // writers produce canonical values: they go to a staging block that block_commit packs into
        // the column at whatever width the column has (or needs) by then.  The staging block is shared by
        // consecutive blocks: in a multi-stream load the others finish first and this block's stream is drained
        // at commit.
        if (t->ctx->load_multi) {
            if ((rc = load_sync_all(t->ctx))) return rc;
            w.serial = true;
        }
        const int64_t need = std::max<int64_t>(w.nrows, 1) * c->canon();
        if (need > c->stage_cap) {
            if (c->d_stage) SYBL_HIP(hipFree(c->d_stage));
            c->d_stage = nullptr;
            SYBL_HIP(hipMalloc(&c->d_stage, (size_t)need));
            c->stage_cap = need;
        }
        *col = c->d_stage;
        if (!all_populated && w.nrows > 0) SYBL_HIP(hipMemsetAsync(*col, 0, (size_t)w.nrows * c->canon(), st));
        w.staged.push_back(BlockWriter::Staged{c, false, 0, 0, 0});
    } else if (c->type != SYBL_SET_VAL) {
        if ((rc = table_reserve(t, c, w.new_phys))) return rc;
        *col = (char *)c->d_data + (size_t)w.start * c->elem;
        if (!all_populated && w.nrows > 0) SYBL_HIP(hipMemsetAsync(*col, 0, (size_t)w.nrows * c->elem, st));
    } else if (col) {
        *col = nullptr;
    }
    return block_valid(w, c, all_populated, valid);
}

int block_col_direct(BlockWriter &w, Column *c, bool all_populated, int64_t mn, int64_t mx, int64_t pop, void **col, uint32_t **valid, bool *ok) {
    Table *t = w.t;
    *ok = false;
    if (!t->compact_mode || c->type == SYBL_SET_VAL || !c->d_data || w.nrows == 0 || env("SYBL_NO_DIRECT_DECODE")) return SYBL_OK;
    if (c->stats_blocks != (int64_t)t->blocks.size()) return SYBL_OK;  // (block statistics are appended at commit)
    if (pop > 0 && c->elem < 8) {
        const __int128 top = (__int128)c->vbase + (((__int128)1 << (8 * c->elem)) - 1);
        if (mn < c->vbase || (__int128)mx > top) return SYBL_OK;  // the block widens the column: staged + repacked at commit
    }
    int rc;
    if ((rc = table_reserve(t, c, w.new_phys))) return rc;
    *col = (char *)c->d_data + (size_t)w.start * (size_t)c->elem;
    if (!all_populated) SYBL_HIP(hipMemsetAsync(*col, 0, (size_t)w.nrows * (size_t)c->elem, t->ctx->stream));
    if ((rc = block_valid(w, c, all_populated, valid))) return rc;
    w.direct.push_back(BlockWriter::Staged{c, true, pop > 0 ? mn : INT64_MAX, pop > 0 ? mx : INT64_MIN, pop});
    *ok = true;
    return SYBL_OK;
}

void block_col_stats(BlockWriter &w, Column *c, int64_t mn, int64_t mx, int64_t pop) {
    for (auto &s : w.staged)
        if (s.c == c) {
            s.have_stats = true;
            s.mn = pop > 0 ? mn : INT64_MAX;
            s.mx = pop > 0 ? mx : INT64_MIN;
            s.pop = pop;
        }
}

// populated: nrows bytes, nullptr = every row; absent column: pass all zero via block_col_absent
static int put_valid_bits(BlockWriter &w, Column *c, const uint8_t *populated, bool absent, void **col_out = nullptr) {
    bool all = !absent;
    if (!absent && populated) {
        for (int64_t r = 0; r < w.nrows; r++)
            if (!populated[r]) { all = false; break; }
    }
    void *col = nullptr;
    uint32_t *valid = nullptr;
    int rc = block_col_device(w, c, all, &col, &valid);
    if (rc) return rc;
    if (col_out) *col_out = col;
    if (valid && !absent) {
        std::vector<uint32_t> bits((size_t)(round_up(w.nrows, 32) / 32), 0);
        for (int64_t r = 0; r < w.nrows; r++)
            if (!populated || populated[r]) bits[(size_t)(r >> 5)] |= 1u << (r & 31);
        if ((rc = host_to_device(w.t->ctx, valid, bits.data(), bits.size() * 4, "validity words"))) return rc;
    }
    return SYBL_OK;
}

static void set_csr_extend(BlockWriter &w, Column *c) {
    // offsets exist for every physical row incl. padding rows (empty sets)
    if (c->h_set_off.empty()) c->h_set_off.push_back(0);
    while ((int64_t)c->h_set_off.size() < w.start + 1) c->h_set_off.push_back(c->h_set_off.back());
}

int block_col_absent(BlockWriter &w, Column *c) {
    int rc = put_valid_bits(w, c, nullptr, true);
    if (rc) return rc;
    if (c->type == SYBL_SET_VAL) {
        set_csr_extend(w, c);
        for (int64_t r = 0; r < w.nrows; r++) c->h_set_off.push_back(c->h_set_off.back());
        c->set_dirty = true;
    }
    return SYBL_OK;
}

int block_col_int_host(BlockWriter &w, Column *c, const int64_t *vals, const uint8_t *populated) {
    void *col = nullptr;
    int rc = put_valid_bits(w, c, populated, false, &col);
    if (rc || w.nrows == 0) return rc;
    hipStream_t st = w.t->ctx->stream;
    if (w.t->compact_mode) {
        int64_t mn = INT64_MAX, mx = INT64_MIN, pop = 0;
        for (int64_t r = 0; r < w.nrows; r++)
            if (!populated || populated[r]) {
                mn = std::min(mn, vals[r]);
                mx = std::max(mx, vals[r]);
                pop++;
            }
        block_col_stats(w, c, mn, mx, pop);
    }
    return host_to_device(w.t->ctx, col, vals, (size_t)w.nrows * 8, c->name.c_str());
}

// ids: table-global dictionary ids
int block_col_str_host(BlockWriter &w, Column *c, const int32_t *global_ids, const uint8_t *populated) {
    void *col = nullptr;
    int rc = put_valid_bits(w, c, populated, false, &col);
    if (rc || w.nrows == 0) return rc;
    hipStream_t st = w.t->ctx->stream;
    if (w.t->compact_mode) {
        int64_t mn = INT64_MAX, mx = INT64_MIN, pop = 0;
        for (int64_t r = 0; r < w.nrows; r++)
            if (!populated || populated[r]) {
                mn = std::min<int64_t>(mn, global_ids[r]);
                mx = std::max<int64_t>(mx, global_ids[r]);
                pop++;
            }
        block_col_stats(w, c, mn, mx, pop);
    }
    return host_to_device(w.t->ctx, col, global_ids, (size_t)w.nrows * 4, c->name.c_str());
}

// CSR over the block's rows; member ids are table-global.  The host keeps the CSR mirror and
// uploads it before the next query that filters on the column.
int block_col_set_host(BlockWriter &w, Column *c, const int64_t *off, const int32_t *global_ids, const uint8_t *populated) {
    int rc = put_valid_bits(w, c, populated, false);
    if (rc) return rc;
    set_csr_extend(w, c);
    for (int64_t r = 0; r < w.nrows; r++) {
        bool pop = !populated || populated[r];
        if (pop)
            for (int64_t k = off[r]; k < off[r + 1]; k++) c->h_set_vals.push_back(global_ids[k]);
        c->h_set_off.push_back((int64_t)c->h_set_vals.size());
    }
    c->set_dirty = true;
    return SYBL_OK;
}

// narrowest (width, base) that holds [lo, hi]; canonical storage when nothing narrower does
static void fit_storage(const Column *c, bool any, int64_t lo, int64_t hi, int *width, int64_t *base) {
    *width = c->canon();
    *base = 0;
    if (!any) {
        *width = 1;  // no populated row: the stored bits are never looked at
        return;
    }
    const unsigned __int128 range = (unsigned __int128)((__int128)hi - (__int128)lo);
    int fit = range < 256 ? 1 : range < 65536 ? 2 : range < ((unsigned __int128)1 << 32) ? 4 : 8;
    if (fit < *width) {
        *width = fit;
        *base = lo;
    }
}

// Compact mode: the staged canonical blocks of the columns -> the columns' compact arrays.  The
// block's exact extrema (one k_block_minmax launch per column, ONE readback for all of them) decide
// whether a column's current (width, base) still holds every value; if not the resident rows are
// re-encoded once at the wider layout.  The block statistics come for free.
static int commit_staged(BlockWriter &w) {
    Table *t = w.t;
    hipStream_t st = t->ctx->stream;
    const int64_t nb = (int64_t)t->blocks.size();
    const size_t ns = w.staged.size();
    if (ns == 0) return SYBL_OK;
    int rc = table_ensure_stats(t);  // extrema of the resident rows (no-op when current)
    if (rc) return rc;
    std::vector<int64_t> h(ns * 3, 0);
    bool need_gpu = false;
    for (size_t k = 0; k < ns; k++) {
        const BlockWriter::Staged &s = w.staged[k];
        if (s.have_stats) {
            h[k * 3] = s.mn;
            h[k * 3 + 1] = s.mx;
            h[k * 3 + 2] = s.pop;
        } else {
            need_gpu = true;
        }
    }
    if (w.nrows > 0 && need_gpu) {
        const size_t need = 2 + ns * 3;  // one Segment, then min / max / pop per staged column
        if ((int64_t)need > t->scratch_words) {
            if (t->d_scratch) SYBL_HIP(hipFree(t->d_scratch));
            t->d_scratch = nullptr;
            SYBL_HIP(hipMalloc((void **)&t->d_scratch, need * 8));
            t->scratch_words = (int64_t)need;
        }
        Segment seg;
        seg.start = w.start;
        seg.n = w.nrows;
        SYBL_HIP(hipMemcpyAsync(t->d_scratch, &seg, sizeof(seg), hipMemcpyHostToDevice, st));
        for (size_t k = 0; k < ns; k++) {
            if (w.staged[k].have_stats) continue;
            Column *c = w.staged[k].c;
            // k_block_minmax indexes values and validity bits by physical row: shift the staging base so
            // that physical row w.start is its first element
            const char *virt = (const char *)c->d_stage - (size_t)w.start * (size_t)c->canon();
            int64_t *out = t->d_scratch + 2 + k * 3;
            hipError_t e = launch_block_minmax(virt, c->canon(), 0, c->d_valid, (const Segment *)t->d_scratch, 1, out, out + 1, out + 2, st);
            if (e != hipSuccess) return hip_fail(e, "k_block_minmax");
        }
        std::vector<int64_t> g(ns * 3, 0);
        SYBL_HIP(hipMemcpyAsync(g.data(), t->d_scratch + 2, ns * 3 * 8, hipMemcpyDeviceToHost, st));
        SYBL_HIP(hipStreamSynchronize(st));
        for (size_t k = 0; k < ns; k++)
            if (!w.staged[k].have_stats)
                for (int j = 0; j < 3; j++) h[k * 3 + j] = g[k * 3 + j];
    }
    for (size_t k = 0; k < ns; k++) {
        Column *c = w.staged[k].c;
        const int64_t bmin = w.nrows > 0 ? h[k * 3] : INT64_MAX, bmax = w.nrows > 0 ? h[k * 3 + 1] : INT64_MIN;
        const int64_t bpop = w.nrows > 0 ? h[k * 3 + 2] : 0;
        const bool any = c->n_pop > 0 || bpop > 0;
        const int64_t lo = std::min(c->n_pop > 0 ? c->exact_min : INT64_MAX, bpop > 0 ? bmin : INT64_MAX);
        const int64_t hi = std::max(c->n_pop > 0 ? c->exact_max : INT64_MIN, bpop > 0 ? bmax : INT64_MIN);
        bool holds = true;
        if (any && c->elem < 8) {
            const __int128 top = (__int128)c->vbase + (((__int128)1 << (8 * c->elem)) - 1);
            holds = lo >= c->vbase && (__int128)hi <= top;
        }
        if (!holds || !c->d_data) {
            int width;
            int64_t base;
            fit_storage(c, any, lo, hi, &width, &base);
            if (c->d_data && width < c->elem) {  // never narrow resident rows here (sybl_table_compact does)
                width = c->elem;
                base = c->elem == c->canon() ? 0 : std::min(lo, c->vbase);
                if (c->elem < 8 && (__int128)hi > (__int128)base + (((__int128)1 << (8 * c->elem)) - 1))
                    fit_storage(c, any, lo, hi, &width, &base);  // the old width cannot span the new range
            }
            if ((rc = column_repack(t, c, width, base))) return rc;
        }
        if ((rc = table_reserve(t, c, w.new_phys))) return rc;
        if (w.nrows > 0) {
            // stream-ordered: the next writer of this column's staging block queues behind the repack
            hipError_t e = launch_repack(c->d_stage, c->canon(), 0, (char *)c->d_data + (size_t)w.start * (size_t)c->elem, c->elem,
                                         c->vbase, w.nrows, st);
            if (e != hipSuccess) return hip_fail(e, "k_repack");
        }
        if (c->stats_blocks == nb) {
            c->blk_min.push_back(bmin);
            c->blk_max.push_back(bmax);
            c->blk_pop.push_back(bpop);
            if (bpop > 0) {
                c->exact_min = std::min(c->exact_min, bmin);
                c->exact_max = std::max(c->exact_max, bmax);
            }
            c->n_pop += bpop;
            if (bpop < w.nrows) c->has_missing = true;
            c->stats_blocks = nb + 1;
        }
    }
    return SYBL_OK;
}

int block_commit(BlockWriter &w) {
    Table *t = w.t;
    int rc = commit_staged(w);
    if (rc) return rc;
    w.staged.clear();
    for (auto &s : w.direct) {  // columns already in place: only their block statistics are left to record
        Column *c = s.c;
        if (c->stats_blocks != (int64_t)t->blocks.size()) continue;
        c->blk_min.push_back(s.mn);
        c->blk_max.push_back(s.mx);
        c->blk_pop.push_back(s.pop);
        if (s.pop > 0) {
            c->exact_min = std::min(c->exact_min, s.mn);
            c->exact_max = std::max(c->exact_max, s.mx);
        }
        c->n_pop += s.pop;
        if (s.pop < w.nrows) c->has_missing = true;
        c->stats_blocks = (int64_t)t->blocks.size() + 1;
    }
    w.direct.clear();
    if (w.serial) {
        SYBL_HIP(hipStreamSynchronize(t->ctx->stream));
        w.serial = false;
    }
    Segment blk;
    blk.start = w.start;
    blk.n = w.nrows;
    t->blocks.push_back(blk);
    t->phys_rows = w.new_phys;
    t->logical_rows += w.nrows;
    t->version++;
    return SYBL_OK;
}

// Gives the rows of trailing blocks that left the scan (Segment::n == 0: sybl_table_refresh) back to the table, so the
// blocks loaded next take their place.  Sybil's ingest rewrites the last, partly filled block on every digest
// (table_ingest.go / table_block_io.go SaveRecordsToBlock): without this every refresh of a long-running host would append
// up to 65 536 rows per column and free nothing.  keep_blocks: blocks [0, keep_blocks) stay whatever their row count (a
// resident block directory refers to them by number).  Returns the blocks dropped.
int64_t table_drop_dead_tail(Table *t, int64_t keep_blocks) {
    int64_t dropped = 0;
    while ((int64_t)t->blocks.size() > keep_blocks && t->blocks.back().n == 0) {
        t->blocks.pop_back();
        dropped++;
    }
    if (!dropped) return 0;
    const int64_t nb = (int64_t)t->blocks.size();
    t->phys_rows = nb ? t->blocks.back().start + t->blocks.back().n : 0;
    for (auto &cp : t->cols) {
        Column *c = cp.get();
        if (c->type == SYBL_SET_VAL) {
            if ((int64_t)c->h_set_off.size() > t->phys_rows + 1) {
                c->h_set_off.resize((size_t)t->phys_rows + 1);
                c->h_set_vals.resize((size_t)c->h_set_off.back());
                c->set_dirty = true;
            }
            continue;
        }
        if ((int64_t)c->blk_min.size() > nb) {
            c->blk_min.resize((size_t)nb);
            c->blk_max.resize((size_t)nb);
            c->blk_pop.resize((size_t)nb);
        }
        c->stats_blocks = std::min(c->stats_blocks, nb);
        // (exact_min / exact_max / the stored width stay: bounds of a superset are still bounds; a dictionary built
        // over the dropped rows is built again -- the block COUNT may come back to what it was with other values)
        if (c->gdict_blocks >= 0) c->gdict_blocks = -1;
    }
    t->version++;
    return dropped;
}

// Blocks that vanished from the middle of a followed table (sybil trim / expire, table_trim.go) leave their rows behind:
// the scan skips them (a block of zero rows), but a host that runs for weeks would keep every expired block in HBM.  Once
// the dead rows are a quarter of the table (and a block's worth), the live blocks close up: every column's rows and
// validity words move down block by block (through a scratch block: a block's new place may overlap its old one), the set
// columns' host CSR is rebuilt, block numbers stay (the loader's directory refers to them).  Extrema and dictionaries
// stay valid: bounds and dictionaries of a superset.  The reference re-lists the directory per query and never holds
// what it does not scan (table_query.go:40-106).
int table_reclaim_dead_rows(Table *t, bool force) {
    const int64_t nb = (int64_t)t->blocks.size();
    std::vector<int64_t> to((size_t)nb, 0);
    int64_t w = 0;
    bool moves = false;
    for (int64_t b = 0; b < nb; b++) {
        to[(size_t)b] = w;
        if (t->blocks[(size_t)b].n > 0) {
            moves = moves || t->blocks[(size_t)b].start != w;
            w += (t->blocks[(size_t)b].n + 31) / 32 * 32;
        }
    }
    int64_t last_live = -1;
    for (int64_t b = 0; b < nb; b++)
        if (t->blocks[(size_t)b].n > 0) last_live = b;
    const int64_t new_phys = last_live >= 0 ? to[(size_t)last_live] + t->blocks[(size_t)last_live].n : 0;
    const int64_t dead = t->phys_rows - new_phys;
    if (!moves || dead <= 0) return SYBL_OK;
    if (!force && !(dead >= 65536 && dead * 4 >= t->phys_rows)) return SYBL_OK;
    Ctx *ctx = t->ctx;
    hipStream_t st = ctx->stream;
    int rc = load_sync_all(ctx);
    if (rc) return rc;
    SYBL_HIP(hipStreamSynchronize(st));
    int64_t max_n = 0;
    for (auto &b : t->blocks) max_n = std::max(max_n, b.n);
    DevOwner scratch;
    SYBL_HIP(hipMalloc(&scratch.p, (size_t)max_n * 8 + (size_t)(max_n / 32 + 2) * 4));
    uint8_t *sc_rows = (uint8_t *)scratch.p;
    uint32_t *sc_valid = (uint32_t *)(sc_rows + (size_t)max_n * 8);
    for (auto &cp : t->cols) {
        Column *c = cp.get();
        if (c->type == SYBL_SET_VAL) {
            if (c->h_set_off.empty()) continue;
            std::vector<int64_t> off;
            std::vector<int32_t> vals;
            off.reserve((size_t)w + 1);
            off.push_back(0);
            for (int64_t b = 0; b < nb; b++) {
                const Segment &blk = t->blocks[(size_t)b];
                if (blk.n <= 0) continue;
                const int64_t rows = b == last_live ? blk.n : (blk.n + 31) / 32 * 32;  // (the table ends with its last row)
                for (int64_t r = 0; r < rows; r++) {
                    const int64_t src = blk.start + r;
                    if (src + 1 < (int64_t)c->h_set_off.size()) {
                        const int64_t lo = c->h_set_off[(size_t)src], hi = c->h_set_off[(size_t)src + 1];
                        vals.insert(vals.end(), c->h_set_vals.begin() + lo, c->h_set_vals.begin() + hi);
                    }
                    off.push_back((int64_t)vals.size());
                }
            }
            c->h_set_off.swap(off);
            c->h_set_vals.swap(vals);
            c->set_dirty = true;
        } else if (c->d_data) {
            for (int64_t b = 0; b < nb; b++) {
                const Segment &blk = t->blocks[(size_t)b];
                if (blk.n <= 0 || blk.start == to[(size_t)b]) continue;
                const size_t bytes = (size_t)blk.n * (size_t)c->elem;
                SYBL_HIP(hipMemcpyAsync(sc_rows, (const uint8_t *)c->d_data + (size_t)blk.start * (size_t)c->elem, bytes, hipMemcpyDeviceToDevice, st));
                SYBL_HIP(hipMemcpyAsync((uint8_t *)c->d_data + (size_t)to[(size_t)b] * (size_t)c->elem, sc_rows, bytes, hipMemcpyDeviceToDevice, st));
                if (c->d_valid) {
                    const size_t words = (size_t)((blk.n + 31) / 32);
                    SYBL_HIP(hipMemcpyAsync(sc_valid, c->d_valid + blk.start / 32, words * 4, hipMemcpyDeviceToDevice, st));
                    SYBL_HIP(hipMemcpyAsync(c->d_valid + to[(size_t)b] / 32, sc_valid, words * 4, hipMemcpyDeviceToDevice, st));
                }
            }
        }
        if (c->gdict_blocks >= 0) c->gdict_blocks = -1;
    }
    SYBL_HIP(hipStreamSynchronize(st));
    for (int64_t b = 0; b < nb; b++) t->blocks[(size_t)b].start = to[(size_t)b];
    t->phys_rows = new_phys;
    t->version++;
    return SYBL_OK;
}

// ------------------------------------------------------------------ group dictionaries
// Direct mapping needs one cell per value of the key RANGE; a sparse key (user ids, raw
// timestamps) gets one cell per DISTINCT value instead: k_distinct collects the distinct values
// into a hash set, the host sorts them (rank order == key order, so results stay in canonical
// order and every rank of a multi-GPU job that installs the same dictionary gets the same
// layout), and the scan looks values up in a value -> rank open-addressing map.

static inline uint32_t host_dict_hash(int64_t x) {
    uint64_t z = (uint64_t)x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

int column_build_gdict(Table *t, Column *c) {
    int64_t nb = (int64_t)t->blocks.size();
    if (c->gdict_blocks == nb || c->gdict_blocks == -2) return SYBL_OK;  // current, or supplied by the host
    hipStream_t st = t->ctx->stream;
    int rc = table_ensure_stats(t);
    if (rc) return rc;
    // (the segments as they are NOW: table_ensure_stats uploads them only when some column's statistics are pending, and
    // a trim-only refresh moves live blocks -- table_reclaim_dead_rows -- without making any pending)
    if ((rc = table_upload_blocks(t))) return rc;
    int64_t want = std::min<int64_t>(4 * kDictMaxDistinct, std::max<int64_t>(1024, 4 * c->n_pop));
    uint32_t cap = 1024;
    while ((int64_t)cap < want) cap <<= 1;
    int64_t *d_keys = nullptr;
    unsigned long long *d_n = nullptr;
    SYBL_HIP(hipMalloc((void **)&d_keys, (size_t)cap * 8));
    SYBL_HIP(hipMalloc((void **)&d_n, 8));
    auto cleanup = [&]() {
        hipFree(d_keys);
        hipFree(d_n);
    };
    hipError_t e = launch_fill64(d_keys, cap, kDictEmpty, st);
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(e, "k_fill64");
    }
    SYBL_HIP(hipMemsetAsync(d_n, 0, 8, st));
    e = launch_distinct(c->d_data, c->elem, c->vbase, c->d_valid, t->d_blocks, (int)nb, d_keys, cap - 1, d_n,
                        (unsigned long long)kDictMaxDistinct, st);
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(e, "k_distinct");
    }
    unsigned long long n = 0;
    SYBL_HIP(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, st));
    SYBL_HIP(hipStreamSynchronize(st));
    if ((int64_t)n > kDictMaxDistinct) {
        cleanup();
        return fail(SYBL_E_INVAL, "column '%s' has more than %lld distinct values: too many groups for this build",
                    c->name.c_str(), (long long)kDictMaxDistinct);
    }
    std::vector<int64_t> h(cap);
    SYBL_HIP(hipMemcpy(h.data(), d_keys, (size_t)cap * 8, hipMemcpyDeviceToHost));
    cleanup();
    // (results whose rows are still to be built turn key digits into values through the dictionary as it is now)
    for (Query *q : t->queries) query_finish_lazy_results(q);
    c->gdict.clear();
    c->gdict.reserve((size_t)n);
    for (int64_t x : h)
        if (x != kDictEmpty) c->gdict.push_back(x);
    if (c->n_pop > 0 && c->exact_min == kDictEmpty) c->gdict.push_back(kDictEmpty);  // the sentinel value itself
    std::sort(c->gdict.begin(), c->gdict.end());
    c->gdict_blocks = nb;
    return column_install_gdict(t, c);
}

int column_install_gdict(Table *t, Column *c) {
    (void)t;
    size_t D = c->gdict.size();
    uint32_t cap = 16;
    while ((size_t)cap < 2 * D + 2) cap <<= 1;
    std::vector<int64_t> keys(cap, kDictEmpty);
    std::vector<int32_t> ranks(cap, -1);
    for (size_t r = 0; r < D; r++) {
        int64_t x = c->gdict[r];
        if (x == kDictEmpty) continue;  // cannot be represented in the map; such rows are reported as overflow
        uint32_t hh = host_dict_hash(x) & (cap - 1);
        while (keys[hh] != kDictEmpty) hh = (hh + 1) & (cap - 1);
        keys[hh] = x;
        ranks[hh] = (int32_t)r;
    }
    if (c->d_gdict_keys) SYBL_HIP(hipFree(c->d_gdict_keys));
    if (c->d_gdict_ranks) SYBL_HIP(hipFree(c->d_gdict_ranks));
    c->d_gdict_keys = nullptr;
    c->d_gdict_ranks = nullptr;
    SYBL_HIP(hipMalloc((void **)&c->d_gdict_keys, (size_t)cap * 8));
    SYBL_HIP(hipMalloc((void **)&c->d_gdict_ranks, (size_t)cap * 4));
    int rc;
    if ((rc = host_to_device(t->ctx, c->d_gdict_keys, keys.data(), (size_t)cap * 8, "group dictionary keys"))) return rc;
    if ((rc = host_to_device(t->ctx, c->d_gdict_ranks, ranks.data(), (size_t)cap * 4, "group dictionary ranks"))) return rc;
    c->gdict_mask = cap - 1;
    c->gdict_gen++;
    return SYBL_OK;
}

int column_build_rank(Table *t, Column *c) {
    const int64_t D = (int64_t)c->gdict.size();
    const int ow = D + 1 <= 256 ? 1 : D + 1 <= 65536 ? 2 : 4;
    auto adopt = [&](Column *r) {
        r->d_valid = c->d_valid;  // borrowed (column_free knows)
        r->has_missing = c->has_missing;
        r->n_pop = c->n_pop;
    };
    if (c->rank_col && c->rank_gen == c->gdict_gen && c->rank_version == t->version && c->rank_col->elem == ow) {
        adopt(c->rank_col.get());
        return SYBL_OK;
    }
    if (c->rank_col) {
        c->rank_col->d_valid = nullptr;
        column_free(c->rank_col.get());
        c->rank_col.reset();
    }
    auto r = std::make_unique<Column>();
    r->name = c->name;
    r->type = SYBL_INT_VAL;
    r->elem = ow;
    r->vbase = 0;
    int rc = table_reserve(t, r.get(), t->phys_rows);
    if (rc) return rc;
    hipError_t e = launch_rank_column(c->d_data, c->elem, c->vbase, c->d_valid, c->d_gdict_keys, c->d_gdict_ranks, c->gdict_mask, t->phys_rows, r->d_data, ow,
                                      (uint32_t)D, t->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->ctx->stream);
    if (e != hipSuccess) {
        column_free(r.get());
        return hip_fail(e, "k_rank_column");
    }
    r->exact_min = 0;
    r->exact_max = D > 0 ? D - 1 : 0;
    r->stats_blocks = (int64_t)t->blocks.size();
    adopt(r.get());
    c->rank_col = std::move(r);
    c->rank_gen = c->gdict_gen;
    c->rank_version = t->version;
    return SYBL_OK;
}

int column_upload_set(Table *t, Column *c) {
    if (c->type != SYBL_SET_VAL || !c->set_dirty) return SYBL_OK;
    // offsets for every physical row + the scan's one-tile slack
    if (c->h_set_off.empty()) c->h_set_off.push_back(0);
    std::vector<int64_t> off = c->h_set_off;
    while ((int64_t)off.size() < t->phys_rows + kTileRows + 2) off.push_back(off.back());
    if (c->d_set_off) SYBL_HIP(hipFree(c->d_set_off));
    if (c->d_set_vals) SYBL_HIP(hipFree(c->d_set_vals));
    c->d_set_off = nullptr;
    c->d_set_vals = nullptr;
    SYBL_HIP(hipMalloc((void **)&c->d_set_off, off.size() * 8));
    {
        int rc = host_to_device(t->ctx, c->d_set_off, off.data(), off.size() * 8, "set offsets");
        if (rc) return rc;
    }
    size_t nv = std::max<size_t>(c->h_set_vals.size(), 1);
    SYBL_HIP(hipMalloc((void **)&c->d_set_vals, nv * 4));
    if (!c->h_set_vals.empty()) {
        int rc = host_to_device(t->ctx, c->d_set_vals, c->h_set_vals.data(), c->h_set_vals.size() * 4, "set members");
        if (rc) return rc;
    }
    c->set_vals_cap = (int64_t)nv;
    c->set_dirty = false;
    return SYBL_OK;
}

}  // namespace sybl

using namespace sybl;

extern "C" {

// ------------------------------------------------------------------ tables

int sybl_table_create(sybl_ctx *ctx, const char *name, sybl_table **out) {
    SYBL_API_GUARD(ctx);
    if (!ctx || !out) return fail(SYBL_E_INVAL, "sybl_table_create: NULL argument");
    sybl_table *t = new sybl_table();
    t->ctx = ctx;
    t->name = name ? name : "";
    *out = t;
    return SYBL_OK;
}

void sybl_table_free(sybl_table *t) {
    SYBL_API_GUARD(t);
    if (!t) return;
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    // queries that outlive their table: results of theirs whose rows are still to be built get them now (result.cpp)
    for (Query *q : t->queries) {
        query_finish_lazy_results(q);
        q->table_gone = true;
    }
    for (auto &c : t->cols) column_free(c.get());
    if (t->d_blocks) hipFree(t->d_blocks);
    if (t->d_scratch) hipFree(t->d_scratch);
    delete t;
}

int sybl_table_add_column(sybl_table *t, const char *name, int type, int64_t info_min, int64_t info_max) {
    SYBL_API_GUARD(t);
    if (!t || !name) return fail(SYBL_E_INVAL, "sybl_table_add_column: NULL argument");
    if (type != SYBL_INT_VAL && type != SYBL_STR_VAL && type != SYBL_SET_VAL) return fail(SYBL_E_INVAL, "bad column type %d", type);
    if (t->col_ix.count(name)) return fail(SYBL_E_INVAL, "column '%s' already exists", name);
    if (!t->blocks.empty()) return fail(SYBL_E_STATE, "columns must be declared before the first block");
    auto c = std::make_unique<Column>();
    c->name = name;
    c->type = type;
    c->elem = type == SYBL_INT_VAL ? 8 : 4;
    c->info_given = info_min <= info_max;
    c->info_min = info_min;
    c->info_max = info_max;
    t->col_ix[name] = (int)t->cols.size();
    t->cols.push_back(std::move(c));
    return SYBL_OK;
}

int sybl_table_append_block(sybl_table *t, int64_t nrows, int32_t ncols, const sybl_col_view *cols) {
    SYBL_API_GUARD(t);
    if (!t || nrows < 0 || (ncols > 0 && !cols)) return fail(SYBL_E_INVAL, "sybl_table_append_block: bad argument");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    for (int i = 0; i < ncols; i++) {
        Column *c = t->find(cols[i].name);
        if (!c) return fail(SYBL_E_BLOCK, "block has unknown column '%s'", cols[i].name ? cols[i].name : "(null)");
        if (c->type != cols[i].type) return fail(SYBL_E_BLOCK, "column '%s' type mismatch", c->name.c_str());
    }
    BlockWriter w;
    int rc = block_begin(t, nrows, &w);
    if (rc) return rc;
    std::vector<int32_t> ids;
    for (auto &cp : t->cols) {
        Column *c = cp.get();
        const sybl_col_view *v = nullptr;
        for (int i = 0; i < ncols; i++)
            if (c->name == cols[i].name) v = &cols[i];
        if (!v) {
            if ((rc = block_col_absent(w, c))) return rc;
            continue;
        }
        if (c->type == SYBL_INT_VAL) {
            if (!v->ints && nrows > 0) return fail(SYBL_E_BLOCK, "int column '%s' without values", c->name.c_str());
            if ((rc = block_col_int_host(w, c, v->ints, v->populated))) return rc;
            continue;
        }
        // block-local dictionary ids -> table-global ids (SURVEY.md 8a note 8)
        std::vector<int32_t> lut((size_t)std::max(v->n_strings, 0));
        for (int k = 0; k < v->n_strings; k++) lut[(size_t)k] = dict_intern(c, v->strings[k] ? v->strings[k] : "");
        if (c->type == SYBL_STR_VAL) {
            if (!v->str_ids && nrows > 0) return fail(SYBL_E_BLOCK, "str column '%s' without ids", c->name.c_str());
            ids.resize((size_t)nrows);
            for (int64_t r = 0; r < nrows; r++) {
                int32_t id = v->str_ids[r];
                bool pop = !v->populated || v->populated[r];
                if (pop && (id < 0 || id >= v->n_strings))
                    return fail(SYBL_E_BLOCK, "str id %d outside the block StringTable of '%s'", id, c->name.c_str());
                ids[(size_t)r] = pop ? lut[(size_t)id] : 0;
            }
            if ((rc = block_col_str_host(w, c, ids.data(), v->populated))) return rc;
        } else {
            if ((!v->set_off || (!v->set_ids && v->set_off[nrows] > 0)) && nrows > 0)
                return fail(SYBL_E_BLOCK, "set column '%s' without offsets/ids", c->name.c_str());
            int64_t total = nrows > 0 ? v->set_off[nrows] : 0;
            ids.resize((size_t)total);
            for (int64_t k = 0; k < total; k++) {
                int32_t id = v->set_ids[k];
                if (id < 0 || id >= v->n_strings)
                    return fail(SYBL_E_BLOCK, "set member id %d outside the block StringTable of '%s'", id, c->name.c_str());
                ids[(size_t)k] = lut[(size_t)id];
            }
            static const int64_t zero_off[1] = {0};
            if ((rc = block_col_set_host(w, c, nrows > 0 ? v->set_off : zero_off, ids.data(), v->populated))) return rc;
        }
    }
    return block_commit(w);
}

int sybl_table_create_synth(sybl_ctx *ctx, const char *name, uint64_t seed, int64_t total_rows, int64_t row0,
                            int64_t nrows, int32_t ncols, const sybl_synth_col *cols, sybl_table **out) {
    SYBL_API_GUARD(ctx);
    if (!ctx || !out || !cols || ncols <= 0 || nrows < 0 || total_rows <= 0 || row0 < 0 || row0 + nrows > total_rows)
        return fail(SYBL_E_INVAL, "sybl_table_create_synth: bad argument");
    SYBL_HIP(hipSetDevice(ctx->device));
    sybl_table *t = nullptr;
    int rc = sybl_table_create(ctx, name, &t);
    if (rc) return rc;
    for (int i = 0; i < ncols; i++) {
        if ((rc = sybl_table_add_column(t, cols[i].name, SYBL_INT_VAL, cols[i].info_min, cols[i].info_max))) {
            sybl_table_free(t);
            return rc;
        }
    }
    for (int i = 0; i < ncols; i++) {
        Column *c = t->cols[(size_t)i].get();
        if ((rc = table_reserve(t, c, nrows))) {
            sybl_table_free(t);
            return rc;
        }
        uint64_t cs = seed ^ ((uint64_t)(cols[i].col_index + 1) * 0x9E3779B97F4A7C15ull);
        hipError_t e = launch_synth((int64_t *)c->d_data, nrows, row0, total_rows, cols[i].kind, cols[i].a, cols[i].b, cs, ctx->stream);
        if (e != hipSuccess) {
            sybl_table_free(t);
            return hip_fail(e, "k_synth");
        }
    }
    for (int64_t r = 0; r < nrows; r += SYBL_BLOCK_ROWS) {
        Segment blk;
        blk.start = r;
        blk.n = std::min<int64_t>(SYBL_BLOCK_ROWS, nrows - r);
        t->blocks.push_back(blk);
    }
    t->phys_rows = nrows;
    t->logical_rows = nrows;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        sybl_table_free(t);
        return hip_fail(e, "synth sync");
    }
    *out = t;
    return SYBL_OK;
}

int64_t sybl_table_rows(const sybl_table *t) { SYBL_API_GUARD(t); return t ? t->logical_rows : 0; }
int64_t sybl_table_blocks(const sybl_table *t) { SYBL_API_GUARD(t); return t ? (int64_t)t->blocks.size() : 0; }

int64_t sybl_table_hbm_bytes(const sybl_table *t) {
    SYBL_API_GUARD(t);
    if (!t) return 0;
    int64_t b = 0;
    for (auto &c : t->cols) {
        b += c->cap_rows * c->elem + c->valid_cap_words * 4 + c->set_vals_cap * 4;
        // ... and the columns derived from it: a sparse key's rank column, a weight column's carried weights
        if (c->rank_col) b += c->rank_col->cap_rows * c->rank_col->elem;
        if (c->carried_weight) b += c->carried_weight->cap_rows * c->carried_weight->elem;
    }
    return b;
}

int sybl_table_column_info(const sybl_table *tc, const char *name, int *type, int64_t *exact_min, int64_t *exact_max,
                           int64_t *info_min, int64_t *info_max, int *has_missing) {
    SYBL_API_GUARD(tc);
    sybl_table *t = const_cast<sybl_table *>(tc);
    if (!t) return fail(SYBL_E_INVAL, "table is NULL");
    Column *c = t->find(name);
    if (!c) return fail(SYBL_E_INVAL, "unknown column '%s'", name ? name : "(null)");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    int rc = table_ensure_stats(t);
    if (rc) return rc;
    if (type) *type = c->type;
    if (exact_min) *exact_min = c->exact_min;
    if (exact_max) *exact_max = c->exact_max;
    if (info_min) *info_min = c->info_given ? c->info_min : c->exact_min;
    if (info_max) *info_max = c->info_given ? c->info_max : c->exact_max;
    if (has_missing) *has_missing = c->has_missing;
    return SYBL_OK;
}

int sybl_table_set_bounds(sybl_table *t, const char *name, int64_t lo, int64_t hi, int has_missing) {
    SYBL_API_GUARD(t);
    if (!t) return fail(SYBL_E_INVAL, "table is NULL");
    Column *c = t->find(name);
    if (!c) return fail(SYBL_E_INVAL, "unknown column '%s'", name ? name : "(null)");
    // lo > hi: no rank holds a value of this column -- nothing to bound, but the agreed has_missing flag still
    // applies (the MISSING key digit / populated-count field must exist on every rank or on none)
    if (lo <= hi) {
        c->bounds_set = true;
        c->bound_lo = lo;
        c->bound_hi = hi;
    }
    if (has_missing) c->has_missing = true;
    t->version++;
    return SYBL_OK;
}

int sybl_table_column_distinct(sybl_table *t, const char *name, const int64_t **values, int64_t *n) {
    SYBL_API_GUARD(t);
    if (!t || !values || !n) return fail(SYBL_E_INVAL, "NULL argument");
    Column *c = t->find(name);
    if (!c || c->type == SYBL_SET_VAL) return fail(SYBL_E_INVAL, "unknown int/str column '%s'", name ? name : "(null)");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    int rc = column_build_gdict(t, c);
    if (rc) return rc;
    *values = c->gdict.data();
    *n = (int64_t)c->gdict.size();
    return SYBL_OK;
}

int sybl_table_set_group_dict(sybl_table *t, const char *name, const int64_t *values, int64_t n) {
    SYBL_API_GUARD(t);
    if (!t || (n > 0 && !values) || n < 0) return fail(SYBL_E_INVAL, "bad argument");
    Column *c = t->find(name);
    if (!c || c->type == SYBL_SET_VAL) return fail(SYBL_E_INVAL, "unknown int/str column '%s'", name ? name : "(null)");
    if (n > kDictMaxDistinct) return fail(SYBL_E_INVAL, "dictionary of %lld values exceeds the limit of %lld", (long long)n, (long long)kDictMaxDistinct);
    SYBL_HIP(hipSetDevice(t->ctx->device));
    std::vector<int64_t> v(values, values + n);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (Query *q : t->queries) query_finish_lazy_results(q);  // (as above: pending rows read the dictionary they were scanned with)
    c->gdict.swap(v);
    c->gdict_blocks = -2;
    t->version++;
    return column_install_gdict(t, c);
}

// Str / set dictionaries across ranks: ids are assigned in first-seen order per process, so ranks
// of a multi-GPU job must agree on ONE dictionary before a str group-by (or a str/set filter
// evaluated per id) means the same thing everywhere.  Hosts gather sybl_table_column_dict from every
// rank and install the (sorted) union with sybl_table_set_dict; resident ids are remapped in place.
int sybl_table_column_dict(sybl_table *t, const char *name, const char *const **strings, int64_t *n) {
    SYBL_API_GUARD(t);
    if (!t || !strings || !n) return fail(SYBL_E_INVAL, "NULL argument");
    Column *c = t->find(name);
    if (!c || c->type == SYBL_INT_VAL) return fail(SYBL_E_INVAL, "unknown str/set column '%s'", name ? name : "(null)");
    c->dict_view.clear();
    for (auto &s : c->dict) c->dict_view.push_back(s.c_str());
    *strings = c->dict_view.data();
    *n = (int64_t)c->dict_view.size();
    return SYBL_OK;
}

int sybl_table_set_dict(sybl_table *t, const char *name, const char *const *strings, int64_t n) {
    SYBL_API_GUARD(t);
    if (!t || n < 0 || (n > 0 && !strings)) return fail(SYBL_E_INVAL, "bad argument");
    Column *c = t->find(name);
    if (!c || c->type == SYBL_INT_VAL) return fail(SYBL_E_INVAL, "unknown str/set column '%s'", name ? name : "(null)");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    // (results whose rows are still to be built read this dictionary by the ids it has now)
    for (Query *q : t->queries) query_finish_lazy_results(q);
    std::vector<std::string> nd;
    std::unordered_map<std::string, int32_t> nix;
    for (int64_t i = 0; i < n; i++) {
        std::string s = strings[i] ? strings[i] : "";
        if (nix.emplace(s, (int32_t)nd.size()).second) nd.push_back(s);
    }
    std::vector<int32_t> lut(c->dict.size());
    for (size_t i = 0; i < c->dict.size(); i++) {
        auto it = nix.find(c->dict[i]);
        if (it == nix.end()) return fail(SYBL_E_INVAL, "new dictionary of '%s' lacks the resident value '%s'", c->name.c_str(), c->dict[i].c_str());
        lut[i] = it->second;
    }
    hipStream_t st = t->ctx->stream;
    if (c->type == SYBL_STR_VAL && t->phys_rows > 0 && !lut.empty()) {
        int rc = column_repack(t, c, c->canon(), 0);  // ids are rewritten as int32
        if (rc) return rc;
        int32_t *d_lut = nullptr;
        SYBL_HIP(hipMalloc((void **)&d_lut, lut.size() * 4));
        {
            int rc2 = host_to_device(t->ctx, d_lut, lut.data(), lut.size() * 4, "dictionary look-up table");
            if (rc2) {
                hipFree(d_lut);
                return rc2;
            }
        }
        hipError_t e = launch_remap_ids((const int32_t *)c->d_data, 4, d_lut, (int32_t)lut.size(), t->phys_rows, (int32_t *)c->d_data, st);
        if (e != hipSuccess) {
            hipFree(d_lut);
            return hip_fail(e, "k_remap_ids");
        }
        SYBL_HIP(hipStreamSynchronize(st));
        SYBL_HIP(hipFree(d_lut));
    } else if (c->type == SYBL_SET_VAL) {
        for (auto &id : c->h_set_vals) id = lut[(size_t)id];
        c->set_dirty = true;
    }
    c->dict.swap(nd);
    c->dict_ix.swap(nix);
    // ids changed: block statistics of a str column (min/max id) and any group dictionary are stale
    if (c->type == SYBL_STR_VAL) {
        c->stats_blocks = 0;
        c->exact_min = INT64_MAX;
        c->exact_max = INT64_MIN;
        c->n_pop = 0;
        c->gdict_blocks = -1;
    }
    t->version++;
    return SYBL_OK;
}

static int compact_column(Table *t, Column *c) {
    if (c->type == SYBL_SET_VAL || !c->d_data) return SYBL_OK;
    int width;
    int64_t base;
    fit_storage(c, c->n_pop > 0, c->exact_min, c->exact_max, &width, &base);
    if (width >= c->elem) return SYBL_OK;  // already this narrow
    return column_repack(t, c, width, base);
}

int sybl_table_compact(sybl_table *t) {
    SYBL_API_GUARD(t);
    if (!t) return fail(SYBL_E_INVAL, "NULL table");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    int rc = table_ensure_stats(t);
    if (rc) return rc;
    for (auto &cp : t->cols) {
        const int before = cp->elem;
        if ((rc = compact_column(t, cp.get()))) return rc;
        if (cp->elem != before) t->version++;
    }
    t->compact_mode = true;  // blocks appended from now on are packed in place
    return SYBL_OK;
}

int sybl_table_column_storage(const sybl_table *t, const char *name, int32_t *width, int64_t *base) {
    SYBL_API_GUARD(t);
    if (!t) return fail(SYBL_E_INVAL, "NULL table");
    Column *c = t->find(name);
    if (!c) return fail(SYBL_E_INVAL, "unknown column '%s'", name ? name : "(null)");
    if (width) *width = c->type == SYBL_SET_VAL ? 0 : c->elem;
    if (base) *base = c->vbase;
    return SYBL_OK;
}

int sybl_table_read_int(const sybl_table *t, const char *name, int64_t row0, int64_t n, int64_t *out) {
    SYBL_API_GUARD(t);
    if (!t || !out) return fail(SYBL_E_INVAL, "NULL argument");
    Column *c = t->find(name);
    if (!c || c->type != SYBL_INT_VAL) return fail(SYBL_E_INVAL, "unknown int column '%s'", name ? name : "(null)");
    if (row0 < 0 || n < 0 || row0 + n > t->logical_rows) return fail(SYBL_E_INVAL, "row range out of bounds");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    SYBL_HIP(hipStreamSynchronize(t->ctx->stream));
    // logical -> physical: walk the blocks
    int64_t lbase = 0, done = 0;
    for (auto &b : t->blocks) {
        int64_t lo = std::max(row0, lbase), hi = std::min(row0 + n, lbase + b.n);
        if (hi > lo) {
            const int64_t first = b.start + (lo - lbase), cnt = hi - lo;
            if (c->elem == 8) {
                SYBL_HIP(hipMemcpy(out + (lo - row0), (const int64_t *)c->d_data + first, (size_t)cnt * 8, hipMemcpyDeviceToHost));
            } else {
                // compact storage: raw bytes to the host, decoded there
                std::vector<uint8_t> raw((size_t)cnt * (size_t)c->elem);
                SYBL_HIP(hipMemcpy(raw.data(), (const char *)c->d_data + (size_t)first * (size_t)c->elem, raw.size(), hipMemcpyDeviceToHost));
                for (int64_t k = 0; k < cnt; k++) {
                    uint64_t u = 0;
                    memcpy(&u, raw.data() + (size_t)k * (size_t)c->elem, (size_t)c->elem);
                    out[lo - row0 + k] = (int64_t)((uint64_t)c->vbase + u);
                }
            }
            done += cnt;
        }
        lbase += b.n;
    }
    return done == n ? SYBL_OK : fail(SYBL_E_INVAL, "short read");
}


}  // extern "C"
