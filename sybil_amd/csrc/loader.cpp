// loader.cpp -- native reader for sybil table directories: the "TableBlock load" half of the
// hot path.
//
// Reference (src/lib/): LoadTableInfo (table_io.go:132-212) reads <dir>/<table>/info.db;
// LoadAndQueryRecords lists the block directories (table_query.go:40-106, file_looks_like_block
// table_io.go:214-239); LoadBlockFromDir (table_block_io.go:225-310) reads <block>/info.db and
// the int_/str_/set_<col>.db[.gz] files named by the LoadSpec and hands each to
// unpackIntCol / unpackStrCol / unpackSetCol (column_store_io.go:493-780).
//
// Here the gob streams are parsed on the host (gob.cpp), the compact decoded form (bin values
// + delta-encoded record ids, or delta-encoded value arrays) is copied to the GPU, and
// k_decode_bins / k_decode_delta / k_remap_ids (kernels.hip) un-delta and scatter it straight
// into the dense column arrays -- no AoS Record slab is ever materialised.
#include <dirent.h>
#include <string.h>
#include <sys/stat.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <future>
#include <thread>

#include "engine.h"
#include "gob.h"

namespace sybl {

static bool ends_with(const std::string &s, const char *suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// file_looks_like_block, table_io.go:214-239
static bool looks_like_block(const std::string &name) {
    if (name == "ingest" || name == ".ingest.temp" || name == "cache") return false;
    if (name.compare(0, 8, "stomache") == 0) return false;
    static const char *bad[] = {"info.db", "old", "broken", "lock", "export", "partial"};
    for (const char *b : bad)
        if (ends_with(name, b)) return false;
    return true;
}

static std::atomic<int64_t> g_file_bytes{0};   // (statistics of the load in progress: sybl_table_load_stats)
static std::atomic<int64_t> g_parse_ns{0};

static bool decode_file(const std::string &path, gob::Value &v, std::string &err) {
    std::vector<uint8_t> data;
    if (!gob::read_file(path, data, err)) return false;
    g_file_bytes += (int64_t)data.size();
    return gob::decode(data.data(), data.size(), v, err);
}

static double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static bool file_exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 || stat((p + ".gz").c_str(), &st) == 0;
}

// Staging for one table load: a pinned host ring and a device ring of the same size, handed out in
// lock step.  The decoded pieces of a column (bins, deltas, dictionary look-up tables, validity words) are
// copied into the pinned slice, sent with a truly asynchronous copy and consumed by the decode kernel on
// the same stream -- nothing waits for them; the stream is synchronised only when the ring wraps around
// (every few dozen blocks) instead of once per column.
struct Stage {
    char *h = nullptr, *d = nullptr;
    size_t cap = 0, off = 0;
    int64_t sent = 0;  // bytes handed to hipMemcpyAsync
    hipStream_t st = nullptr;
    int take(size_t bytes, void **hp, void **dp) {
        bytes = (bytes + 255) / 256 * 256;
        if (bytes > cap) {
            SYBL_HIP(hipStreamSynchronize(st));
            if (h) SYBL_HIP(hipHostFree(h));
            if (d) SYBL_HIP(hipFree(d));
            h = d = nullptr;
            cap = std::max<size_t>(std::max(bytes * 4, cap * 2), (size_t)32 << 20);
            SYBL_HIP(hipHostMalloc((void **)&h, cap, hipHostMallocDefault));
            SYBL_HIP(hipMalloc((void **)&d, cap));
            off = 0;
        }
        if (off + bytes > cap) {
            SYBL_HIP(hipStreamSynchronize(st));  // everything queued so far has consumed its slices
            off = 0;
        }
        *hp = h + off;
        *dp = d + off;
        off += bytes;
        return SYBL_OK;
    }
    // host -> pinned slice -> device slice (async); returns the device pointer
    int send(const void *src, size_t bytes, void **dp) {
        void *hp;
        int rc = take(std::max<size_t>(bytes, 16), &hp, dp);
        if (rc) return rc;
        if (bytes) {
            memcpy(hp, src, bytes);
            SYBL_HIP(hipMemcpyAsync(*dp, hp, bytes, hipMemcpyHostToDevice, st));
            sent += (int64_t)bytes;
        }
        return SYBL_OK;
    }
    ~Stage() {
        if (st && (h || d)) (void)hipStreamSynchronize(st);  // the last slices may still be in use
        if (h) hipHostFree(h);
        if (d) hipFree(d);
    }
};

// ---- phase 1 (worker threads, pure CPU): read + gob-decode + flatten one block's column files.
// Validation follows the reference: a record id or value count beyond NumRecords marks the block
// broken ("BLOCK SIZE CHANGED DURING QUERY", column_store_io.go:524-526,572-574,733-735).

struct FlatBins {
    std::vector<int64_t> val, off;
    std::vector<uint32_t> recs;
};

struct PreparedCol {
    enum Kind { kAbsent, kIntBins, kIntValues, kStrBins, kStrValues, kSet } kind = kAbsent;
    FlatBins fb;                       // *Bins
    bool delta = false, venc = false;
    std::vector<int64_t> values;       // kIntValues (delta-encoded when venc)
    std::vector<int32_t> local;        // kStrValues: block-local ids per row
    std::vector<std::string> strings;  // block StringTable (str / set)
    std::vector<int64_t> set_off;      // kSet: CSR over the block's rows, block-local member ids
    std::vector<int32_t> set_ids;
    std::vector<uint8_t> set_pop;
    // int columns: extrema over the populated rows, found by the worker (compact mode packs the block
    // without asking the GPU for them)
    bool have_stats = false;
    int64_t vmin = INT64_MAX, vmax = INT64_MIN, vpop = 0;
};

struct PreparedBlock {
    int64_t nrows = 0;
    bool unreadable = false;  // block info.db missing / undecodable / NumRecords <= 0
    bool broken = false;      // a column failed validation
    std::string why;
    std::vector<PreparedCol> cols;
};

static bool flatten_bins(const gob::Value *bins, bool delta, int64_t num_records, FlatBins &fb) {
    fb.off.assign(1, 0);
    if (!bins) return true;
    for (auto &b : bins->items) {
        const gob::Value *v = b->field("Value"), *r = b->field("Records");
        fb.val.push_back(v ? v->as_int() : 0);
        if (r && r->kind == gob::Value::kIntVec) {
            uint64_t abs = 0;
            for (int64_t x : r->ints) {
                abs = delta ? abs + (uint64_t)x : (uint64_t)x;
                if (abs >= (uint64_t)num_records) return false;
                fb.recs.push_back((uint32_t)x);
            }
        }
        fb.off.push_back((int64_t)fb.recs.size());
    }
    return true;
}

static void string_table(const gob::Value &v, std::vector<std::string> &out) {
    const gob::Value *st = v.field("StringTable");
    if (!st) return;
    for (auto &s : st->items) out.push_back(s->s);
}

struct ColSpec {
    std::string name;
    int type;
};

constexpr int64_t kMaxBlockRows = (int64_t)1 << 24;  // 256 x the reference's block size

static PreparedBlock prepare_block_unguarded(const std::string &bdir, const std::vector<ColSpec> &specs);
// A worker thread must not let an exception escape (std::bad_alloc / length_error from a damaged file): it
// would be rethrown by future::get() and leave the extern "C" entry point.  The block is skipped instead,
// like every other block the reference cannot read.
static PreparedBlock prepare_block(const std::string &bdir, const std::vector<ColSpec> &specs) {
    const auto t0 = std::chrono::steady_clock::now();
    struct Tally {
        std::chrono::steady_clock::time_point t0;
        ~Tally() { g_parse_ns += (int64_t)(seconds_since(t0) * 1e9); }
    } tally{t0};
    try {
        return prepare_block_unguarded(bdir, specs);
    } catch (const std::exception &) {
        PreparedBlock pb;
        pb.unreadable = true;
        return pb;
    }
}

static PreparedBlock prepare_block_unguarded(const std::string &bdir, const std::vector<ColSpec> &specs) {
    static const char *prefix[] = {"", "int_", "str_", "set_"};
    PreparedBlock pb;
    std::string err;
    gob::Value binfo;
    if (!decode_file(bdir + "/info.db", binfo, err)) {
        pb.unreadable = true;  // "COULDNT READ BLOCK INFO" -> block skipped (table_block_io.go:234-237)
        return pb;
    }
    const gob::Value *nr = binfo.field("NumRecords");
    pb.nrows = nr ? nr->as_int() : 0;
    // NumRecords is an int32 in the reference and blocks hold CHUNK_SIZE = 65536 rows (table.go:44); a larger value
    // can only come from a damaged or hostile info.db and would size every per-row array of this block
    if (pb.nrows <= 0 || pb.nrows > kMaxBlockRows) {
        pb.unreadable = true;  // "NUM RECORDS BELOW 0"
        return pb;
    }
    pb.cols.resize(specs.size());
    for (size_t ci = 0; ci < specs.size(); ci++) {
        PreparedCol &pc = pb.cols[ci];
        std::string path = bdir + "/" + prefix[specs[ci].type] + specs[ci].name + ".db";
        gob::Value v;
        // a missing file = column unpopulated in this block; "DECODE COL ERR": the reference logs and
        // carries on with an empty column
        if (!file_exists(path) || !decode_file(path, v, err)) continue;
        const gob::Value *f;
        bool bucket = (f = v.field("BucketEncoded")) && f->as_bool();
        pc.delta = (f = v.field("DeltaEncodedIDs")) && f->as_bool();
        pc.venc = (f = v.field("ValueEncoded")) && f->as_bool();
        bool ok = true;
        if (specs[ci].type == SYBL_INT_VAL) {  // unpackIntCol, column_store_io.go:690-780
            if (bucket) {
                pc.kind = PreparedCol::kIntBins;
                ok = flatten_bins(v.field("Bins"), pc.delta, pb.nrows, pc.fb);
                for (size_t k = 0; ok && k < pc.fb.val.size(); k++)
                    if (pc.fb.off[k + 1] > pc.fb.off[k]) {
                        pc.vmin = std::min(pc.vmin, pc.fb.val[k]);
                        pc.vmax = std::max(pc.vmax, pc.fb.val[k]);
                    }
                // (record ids of different bins are disjoint in a well-formed file; if they are not, a row is
                // only counted twice here, which makes the column look less populated than it is -- harmless)
                pc.vpop = std::min<int64_t>((int64_t)pc.fb.recs.size(), pb.nrows);
                pc.have_stats = ok;
            } else {
                pc.kind = PreparedCol::kIntValues;
                const gob::Value *vals = v.field("Values");
                if (vals && vals->kind == gob::Value::kIntVec) pc.values = vals->ints;
                ok = (int64_t)pc.values.size() <= pb.nrows;
                int64_t run = 0;  // every row below len(Values) is populated (column_store_io.go:758-766)
                for (int64_t x : pc.values) {
                    run = pc.venc ? (int64_t)((uint64_t)run + (uint64_t)x) : x;
                    pc.vmin = std::min(pc.vmin, run);
                    pc.vmax = std::max(pc.vmax, run);
                }
                pc.vpop = (int64_t)pc.values.size();
                pc.have_stats = ok;
            }
        } else if (specs[ci].type == SYBL_STR_VAL) {  // unpackStrCol, :493-609 (without -str-replace)
            string_table(v, pc.strings);
            ok = (int64_t)pc.strings.size() <= pb.nrows;
            if (ok && bucket) {
                pc.kind = PreparedCol::kStrBins;
                ok = flatten_bins(v.field("Bins"), pc.delta, pb.nrows, pc.fb);
                for (auto x : pc.fb.val) ok = ok && x >= 0 && x < (int64_t)pc.strings.size();
            } else if (ok) {
                pc.kind = PreparedCol::kStrValues;
                const gob::Value *vals = v.field("Values");
                if (vals && vals->kind == gob::Value::kIntVec) {
                    pc.local.resize(vals->ints.size());
                    for (size_t r = 0; r < vals->ints.size(); r++) pc.local[r] = (int32_t)vals->ints[r];
                }
                ok = (int64_t)pc.local.size() <= pb.nrows;
            }
        } else {  // unpackSetCol, :611-688: variable-length sets become CSR (member order is immaterial)
            pc.kind = PreparedCol::kSet;
            string_table(v, pc.strings);
            std::vector<std::vector<int32_t>> rows((size_t)pb.nrows);
            pc.set_pop.assign((size_t)pb.nrows, 0);
            if (bucket) {
                const gob::Value *bins = v.field("Bins");
                if (bins)
                    for (auto &b : bins->items) {
                        const gob::Value *bv = b->field("Value"), *br = b->field("Records");
                        int64_t id = bv ? bv->as_int() : 0;
                        if (id < 0 || id >= (int64_t)pc.strings.size()) ok = false;
                        uint64_t abs = 0;
                        if (ok && br && br->kind == gob::Value::kIntVec)
                            for (int64_t x : br->ints) {
                                abs = pc.delta ? abs + (uint64_t)x : (uint64_t)x;
                                if (abs >= (uint64_t)pb.nrows) {
                                    ok = false;
                                    break;
                                }
                                rows[(size_t)abs].push_back((int32_t)id);
                                pc.set_pop[(size_t)abs] = 1;
                            }
                    }
            } else {
                const gob::Value *vals = v.field("Values");
                int64_t n = vals && vals->kind == gob::Value::kSlice ? (int64_t)vals->items.size() : 0;
                ok = n <= pb.nrows;
                for (int64_t r = 0; ok && r < n; r++) {
                    pc.set_pop[(size_t)r] = 1;  // Populated = SET_VAL for every row below len(Values) (:681-684)
                    const gob::Value &m = *vals->items[(size_t)r];
                    if (m.kind == gob::Value::kIntVec)
                        for (int64_t id : m.ints) {
                            if (id < 0 || id >= (int64_t)pc.strings.size()) ok = false;
                            else rows[(size_t)r].push_back((int32_t)id);
                        }
                }
            }
            pc.set_off.assign(1, 0);
            for (int64_t r = 0; ok && r < pb.nrows; r++) {
                pc.set_ids.insert(pc.set_ids.end(), rows[(size_t)r].begin(), rows[(size_t)r].end());
                pc.set_off.push_back((int64_t)pc.set_ids.size());
            }
        }
        if (!ok) {
            pb.broken = true;  // "ERROR DURING COLUMN UNPACK ... SKIPPING BLOCK" (table_block_io.go:297-301)
            pb.why = "BLOCK SIZE CHANGED DURING QUERY in column '" + specs[ci].name + "'";
            return pb;
        }
    }
    return pb;
}

// ---- phase 2 (serial, in block order): dictionaries, PCIe, decode kernels

static int upload_bins(Stage &stage, const FlatBins &fb, const uint32_t **d_recs, const int64_t **d_off, const int64_t **d_val) {
    int rc;
    void *p;
    if ((rc = stage.send(fb.recs.data(), fb.recs.size() * 4, &p))) return rc;
    *d_recs = (const uint32_t *)p;
    if ((rc = stage.send(fb.off.data(), fb.off.size() * 8, &p))) return rc;
    *d_off = (const int64_t *)p;
    if ((rc = stage.send(fb.val.data(), fb.val.size() * 8, &p))) return rc;
    *d_val = (const int64_t *)p;
    return SYBL_OK;
}

static int put_prefix_valid(BlockWriter &w, Stage &stage, uint32_t *valid, int64_t n) {
    // every row below len(Values) becomes populated, holes included (column_store_io.go:758-766)
    if (!valid || n <= 0) return SYBL_OK;
    std::vector<uint32_t> bits((size_t)((w.nrows + 31) / 32), 0);
    for (int64_t r = 0; r < n; r++) bits[(size_t)(r >> 5)] |= 1u << (r & 31);
    void *hp, *dp;
    int rc = stage.take(bits.size() * 4, &hp, &dp);
    if (rc) return rc;
    memcpy(hp, bits.data(), bits.size() * 4);
    // (pinned source: the copy is asynchronous and ordered behind the memset block_col_device queued)
    SYBL_HIP(hipMemcpyAsync(valid, hp, bits.size() * 4, hipMemcpyHostToDevice, stage.st));
    return SYBL_OK;
}

static int apply_col(BlockWriter &w, Column *c, PreparedCol &pc, Stage &stage) {
    Table *t = w.t;
    hipStream_t st = t->ctx->stream;
    void *col = nullptr;
    uint32_t *valid = nullptr;
    int rc;
    std::vector<int32_t> lut;
    for (auto &s : pc.strings) lut.push_back(dict_intern(c, s));  // block-local id -> table-global id
    switch (pc.kind) {
    case PreparedCol::kAbsent: return block_col_absent(w, c);
    case PreparedCol::kIntBins:
    case PreparedCol::kStrBins: {
        bool w32 = pc.kind == PreparedCol::kStrBins;
        if (w32)
            for (auto &x : pc.fb.val) x = lut[(size_t)x];
        bool all = (int64_t)pc.fb.recs.size() == w.nrows;
        if ((rc = block_col_device(w, c, all, &col, &valid))) return rc;
        const uint32_t *d_recs;
        const int64_t *d_off, *d_val;
        if ((rc = upload_bins(stage, pc.fb, &d_recs, &d_off, &d_val))) return rc;
        hipError_t e = launch_decode_bins(d_recs, d_off, d_val, (int)pc.fb.val.size(), pc.delta, col, w32, valid, (uint32_t)w.nrows, st);
        if (e != hipSuccess) return hip_fail(e, "k_decode_bins");
        if (w32) {  // str bins: the values are table-global ids now
            int64_t mn = INT64_MAX, mx = INT64_MIN;
            for (size_t k = 0; k < pc.fb.val.size(); k++)
                if (pc.fb.off[k + 1] > pc.fb.off[k]) {
                    mn = std::min(mn, pc.fb.val[k]);
                    mx = std::max(mx, pc.fb.val[k]);
                }
            block_col_stats(w, c, mn, mx, std::min<int64_t>((int64_t)pc.fb.recs.size(), w.nrows));
        } else if (pc.have_stats) {
            block_col_stats(w, c, pc.vmin, pc.vmax, pc.vpop);
        }
        return SYBL_OK;
    }
    case PreparedCol::kIntValues: {
        int64_t n = (int64_t)pc.values.size();
        if ((rc = block_col_device(w, c, n == w.nrows, &col, &valid))) return rc;
        if ((rc = put_prefix_valid(w, stage, valid, n))) return rc;
        if (n > 0) {
            void *d_vals;
            if ((rc = stage.send(pc.values.data(), (size_t)n * 8, &d_vals))) return rc;
            hipError_t e = launch_decode_delta((const int64_t *)d_vals, n, pc.venc, (int64_t *)col, st);
            if (e != hipSuccess) return hip_fail(e, "k_decode_delta");
        }
        if (pc.have_stats) block_col_stats(w, c, pc.vmin, pc.vmax, pc.vpop);
        return SYBL_OK;
    }
    case PreparedCol::kStrValues: {
        int64_t n = (int64_t)pc.local.size();
        if ((rc = block_col_device(w, c, n == w.nrows, &col, &valid))) return rc;
        if ((rc = put_prefix_valid(w, stage, valid, n))) return rc;
        if (n > 0) {
            void *d_local, *d_lut;
            if ((rc = stage.send(pc.local.data(), (size_t)n * 4, &d_local))) return rc;
            if ((rc = stage.send(lut.data(), lut.size() * 4, &d_lut))) return rc;
            hipError_t e = launch_remap_ids((const int32_t *)d_local, (const int32_t *)d_lut, (int32_t)lut.size(), n, (int32_t *)col, st);
            if (e != hipSuccess) return hip_fail(e, "k_remap_ids");
        }
        return SYBL_OK;
    }
    case PreparedCol::kSet: {
        for (auto &id : pc.set_ids) id = lut[(size_t)id];
        return block_col_set_host(w, c, pc.set_off.data(), pc.set_ids.data(), pc.set_pop.data());
    }
    }
    return SYBL_OK;
}

static int open_table(Ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                      int32_t rank, int32_t nranks, int32_t flags, sybl_table **out) {
    std::string tdir = std::string(dir ? dir : ".") + "/" + table;
    std::string err;
    // ---- table info.db: KeyTable, KeyTypes, IntInfo (table_io.go:145-180)
    gob::Value info;
    if (!decode_file(tdir + "/info.db", info, err)) return fail(SYBL_E_IO, "%s", err.c_str());
    std::map<std::string, int64_t> key_id;
    std::map<int64_t, int> key_type;
    std::map<int64_t, std::pair<int64_t, int64_t>> int_info;
    if (const gob::Value *kt = info.field("KeyTable"))
        for (auto &e : kt->entries) key_id[e.first->s] = e.second->as_int();
    if (const gob::Value *ky = info.field("KeyTypes"))
        for (auto &e : ky->entries) key_type[e.first->as_int()] = (int)e.second->as_int();
    if (const gob::Value *ii = info.field("IntInfo"))
        for (auto &e : ii->entries) {
            const gob::Value *mn = e.second->field("Min"), *mx = e.second->field("Max");
            int_info[e.first->as_int()] = {mn ? mn->as_int() : 0, mx ? mx->as_int() : 0};
        }
    if (key_id.empty()) return fail(SYBL_E_IO, "%s/info.db has no KeyTable", tdir.c_str());

    sybl_table *t = nullptr;
    int rc = sybl_table_create((sybl_ctx *)ctx, table, &t);
    if (rc == SYBL_OK && (flags & SYBL_OPEN_COMPACT)) t->compact_mode = true;  // every block is packed as it arrives
    if (rc) return rc;
    auto bail = [&](int code) {
        sybl_table_free(t);
        return code;
    };
    std::vector<std::string> want;
    if (columns && n_columns > 0) {
        for (int i = 0; i < n_columns; i++) want.push_back(columns[i] ? columns[i] : "");
    } else {
        for (auto &kv : key_id) want.push_back(kv.first);
    }
    for (auto &name : want) {
        auto it = key_id.find(name);
        if (it == key_id.end()) return bail(fail(SYBL_E_INVAL, "column '%s' is not in the table's KeyTable", name.c_str()));
        auto ty = key_type.find(it->second);
        int type = ty == key_type.end() ? SYBL_NO_VAL : ty->second;
        if (type != SYBL_INT_VAL && type != SYBL_STR_VAL && type != SYBL_SET_VAL)
            return bail(fail(SYBL_E_IO, "column '%s' has unknown key type %d", name.c_str(), type));
        int64_t imin = 1, imax = 0;
        auto ii = int_info.find(it->second);
        if (type == SYBL_INT_VAL && ii != int_info.end()) {
            imin = ii->second.first;
            imax = ii->second.second;
        }
        if ((rc = sybl_table_add_column(t, name.c_str(), type, imin, imax))) return bail(rc);
    }

    // ---- block directories, in name order (ioutil.ReadDir sorts), sharded contiguously over ranks
    std::vector<std::string> blocks;
    DIR *d = opendir(tdir.c_str());
    if (!d) return bail(fail(SYBL_E_IO, "cannot list %s", tdir.c_str()));
    while (struct dirent *e = readdir(d)) {
        std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        struct stat st;
        if (stat((tdir + "/" + name).c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) continue;
        if (looks_like_block(name)) blocks.push_back(name);
    }
    closedir(d);
    std::sort(blocks.begin(), blocks.end());
    if (nranks < 1) nranks = 1;
    size_t b0 = blocks.size() * (size_t)rank / (size_t)nranks, b1 = blocks.size() * (size_t)(rank + 1) / (size_t)nranks;

    Stage stage;
    stage.st = ctx->stream;
    std::vector<ColSpec> specs;
    for (auto &cp : t->cols) specs.push_back({cp->name, cp->type});
    // worker threads decode a window of blocks ahead of the (serial, in-order) GPU phase
    // (the calling thread's serial phase sustains ~600 M rows/s; round 1's cap of 32 workers left the load parse-bound at
    // 180 M rows/s on a 256-thread host)
    size_t n_workers = std::min<size_t>(128, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("SYBL_LOADER_THREADS")) n_workers = (size_t)std::max(1, atoi(e));
    const size_t window = n_workers * 2;
    std::deque<std::future<PreparedBlock>> inflight;
    size_t next = b0;
    auto submit = [&]() {
        while (next < b1 && inflight.size() < window) {
            std::string bdir = tdir + "/" + blocks[next++];
            inflight.push_back(std::async(n_workers > 1 ? std::launch::async : std::launch::deferred,
                                          [bdir, &specs]() { return prepare_block(bdir, specs); }));
        }
    };
    const auto t_open = std::chrono::steady_clock::now();
    g_file_bytes = 0;
    g_parse_ns = 0;
    double wait_s = 0, apply_s = 0;
    submit();
    while (!inflight.empty()) {
        auto tw = std::chrono::steady_clock::now();
        PreparedBlock pb = inflight.front().get();
        wait_s += seconds_since(tw);
        inflight.pop_front();
        submit();
        struct Apply {
            std::chrono::steady_clock::time_point t0;
            double *acc;
            ~Apply() { *acc += seconds_since(t0); }
        } apply{std::chrono::steady_clock::now(), &apply_s};
        if (pb.unreadable || pb.broken) {
            t->broken_blocks++;
            continue;
        }
        BlockWriter w;
        if ((rc = block_begin(t, pb.nrows, &w))) return bail(rc);
        for (size_t ci = 0; ci < t->cols.size(); ci++)
            if ((rc = apply_col(w, t->cols[ci].get(), pb.cols[ci], stage))) {
                for (auto &f : inflight) f.wait();
                return bail(rc);
            }
        if ((rc = block_commit(w))) return bail(rc);
    }
    SYBL_HIP(hipStreamSynchronize(ctx->stream));  // the table is resident when the call returns (and the wall time says so)
    t->load_stats.wall_s = seconds_since(t_open);
    t->load_stats.parse_cpu_s = (double)g_parse_ns.load() * 1e-9;
    t->load_stats.wait_s = wait_s;
    t->load_stats.apply_s = apply_s;
    t->load_stats.file_bytes = g_file_bytes.load();
    t->load_stats.h2d_bytes = stage.sent;
    t->load_stats.workers = (int32_t)n_workers;
    t->load_stats.blocks = (int32_t)(b1 - b0);
    *out = t;
    return SYBL_OK;
}

}  // namespace sybl

using namespace sybl;

extern "C" {

int sybl_table_open(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                    int32_t rank, int32_t nranks, sybl_table **out) {
    if (!ctx || !table || !out || rank < 0 || (nranks > 0 && rank >= nranks)) return fail(SYBL_E_INVAL, "sybl_table_open: bad argument");
    *out = nullptr;
    SYBL_HIP(hipSetDevice(ctx->device));
    return open_table(ctx, dir, table, columns, n_columns, rank, nranks, 0, out);
}

int sybl_table_open_flags(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                          int32_t rank, int32_t nranks, int32_t flags, sybl_table **out) {
    if (!ctx || !table || !out || rank < 0 || (nranks > 0 && rank >= nranks)) return fail(SYBL_E_INVAL, "sybl_table_open: bad argument");
    *out = nullptr;
    SYBL_HIP(hipSetDevice(ctx->device));
    try {
        return open_table(ctx, dir, table, columns, n_columns, rank, nranks, flags, out);
    } catch (const std::exception &e) {
        *out = nullptr;
        return fail(SYBL_E_IO, "sybl_table_open: %s", e.what());
    }
}

int64_t sybl_table_broken_blocks(const sybl_table *t) { return t ? t->broken_blocks : 0; }

int sybl_table_load_stats(const sybl_table *t, sybl_load_stats *out) {
    if (!t || !out) return fail(SYBL_E_INVAL, "NULL argument");
    *out = t->load_stats;
    return SYBL_OK;
}

// Test hook: decodes a gob file (optionally gzipped) into JSON; the buffer is owned by the
// library and valid until the next call on this thread.
const char *sybl_debug_gob_to_json(const char *path) {
    static thread_local std::string out;
    std::string err;
    gob::Value v;
    if (!path || !decode_file(path, v, err)) {
        set_error("%s", err.empty() ? "sybl_debug_gob_to_json: bad argument" : err.c_str());
        return nullptr;
    }
    out.clear();
    gob::to_json(v, out);
    return out.c_str();
}

}  // extern "C"
