// writer.cpp -- sybl_table_save: the resident table written back in the reference's on-disk format, so a
// real `sybil` binary (or this library's own loader) can open it.
//
// Mirrors SaveToColumns / SaveIntsToColumns / SaveStrsToColumns / SaveSetsToColumns / SaveInfoToColumns
// and SaveTableInfo (column_store_io.go:64-358,419-491, table_io.go:40-78): one directory per block with
// info.db + int_/str_/set_<col>.db gob files; a column with <= CARDINALITY_THRESHOLD (5000,
// column_store_io.go:18) distinct values in the block is bucket encoded -- per value the ascending,
// delta-encoded record ids (delta_encode_col, :21-30) -- otherwise value encoded (per-row Values, delta
// encoded for ints); str / set values go through a block-local StringTable; VERSION = 1.  The gob bytes
// follow encoding/gob's own rules (gobenc.h); tests/test_gpu_writer.py compares them byte for byte with the
// Python writer the loader tests use (tests/sybil_fixture.py) and reads the table back with sybl_table_open.
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <unordered_map>

#include "engine.h"
#include "gobenc.h"

namespace sybl {

namespace {

using gobenc::Buf;
using gobenc::Fields;
using gobenc::Schema;
using gobenc::Type;

constexpr size_t kCardinalityThreshold = 5000;  // column_store_io.go:18

int write_file(const std::string &path, const std::string &data) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return fail(SYBL_E_IO, "cannot write %s: %s", path.c_str(), strerror(errno));
    size_t n = fwrite(data.data(), 1, data.size(), f);
    int rc = fclose(f);
    if (n != data.size() || rc != 0) return fail(SYBL_E_IO, "short write to %s", path.c_str());
    return SYBL_OK;
}

int make_dir(const std::string &path) {
    if (mkdir(path.c_str(), 0755) != 0 && errno != EEXIST) return fail(SYBL_E_IO, "cannot create %s: %s", path.c_str(), strerror(errno));
    return SYBL_OK;
}

// decoded values and validity of one block of an int / str column
int fetch_block(const Table *t, const Column *c, const Segment &blk, std::vector<int64_t> &vals, std::vector<uint8_t> &pop) {
    const size_t n = (size_t)blk.n;
    vals.assign(n, 0);
    pop.assign(n, 1);
    if (n == 0) return SYBL_OK;
    std::vector<uint8_t> raw(n * (size_t)c->elem);
    SYBL_HIP(hipMemcpy(raw.data(), (const char *)c->d_data + (size_t)blk.start * (size_t)c->elem, raw.size(), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; k++) {
        uint64_t u = 0;
        memcpy(&u, raw.data() + k * (size_t)c->elem, (size_t)c->elem);
        vals[k] = (int64_t)((uint64_t)c->vbase + u);  // elem 8: vbase 0, the value itself
    }
    if (c->d_valid) {
        const size_t w0 = (size_t)(blk.start / 32), w1 = (size_t)((blk.start + blk.n + 31) / 32);
        std::vector<uint32_t> bits(w1 - w0);
        SYBL_HIP(hipMemcpy(bits.data(), c->d_valid + w0, bits.size() * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) {
            const size_t row = (size_t)blk.start + k;
            pop[k] = (bits[row / 32 - w0] >> (row & 31)) & 1u;
        }
    }
    return SYBL_OK;
}

// {value: ascending record ids} in first-seen order -> bins of delta-encoded ids
struct Bins {
    std::vector<int64_t> value;
    std::vector<std::vector<uint32_t>> recs;  // delta encoded
    std::vector<uint32_t> last;
    std::unordered_map<int64_t, size_t> ix;
    void add(int64_t v, uint32_t row) {
        auto it = ix.find(v);
        size_t k;
        if (it == ix.end()) {
            k = value.size();
            ix.emplace(v, k);
            value.push_back(v);
            recs.emplace_back();
            last.push_back(0);
        } else {
            k = it->second;
        }
        recs[k].push_back(row - last[k]);
        last[k] = row;
    }
};

void put_bins(Fields &f, Buf &w, int field, const Bins &bins) {
    if (bins.value.empty()) return;
    f.at(field);
    w.u(bins.value.size());
    for (size_t k = 0; k < bins.value.size(); k++) {
        Fields b(w);  // Saved*Bucket{Value, Records []uint32}
        b.put_int(0, bins.value[k]);
        b.at(1);
        w.u(bins.recs[k].size());
        for (uint32_t d : bins.recs[k]) w.u(d);
        b.end();
    }
}

void put_strings(Fields &f, Buf &w, int field, const std::vector<std::string> &table) {
    if (table.empty()) return;
    f.at(field);
    w.u(table.size());
    for (auto &s : table) w.s(s);
}

std::string encode_int_column(const std::string &name, const std::vector<int64_t> &vals, const std::vector<uint8_t> &pop) {
    Schema S;
    Type *Bool = S.basic(gobenc::kBool), *Int = S.basic(gobenc::kInt), *Uint = S.basic(gobenc::kUint), *String = S.basic(gobenc::kString);
    Type *bucket = S.strukt("SavedIntBucket", {{"Value", Int}, {"Records", S.slice(Uint, "[]uint32")}});
    Type *top = S.strukt("SavedIntColumn", {{"Name", String}, {"DeltaEncodedIDs", Bool}, {"ValueEncoded", Bool}, {"BucketEncoded", Bool},
                                            {"Bins", S.slice(bucket, "[]sybil.SavedIntBucket")}, {"Values", S.slice(Int, "[]int64")},
                                            {"VERSION", Int}});
    Bins bins;
    size_t max_r = 0;
    for (size_t r = 0; r < vals.size(); r++)
        if (pop[r]) {
            bins.add(vals[r], (uint32_t)r);
            max_r = r + 1;
        }
    Buf w;
    Fields f(w);
    f.put_str(0, name);
    f.put_bool(1, true);
    if (bins.value.size() <= kCardinalityThreshold) {
        f.put_bool(3, true);
        put_bins(f, w, 4, bins);
    } else {
        // SaveIntsToColumns :97-114: Values[max_r], rows without a value hold 0, then delta encoded
        f.put_bool(2, true);
        f.at(5);
        w.u(max_r);
        int64_t prev = 0;
        for (size_t r = 0; r < max_r; r++) {
            const int64_t v = pop[r] ? vals[r] : 0;
            w.i((int64_t)((uint64_t)v - (uint64_t)prev));
            prev = v;
        }
    }
    f.put_int(6, 1);
    f.end();
    return gobenc::Encoder().finish(top, w.b);
}

std::string encode_str_column(const std::string &name, const std::vector<int64_t> &ids, const std::vector<uint8_t> &pop,
                              const std::vector<std::string> &dict) {
    Schema S;
    Type *Bool = S.basic(gobenc::kBool), *Int = S.basic(gobenc::kInt), *Uint = S.basic(gobenc::kUint), *String = S.basic(gobenc::kString);
    Type *bucket = S.strukt("SavedStrBucket", {{"Value", Int}, {"Records", S.slice(Uint, "[]uint32")}});
    Type *top = S.strukt("SavedStrColumn", {{"Name", String}, {"DeltaEncodedIDs", Bool}, {"BucketEncoded", Bool},
                                            {"Bins", S.slice(bucket, "[]sybil.SavedStrBucket")}, {"Values", S.slice(Int, "[]int32")},
                                            {"StringTable", S.slice(String, "[]string")}, {"VERSION", Int}});
    // block-local string table in first-seen order
    std::vector<std::string> table;
    std::unordered_map<int64_t, int64_t> local;
    std::vector<int64_t> lid(ids.size(), 0);
    size_t max_r = 0;
    for (size_t r = 0; r < ids.size(); r++) {
        if (!pop[r]) continue;
        auto it = local.find(ids[r]);
        if (it == local.end()) {
            it = local.emplace(ids[r], (int64_t)table.size()).first;
            table.push_back((size_t)ids[r] < dict.size() ? dict[(size_t)ids[r]] : std::string());
        }
        lid[r] = it->second;
        max_r = r + 1;
    }
    Buf w;
    Fields f(w);
    f.put_str(0, name);
    f.put_bool(1, true);
    if (table.size() <= kCardinalityThreshold) {
        Bins bins;
        for (size_t r = 0; r < ids.size(); r++)
            if (pop[r]) bins.add(lid[r], (uint32_t)r);
        f.put_bool(2, true);
        put_bins(f, w, 3, bins);
    } else if (max_r > 0) {
        f.at(4);
        w.u(max_r);
        for (size_t r = 0; r < max_r; r++) w.i(pop[r] ? lid[r] : 0);
    }
    put_strings(f, w, 5, table);
    f.put_int(6, 1);
    f.end();
    return gobenc::Encoder().finish(top, w.b);
}

std::string encode_set_column(const std::string &name, const Table *t, const Column *c, const Segment &blk) {
    Schema S;
    Type *Bool = S.basic(gobenc::kBool), *Int = S.basic(gobenc::kInt), *Uint = S.basic(gobenc::kUint), *String = S.basic(gobenc::kString);
    Type *bucket = S.strukt("SavedSetBucket", {{"Value", Int}, {"Records", S.slice(Uint, "[]uint32")}});
    Type *top = S.strukt("SavedSetColumn", {{"Name", String}, {"Bins", S.slice(bucket, "[]sybil.SavedSetBucket")},
                                            {"Values", S.slice(S.slice(Int, ""), "[][]int32")}, {"StringTable", S.slice(String, "[]string")},
                                            {"DeltaEncodedIDs", Bool}, {"BucketEncoded", Bool}, {"VERSION", Int}});
    std::vector<std::string> table;
    std::unordered_map<int32_t, int64_t> local;
    Bins bins;  // per member value: the rows holding it
    std::vector<std::vector<int64_t>> rows((size_t)blk.n);
    size_t max_r = 0;
    for (int64_t r = 0; r < blk.n; r++) {
        const size_t row = (size_t)(blk.start + r);
        if (row + 1 >= c->h_set_off.size()) break;
        for (int64_t m = c->h_set_off[row]; m < c->h_set_off[row + 1]; m++) {
            const int32_t gid = c->h_set_vals[(size_t)m];
            auto it = local.find(gid);
            if (it == local.end()) {
                it = local.emplace(gid, (int64_t)table.size()).first;
                table.push_back((size_t)gid < c->dict.size() ? c->dict[(size_t)gid] : std::string());
            }
            bins.add(it->second, (uint32_t)r);
            rows[(size_t)r].push_back(it->second);
            max_r = (size_t)r + 1;
        }
    }
    (void)t;
    Buf w;
    Fields f(w);
    f.put_str(0, name);
    const bool bucketed = table.size() <= kCardinalityThreshold;
    if (bucketed) {
        put_bins(f, w, 1, bins);
    } else if (max_r > 0) {
        f.at(2);
        w.u(max_r);
        for (size_t r = 0; r < max_r; r++) {
            w.u(rows[r].size());
            for (int64_t x : rows[r]) w.i(x);
        }
    }
    put_strings(f, w, 3, table);
    f.put_bool(4, true);
    f.put_bool(5, bucketed);
    f.put_int(6, 1);
    f.end();
    return gobenc::Encoder().finish(top, w.b);
}

struct IntStat {
    bool any = false;
    int64_t mn = 0, mx = 0, count = 0;
    long double sum = 0, m2 = 0;
    __int128 isum = 0;
};

void put_int_info(Buf &w, int64_t mn, int64_t mx, double avg, double m2, int64_t count) {
    Fields f(w);  // IntInfo{Min, Max, Avg, M2, Count} (table_column_info.go:18-24)
    f.put_int(0, mn);
    f.put_int(1, mx);
    f.put_float(2, avg);
    f.put_float(3, m2);
    f.put_int(4, count);
    f.end();
}

}  // namespace

}  // namespace sybl

using namespace sybl;

// Test hook (no GPU needed): the gob bytes sybl_table_save would write for one block of an int column
// (kind 1) or a str column (kind 2; vals = dictionary ids into dict).  Library-owned buffer, valid until the
// next call on the thread.
extern "C" const void *sybl_debug_encode_column(int kind, const char *name, const int64_t *vals, const uint8_t *populated, int64_t n,
                                                const char *const *dict, int64_t n_dict, int64_t *n_bytes) {
    static thread_local std::string out;
    if (!name || n < 0 || (n > 0 && !vals) || !n_bytes || (kind != SYBL_INT_VAL && kind != SYBL_STR_VAL)) {
        set_error("sybl_debug_encode_column: bad argument");
        return nullptr;
    }
    std::vector<int64_t> v(vals, vals + n);
    std::vector<uint8_t> pop((size_t)n, 1);
    if (populated) pop.assign(populated, populated + n);
    if (kind == SYBL_INT_VAL) {
        out = encode_int_column(name, v, pop);
    } else {
        std::vector<std::string> d;
        for (int64_t i = 0; i < n_dict; i++) d.push_back(dict[i] ? dict[i] : "");
        out = encode_str_column(name, v, pop, d);
    }
    *n_bytes = (int64_t)out.size();
    return out.data();
}

extern "C" int sybl_table_save(sybl_table *t, const char *dir) {
    SYBL_API_GUARD(t);
    if (!t || !dir) return fail(SYBL_E_INVAL, "sybl_table_save: NULL argument");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    SYBL_HIP(hipStreamSynchronize(t->ctx->stream));
    int rc;
    for (auto &cp : t->cols)
        if (cp->type == SYBL_SET_VAL && (rc = column_upload_set(t, cp.get()))) return rc;  // (keeps the host mirror authoritative)
    if ((rc = make_dir(dir))) return rc;
    const std::string tdir = std::string(dir) + "/" + t->name;
    if ((rc = make_dir(tdir))) return rc;

    // blocks are independent: worker threads fetch, encode and write them (one thread took 16 s for a 105 M-row, 4-column
    // table); the table-level IntInfo is merged from the threads' partial statistics afterwards
    std::vector<IntStat> tstat(t->cols.size());
    auto save_block = [&](size_t b, std::vector<IntStat> &tstat, std::vector<int64_t> &vals, std::vector<uint8_t> &pop) -> int {
        int rc;
        const Segment &blk = t->blocks[b];
        // blocks sybl_table_refresh dropped (vanished / rewritten on disk: n = 0, their rows unreferenced) are not written:
        // the reference treats a block directory with NumRecords = 0 as broken
        if (blk.n <= 0) return SYBL_OK;
        char bname[32];
        snprintf(bname, sizeof(bname), "block%09lld", (long long)(b + 1));
        const std::string bdir = tdir + "/" + bname;
        if ((rc = make_dir(bdir))) return rc;
        // SavedColumnInfo{NumRecords, StrInfoMap, IntInfoMap} (column_store.go:39-44)
        gobenc::Schema S;
        Type *Int = S.basic(gobenc::kInt), *Float = S.basic(gobenc::kFloat), *String = S.basic(gobenc::kString);
        Type *intinfo = S.strukt("", {{"Min", Int}, {"Max", Int}, {"Avg", Float}, {"M2", Float}, {"Count", Int}});
        Type *strinfo = S.strukt("", {{"TopStringCount", S.map(Int, Int, "map[int32]int")}, {"Cardinality", Int}});
        Type *info_t = S.strukt("SavedColumnInfo", {{"NumRecords", Int}, {"StrInfoMap", S.map(String, strinfo, "SavedStrInfo")},
                                                    {"IntInfoMap", S.map(String, intinfo, "SavedIntInfo")}});
        Buf infos;  // the IntInfoMap entries
        size_t n_infos = 0;
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            const Column *c = t->cols[ci].get();
            if (c->type == SYBL_SET_VAL) {
                if ((rc = write_file(bdir + "/set_" + c->name + ".db", encode_set_column(c->name, t, c, blk)))) return rc;
                continue;
            }
            if ((rc = fetch_block(t, c, blk, vals, pop))) return rc;
            if (c->type == SYBL_STR_VAL) {
                if ((rc = write_file(bdir + "/str_" + c->name + ".db", encode_str_column(c->name, vals, pop, c->dict)))) return rc;
                continue;
            }
            if ((rc = write_file(bdir + "/int_" + c->name + ".db", encode_int_column(c->name, vals, pop)))) return rc;
            IntStat st;
            for (size_t r = 0; r < vals.size(); r++) {
                if (!pop[r]) continue;
                st.mn = st.any ? std::min(st.mn, vals[r]) : vals[r];
                st.mx = st.any ? std::max(st.mx, vals[r]) : vals[r];
                st.any = true;
                st.count++;
                st.isum += vals[r];
            }
            if (!st.any) continue;
            const long double mean = (long double)st.isum / (long double)st.count;
            for (size_t r = 0; r < vals.size(); r++)
                if (pop[r]) st.m2 += ((long double)vals[r] - mean) * ((long double)vals[r] - mean);
            infos.s(c->name);
            put_int_info(infos, st.mn, st.mx, (double)mean, (double)st.m2, st.count);
            n_infos++;
            IntStat &ts = tstat[ci];
            ts.mn = ts.any ? std::min(ts.mn, st.mn) : st.mn;
            ts.mx = ts.any ? std::max(ts.mx, st.mx) : st.mx;
            ts.any = true;
            ts.count += st.count;
            ts.isum += st.isum;
        }
        Buf w;
        Fields f(w);
        f.put_int(0, blk.n);
        if (n_infos) {
            f.at(2);
            w.u(n_infos);
            w.b += infos.b;
        }
        f.end();
        return write_file(bdir + "/info.db", gobenc::Encoder().finish(info_t, w.b));
    };
    {
        size_t n_threads = std::min<size_t>(std::max<unsigned>(1, std::thread::hardware_concurrency()), 16);
        if (const char *e = env("SYBL_WRITER_THREADS")) n_threads = (size_t)std::max(1, atoi(e));
        n_threads = std::min(n_threads, std::max<size_t>(t->blocks.size(), 1));
        std::vector<std::vector<IntStat>> part(n_threads, std::vector<IntStat>(t->cols.size()));
        std::vector<int> rcs(n_threads, SYBL_OK);
        std::vector<std::string> errs(n_threads);
        std::atomic<size_t> next{0};
        const int device = t->ctx->device;
        auto work = [&](size_t k) {
            if (hipSetDevice(device) != hipSuccess) {
                rcs[k] = SYBL_E_NODEVICE;
                return;
            }
            std::vector<int64_t> vals;
            std::vector<uint8_t> pop;
            for (size_t b = next++; b < t->blocks.size() && rcs[k] == SYBL_OK; b = next++) {
                rcs[k] = save_block(b, part[k], vals, pop);
                if (rcs[k]) errs[k] = sybl_last_error();
            }
        };
        std::vector<std::thread> threads;
        for (size_t k = 1; k < n_threads; k++) threads.emplace_back(work, k);
        work(0);
        for (auto &th : threads) th.join();
        for (size_t k = 0; k < n_threads; k++)
            if (rcs[k]) return fail(rcs[k], "%s", errs[k].c_str());
        for (size_t k = 0; k < n_threads; k++)
            for (size_t ci = 0; ci < t->cols.size(); ci++) {
                const IntStat &st = part[k][ci];
                if (!st.any) continue;
                IntStat &ts = tstat[ci];
                ts.mn = ts.any ? std::min(ts.mn, st.mn) : st.mn;
                ts.mx = ts.any ? std::max(ts.mx, st.mx) : st.mx;
                ts.any = true;
                ts.count += st.count;
                ts.isum += st.isum;
            }
    }

    // table info.db: getSaveTable (table_io.go:72-78): Name, KeyTable, KeyTypes, StrInfo, IntInfo
    gobenc::Schema S;
    Type *Int = S.basic(gobenc::kInt), *Float = S.basic(gobenc::kFloat), *String = S.basic(gobenc::kString);
    Type *intinfo = S.strukt("", {{"Min", Int}, {"Max", Int}, {"Avg", Float}, {"M2", Float}, {"Count", Int}});
    Type *strinfo = S.strukt("", {{"TopStringCount", S.map(Int, Int, "map[int32]int")}, {"Cardinality", Int}});
    Type *table_t = S.strukt("Table", {{"Name", String}, {"KeyTable", S.map(String, Int, "map[string]int16")},
                                       {"KeyTypes", S.map(Int, Int, "map[int16]int8")}, {"StrInfo", S.map(Int, strinfo, "StrInfoTable")},
                                       {"IntInfo", S.map(Int, intinfo, "IntInfoTable")}});
    Buf w;
    Fields f(w);
    f.put_str(0, t->name);
    if (!t->cols.empty()) {
        f.at(1);
        w.u(t->cols.size());
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            w.s(t->cols[ci]->name);
            w.i((int64_t)ci);
        }
        f.at(2);
        w.u(t->cols.size());
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            w.i((int64_t)ci);
            w.i(t->cols[ci]->type);  // INT_VAL = 1, STR_VAL = 2, SET_VAL = 3 (record.go:14-19)
        }
    }
    size_t n_int = 0;
    for (auto &s : tstat) n_int += s.any ? 1 : 0;
    if (n_int) {
        f.at(4);
        w.u(n_int);
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            const IntStat &s = tstat[ci];
            if (!s.any) continue;
            const Column *c = t->cols[ci].get();
            w.i((int64_t)ci);
            // the IntInfo the queries' histograms use: the declared one when the host gave one
            put_int_info(w, c->info_given ? c->info_min : s.mn, c->info_given ? c->info_max : s.mx,
                         (double)((long double)s.isum / (long double)s.count), 0.0, s.count);
        }
    }
    f.end();
    return write_file(tdir + "/info.db", gobenc::Encoder().finish(table_t, w.b));
}
