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
#include <time.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
static std::atomic<int64_t> g_parse_ns{0};      // CPU time of the workers (CLOCK_THREAD_CPUTIME_ID)
static std::atomic<int64_t> g_parse_wall_ns{0};  // their elapsed time: more than the CPU time when the cgroup throttles them
static int64_t thread_cpu_ns() {
    struct timespec ts;
    if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0;
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
// CPU time of the whole process so far (every thread: the workers, the loading thread, the HIP runtime's own)
static double process_cpu_s() {
    struct timespec ts;
    if (clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) != 0) return 0;
    return (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9;
}
// the cgroup's CPU throttling so far (cgroup v2 cpu.stat: periods throttled, microseconds throttled); zeros when unreadable
static void cgroup_throttle(int64_t *periods, int64_t *usec) {
    *periods = *usec = 0;
    FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r");
    if (!f) return;
    char key[64];
    long long v;
    while (fscanf(f, "%63s %lld", key, &v) == 2) {
        if (!strcmp(key, "nr_throttled")) *periods = v;
        else if (!strcmp(key, "throttled_usec")) *usec = v;
    }
    fclose(f);
}

// narrow: the int slices in the form they travel in (gob::DecodeOpts) -- the int / str column files of a block
// raw: the file's top-level `Values` slice is located, not decoded (gob::RawInts: pointers into this thread's file buffer, valid
// until the thread's next decode_file)
static bool decode_file(const std::string &path, gob::Value &v, std::string &err, bool narrow = false, gob::RawInts *raw = nullptr,
                        gob::RawInts *raw_bins = nullptr) {
    static thread_local gob::FileBuf data;  // (reused: no allocation / page faults / zero-fill per file)
    if (!gob::read_file(path, data, err)) return false;
    g_file_bytes += (int64_t)data.n;
    gob::DecodeOpts opts;
    opts.narrow = narrow;
    opts.raw_values = raw;
    opts.raw_bins = raw_bins;
    return gob::decode(data.p, data.n, v, err, &opts);
}

static double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Staging for one table load.  Everything a block sends across PCIe -- bin values, record ids, delta-encoded
// value arrays, block-local str ids, look-up tables, validity prefixes -- is laid out by the WORKER that decoded the
// block in a pinned host slab (narrowed on the way: record ids of a block of <= 65536 rows travel as uint16, value
// deltas as int32 when they fit), crosses with ONE asynchronous copy into the slab's device twin and is consumed there
// by the decode kernels.  The calling thread only interns dictionaries, queues the copy and launches kernels: round 2
// measured its share at 0.38 s of a 0.46 s load when it still copied 2.1 GB of decoded pieces into a pinned ring itself.
struct Slab {
    char *h = nullptr, *d = nullptr;
    size_t cap = 0;             // of both; the device twin has SlabPool::scratch_bytes more behind it
    hipEvent_t done = nullptr;  // recorded behind the last kernel that reads d
};

// The slabs are carved out of ONE pinned and ONE device allocation: pinning and unpinning 190 slabs one by one cost
// 0.1 s of a 0.4 s load (hipHostFree alone ~0.3 ms each).
struct SlabPool {
    std::vector<Slab> slabs;
    std::vector<int> free_;
    std::deque<int> pending;  // applied, event recorded, not yet known to be finished
    size_t slab_bytes = 0, max_slabs = 0;
    size_t scratch_bytes = 0;  // device only, behind every slab's twin (the GPU varint walk's values)
    Ctx *ctx = nullptr;
    static constexpr size_t kChunkSlabs = 8;
    // (the arena belongs to the context and outlives the load: Ctx::load_chunks)
    int init(Ctx *c) {
        ctx = c;
        if (c->load_slab_bytes != slab_bytes || c->load_scratch_bytes != scratch_bytes) {
            ctx_free_load_arena(c);
            c->load_slab_bytes = slab_bytes;
            c->load_scratch_bytes = scratch_bytes;
        }
        return SYBL_OK;
    }
    // the k-th slab's pinned / device pair; the chunk that holds it is allocated now if this load is the first to get there
    int slab_memory(size_t k, char **h, char **d) {
        const size_t ci = k / kChunkSlabs;
        while (ctx->load_chunks.size() <= ci) {
            Ctx::ArenaChunk ch;
            ch.slabs = kChunkSlabs;
            SYBL_HIP(hipHostMalloc((void **)&ch.h, slab_bytes * kChunkSlabs, hipHostMallocDefault));
            hipError_t e = hipMalloc((void **)&ch.d, (slab_bytes + scratch_bytes) * kChunkSlabs);
            if (e != hipSuccess) {
                (void)hipHostFree(ch.h);
                return hip_fail(e, "hipMalloc(loader arena)");
            }
            ctx->load_chunks.push_back(ch);
        }
        *h = ctx->load_chunks[ci].h + (k % kChunkSlabs) * slab_bytes;
        *d = ctx->load_chunks[ci].d + (k % kChunkSlabs) * (slab_bytes + scratch_bytes);
        return SYBL_OK;
    }
    // *out = -1 when no slab can be had right now (must_wait: block until the oldest pending one is done)
    int acquire(bool must_wait, int *out) {
        *out = -1;
        if (free_.empty() && !pending.empty()) {
            const int i = pending.front();
            hipError_t e = must_wait ? hipEventSynchronize(slabs[(size_t)i].done) : hipEventQuery(slabs[(size_t)i].done);
            if (e == hipSuccess) {
                pending.pop_front();
                free_.push_back(i);
            } else if (e != hipErrorNotReady) {
                return hip_fail(e, "slab event");
            } else {
                (void)hipGetLastError();  // "not ready" is an answer, not an error a later hipGetLastError() should report
            }
        }
        if (free_.empty() && slabs.size() < max_slabs) {
            Slab s;
            int rc = slab_memory(slabs.size(), &s.h, &s.d);
            if (rc) return rc;
            SYBL_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
            s.cap = slab_bytes;
            slabs.push_back(s);
            free_.push_back((int)slabs.size() - 1);
        }
        if (free_.empty()) return SYBL_OK;
        *out = free_.back();
        free_.pop_back();
        return SYBL_OK;
    }
    void release_after(int i, hipStream_t st) {
        (void)hipEventRecord(slabs[(size_t)i].done, st);
        pending.push_back(i);
    }
    void give_back(int i) { free_.push_back(i); }
    ~SlabPool() {
        for (auto &s : slabs)
            if (s.done) {
                (void)hipEventSynchronize(s.done);
                (void)hipEventDestroy(s.done);
            }
        if (ctx && env("SYBL_LOADER_KEEP_ARENA") && atoi(env("SYBL_LOADER_KEEP_ARENA")) == 0) ctx_free_load_arena(ctx);
    }
};

void ctx_free_load_arena(Ctx *ctx) {
    for (auto &ch : ctx->load_chunks) {
        if (ch.h) (void)hipHostFree(ch.h);
        if (ch.d) (void)hipFree(ch.d);
    }
    ctx->load_chunks.clear();
    ctx->load_slab_bytes = ctx->load_scratch_bytes = 0;
}

// Worker threads of one table load (std::async started a thread per block: 1600 thread creations, 27 us each on the
// calling thread, for a 100 M-row table).
class LoadWorkers {
   public:
    explicit LoadWorkers(size_t n) {
        for (size_t i = 0; i < n; i++) threads_.emplace_back([this]() { loop(); });
    }
    ~LoadWorkers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(std::function<void()> job) {
        if (threads_.empty()) {
            job();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            jobs_.push_back(std::move(job));
        }
        cv_.notify_one();
    }

   private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;  // (stop requested and nothing left)
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> jobs_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

// ---- phase 1 (worker threads, pure CPU): read + gob-decode one block's column files and lay the decoded pieces out
// in the block's slab.  Validation follows the reference: a record id or value count beyond NumRecords marks the
// block broken ("BLOCK SIZE CHANGED DURING QUERY", column_store_io.go:524-526,572-574,733-735).

struct PreparedCol {
    enum Kind { kAbsent, kIntBins, kIntValues, kStrBins, kStrValues, kSet } kind = kAbsent;
    bool delta = false, venc = false;
    // offsets into the slab (16-byte aligned) and element widths
    size_t rec_at = 0, binoff_at = 0, binval_at = 0, val_at = 0, local_at = 0, lut_at = 0, bits_at = 0;
    int rec_w = 4, val_w = 8, local_w = 4;
    int64_t n_recs = 0, n_bins = 0, n_vals = 0, n_local = 0, bits_words = 0;
    std::vector<std::string> strings;  // block StringTable (str / set)
    std::vector<int64_t> set_off;      // kSet: CSR over the block's rows, block-local member ids
    std::vector<int32_t> set_ids;
    std::vector<uint8_t> set_pop;
    // int columns: extrema over the populated rows, found by the worker (compact mode packs the block
    // without asking the GPU for them)
    bool have_stats = false;
    int64_t vmin = INT64_MAX, vmax = INT64_MIN, vpop = 0;
    // SYBL_LOADER_GPU_VARINT (round 6, gobgpu.hip): a value-encoded int column whose file bytes travel as they are --
    // raw_at / raw_len: the `Values` region in the slab; val_at (int64 values) and tok_at (uint32 offsets) are DEVICE-ONLY
    // scratch, relative to PreparedBlock::scratch_at (behind the bytes that cross PCIe).  vmin / vmax are then the block
    // info.db's IntInfo, which the kernel's own extrema are checked against when the load ends.
    // A bucket-encoded one likewise (raw_len > 0, rec_w == 8): the `Bins` region travels; the region's values (val_at), the ranks
    // of its zeros (tok_at), the buckets' values (binval_at) and record ranges (binoff_at) are device-only scratch; n_recs is the
    // block info.db's Count, which k_gob_bins holds the buckets' records against.
    int64_t raw_len = 0, tok_cap = 0;
    size_t raw_at = 0, tok_at = 0;
};

struct PreparedBlock {
    int64_t nrows = 0;
    bool unreadable = false;  // block info.db missing / undecodable / NumRecords <= 0
    bool broken = false;      // a column failed validation
    std::string why;
    std::vector<PreparedCol> cols;
    size_t bytes = 0;          // of the slab that are in use
    size_t scratch_at = 0;     // device-only scratch behind them (GPU varint walk): not copied
    char *own_h = nullptr, *own_d = nullptr;  // a block too large for the pool's slabs brings its own pair
    std::pair<int64_t, int64_t> sig{-1, -1};  // block_signature, taken by the worker before it reads the block
};

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

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

static PreparedBlock prepare_block_unguarded(const std::string &bdir, const std::vector<ColSpec> &specs, char *slab_h, size_t slab_cap, int device,
                                             bool streamed = true, size_t scratch_cap = 0);
static std::pair<int64_t, int64_t> block_signature(const std::string &bdir);
// A worker thread must not let an exception escape (std::bad_alloc / length_error from a damaged file): it
// would be rethrown by future::get() and leave the extern "C" entry point.  The block is skipped instead,
// like every other block the reference cannot read.
static PreparedBlock prepare_block(const std::string &bdir, const std::vector<ColSpec> &specs, char *slab_h, size_t slab_cap, int device,
                                   size_t scratch_cap = 0) {
    const auto t0 = std::chrono::steady_clock::now();
    struct Tally {
        std::chrono::steady_clock::time_point t0;
        int64_t c0;
        ~Tally() {
            g_parse_wall_ns += (int64_t)(seconds_since(t0) * 1e9);
            g_parse_ns += thread_cpu_ns() - c0;
        }
    } tally{t0, thread_cpu_ns()};
    // (what the block looked like BEFORE it was read: a rewrite in between makes the next refresh load it again)
    const std::pair<int64_t, int64_t> sig = block_signature(bdir);
    try {
        PreparedBlock pb = prepare_block_unguarded(bdir, specs, slab_h, slab_cap, device, true, scratch_cap);
        pb.sig = sig;
        return pb;
    } catch (const std::exception &) {
        PreparedBlock pb;
        pb.unreadable = true;
        pb.sig = sig;
        return pb;
    }
}

// The Bins of a bucket-encoded column as flat arrays.  The gob reader produces them directly (Value::kBinVec); a file
// whose bucket struct carries other fields went through the generic tree and is flattened here.
struct BinsView {
    int64_t n = 0;
    const int64_t *val = nullptr, *off = nullptr, *recs = nullptr;
    const uint16_t *recs16 = nullptr;  // the records when the reader produced them as uint16 (then recs is nullptr)
    std::vector<int64_t> own_val, own_off, own_recs;
    explicit BinsView(const gob::Value *bins) {
        static const int64_t zero = 0;
        off = &zero;
        if (!bins) return;
        if (bins->kind == gob::Value::kBinVec) {
            n = (int64_t)bins->bin_val.size();
            val = bins->bin_val.data();
            off = bins->bin_off.data();
            if (bins->ints.w == 2) recs16 = bins->ints.data16();
            else recs = bins->ints.data();
            return;
        }
        own_off.push_back(0);
        for (auto &b : bins->items) {
            const gob::Value *v = b->field("Value"), *r = b->field("Records");
            own_val.push_back(v ? v->as_int() : 0);
            if (r && r->kind == gob::Value::kIntVec) own_recs.insert(own_recs.end(), r->ints.begin(), r->ints.end());
            own_off.push_back((int64_t)own_recs.size());
        }
        n = (int64_t)own_val.size();
        val = own_val.data();
        off = own_off.data();
        recs = own_recs.data();
    }
};

// ---- the copies that narrow a decoded int64 array on its way into the slab, and the extrema of one.  Plain loops, plus
// AVX-512 forms picked at run time (the baseline x86-64 the library is built for has no 64-bit min / max and narrows
// through shuffles): with the varint walk out of the way these passes were a quarter of a worker's time per block.
#if defined(__x86_64__)
static const bool g_cpu512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && !env("SYBL_LOADER_NO_AVX512");
__attribute__((target("avx512f,avx512bw"))) static uint64_t narrow16_512(const int64_t *src, uint16_t *dst, int64_t n) {
    __m512i acc = _mm512_setzero_si512();
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m512i v = _mm512_loadu_si512((const void *)(src + i));
        acc = _mm512_or_si512(acc, v);
        _mm_storeu_si128((__m128i *)(dst + i), _mm512_cvtepi64_epi16(v));
    }
    uint64_t bits = (uint64_t)_mm512_reduce_or_epi64(acc);
    for (; i < n; i++) {
        bits |= (uint64_t)src[i];
        dst[i] = (uint16_t)src[i];
    }
    return bits;
}
__attribute__((target("avx512f,avx512bw"))) static void narrow32_512(const int64_t *src, int32_t *dst, int64_t n) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_si256((__m256i *)(dst + i), _mm512_cvtepi64_epi32(_mm512_loadu_si512((const void *)(src + i))));
    for (; i < n; i++) dst[i] = (int32_t)src[i];
}
__attribute__((target("avx512f,avx512bw"))) static void minmax_512(const int64_t *src, int64_t n, int64_t *lo, int64_t *hi) {
    __m512i mn = _mm512_set1_epi64(INT64_MAX), mx = _mm512_set1_epi64(INT64_MIN);
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m512i v = _mm512_loadu_si512((const void *)(src + i));
        mn = _mm512_min_epi64(mn, v);
        mx = _mm512_max_epi64(mx, v);
    }
    int64_t a = _mm512_reduce_min_epi64(mn), b = _mm512_reduce_max_epi64(mx);
    for (; i < n; i++) {
        a = std::min(a, src[i]);
        b = std::max(b, src[i]);
    }
    *lo = a;
    *hi = b;
}
// extrema of the running sums of src[0, n) (a value-encoded column's values are the running sums of its deltas): the scan
// of eight lanes is three shifted adds, the carry into the next eight one broadcast -- the scalar loop's add / min / max
// chain was 1 ns per value, a tenth of a worker's time per block
template <typename T>
__attribute__((target("avx512f,avx512bw"))) static void scan_minmax_512(const T *src, int64_t n, int64_t *lo, int64_t *hi) {
    const __m512i zero = _mm512_setzero_si512(), last = _mm512_set1_epi64(7);
    __m512i mn = _mm512_set1_epi64(INT64_MAX), mx = _mm512_set1_epi64(INT64_MIN), carry = zero;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        __m512i d = sizeof(T) == 4 ? _mm512_cvtepi32_epi64(_mm256_loadu_si256((const __m256i *)(src + i))) : _mm512_loadu_si512((const void *)(src + i));
        d = _mm512_add_epi64(d, _mm512_alignr_epi64(d, zero, 7));
        d = _mm512_add_epi64(d, _mm512_alignr_epi64(d, zero, 6));
        d = _mm512_add_epi64(d, _mm512_alignr_epi64(d, zero, 4));
        d = _mm512_add_epi64(d, carry);
        mn = _mm512_min_epi64(mn, d);
        mx = _mm512_max_epi64(mx, d);
        carry = _mm512_permutexvar_epi64(last, d);
    }
    int64_t a = _mm512_reduce_min_epi64(mn), b = _mm512_reduce_max_epi64(mx);
    uint64_t run = (uint64_t)_mm_cvtsi128_si64(_mm512_castsi512_si128(carry));
    for (; i < n; i++) {
        run += (uint64_t)(int64_t)src[i];
        a = std::min(a, (int64_t)run);
        b = std::max(b, (int64_t)run);
    }
    *lo = a;
    *hi = b;
}
#else
static const bool g_cpu512 = false;
template <typename T>
static void scan_minmax_512(const T *, int64_t, int64_t *, int64_t *) {}
static uint64_t narrow16_512(const int64_t *, uint16_t *, int64_t) { return 0; }
static void narrow32_512(const int64_t *, int32_t *, int64_t) {}
static void minmax_512(const int64_t *, int64_t, int64_t *, int64_t *) {}
#endif
// dst[i] = (uint16_t)src[i]; returns the OR of the source values (anything beyond 16 bits shows there)
static uint64_t narrow16(const int64_t *src, uint16_t *dst, int64_t n) {
    if (g_cpu512) return narrow16_512(src, dst, n);
    uint64_t bits = 0;
    for (int64_t i = 0; i < n; i++) {
        bits |= (uint64_t)src[i];
        dst[i] = (uint16_t)src[i];
    }
    return bits;
}
static void narrow32(const int64_t *src, int32_t *dst, int64_t n) {
    if (g_cpu512) return narrow32_512(src, dst, n);
    for (int64_t i = 0; i < n; i++) dst[i] = (int32_t)src[i];
}
// extrema of src[0, n) (n = 0: INT64_MAX, INT64_MIN)
static void minmax64(const int64_t *src, int64_t n, int64_t *lo, int64_t *hi) {
    if (g_cpu512) return minmax_512(src, n, lo, hi);
    int64_t a = INT64_MAX, b = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        a = std::min(a, src[i]);
        b = std::max(b, src[i]);
    }
    *lo = a;
    *hi = b;
}

// extrema of the running sums of src[0, n) (n = 0: INT64_MAX, INT64_MIN); sums wrap like the reference's int64
template <typename T>
static void scan_minmax(const T *src, int64_t n, int64_t *lo, int64_t *hi) {
    if (g_cpu512) return scan_minmax_512<T>(src, n, lo, hi);
    int64_t a = INT64_MAX, b = INT64_MIN;
    uint64_t run = 0;
    for (int64_t i = 0; i < n; i++) {
        run += (uint64_t)(int64_t)src[i];
        a = std::min(a, (int64_t)run);
        b = std::max(b, (int64_t)run);
    }
    *lo = a;
    *hi = b;
}

// Bins -> slab: bin values (int64), bin offsets (int64, n_bins + 1) and the record ids as they are in the file
// (absolute or delta-encoded) at rec_w bytes each.  False: an id beyond NumRecords.
static bool fill_bins(const BinsView &bv, bool delta, int64_t num_records, char *base, PreparedCol &pc) {
    int64_t *off = (int64_t *)(base + pc.binoff_at), *val = (int64_t *)(base + pc.binval_at);
    uint16_t *r16 = (uint16_t *)(base + pc.rec_at);
    uint32_t *r32 = (uint32_t *)(base + pc.rec_at);
    if (bv.n > 0) memcpy(val, bv.val, (size_t)bv.n * 8);
    memcpy(off, bv.off, (size_t)(bv.n + 1) * 8);
    if (bv.recs16) {
        // the reader produced uint16 already (every id / delta is below 65536): what is left is NumRecords
        const int64_t total = bv.off[bv.n];
        for (int64_t k = 0; k < bv.n; k++) {
            const uint16_t *r = bv.recs16 + bv.off[k];
            const int64_t m = bv.off[k + 1] - bv.off[k];
            uint64_t top = 0;  // the bin's largest id: the sum of its (unsigned) deltas, or the maximum of its ids
            if (delta) {
                for (int64_t i = 0; i < m; i++) top += r[i];
            } else {
                for (int64_t i = 0; i < m; i++) top = std::max<uint64_t>(top, r[i]);
            }
            if (m > 0 && top >= (uint64_t)num_records) return false;
        }
        if (pc.rec_w == 2) {
            if (total > 0) memcpy(r16, bv.recs16, (size_t)total * 2);
        } else {
            for (int64_t i = 0; i < total; i++) r32[i] = bv.recs16[i];
        }
        return true;
    }
    if (delta && pc.rec_w == 2) {
        // the common case (delta-encoded ids, a block of <= 65536 rows) without a branch per id: the deltas are unsigned,
        // so a bin's ids are ascending and its last one -- the sum -- is the one to hold against NumRecords; an id that
        // does not fit 32 bits (a negative delta of a damaged file among them) shows in the OR.  Both loops vectorise;
        // the id-by-id loop below was a quarter of a worker's time per block.
        const int64_t total = bv.off[bv.n];
        uint64_t any = 0;
        for (int64_t k = 0; k < bv.n; k++) {
            uint64_t sum = 0;
            for (int64_t i = bv.off[k]; i < bv.off[k + 1]; i++) sum += (uint64_t)bv.recs[i];
            any |= sum;  // (a sum of < 2^32 values below 2^32 does not wrap; one that is not below 2^32 shows here or below)
        }
        const uint64_t bits = narrow16(bv.recs, r16, total);
        if ((bits >> 32) != 0) return false;
        if (any >= (uint64_t)num_records) {
            // (some bin's last id is out of range -- or the OR of in-range sums merely looks so: ask bin by bin)
            for (int64_t k = 0; k < bv.n; k++) {
                uint64_t sum = 0;
                for (int64_t i = bv.off[k]; i < bv.off[k + 1]; i++) sum += (uint64_t)bv.recs[i];
                if (sum >= (uint64_t)num_records) return false;
            }
        }
        return true;
    }
    for (int64_t k = 0; k < bv.n; k++) {
        uint64_t abs = 0;
        for (int64_t i = bv.off[k]; i < bv.off[k + 1]; i++) {
            const int64_t x = bv.recs[i];
            abs = delta ? abs + (uint64_t)x : (uint64_t)x;
            if (abs >= (uint64_t)num_records) return false;
            // (an id or delta below NumRecords <= 65536 fits 16 bits; a NEGATIVE delta in a damaged file
            // wraps `abs` and was caught above)
            if (pc.rec_w == 2) r16[i] = (uint16_t)x;
            else r32[i] = (uint32_t)x;
        }
    }
    return true;
}

// <block>/info.db's IntInfoMap[name] (Min / Max: what the reference itself goes by when it skips blocks, table_block_io.go:120-135)
static bool block_int_bounds(const gob::Value &binfo, const std::string &name, int64_t *mn, int64_t *mx, int64_t *count) {
    const gob::Value *m = binfo.field("IntInfoMap");
    if (!m) return false;
    for (auto &e : m->entries) {
        if (!e.first || e.first->s != name || !e.second) continue;
        const gob::Value *a = e.second->field("Min"), *b = e.second->field("Max");
        *mn = a ? a->as_int() : 0;  // (gob leaves a zero field out)
        *mx = b ? b->as_int() : 0;
        const gob::Value *n = e.second->field("Count");
        *count = n ? n->as_int() : 0;
        return *mn <= *mx;
    }
    return false;
}

// scratch_cap > 0: the GPU varint walk is on (SYBL_LOADER_GPU_VARINT) and the slab's device twin has that many bytes behind it
static PreparedBlock prepare_block_unguarded(const std::string &bdir, const std::vector<ColSpec> &specs, char *slab_h, size_t slab_cap, int device,
                                             bool streamed, size_t scratch_cap) {
    const bool gpu_varint = scratch_cap > 0;
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
    // Column by column (streamed, the normal case): a file is decoded, its pieces get their place in the slab -- the layout
    // is the running total of the columns before it -- and are written there at once, while the decoder's arrays are still in
    // this core's cache; the tree then goes back to the thread's stock and the next file decodes into the same memory.  Round
    // 5: decoding all files first and filling the slab in a second pass kept 1.2 MB of decoded arrays alive per worker, read
    // back cold (profiles/r05_loader_streamed.txt).  A block that outgrows the pool's slab starts over in two passes
    // (plan everything, then a buffer of its own): SYBL_LOADER_TWO_PASS=1 does that for every block (A/B).
    if (env("SYBL_LOADER_TWO_PASS")) streamed = false;  // (read per call: under SYBL_ENV_LIVE a test may flip it between loads)
    std::vector<gob::Value> trees(streamed ? 0 : specs.size());
    std::vector<char> have(specs.size(), 0), bucketed(specs.size(), 0);
    size_t total = 0, scratch = 0;  // bytes that travel; device-only bytes behind them (GPU varint walk)
    gob::RawInts raw, raw_bins;     // (streamed only: a column is planned and filled before the next file is read)
    auto reserve = [&](size_t bytes) {
        total = align16(total);
        const size_t at = total;
        total += bytes;
        return at;
    };
    // ---- plan: decode column ci's file into v, decide what travels and where in the slab.  False: the block is broken.
    auto plan = [&](size_t ci, gob::Value &v) -> bool {
        PreparedCol &pc = pb.cols[ci];
        std::string path = bdir + "/" + prefix[specs[ci].type] + specs[ci].name + ".db";
        // a missing file = column unpopulated in this block; "DECODE COL ERR": the reference logs and
        // carries on with an empty column
        const bool wide = env("SYBL_LOADER_WIDE_DECODE") != nullptr;  // (A/B: int64 slices, narrowed afterwards; read per call)
        // GPU varint walk (gobgpu.hip): an int column's `Values` slice is located, not decoded -- when the block's info.db
        // says what its extrema are (the decode kernels' destinations are chosen before they run) and the block goes column
        // by column
        int64_t info_mn = 0, info_mx = 0, info_n = 0;
        const bool try_raw = gpu_varint && streamed && specs[ci].type == SYBL_INT_VAL && block_int_bounds(binfo, specs[ci].name, &info_mn, &info_mx, &info_n);
        raw = raw_bins = gob::RawInts();
        if (!decode_file(path, v, err, specs[ci].type != SYBL_SET_VAL && !wide, try_raw ? &raw : nullptr, try_raw ? &raw_bins : nullptr)) return true;
        have[ci] = 1;
        const gob::Value *f;
        const bool bucket = (f = v.field("BucketEncoded")) && f->as_bool();
        bucketed[ci] = bucket;
        pc.delta = (f = v.field("DeltaEncodedIDs")) && f->as_bool();
        pc.venc = (f = v.field("ValueEncoded")) && f->as_bool();
        bool ok = true;
        if (specs[ci].type == SYBL_SET_VAL) return true;  // sets stay on the host (CSR mirror)
        if (specs[ci].type == SYBL_STR_VAL) {
            string_table(v, pc.strings);
            ok = (int64_t)pc.strings.size() <= pb.nrows;
        }
        const gob::Value *bins = v.field("Bins"), *vals = v.field("Values");
        // (the walk's state words hold a workgroup each for 1 MB of file; the bucket parser's workgroup 8192 buckets; the records
        // of all buckets are the column's set rows, which info.db counted)
        const size_t bins_len = raw_bins.hit && raw_bins.end > raw_bins.p ? (size_t)(raw_bins.end - raw_bins.p) : 0;
        if (ok && bucket && specs[ci].type == SYBL_INT_VAL && bins_len > 0 && bins_len <= (size_t)kGobMaxWgs * kGobWgBytes && raw_bins.n >= 1 &&
            raw_bins.n <= (uint64_t)kGobMaxBins && (int64_t)raw_bins.n <= pb.nrows && info_n >= 1 && info_n <= pb.nrows) {
            pc.kind = PreparedCol::kIntBins;
            pc.n_bins = (int64_t)raw_bins.n;
            pc.n_recs = info_n;
            pc.rec_w = 8;
            pc.raw_len = (int64_t)bins_len;
            pc.raw_at = reserve(((bins_len + 63) & ~(size_t)63) + 16);
            pc.tok_cap = info_n + 5 * pc.n_bins + 16;
            auto take = [&](size_t bytes) {
                const size_t at = scratch;
                scratch += align16(bytes);
                return at;
            };
            pc.val_at = take((size_t)pc.tok_cap * 8);          // the region's values
            pc.tok_at = take((size_t)(pc.n_bins + 8) * 4);     // ranks of the zeros
            pc.binval_at = take((size_t)pc.n_bins * 8);
            pc.binoff_at = take((size_t)pc.n_bins * 16);       // [first, one-past-last) of every bucket's records
            pc.vmin = info_mn;
            pc.vmax = info_mx;
            pc.vpop = std::min<int64_t>(info_n, pb.nrows);
            pc.have_stats = true;
        } else if (ok && bucket) {
            if (raw_bins.hit) {
                // (not a case for the GPU: the file once more, through the whole reader)
                raw = raw_bins = gob::RawInts();
                v = gob::Value();
                if (!decode_file(path, v, err, specs[ci].type != SYBL_SET_VAL && !wide)) {
                    have[ci] = 0;
                    return true;
                }
                bins = v.field("Bins");
                vals = v.field("Values");
            }
            pc.kind = specs[ci].type == SYBL_INT_VAL ? PreparedCol::kIntBins : PreparedCol::kStrBins;
            const BinsView bv(bins);
            pc.n_bins = bv.n;
            pc.n_recs = bv.off[bv.n];
            pc.rec_w = pb.nrows <= 65536 ? 2 : 4;
            pc.binval_at = reserve((size_t)std::max<int64_t>(pc.n_bins, 1) * 8);
            pc.binoff_at = reserve((size_t)(pc.n_bins + 1) * 8);
            pc.rec_at = reserve((size_t)std::max<int64_t>(pc.n_recs, 1) * (size_t)pc.rec_w);
        } else if (ok) {
            const int64_t n = vals && vals->kind == gob::Value::kIntVec ? (int64_t)vals->ints.size() : 0;
            ok = n <= pb.nrows;  // unpackIntCol / unpackStrCol: more values than NumRecords
            if (specs[ci].type == SYBL_INT_VAL && raw.hit && raw.n > 0 && raw.end > raw.p && (size_t)(raw.end - raw.p) <= (size_t)kGobMaxWgs * kGobWgBytes) {
                // the file's bytes travel; the values exist on the device only (int64, behind the slab's travelling part)
                pc.kind = PreparedCol::kIntValues;
                ok = raw.n <= (uint64_t)pb.nrows;
                pc.n_vals = (int64_t)raw.n;
                pc.val_w = 8;
                pc.raw_len = (int64_t)(raw.end - raw.p);
                // (k_gob_values reads whole 64-byte chunks and sixteen bytes behind the last one)
                pc.raw_at = reserve((((size_t)pc.raw_len + 63) & ~(size_t)63) + 16);
                pc.val_at = scratch;
                scratch += align16((size_t)pc.n_vals * 8);
                pc.tok_cap = pc.n_vals;
                pc.vmin = info_mn;
                pc.vmax = info_mx;
                // fewer set values than the slice is long: the rows in between hold 0 and become populated with it
                // (column_store_io.go:97-114, 758-766), which info.db's extrema -- kept over the set values -- do not say
                if (info_n < pc.n_vals) pc.vmin = std::min<int64_t>(pc.vmin, 0), pc.vmax = std::max<int64_t>(pc.vmax, 0);
                if (ok && pc.n_vals < pb.nrows) {
                    pc.bits_words = (pb.nrows + 31) / 32;
                    pc.bits_at = reserve((size_t)pc.bits_words * 4);
                }
            } else if (specs[ci].type == SYBL_INT_VAL) {
                pc.kind = PreparedCol::kIntValues;
                pc.n_vals = n;
                // (the stored values / deltas travel as int32 when they all fit; without an exit from the loop it vectorises)
                // (decode_file asked the reader for int32 where they fit: IntBuf::w says whether they did)
                int64_t lo = 0, hi = 0;
                if (ok && n > 0 && vals->ints.w == 8) minmax64(vals->ints.data(), n, &lo, &hi);
                pc.val_w = lo < INT32_MIN || hi > INT32_MAX ? 8 : 4;
                pc.val_at = reserve((size_t)std::max<int64_t>(n, 1) * (size_t)pc.val_w);
            } else {
                pc.kind = PreparedCol::kStrValues;
                pc.n_local = n;
                pc.local_w = 2;
                for (int64_t k = 0; ok && k < n; k++) {
                    const int64_t x = vals->ints.at((size_t)k);
                    if (x < 0 || x > 65535) {
                        pc.local_w = 4;
                        break;
                    }
                }
                pc.local_at = reserve((size_t)std::max<int64_t>(n, 1) * (size_t)pc.local_w);
                pc.lut_at = reserve(std::max<size_t>(pc.strings.size(), 1) * 4);
            }
            if (ok && n < pb.nrows && n > 0 && pc.raw_len == 0) {
                // every row below len(Values) becomes populated, holes included (column_store_io.go:758-766)
                pc.bits_words = (pb.nrows + 31) / 32;
                pc.bits_at = reserve((size_t)pc.bits_words * 4);
            }
        }
        if (!ok) {
            pb.broken = true;  // "ERROR DURING COLUMN UNPACK ... SKIPPING BLOCK" (table_block_io.go:297-301)
            pb.why = "BLOCK SIZE CHANGED DURING QUERY in column '" + specs[ci].name + "'";
            return false;
        }
        return true;
    };
    // ---- fill: column ci's pieces into their places under `base`.  False: the block is broken.
    auto fill = [&](size_t ci, const gob::Value &v, char *base) -> bool {
        PreparedCol &pc = pb.cols[ci];
        if (!have[ci]) return true;
        const gob::Value *bins = v.field("Bins"), *vals = v.field("Values");
        bool ok = true;
        switch (pc.kind) {
        case PreparedCol::kIntBins:
        case PreparedCol::kStrBins: {
            if (pc.raw_len > 0) {
                memcpy(base + pc.raw_at, raw_bins.p, (size_t)pc.raw_len);
                break;
            }
            ok = fill_bins(BinsView(bins), pc.delta, pb.nrows, base, pc);
            const int64_t *off = (const int64_t *)(base + pc.binoff_at), *val = (const int64_t *)(base + pc.binval_at);
            if (ok && pc.kind == PreparedCol::kIntBins) {
                for (int64_t k = 0; k < pc.n_bins; k++)
                    if (off[k + 1] > off[k]) {
                        pc.vmin = std::min(pc.vmin, val[k]);
                        pc.vmax = std::max(pc.vmax, val[k]);
                    }
                // (record ids of different bins are disjoint in a well-formed file; if they are not, a row is
                // only counted twice here, which makes the column look less populated than it is -- harmless)
                pc.vpop = std::min<int64_t>(pc.n_recs, pb.nrows);
                pc.have_stats = true;
            } else if (ok) {
                for (int64_t k = 0; k < pc.n_bins; k++) ok = ok && val[k] >= 0 && val[k] < (int64_t)pc.strings.size();
            }
            break;
        }
        case PreparedCol::kIntValues: {
            if (pc.raw_len > 0) {
                memcpy(base + pc.raw_at, raw.p, (size_t)pc.raw_len);
                pc.vpop = pc.n_vals;
                pc.have_stats = true;
                break;
            }
            // every row below len(Values) is populated (column_store_io.go:758-766)
            int32_t *o32 = (int32_t *)(base + pc.val_at);
            int64_t *o64 = (int64_t *)(base + pc.val_at);
            if (vals && vals->kind == gob::Value::kIntVec && vals->ints.size() > 0 && vals->ints.w == 4) {
                // the reader produced int32 (the form they travel in): extrema from the 256 KB array, then one copy
                const int32_t *src = vals->ints.data32();
                const int64_t nv = (int64_t)vals->ints.size();
                int64_t mn = INT64_MAX, mx = INT64_MIN;
                if (pc.venc) {
                    scan_minmax(src, nv, &mn, &mx);
                } else {
                    int32_t a = INT32_MAX, b = INT32_MIN;
                    for (int64_t k = 0; k < nv; k++) {
                        a = std::min(a, src[k]);
                        b = std::max(b, src[k]);
                    }
                    mn = a, mx = b;
                }
                pc.vmin = std::min(pc.vmin, mn);
                pc.vmax = std::max(pc.vmax, mx);
                memcpy(o32, src, (size_t)nv * 4);
            } else if (vals && vals->kind == gob::Value::kIntVec && vals->ints.size() > 0) {
                const int64_t *src = vals->ints.data();
                const int64_t nv = (int64_t)vals->ints.size();
                int64_t mn = INT64_MAX, mx = INT64_MIN;
                if (pc.venc) {
                    scan_minmax(src, nv, &mn, &mx);
                } else {
                    minmax64(src, nv, &mn, &mx);
                }
                pc.vmin = std::min(pc.vmin, mn);
                pc.vmax = std::max(pc.vmax, mx);
                if (pc.val_w == 4) {
                    narrow32(src, o32, nv);
                } else {
                    memcpy(o64, src, (size_t)nv * 8);
                }
            }
            pc.vpop = pc.n_vals;
            pc.have_stats = true;
            break;
        }
        case PreparedCol::kStrValues: {
            int64_t k = 0;
            uint16_t *o16 = (uint16_t *)(base + pc.local_at);
            int32_t *o32 = (int32_t *)(base + pc.local_at);
            if (vals && vals->kind == gob::Value::kIntVec) {
                const int64_t nv = (int64_t)vals->ints.size();
                for (; k < nv; k++) {
                    const int64_t x = vals->ints.at((size_t)k);
                    if (pc.local_w == 2) o16[k] = (uint16_t)x;
                    else o32[k] = (int32_t)x;
                }
            }
            break;
        }
        default: break;
        }
        if (ok && pc.bits_words > 0) {
            uint32_t *bits = (uint32_t *)(base + pc.bits_at);
            const int64_t n = pc.kind == PreparedCol::kIntValues ? pc.n_vals : pc.n_local;
            for (int64_t wd = 0; wd < pc.bits_words; wd++) {
                const int64_t lo = wd * 32;
                bits[wd] = n >= lo + 32 ? 0xFFFFFFFFu : (n > lo ? (1u << (n - lo)) - 1u : 0u);
            }
        }
        if (ok && specs[ci].type == SYBL_SET_VAL) {  // unpackSetCol, :611-688: variable-length sets become CSR (member order is immaterial)
            pc.kind = PreparedCol::kSet;
            string_table(v, pc.strings);
            std::vector<std::vector<int32_t>> rows((size_t)pb.nrows);
            pc.set_pop.assign((size_t)pb.nrows, 0);
            if (bucketed[ci]) {
                const BinsView bv(bins);
                for (int64_t k = 0; ok && k < bv.n; k++) {
                    const int64_t id = bv.val[k];
                    if (id < 0 || id >= (int64_t)pc.strings.size()) ok = false;
                    uint64_t abs = 0;
                    for (int64_t i = bv.off[k]; ok && i < bv.off[k + 1]; i++) {
                        abs = pc.delta ? abs + (uint64_t)bv.recs[i] : (uint64_t)bv.recs[i];
                        if (abs >= (uint64_t)pb.nrows) {
                            ok = false;
                            break;
                        }
                        rows[(size_t)abs].push_back((int32_t)id);
                        pc.set_pop[(size_t)abs] = 1;
                    }
                }
            } else {
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
            pb.broken = true;
            pb.why = "BLOCK SIZE CHANGED DURING QUERY in column '" + specs[ci].name + "'";
            return false;
        }
        return true;
    };
    if (streamed) {
        for (size_t ci = 0; ci < specs.size(); ci++) {
            gob::Value v;
            if (!plan(ci, v)) return pb;
            if (align16(total) > slab_cap || scratch > scratch_cap) return prepare_block_unguarded(bdir, specs, slab_h, slab_cap, device, false);  // (an over-sized block)
            if (!fill(ci, v, slab_h)) return pb;
        }
        pb.bytes = align16(total);
        pb.scratch_at = slab_cap;  // (the device twin's scratch lies behind the whole slab)
        return pb;
    }
    // ---- two passes: plan every column, then fill
    for (size_t ci = 0; ci < specs.size(); ci++)
        if (!plan(ci, trees[ci])) return pb;
    pb.bytes = align16(total);
    char *base = slab_h;
    if (pb.bytes > slab_cap) {
        // larger than the pool's slabs (an over-sized block): a pinned / device pair of its own
        // (device < 0: sybl_debug_block_layout, which has no device to ask -- the caller reads pb.bytes and stops)
        if (device < 0 || hipSetDevice(device) != hipSuccess || hipHostMalloc((void **)&pb.own_h, pb.bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&pb.own_d, pb.bytes) != hipSuccess) {
            if (pb.own_h) (void)hipHostFree(pb.own_h);
            pb.own_h = pb.own_d = nullptr;
            pb.unreadable = true;
            return pb;
        }
        base = pb.own_h;
    }
    // ---- pass 2: fill
    for (size_t ci = 0; ci < specs.size(); ci++)
        if (!fill(ci, trees[ci], base)) return pb;
    return pb;
}

// ---- phase 2 (serial, in block order): dictionaries, ONE copy across PCIe, decode kernels

// dictionaries first (they decide what some slab bytes are): block-local id -> table-global id tables, written into
// the slab where the kernels expect them
static void apply_dictionaries(Table *t, PreparedBlock &pb, char *H, std::vector<std::vector<int32_t>> &luts) {
    luts.resize(t->cols.size());
    for (size_t ci = 0; ci < t->cols.size(); ci++) {
        PreparedCol &pc = pb.cols[ci];
        Column *c = t->cols[ci].get();
        std::vector<int32_t> &lut = luts[ci];
        lut.clear();
        if (pc.strings.empty() && pc.kind != PreparedCol::kStrBins && pc.kind != PreparedCol::kStrValues && pc.kind != PreparedCol::kSet) continue;
        for (auto &s : pc.strings) lut.push_back(dict_intern(c, s));
        if (pc.kind == PreparedCol::kStrBins) {
            int64_t *val = (int64_t *)(H + pc.binval_at);
            for (int64_t k = 0; k < pc.n_bins; k++) val[k] = lut[(size_t)val[k]];  // the bins' values are table-global ids now
        } else if (pc.kind == PreparedCol::kStrValues) {
            if (!lut.empty()) memcpy(H + pc.lut_at, lut.data(), lut.size() * 4);
        } else if (pc.kind == PreparedCol::kSet) {
            for (auto &id : pc.set_ids) id = lut[(size_t)id];
        }
    }
}

// The decode launches of one block, gathered while its columns are placed and issued together (DecodeBinsBatch /
// DecodeDeltaBatch: a launch per kind instead of one per column).  SYBL_LOADER_FUSED=0: a launch per column, as before
// round 4.
// One GPU varint walk of a load (SYBL_LOADER_GPU_VARINT): what the calling thread holds the kernel's status words against
// when the load ends.
struct GobCheck {
    size_t name_ix;      // of the block, in the load's list of names
    int64_t block;       // its index in Table::blocks (set at the commit)
    size_t col;
    int64_t n, mn, mx;   // the count the slice header announced; the info.db bounds the block was placed by
};

struct DecodeBatches {
    bool on = true;
    DecodeBinsBatch bins;
    DecodeDeltaBatch deltas;
    GobValuesBatch gobs;
    GobBinsBatch gbins;
    char *scratch = nullptr;                  // this block's device-only scratch (PreparedBlock::scratch_at)
    unsigned long long *d_state = nullptr;    // [kGobStateWords] per walk of the load, zeroed
    std::vector<GobCheck> checks;
    size_t name_ix = 0, col_ix = 0;  // the block / column being applied
    uint32_t nrows = 0;  // of the block being applied
    DecodeBatches() { begin(); }  // (a load without a block flushes, too)
    void begin() { bins.n = deltas.n = gobs.n = gbins.n = 0; }
    // another block's worth of jobs fits every batch
    bool room_for(size_t n_cols) const {
        return (size_t)bins.n + n_cols <= (size_t)kDecodeBatchMax && (size_t)deltas.n + n_cols <= (size_t)kDecodeBatchMax &&
               (size_t)gobs.n + n_cols <= (size_t)kGobBatchMax;
    }
    bool empty() const { return bins.n == 0 && deltas.n == 0 && gobs.n == 0 && gbins.n == 0; }
    int flush_gobs(hipStream_t st) {
        hipError_t e = launch_gob_values(gobs, st);
        gobs.n = 0;
        return e == hipSuccess ? SYBL_OK : hip_fail(e, "k_gob_values");
    }
    int flush_bins(hipStream_t st) {
        if (gobs.n > 0 || gbins.n > 0) {  // (the walks and the bucket parsers write what the bins kernel reads)
            int rc = flush_deltas(st);
            if (rc) return rc;
        }
        hipError_t e = launch_decode_bins_multi(bins, st);
        bins.n = 0;
        return e == hipSuccess ? SYBL_OK : hip_fail(e, "k_decode_bins_multi");
    }
    int flush_deltas(hipStream_t st) {
        if (gobs.n > 0) {  // (the walks write what the delta kernel reads)
            int rc = flush_gobs(st);
            if (rc) return rc;
        }
        // (the bucket parsers of walked `Bins` regions ride along: they wait for the walk only, like the delta jobs)
        hipError_t e = launch_decode_delta_multi(deltas, gbins, st);
        deltas.n = gbins.n = 0;
        return e == hipSuccess ? SYBL_OK : hip_fail(e, "k_decode_delta_multi");
    }
    int flush(hipStream_t st) {
        int rc = flush_deltas(st);
        return rc ? rc : flush_bins(st);
    }
};

static int apply_col(BlockWriter &w, Column *c, PreparedCol &pc, const char *H, char *D, size_t n_strings, DecodeBatches &batch) {
    Table *t = w.t;
    hipStream_t st = t->ctx->stream;
    void *col = nullptr;
    uint32_t *valid = nullptr;
    int rc;
    switch (pc.kind) {
    case PreparedCol::kAbsent: return block_col_absent(w, c);
    case PreparedCol::kIntBins:
    case PreparedCol::kStrBins: {
        const bool w32 = pc.kind == PreparedCol::kStrBins;
        const bool all = pc.n_recs == w.nrows;
        int64_t mn = pc.vmin, mx = pc.vmax, pop = pc.vpop;
        if (w32) {  // str bins: the values are table-global ids now
            const int64_t *off = (const int64_t *)(H + pc.binoff_at), *val = (const int64_t *)(H + pc.binval_at);
            mn = INT64_MAX;
            mx = INT64_MIN;
            for (int64_t k = 0; k < pc.n_bins; k++)
                if (off[k + 1] > off[k]) {
                    mn = std::min(mn, val[k]);
                    mx = std::max(mx, val[k]);
                }
            pop = std::min<int64_t>(pc.n_recs, w.nrows);
        }
        const bool stats = w32 || pc.have_stats;
        bool direct = false;
        if (stats && (rc = block_col_direct(w, c, all, mn, mx, pop, &col, &valid, &direct))) return rc;
        if (!direct && (rc = block_col_device(w, c, all, &col, &valid))) return rc;
        const bool walked = pc.raw_len > 0;
        unsigned long long *gob_state = nullptr;
        if (walked) {
            // the varint walk of the `Bins` region, then its buckets: file bytes -> values -> bucket values and record ranges,
            // all in the block's device-only scratch
            if ((batch.gobs.n == kGobBatchMax || batch.bins.n == kDecodeBatchMax) && (rc = batch.flush(st))) return rc;
            gob_state = batch.d_state + (size_t)kGobStateWords * batch.checks.size();
            GobValuesJob &G = batch.gobs.job[batch.gobs.n++];
            memset(&G, 0, sizeof(G));
            G.bytes = (const uint8_t *)(D + pc.raw_at);
            G.n_bytes = (uint32_t)pc.raw_len;
            G.n = (uint32_t)pc.tok_cap;
            G.out = (long long *)(batch.scratch + pc.val_at);
            G.state = gob_state;
            G.zpos = (uint32_t *)(batch.scratch + pc.tok_at);
            G.n_zpos = (uint32_t)(pc.n_bins + 8);
            G.n_wgs = (int32_t)(((pc.raw_len + 63) / 64 + kGobWgThreads - 1) / kGobWgThreads);
            GobBinsJob &P = batch.gbins.job[batch.gbins.n++];
            memset(&P, 0, sizeof(P));
            P.tok = (const unsigned long long *)G.out;
            P.zpos = G.zpos;
            P.state = gob_state;
            P.bin_val = (long long *)(batch.scratch + pc.binval_at);
            P.bin_rng = (long long *)(batch.scratch + pc.binoff_at);
            P.n_bins = (uint32_t)pc.n_bins;
            P.tok_cap = (uint32_t)pc.tok_cap;
            P.zpos_cap = G.n_zpos;
            P.n_recs = pc.n_recs;
            P.chk_min = pc.vmin;
            P.chk_max = pc.vmax;
            batch.checks.push_back(GobCheck{batch.name_ix, -1, batch.col_ix, 0, pc.vmin, pc.vmax});
        }
        if ((batch.on || walked) && pc.n_bins > 0) {
            if (batch.bins.n == kDecodeBatchMax && (rc = batch.flush_bins(st))) return rc;
            DecodeBinsJob &J = batch.bins.job[batch.bins.n++];
            memset(&J, 0, sizeof(J));
            J.recs = walked ? batch.scratch + pc.val_at : D + pc.rec_at;
            J.bin_off = (const int64_t *)(walked ? batch.scratch + pc.binoff_at : D + pc.binoff_at);
            J.bin_val = (const int64_t *)(walked ? batch.scratch + pc.binval_at : D + pc.binval_at);
            J.chk_flags = walked ? gob_state + kGobStateFlags : nullptr;
            J.col = col;
            J.valid = valid;
            J.vbase = direct ? c->vbase : 0;
            J.n_bins = (int32_t)pc.n_bins;
            J.rec_w = (uint8_t)pc.rec_w;
            J.out_w = (uint8_t)(direct ? c->elem : c->canon());
            J.delta = pc.delta ? 1 : 0;
            J.pad = 0;
            J.nrows = (uint32_t)w.nrows;
            if (!batch.on && (rc = batch.flush_bins(st))) return rc;
        } else {
            hipError_t e = launch_decode_bins(D + pc.rec_at, pc.rec_w, (const int64_t *)(D + pc.binoff_at), (const int64_t *)(D + pc.binval_at),
                                              (int)pc.n_bins, pc.delta, col, direct ? c->elem : c->canon(), direct ? c->vbase : 0, valid,
                                              (uint32_t)w.nrows, st);
            if (e != hipSuccess) return hip_fail(e, "k_decode_bins");
        }
        if (!direct && stats) block_col_stats(w, c, mn, mx, pop);
        return SYBL_OK;
    }
    case PreparedCol::kIntValues:
    case PreparedCol::kStrValues: {
        const bool ints = pc.kind == PreparedCol::kIntValues;
        const int64_t n = ints ? pc.n_vals : pc.n_local;
        bool direct = false;
        if (ints && pc.have_stats && (rc = block_col_direct(w, c, n == w.nrows, pc.vmin, pc.vmax, pc.vpop, &col, &valid, &direct))) return rc;
        if (!direct && (rc = block_col_device(w, c, n == w.nrows, &col, &valid))) return rc;
        if (valid && pc.bits_words > 0)
            SYBL_HIP(hipMemcpyAsync(valid, D + pc.bits_at, (size_t)pc.bits_words * 4, hipMemcpyDeviceToDevice, st));
        const char *values = D + pc.val_at;
        unsigned long long *gob_state = nullptr;
        if (ints && pc.raw_len > 0) {
            // the varint walk first: file bytes -> int64 values in the block's device-only scratch
            if (batch.gobs.n == kGobBatchMax && (rc = batch.flush(st))) return rc;
            GobValuesJob &G = batch.gobs.job[batch.gobs.n++];
            memset(&G, 0, sizeof(G));
            G.bytes = (const uint8_t *)(D + pc.raw_at);
            G.n_bytes = (uint32_t)pc.raw_len;
            G.n = (uint32_t)n;
            G.out = (long long *)(batch.scratch + pc.val_at);
            G.state = batch.d_state + (size_t)kGobStateWords * batch.checks.size();
            G.n_wgs = (int32_t)(((pc.raw_len + 63) / 64 + kGobWgThreads - 1) / kGobWgThreads);
            gob_state = G.state;
            batch.checks.push_back(GobCheck{batch.name_ix, -1, batch.col_ix, n, pc.vmin, pc.vmax});
            values = batch.scratch + pc.val_at;
        }
        if (n > 0 && ints && (batch.on || gob_state)) {
            if (batch.deltas.n == kDecodeBatchMax && (rc = batch.flush_deltas(st))) return rc;
            DecodeDeltaJob &J = batch.deltas.job[batch.deltas.n++];
            memset(&J, 0, sizeof(J));
            J.deltas = values;
            J.col = col;
            J.n = n;
            J.vbase = direct ? c->vbase : 0;
            J.val_w = (uint8_t)pc.val_w;
            J.out_w = (uint8_t)(direct ? c->elem : 8);
            J.venc = pc.venc ? 1 : 0;
            if (gob_state) {
                // (the block was placed by its info.db bounds: the kernel that first sees the column's values checks them)
                J.chk_flags = gob_state + kGobStateFlags;
                J.chk_min = pc.vmin;
                J.chk_max = pc.vmax;
                if (!batch.on && (rc = batch.flush_deltas(st))) return rc;
            }
        } else if (n > 0) {
            hipError_t e = ints ? launch_decode_delta(values, pc.val_w, n, pc.venc, col, direct ? c->elem : 8, direct ? c->vbase : 0, st)
                                : launch_remap_ids(D + pc.local_at, pc.local_w, (const int32_t *)(D + pc.lut_at), (int32_t)n_strings, n,
                                                   (int32_t *)col, st);
            if (e != hipSuccess) return hip_fail(e, ints ? "k_decode_delta" : "k_remap_ids");
        }
        if (ints && !direct && pc.have_stats) block_col_stats(w, c, pc.vmin, pc.vmax, pc.vpop);
        return SYBL_OK;
    }
    case PreparedCol::kSet: return block_col_set_host(w, c, pc.set_off.data(), pc.set_ids.data(), pc.set_pop.data());
    }
    return SYBL_OK;
}

struct TableInfo {
    std::map<std::string, int64_t> key_id;
    std::map<int64_t, int> key_type;
    std::map<int64_t, std::pair<int64_t, int64_t>> int_info;
};

// <table>/info.db: KeyTable, KeyTypes, IntInfo (table_io.go:145-180)
static int read_table_info(const std::string &tdir, TableInfo &ti) {
    std::string err;
    gob::Value info;
    if (!decode_file(tdir + "/info.db", info, err)) return fail(SYBL_E_IO, "%s", err.c_str());
    if (const gob::Value *kt = info.field("KeyTable"))
        for (auto &e : kt->entries) ti.key_id[e.first->s] = e.second->as_int();
    if (const gob::Value *ky = info.field("KeyTypes"))
        for (auto &e : ky->entries) ti.key_type[e.first->as_int()] = (int)e.second->as_int();
    if (const gob::Value *ii = info.field("IntInfo"))
        for (auto &e : ii->entries) {
            const gob::Value *mn = e.second->field("Min"), *mx = e.second->field("Max");
            ti.int_info[e.first->as_int()] = {mn ? mn->as_int() : 0, mx ? mx->as_int() : 0};
        }
    if (ti.key_id.empty()) return fail(SYBL_E_IO, "%s/info.db has no KeyTable", tdir.c_str());
    return SYBL_OK;
}

// block directories in name order (ioutil.ReadDir sorts), this rank's contiguous share of them
static int list_blocks(const std::string &tdir, int32_t rank, int32_t nranks, std::vector<std::string> &mine) {
    std::vector<std::string> blocks;
    DIR *d = opendir(tdir.c_str());
    if (!d) return fail(SYBL_E_IO, "cannot list %s", tdir.c_str());
    while (struct dirent *e = readdir(d)) {
        std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        if (!looks_like_block(name)) continue;
        // (the directory entry says what it is on most file systems: 1600 stat calls were part of every open)
        if (e->d_type != DT_DIR) {
            struct stat st;
            if (e->d_type != DT_UNKNOWN && e->d_type != DT_LNK) continue;
            if (stat((tdir + "/" + name).c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) continue;
        }
        blocks.push_back(name);
    }
    closedir(d);
    std::sort(blocks.begin(), blocks.end());
    if (nranks < 1) nranks = 1;
    const size_t b0 = blocks.size() * (size_t)rank / (size_t)nranks, b1 = blocks.size() * (size_t)(rank + 1) / (size_t)nranks;
    mine.assign(blocks.begin() + (long)b0, blocks.begin() + (long)b1);
    return SYBL_OK;
}

// what a block looked like when it was loaded: <block>/info.db's modification time (ns) and size -- digest rewrites
// it whenever it rewrites the block (column_store_io.go:308-358)
static std::pair<int64_t, int64_t> block_signature(const std::string &bdir) {
    struct stat st;
    if (stat((bdir + "/info.db").c_str(), &st) != 0 && stat((bdir + "/info.db.gz").c_str(), &st) != 0) return {-1, -1};
    return {(int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec, (int64_t)st.st_size};
}

// A resident block's rows leave the scan (its directory vanished, was rewritten, or its load has to be done again)
static void table_retire_block(Table *t, int64_t index) {
    if (index < 0 || index >= (int64_t)t->blocks.size() || t->blocks[(size_t)index].n <= 0) return;
    const int64_t n = t->blocks[(size_t)index].n;
    t->blocks[(size_t)index].n = 0;
    t->logical_rows -= n;
    for (auto &cp : t->cols) {
        Column *c = cp.get();
        if (c->type == SYBL_SET_VAL || (int64_t)c->blk_pop.size() <= index) continue;
        c->n_pop -= c->blk_pop[(size_t)index];
        c->blk_pop[(size_t)index] = 0;  // (exact_min / exact_max stay: bounds of a superset are still bounds)
    }
}

// Loads the named block directories of tdir, in order, behind the table's resident blocks (the pipeline of
// sybl_table_open and sybl_table_refresh).  Every block is committed atomically; on an error the blocks appended so far
// stay.
static int load_blocks_once(Ctx *ctx, Table *t, const std::string &tdir, const std::vector<std::string> &names, bool allow_gpu_varint,
                            std::vector<std::string> &again) {
    int rc;
    const size_t n_names = names.size();
    std::vector<ColSpec> specs;
    for (auto &cp : t->cols) specs.push_back({cp->name, cp->type});
    // Round 6: the `Values` / `Bins` slices of int column files cross PCIe as file bytes and are walked on the GPU (gobgpu.hip);
    // a block whose walk reports anything unexpected is loaded again by the host parser when the load ends.
    // SYBL_LOADER_GPU_VARINT=0: every file through the host parser (gob.cpp), as before.
    bool gpu_varint = allow_gpu_varint;
    if (const char *e = env("SYBL_LOADER_GPU_VARINT")) gpu_varint = allow_gpu_varint && atoi(e) != 0;
    size_t n_int_cols = 0;
    for (auto &sp : specs) n_int_cols += sp.type == SYBL_INT_VAL ? 1 : 0;
    if (n_int_cols == 0) gpu_varint = false;
    // worker threads decode a window of blocks ahead of the (serial, in-order) GPU phase
    // (twice the CPUs this process may use: a container's CFS quota -- 16 CPUs on the 256-thread GPU boxes of round 2 --
    // throttles every thread of the group once a burst of 128 workers has spent the period's budget)
    size_t n_workers = std::min<size_t>(128, std::max<size_t>(2, 2 * usable_cpus()));
    if (const char *e = env("SYBL_LOADER_THREADS")) n_workers = (size_t)std::max(1, atoi(e));
    // one slab per block in flight: sized for a reference block (65536 rows) of every requested column -- a value-
    // encoded int column is at most 8 bytes per row, a str column adds its look-up table; larger blocks bring their own
    SlabPool pool;
    {
        size_t per_block = 65536;
        for (auto &sp : specs) per_block += sp.type == SYBL_SET_VAL ? 0 : (size_t)65536 * (sp.type == SYBL_STR_VAL ? 12 : 8) + ((size_t)96 << 10);
        // (an int column's file bytes travel instead of what a worker would have made of them -- at most nine per value; the
        // int64 values they become, the buckets' values and record ranges and the ranks of the zeros exist behind the slab's
        // device twin only)
        if (gpu_varint) {
            per_block += n_int_cols * ((size_t)65536 + 4096);
            pool.scratch_bytes = align16(n_int_cols * ((size_t)65536 * 8 + (size_t)kGobMaxBins * 72 + 4096));
        }
        pool.slab_bytes = align16(per_block);
        if (const char *e = env("SYBL_LOADER_SLAB_BYTES")) pool.slab_bytes = align16((size_t)std::max(16, atoi(e)));  // (tests: over-sized blocks)
        pool.max_slabs = std::min<size_t>(std::max<size_t>(((size_t)512 << 20) / pool.slab_bytes, 4), std::max<size_t>(2 * n_workers, 4));
        if (const char *e = env("SYBL_LOADER_SLABS")) pool.max_slabs = (size_t)std::max(2, atoi(e));  // (tuning)
        pool.max_slabs = std::min<size_t>(pool.max_slabs, std::max<size_t>(n_names, 1));
        if ((rc = pool.init(ctx))) return (rc);
    }
    const size_t window = std::min(n_workers * 2, pool.max_slabs);
    struct InFlight {
        std::future<PreparedBlock> fut;
        int slab;
        size_t name_ix;
    };
    std::deque<InFlight> inflight;
    LoadWorkers workers(n_workers > 1 ? n_workers : 0);  // (declared after the pool and the queue: joined first)
    size_t next = 0;
    const int device = ctx->device;
    auto submit = [&]() -> int {
        while (next < n_names && inflight.size() < window) {
            int slab = -1;
            int rc2 = pool.acquire(inflight.empty(), &slab);
            if (rc2) return rc2;
            if (slab < 0) break;  // every slab is in flight or still being read by the GPU
            std::string bdir = tdir + "/" + names[next++];
            char *sh = pool.slabs[(size_t)slab].h;
            const size_t scap = pool.slabs[(size_t)slab].cap;
            auto prom = std::make_shared<std::promise<PreparedBlock>>();
            inflight.push_back(InFlight{prom->get_future(), slab, next - 1});
            const size_t xcap = pool.scratch_bytes;
            workers.run([prom, bdir, &specs, sh, scap, device, xcap]() { prom->set_value(prepare_block(bdir, specs, sh, scap, device, xcap)); });
        }
        return SYBL_OK;
    };
    auto drain = [&]() {
        for (auto &f : inflight) {
            PreparedBlock pb = f.fut.get();
            if (pb.own_h) (void)hipHostFree(pb.own_h);
            if (pb.own_d) (void)hipFree(pb.own_d);
        }
        inflight.clear();
    };
    // the columns grow to about this many rows: reserve once instead of doubling through ~20 reallocations (each a
    // device malloc + copy + synchronise + free)
    t->reserve_hint_rows = std::max(t->reserve_hint_rows, t->phys_rows + (int64_t)n_names * (SYBL_BLOCK_ROWS + 32));
    // consecutive blocks go to different streams (Ctx::load_streams): restored, and everything drained, on every way out
    struct StreamGuard {
        Ctx *ctx;
        hipStream_t saved;
        ~StreamGuard() {
            (void)load_sync_all(ctx);
            ctx->load_multi = false;
            ctx->stream = saved;
        }
    } stream_guard{ctx, ctx->stream};
    {
        // (sixteen: a block's copy -> decode kernels -> event chain is ~100 us of GPU timeline on its stream, mostly the gaps
        // between dependent commands, and a slab is only handed to the next block when that chain has finished -- with four
        // streams the workers ran out of slabs: the 7-column bench table loaded in 0.116 s with four, 0.084 s with eight,
        // 0.063 s with sixteen, tools/r04 sweep in profiles/r04_loader_sweep.txt)
        int ns = 16;
        if (const char *e = env("SYBL_LOADER_STREAMS")) ns = std::max(1, std::min((int)Ctx::kMaxLoadStreams, atoi(e)));
        if (ns > 1) {
            SYBL_HIP(hipStreamSynchronize(ctx->stream));  // whatever the caller queued comes first
            for (int i = 0; i < ns; i++)
                if (!ctx->load_streams[i]) SYBL_HIP(hipStreamCreateWithFlags(&ctx->load_streams[i], hipStreamNonBlocking));
            ctx->n_load_streams = ns;
            ctx->load_multi = true;
        }
    }
    int64_t block_no = 0;
    const auto t_open = std::chrono::steady_clock::now();
    g_file_bytes = 0;
    g_parse_ns = 0;
    g_parse_wall_ns = 0;
    int64_t thr_n0, thr_us0;
    cgroup_throttle(&thr_n0, &thr_us0);
    const double cpu0 = process_cpu_s();
    double wait_s = 0, apply_s = 0;
    int64_t h2d_bytes = 0;
    std::vector<std::vector<int32_t>> luts;
    DecodeBatches batch;
    if (const char *e = env("SYBL_LOADER_FUSED")) batch.on = atoi(e) != 0;
    // the walks' state words (flags, values found, the workgroups' look-back words), zero before a kernel has seen them
    struct StateGuard {
        unsigned long long *d = nullptr;
        ~StateGuard() {
            if (d) (void)hipFree(d);
        }
    } status;
    const size_t n_status = gpu_varint ? n_names * n_int_cols : 0;
    if (n_status > 0) {
        SYBL_HIP(hipMalloc((void **)&status.d, n_status * kGobStateWords * 8));
        // (the load's streams do not wait for the null stream: the zeros are there before the first walk is queued)
        SYBL_HIP(hipMemsetAsync(status.d, 0, n_status * kGobStateWords * 8, ctx->stream));
        SYBL_HIP(hipStreamSynchronize(ctx->stream));
        batch.d_state = status.d;
    }
    // SYBL_LOADER_TRACE=1: where the calling thread's time goes (stderr)
    const bool trace = env("SYBL_LOADER_TRACE") != nullptr;
    double tr[5] = {0, 0, 0, 0, 0};  // dictionaries, copy, column kernels, commit, submit
    auto lap = [&](int k, std::chrono::steady_clock::time_point &t0) {
        if (!trace) return;
        auto t1 = std::chrono::steady_clock::now();
        tr[k] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    // (workers still hold references to this frame's state: every way out drains what is in flight)
    struct DrainGuard {
        decltype(drain) &d;
        ~DrainGuard() { d(); }
    } drain_guard{drain};
    // Consecutive blocks share a stream and their decode launches: the columns of up to kGroupBlocks blocks (as many as the job
    // tables hold) go into ONE walk, ONE delta + bucket and ONE bins launch -- a block's chain of copy -> walk -> delta -> bins
    // is ~90 us of mostly latency whatever it carries, the load's sixteen streams share a handful of hardware queues, and the
    // calling thread pays every launch.  SYBL_LOADER_GROUP=1: a launch set per block, as before.  A block that needs its
    // kernels issued before its commit (columns staged for a repack, a serial block, a slab of its own) ends its group.
    size_t kGroupBlocks = 4;
    if (const char *e = env("SYBL_LOADER_GROUP")) kGroupBlocks = (size_t)std::max(1, atoi(e));
    struct Group {
        hipStream_t st = nullptr;
        std::vector<int> slabs;
        size_t blocks = 0;
    } group;
    auto flush_group = [&]() -> int {
        int frc = SYBL_OK;
        if (!batch.empty()) frc = batch.flush(group.st);
        for (int sl : group.slabs) pool.release_after(sl, group.st);  // (behind the kernels that read them)
        group.slabs.clear();
        group.blocks = 0;
        return frc;
    };
    // (whoever waits for the load's streams -- a column about to grow, a widening repack -- has the held launches issued first)
    struct FlushHook {
        Ctx *ctx;
        ~FlushHook() { ctx->load_flush = nullptr; }
    } flush_hook{ctx};
    ctx->load_flush = flush_group;
    if ((rc = submit())) return (rc);
    while (!inflight.empty()) {
        auto tw = std::chrono::steady_clock::now();
        PreparedBlock pb = inflight.front().fut.get();
        const int slab = inflight.front().slab;
        const size_t name_ix = inflight.front().name_ix;
        wait_s += seconds_since(tw);
        inflight.pop_front();
        struct Apply {
            std::chrono::steady_clock::time_point t0;
            double *acc;
            ~Apply() { *acc += seconds_since(t0); }
        } apply{std::chrono::steady_clock::now(), &apply_s};
        auto fail_out = [&](int code) {
            (void)flush_group();  // (blocks committed earlier in the group keep their decode)
            if (pb.own_h) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipHostFree(pb.own_h);
            }
            if (pb.own_d) (void)hipFree(pb.own_d);
            drain();
            return code;
        };
        if (pb.unreadable || pb.broken) {
            t->broken_blocks++;
            t->loaded.push_back(LoadedBlock{names[name_ix], pb.sig.first, pb.sig.second, -1});
            pool.give_back(slab);
            if ((rc = submit())) return fail_out(rc);
            continue;
        }
        char *H = pb.own_h ? pb.own_h : pool.slabs[(size_t)slab].h, *D = pb.own_h ? pb.own_d : pool.slabs[(size_t)slab].d;
        if (group.blocks == 0) {
            group.st = ctx->load_multi ? ctx->load_streams[block_no++ % ctx->n_load_streams] : ctx->stream;
            batch.begin();
        }
        ctx->stream = group.st;
        auto tl = std::chrono::steady_clock::now();
        apply_dictionaries(t, pb, H, luts);
        lap(0, tl);
        if (pb.bytes > 0) {
            hipError_t e = hipMemcpyAsync(D, H, pb.bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return fail_out(hip_fail(e, "hipMemcpyAsync(block slab)"));
            h2d_bytes += (int64_t)pb.bytes;
        }
        lap(1, tl);
        BlockWriter w;
        if ((rc = block_begin(t, pb.nrows, &w))) return fail_out(rc);
        batch.nrows = (uint32_t)pb.nrows;
        batch.scratch = D + pb.scratch_at;
        batch.name_ix = name_ix;
        const size_t checks0 = batch.checks.size();
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            batch.col_ix = ci;
            if ((rc = apply_col(w, t->cols[ci].get(), pb.cols[ci], H, D, luts[ci].size(), batch))) return fail_out(rc);
        }
        // (a commit that packs staged columns or drains the stream reads what the decode kernels wrote: they go first)
        const bool ends_group = !w.staged.empty() || w.serial || pb.own_h != nullptr || !batch.on;
        if (ends_group && (rc = batch.flush(group.st))) return fail_out(rc);
        lap(2, tl);
        if ((rc = block_commit(w))) return fail_out(rc);
        for (size_t k = checks0; k < batch.checks.size(); k++) batch.checks[k].block = (int64_t)t->blocks.size() - 1;
        t->loaded.push_back(LoadedBlock{names[name_ix], pb.sig.first, pb.sig.second, (int64_t)t->blocks.size() - 1});
        lap(3, tl);
        if (pb.own_h) {
            // an over-sized block's private pair: wait for its kernels, then let it go
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipHostFree(pb.own_h);
            (void)hipFree(pb.own_d);
            pb.own_h = pb.own_d = nullptr;
            pool.give_back(slab);
        } else {
            group.slabs.push_back(slab);
        }
        group.blocks++;
        // (the queue running dry also ends a group: nothing is gained by holding launches back while the workers are behind)
        if (ends_group || group.blocks >= kGroupBlocks || !batch.room_for(t->cols.size()) || inflight.empty() ||
            inflight.front().fut.wait_for(std::chrono::seconds(0)) != std::future_status::ready) {
            if ((rc = flush_group())) return fail_out(rc);
        }
        if ((rc = submit())) return fail_out(rc);
        lap(4, tl);
    }
    if ((rc = flush_group())) return rc;
    ctx->load_flush = nullptr;
    if (trace) {
        int64_t thr_n1, thr_us1;
        cgroup_throttle(&thr_n1, &thr_us1);
        fprintf(stderr, "loader: %zu workers, parse %.3f s CPU in %.3f s of worker time, the whole process %.3f s CPU; cgroup throttled %lld periods, %.3f s\n",
                n_workers, (double)g_parse_ns.load() * 1e-9, (double)g_parse_wall_ns.load() * 1e-9, process_cpu_s() - cpu0, (long long)(thr_n1 - thr_n0),
                (double)(thr_us1 - thr_us0) * 1e-6);
    }
    if (trace)
        fprintf(stderr, "loader: dictionaries %.3f s, copy %.3f s, column kernels %.3f s, commit %.3f s, submit %.3f s, wait %.3f s, slabs %zu x %zu KB\n",
                tr[0], tr[1], tr[2], tr[3], tr[4], wait_s, pool.slabs.size(), pool.slab_bytes >> 10);
    if ((rc = load_sync_all(ctx))) return (rc);
    SYBL_HIP(hipStreamSynchronize(ctx->stream));  // the table is resident when the call returns (and the wall time says so)
    // the GPU varint walks' verdicts: anything but "every announced value found, all of them within the bounds the block was
    // placed by" sends the block through the host parser again (its rows here leave the scan, like a rewritten block's)
    t->load_stats.gpu_varint_cols = t->load_stats.gpu_varint_redone = 0;
    if (!batch.checks.empty()) {
        std::vector<unsigned long long> got(batch.checks.size() * kGobStateWords);
        SYBL_HIP(hipMemcpy(got.data(), status.d, got.size() * 8, hipMemcpyDeviceToHost));
        std::vector<char> bad_name(n_names, 0);
        for (size_t k = 0; k < batch.checks.size(); k++) {
            const GobCheck &ck = batch.checks[k];
            const unsigned long long *g = &got[(size_t)kGobStateWords * k];
            // (`Values`: every announced value found; `Bins`: n = 0 -- k_gob_bins has held the buckets against what was announced)
            // `Values`: behind the n-th value the struct ends -- [0], or [1][VERSION][0] (SavedIntColumn's last field) -- and with
            // it the region: what the host parser would have read there
            const unsigned long long *tl = g + kGobStateTail;
            const bool ends = ck.n == 0 || (g[kGobStateFound] == (unsigned long long)ck.n + 1 && tl[0] == 1) ||
                              (g[kGobStateFound] == (unsigned long long)ck.n + 3 && tl[0] == 2 && tl[2] == 1);
            if (g[kGobStateFlags] == 0 && ends) continue;
            if (trace)
                fprintf(stderr, "loader: gpu varint walk of block %s column %zu: flags 0x%llx, found %llu of %lld values, %llu zeros, %llu records\n",
                        names[ck.name_ix].c_str(), ck.col, (unsigned long long)g[kGobStateFlags], (unsigned long long)g[kGobStateFound], (long long)ck.n,
                        (unsigned long long)g[kGobStateZeros], (unsigned long long)g[kGobStateRecs]);
            if (bad_name[ck.name_ix]) continue;
            bad_name[ck.name_ix] = 1;
            table_retire_block(t, ck.block);
            for (auto &lb : t->loaded)
                if (lb.index == ck.block) lb.index = -2;  // (forgotten below)
            again.push_back(names[ck.name_ix]);
        }
        if (!again.empty()) {
            t->loaded.erase(std::remove_if(t->loaded.begin(), t->loaded.end(), [](const LoadedBlock &lb) { return lb.index == -2; }), t->loaded.end());
            t->version++;
        }
        t->load_stats.gpu_varint_cols = (int32_t)std::min<size_t>(batch.checks.size(), INT32_MAX);
        t->load_stats.gpu_varint_redone = (int32_t)again.size();
    }
    t->load_stats.wall_s = seconds_since(t_open);
    t->load_stats.parse_cpu_s = (double)g_parse_ns.load() * 1e-9;
    t->load_stats.wait_s = wait_s;
    t->load_stats.apply_s = apply_s;
    t->load_stats.file_bytes = g_file_bytes.load();
    t->load_stats.h2d_bytes = h2d_bytes;
    t->load_stats.workers = (int32_t)n_workers;
    t->load_stats.blocks = (int32_t)n_names;
    return SYBL_OK;
}

static int load_blocks(Ctx *ctx, Table *t, const std::string &tdir, const std::vector<std::string> &names) {
    std::vector<std::string> again, none;
    int rc = load_blocks_once(ctx, t, tdir, names, true, again);
    if (rc || again.empty()) return rc;
    // blocks whose GPU varint walk did not come out as announced: the host parser takes them (behind the others)
    const sybl_load_stats first = t->load_stats;
    if ((rc = load_blocks_once(ctx, t, tdir, again, false, none))) return rc;
    sybl_load_stats &ls = t->load_stats;
    ls.wall_s += first.wall_s, ls.parse_cpu_s += first.parse_cpu_s, ls.wait_s += first.wait_s, ls.apply_s += first.apply_s;
    ls.file_bytes += first.file_bytes, ls.h2d_bytes += first.h2d_bytes;
    ls.workers = first.workers, ls.blocks = first.blocks;
    ls.gpu_varint_cols = first.gpu_varint_cols, ls.gpu_varint_redone = first.gpu_varint_redone;
    return SYBL_OK;
}

static int open_table(Ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                      int32_t rank, int32_t nranks, int32_t flags, sybl_table **out) {
    std::string tdir = std::string(dir ? dir : ".") + "/" + table;
    TableInfo ti;
    int rc = read_table_info(tdir, ti);
    if (rc) return rc;
    sybl_table *t = nullptr;
    rc = sybl_table_create((sybl_ctx *)ctx, table, &t);
    if (rc == SYBL_OK && (flags & SYBL_OPEN_COMPACT)) t->compact_mode = true;  // every block is packed as it arrives
    if (rc) return rc;
    auto bail = [&](int code) {
        (void)load_sync_all(ctx);  // (blocks of a multi-stream load may still be decoding into the table)
        sybl_table_free(t);
        return code;
    };
    std::vector<std::string> want;
    if (columns && n_columns > 0) {
        for (int i = 0; i < n_columns; i++) want.push_back(columns[i] ? columns[i] : "");
    } else {
        for (auto &kv : ti.key_id) want.push_back(kv.first);
    }
    for (auto &name : want) {
        auto it = ti.key_id.find(name);
        if (it == ti.key_id.end()) return bail(fail(SYBL_E_INVAL, "column '%s' is not in the table's KeyTable", name.c_str()));
        auto ty = ti.key_type.find(it->second);
        int type = ty == ti.key_type.end() ? SYBL_NO_VAL : ty->second;
        if (type != SYBL_INT_VAL && type != SYBL_STR_VAL && type != SYBL_SET_VAL)
            return bail(fail(SYBL_E_IO, "column '%s' has unknown key type %d", name.c_str(), type));
        int64_t imin = 1, imax = 0;
        auto ii = ti.int_info.find(it->second);
        if (type == SYBL_INT_VAL && ii != ti.int_info.end()) {
            imin = ii->second.first;
            imax = ii->second.second;
        }
        if ((rc = sybl_table_add_column(t, name.c_str(), type, imin, imax))) return bail(rc);
    }
    std::vector<std::string> names;
    if ((rc = list_blocks(tdir, rank, nranks, names))) return bail(rc);
    t->src_dir = tdir;
    t->src_rank = rank;
    t->src_nranks = nranks < 1 ? 1 : nranks;
    if ((rc = load_blocks(ctx, t, tdir, names))) return bail(rc);
    *out = t;
    return SYBL_OK;
}

// sybl_table_refresh: the resident table follows its directory.  The reference re-lists and re-reads the block
// directories on every query (table_query.go:40-106) and keeps per-block results in a cache keyed by the block
// (query_cache.go:30-64); here the blocks stay decoded in HBM, so what has to be noticed is a block that appeared
// (digest wrote a new one), vanished (trim / expire) or was rewritten (digest into a partly filled block:
// <block>/info.db changed).
static int refresh_table(Table *t, int64_t *n_added, int64_t *n_dropped, int64_t *n_reloaded) {
    Ctx *ctx = t->ctx;
    if (t->src_dir.empty()) return fail(SYBL_E_STATE, "the table was not opened from a directory (sybl_table_open)");
    TableInfo ti;
    int rc = read_table_info(t->src_dir, ti);
    if (rc) return rc;
    // IntInfo may have moved with the new records (table_column_info.go:75-131)
    bool info_changed = false;
    for (auto &cp : t->cols) {
        auto it = ti.key_id.find(cp->name);
        if (it == ti.key_id.end() || cp->type != SYBL_INT_VAL) continue;
        auto ii = ti.int_info.find(it->second);
        if (ii == ti.int_info.end()) continue;
        if (!cp->info_given || cp->info_min != ii->second.first || cp->info_max != ii->second.second) info_changed = true;
        cp->info_given = true;
        cp->info_min = ii->second.first;
        cp->info_max = ii->second.second;
    }
    std::vector<std::string> names;
    if ((rc = list_blocks(t->src_dir, t->src_rank, t->src_nranks, names))) return rc;
    std::map<std::string, size_t> wanted;
    for (size_t i = 0; i < names.size(); i++) wanted[names[i]] = i;
    if ((rc = table_ensure_stats(t))) return rc;
    int64_t added = 0, dropped = 0, reloaded = 0;
    std::vector<LoadedBlock> keep;
    std::vector<std::string> to_load;
    std::map<std::string, bool> resident;
    for (auto &lb : t->loaded) {
        auto w = wanted.find(lb.name);
        bool same = false;
        if (w != wanted.end()) {
            auto sig = block_signature(t->src_dir + "/" + lb.name);
            same = sig.first == lb.mtime_ns && sig.second == lb.size;
        }
        if (same) {
            keep.push_back(lb);
            resident[lb.name] = true;
            continue;
        }
        // vanished, moved to another rank's share, or rewritten: its rows leave the scan.  Those of trailing blocks are
        // reused by the blocks loaded below (table_drop_dead_tail: the rewritten last block of an ingest loop); rows in
        // the middle of the table are given back once there are enough of them (table_reclaim_dead_rows)
        if (lb.index >= 0 && lb.index < (int64_t)t->blocks.size() && t->blocks[(size_t)lb.index].n > 0) {
            table_retire_block(t, lb.index);
        } else if (lb.index < 0) {
            t->broken_blocks--;
        }
        if (w != wanted.end()) reloaded++;
        else dropped++;
    }
    for (auto &nm : names)
        if (!resident.count(nm)) to_load.push_back(nm);
    added = (int64_t)to_load.size() - reloaded;
    t->loaded.swap(keep);
    if (info_changed || dropped + reloaded > 0) t->version++;  // (appended blocks bump it themselves)
    int64_t keep_blocks = 0;
    for (auto &lb : t->loaded) keep_blocks = std::max(keep_blocks, lb.index + 1);
    table_drop_dead_tail(t, keep_blocks);
    // (blocks that vanished from the middle -- trim / expire: once their rows are a quarter of the table the rest closes up)
    if ((rc = table_reclaim_dead_rows(t, env("SYBL_RECLAIM_ALWAYS") != nullptr))) return rc;
    if (!to_load.empty() && (rc = load_blocks(ctx, t, t->src_dir, to_load))) return rc;
    if (n_added) *n_added = added;
    if (n_dropped) *n_dropped = dropped;
    if (n_reloaded) *n_reloaded = reloaded;
    return SYBL_OK;
}

}  // namespace sybl

using namespace sybl;

extern "C" {

int sybl_table_open(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                    int32_t rank, int32_t nranks, sybl_table **out) {
    SYBL_API_GUARD(ctx);
    return sybl_table_open_flags(ctx, dir, table, columns, n_columns, rank, nranks, 0, out);
}

int sybl_table_open_flags(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns, int32_t n_columns,
                          int32_t rank, int32_t nranks, int32_t flags, sybl_table **out) {
    SYBL_API_GUARD(ctx);
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

int sybl_table_refresh(sybl_table *t, int64_t *n_added, int64_t *n_dropped, int64_t *n_reloaded) {
    SYBL_API_GUARD(t);
    if (!t) return fail(SYBL_E_INVAL, "sybl_table_refresh: NULL table");
    SYBL_HIP(hipSetDevice(t->ctx->device));
    try {
        return refresh_table(t, n_added, n_dropped, n_reloaded);
    } catch (const std::exception &e) {
        return fail(SYBL_E_IO, "sybl_table_refresh: %s", e.what());
    }
}

int64_t sybl_table_broken_blocks(const sybl_table *t) { SYBL_API_GUARD(t); return t ? t->broken_blocks : 0; }

int sybl_ctx_trim(sybl_ctx *ctx) {
    SYBL_API_GUARD(ctx);
    if (!ctx) return fail(SYBL_E_INVAL, "ctx is NULL");
    SYBL_HIP(hipSetDevice(ctx->device));
    SYBL_HIP(hipStreamSynchronize(ctx->stream));
    ctx_free_load_arena(ctx);  // (the next sybl_table_open / sybl_table_refresh allocates it again)
    return SYBL_OK;
}

int sybl_table_load_stats(const sybl_table *t, sybl_load_stats *out) {
    SYBL_API_GUARD(t);
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
    // (SYBL_DEBUG_GOB_NARROW: the reader's narrow slices, as the loader asks for them -- the JSON must not change)
    if (!path || !decode_file(path, v, err, env("SYBL_DEBUG_GOB_NARROW") != nullptr)) {
        set_error("%s", err.empty() ? "sybl_debug_gob_to_json: bad argument" : err.c_str());
        return nullptr;
    }
    out.clear();
    gob::to_json(v, out);
    return out.c_str();
}

const char *sybl_debug_block_layout(const char *block_dir, const char *const *columns, const int32_t *types, int32_t n_columns) {
    static thread_local std::string out;
    if (!block_dir || n_columns < 0 || (n_columns > 0 && (!columns || !types))) {
        set_error("sybl_debug_block_layout: bad argument");
        return nullptr;
    }
    std::vector<ColSpec> specs;
    for (int32_t i = 0; i < n_columns; i++) {
        if (!columns[i] || (types[i] != SYBL_INT_VAL && types[i] != SYBL_STR_VAL && types[i] != SYBL_SET_VAL)) {
            set_error("sybl_debug_block_layout: bad column %d", (int)i);
            return nullptr;
        }
        specs.push_back({columns[i], (int)types[i]});
    }
    // (a plain host buffer of the size load_blocks gives a slab; a block that needs more is reported, not loaded: the
    // pinned pair of its own would ask for a device)
    size_t cap = 65536;
    for (auto &sp : specs) cap += sp.type == SYBL_SET_VAL ? 0 : (size_t)65536 * (sp.type == SYBL_STR_VAL ? 12 : 8) + ((size_t)96 << 10);
    // (SYBL_LOADER_GPU_VARINT: the worker half of the GPU varint walk -- the lines gain raw=<bytes>:<digest of the file's
    // `Values` region as it would travel>; val= is then empty, the values exist on the device only)
    bool gpu_varint = false;
    if (const char *e = env("SYBL_LOADER_GPU_VARINT")) gpu_varint = atoi(e) != 0;
    size_t xcap = 0;
    if (gpu_varint)
        for (auto &sp : specs)
            if (sp.type == SYBL_INT_VAL) cap += (size_t)65536 + 4096, xcap += (size_t)65536 * 8 + (size_t)kGobMaxBins * 72 + 4096;
    std::vector<char> slab(cap);
    PreparedBlock pb;
    try {
        pb = prepare_block_unguarded(block_dir, specs, slab.data(), cap, -1, true, xcap);
    } catch (const std::exception &e) {
        out = std::string("exception: ") + e.what();
        return out.c_str();
    }
    char b[512];
    auto fnv = [](const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
        const unsigned char *c = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
        return h;
    };
    snprintf(b, sizeof(b), "rows=%lld unreadable=%d broken=%d bytes=%zu%s\n", (long long)pb.nrows, (int)pb.unreadable, (int)pb.broken, pb.bytes,
             pb.bytes > cap ? " (beyond a slab)" : "");
    out = b;
    if (pb.unreadable || pb.broken || pb.bytes > cap) return out.c_str();  // (beyond a slab: marked unreadable above, nothing was laid out)
    const char *H = slab.data();
    for (size_t ci = 0; ci < pb.cols.size(); ci++) {
        const PreparedCol &pc = pb.cols[ci];
        uint64_t hs = 1469598103934665603ull;
        for (auto &st : pc.strings) hs = fnv(st.data(), st.size() + 1, hs);
        uint64_t hset = fnv(pc.set_off.data(), pc.set_off.size() * 8);
        hset = fnv(pc.set_ids.data(), pc.set_ids.size() * 4, hset);
        hset = fnv(pc.set_pop.data(), pc.set_pop.size(), hset);
        const bool bins = (pc.kind == PreparedCol::kIntBins || pc.kind == PreparedCol::kStrBins) && pc.raw_len == 0;  // (walked on the GPU: device-only)
        snprintf(b, sizeof(b),
                 "%s kind=%d delta=%d venc=%d rec_w=%d val_w=%d local_w=%d recs=%lld bins=%lld vals=%lld local=%lld bits=%lld stats=%d min=%lld max=%lld pop=%lld "
                 "strings=%zu:%016llx binval=%016llx binoff=%016llx rec=%016llx val=%016llx loc=%016llx valid=%016llx set=%016llx\n",
                 specs[ci].name.c_str(), (int)pc.kind, (int)pc.delta, (int)pc.venc, pc.rec_w, pc.val_w, pc.local_w, (long long)pc.n_recs, (long long)pc.n_bins,
                 (long long)pc.n_vals, (long long)pc.n_local, (long long)pc.bits_words, (int)pc.have_stats, (long long)(pc.have_stats ? pc.vmin : 0),
                 (long long)(pc.have_stats ? pc.vmax : 0), (long long)pc.vpop, pc.strings.size(), (unsigned long long)hs,
                 (unsigned long long)(bins ? fnv(H + pc.binval_at, (size_t)pc.n_bins * 8) : 0), (unsigned long long)(bins ? fnv(H + pc.binoff_at, (size_t)(pc.n_bins + 1) * 8) : 0),
                 (unsigned long long)(bins ? fnv(H + pc.rec_at, (size_t)pc.n_recs * (size_t)pc.rec_w) : 0),
                 (unsigned long long)(pc.kind == PreparedCol::kIntValues && pc.raw_len == 0 ? fnv(H + pc.val_at, (size_t)pc.n_vals * (size_t)pc.val_w) : 0),
                 (unsigned long long)(pc.kind == PreparedCol::kStrValues ? fnv(H + pc.local_at, (size_t)pc.n_local * (size_t)pc.local_w) : 0),
                 (unsigned long long)(pc.bits_words > 0 ? fnv(H + pc.bits_at, (size_t)pc.bits_words * 4) : 0), (unsigned long long)hset);
        out += b;
        if (pc.raw_len > 0) {
            out.pop_back();
            snprintf(b, sizeof(b), " raw=%lld:%016llx\n", (long long)pc.raw_len, (unsigned long long)fnv(H + pc.raw_at, (size_t)pc.raw_len));
            out += b;
        }
    }
    return out.c_str();
}

}  // extern "C"
