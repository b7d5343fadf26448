// rccl_standin.cpp -- TEST INFRASTRUCTURE ONLY.  Never linked by the product (sybil_amd/csrc/Makefile links the real
// -lrccl); tests/test_gpu_multirank.py puts this library in LD_PRELOAD of its worker processes so that the N > 1 branches
// of csrc/rccl.cpp (reduce-scatter slices, k_pack32 / k_unpack32, the outlier all-gather, the hash key union, the
// collective finalize) execute on a box with ONE GPU: R processes share the device, and the nine RCCL entry points the
// engine imports (`nm -D libsybilgpu.so | grep ' U nccl'`) are implemented between them over POSIX shared memory with
// host-staged copies.
//
// Semantics kept from RCCL: a collective is ENQUEUED on the caller's stream and returns at once -- each step is
// [hipMemcpyAsync D2H into pinned memory] -> [hipLaunchHostFunc: publish to the shared segment, barrier, reduce / gather
// the ranks' pieces, barrier] -> [hipMemcpyAsync H2D] on that stream -- so code that reads a result without waiting for
// the stream fails here as it would on a node.  In-place operands work (all-reduce with send == recv, all-gather with
// send == recv + rank * count).  ncclGroupStart / End only count: the ranks issue the same collectives in the same
// order (SPMD), which is checked -- every step publishes (sequence number, kind, count, type, op) and a rank that finds a
// peer on a different step aborts with a message instead of hanging.  A barrier that is not met within
// SYBL_STANDIN_TIMEOUT_S (default 120) seconds aborts the process too.  SYBL_STANDIN_SYNC=1 runs every step on the calling
// thread behind a hipStreamSynchronize (no host functions): a fallback for debugging.
//
// Not modelled: topology, bandwidth, several communicators per process sharing a stream concurrently, floating-point types.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {

constexpr int kMaxRanks = 16;
constexpr uint32_t kMagic = 0x5359424cu;  // "SYBL"

enum Kind : int32_t { ALLREDUCE = 1, REDUCESCATTER = 2, ALLGATHER = 3 };

struct Desc {
    uint64_t seq;
    int32_t kind, dtype, op, pad;
    uint64_t count, offset, m;
};

struct Shm {
    std::atomic<uint32_t> ready, attached, bar_count, bar_gen;
    uint32_t nranks, pad;
    uint64_t slot_bytes;
    Desc desc[kMaxRanks];
};
constexpr size_t kHeaderBytes = 4096;
static_assert(sizeof(Shm) <= kHeaderBytes, "header fits a page");

struct Comm {
    Shm *shm = nullptr;
    size_t map_bytes = 0;
    int rank = 0, nranks = 1;
    size_t slot_bytes = 0;
    char *pin_in = nullptr, *pin_out = nullptr;
    uint64_t seq = 0;
    double timeout_s = 120;
    bool sync = false;
    char *slot(int r) const { return (char *)shm + kHeaderBytes + (size_t)r * slot_bytes; }
};

struct Step {
    Comm *c;
    Desc d;
    size_t es;
};

[[noreturn]] void die(const Comm *c, const char *fmt, ...) {
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[rccl_standin rank %d/%d] FATAL: %s\n", c ? c->rank : -1, c ? c->nranks : 0, msg);
    fflush(stderr);
    _exit(86);
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void barrier(Comm *c, const char *where) {
    Shm *s = c->shm;
    const uint32_t gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == (uint32_t)c->nranks) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return;
    }
    const double t0 = now_s();
    for (uint64_t spin = 0; s->bar_gen.load() == gen; spin++) {
        if (spin > 2000) usleep(50);
        if ((spin & 1023) == 1023 && now_s() - t0 > c->timeout_s) die(c, "barrier timed out after %.0f s in %s (step %llu)", c->timeout_s, where, (unsigned long long)c->seq);
    }
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: return 4;
        case ncclInt64: case ncclUint64: return 8;
        default: return 0;
    }
}

template <typename T>
void reduce_t(T *out, const Comm *c, size_t slot_off_elems, size_t m, int op) {
    const T *first = (const T *)c->slot(0) + slot_off_elems;
    for (size_t i = 0; i < m; i++) out[i] = first[i];
    for (int r = 1; r < c->nranks; r++) {
        const T *src = (const T *)c->slot(r) + slot_off_elems;
        if (op == ncclSum)
            for (size_t i = 0; i < m; i++) out[i] = (T)(out[i] + src[i]);
        else if (op == ncclMax)
            for (size_t i = 0; i < m; i++) out[i] = src[i] > out[i] ? src[i] : out[i];
        else
            for (size_t i = 0; i < m; i++) out[i] = src[i] < out[i] ? src[i] : out[i];
    }
}

void reduce_any(void *out, const Comm *c, size_t off, size_t m, int dtype, int op) {
    switch (dtype) {
        case ncclInt8: reduce_t((int8_t *)out, c, off, m, op); break;
        case ncclUint8: reduce_t((uint8_t *)out, c, off, m, op); break;
        case ncclInt32: reduce_t((int32_t *)out, c, off, m, op); break;
        case ncclUint32: reduce_t((uint32_t *)out, c, off, m, op); break;
        case ncclInt64: reduce_t((int64_t *)out, c, off, m, op); break;
        case ncclUint64: reduce_t((uint64_t *)out, c, off, m, op); break;
    }
}

// The host half of one step: pin_in holds this rank's piece, pin_out receives what the H2D copy behind it delivers.
void step_host(void *arg) {
    Step *st = (Step *)arg;
    Comm *c = st->c;
    const Desc &d = st->d;
    const size_t es = st->es, R = (size_t)c->nranks;
    const size_t in_elems = d.kind == ALLREDUCE ? d.m : d.kind == REDUCESCATTER ? d.m * R : d.m;
    memcpy(c->slot(c->rank), c->pin_in, in_elems * es);
    c->shm->desc[c->rank] = d;
    barrier(c, "publish");
    for (int r = 0; r < c->nranks; r++) {
        const Desc &o = c->shm->desc[r];
        if (o.seq != d.seq || o.kind != d.kind || o.dtype != d.dtype || o.op != d.op || o.count != d.count || o.offset != d.offset)
            die(c, "collective mismatch: this rank is at step %llu (kind %d, type %d, op %d, count %llu, offset %llu), rank %d at step %llu (kind %d, type %d, op %d, count %llu, offset %llu)",
                (unsigned long long)d.seq, d.kind, d.dtype, d.op, (unsigned long long)d.count, (unsigned long long)d.offset, r,
                (unsigned long long)o.seq, o.kind, o.dtype, o.op, (unsigned long long)o.count, (unsigned long long)o.offset);
    }
    if (d.kind == ALLREDUCE) {
        reduce_any(c->pin_out, c, 0, d.m, d.dtype, d.op);
    } else if (d.kind == REDUCESCATTER) {
        reduce_any(c->pin_out, c, (size_t)c->rank * d.m, d.m, d.dtype, d.op);
    } else {
        for (int r = 0; r < c->nranks; r++) memcpy(c->pin_out + (size_t)r * d.m * es, c->slot(r), d.m * es);
    }
    barrier(c, "consume");
    delete st;
}

ncclResult_t hip_err(const Comm *c, hipError_t e, const char *what) {
    fprintf(stderr, "[rccl_standin rank %d] %s: %s\n", c->rank, what, hipGetErrorString(e));
    return ncclUnhandledCudaError;
}
#define SI_HIP(expr)                                         \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return hip_err(c, e__, #expr); \
    } while (0)

ncclResult_t run_step(Comm *c, const Desc &d, size_t es, hipStream_t st) {
    Step *s = new Step{c, d, es};
    if (c->sync) {
        SI_HIP(hipStreamSynchronize(st));
        step_host(s);
    } else {
        SI_HIP(hipLaunchHostFunc(st, step_host, s));
    }
    return ncclSuccess;
}

ncclResult_t collective(Comm *c, Kind kind, const void *send, void *recv, size_t count, ncclDataType_t dtype, int op, hipStream_t st) {
    if (!c || !c->shm) return ncclInvalidArgument;
    const size_t es = type_size(dtype), R = (size_t)c->nranks;
    if (!es) return ncclInvalidArgument;
    if (count == 0) return ncclSuccess;
    const size_t per = kind == ALLREDUCE ? c->slot_bytes / es : c->slot_bytes / (es * R);
    const char *S = (const char *)send;
    char *D = (char *)recv;
    for (size_t o = 0; o < count; o += per) {
        const size_t m = count - o < per ? count - o : per;
        Desc d{c->seq++, kind, (int32_t)dtype, op, 0, count, o, m};
        if (kind == ALLREDUCE) {
            SI_HIP(hipMemcpyAsync(c->pin_in, S + o * es, m * es, hipMemcpyDeviceToHost, st));
        } else if (kind == REDUCESCATTER) {
            for (size_t r = 0; r < R; r++) SI_HIP(hipMemcpyAsync(c->pin_in + r * m * es, S + (r * count + o) * es, m * es, hipMemcpyDeviceToHost, st));
        } else {
            SI_HIP(hipMemcpyAsync(c->pin_in, S + o * es, m * es, hipMemcpyDeviceToHost, st));
        }
        ncclResult_t rc = run_step(c, d, es, st);
        if (rc != ncclSuccess) return rc;
        if (kind == ALLGATHER) {
            for (size_t r = 0; r < R; r++) SI_HIP(hipMemcpyAsync(D + (r * count + o) * es, c->pin_out + r * m * es, m * es, hipMemcpyHostToDevice, st));
        } else {
            SI_HIP(hipMemcpyAsync(D + o * es, c->pin_out, m * es, hipMemcpyHostToDevice, st));
        }
        if (c->sync) SI_HIP(hipStreamSynchronize(st));
    }
    return ncclSuccess;
}

void shm_name(const ncclUniqueId &id, char *out, size_t n) {
    static const char hex[] = "0123456789abcdef";
    size_t at = (size_t)snprintf(out, n, "/sybl_rccl_standin_");
    for (int i = 0; i < 16 && at + 2 < n; i++) {
        out[at++] = hex[((unsigned char)id.internal[i]) >> 4];
        out[at++] = hex[((unsigned char)id.internal[i]) & 15];
    }
    out[at] = 0;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) {
        uint64_t a = (uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)(now_s() * 1e9), b = a * 0xD6E8FEB86659FD93ull;
        memcpy(id->internal, &a, 8);
        memcpy(id->internal + 8, &b, 8);
    }
    if (fd >= 0) close(fd);
    memcpy(id->internal + 16, "sybl-rccl-standin", 17);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->rank = rank;
    c->nranks = nranks;
    const char *e = getenv("SYBL_STANDIN_SLOT_MB");
    c->slot_bytes = (size_t)(e && atoi(e) > 0 ? atoi(e) : 8) << 20;
    e = getenv("SYBL_STANDIN_TIMEOUT_S");
    if (e && atof(e) > 0) c->timeout_s = atof(e);
    e = getenv("SYBL_STANDIN_SYNC");
    c->sync = e && *e && *e != '0';
    c->map_bytes = kHeaderBytes + (size_t)nranks * c->slot_bytes;
    char name[96];
    shm_name(id, name, sizeof name);
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) die(c, "shm_open(%s): %s", name, strerror(errno));
        if (ftruncate(fd, (off_t)c->map_bytes) != 0) die(c, "ftruncate: %s", strerror(errno));
    } else {
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= c->map_bytes) break;
            if (fd >= 0) close(fd);
            if (now_s() - t0 > c->timeout_s) die(c, "rank 0 never created %s", name);
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) die(c, "mmap: %s", strerror(errno));
    c->shm = (Shm *)p;
    if (rank == 0) {
        c->shm->nranks = (uint32_t)nranks;
        c->shm->slot_bytes = c->slot_bytes;
        c->shm->bar_count.store(0);
        c->shm->bar_gen.store(0);
        c->shm->attached.store(0);
        c->shm->ready.store(kMagic);
    } else {
        while (c->shm->ready.load() != kMagic) {
            if (now_s() - t0 > c->timeout_s) die(c, "segment never became ready");
            usleep(200);
        }
        if (c->shm->nranks != (uint32_t)nranks || c->shm->slot_bytes != c->slot_bytes) die(c, "ranks disagree on the communicator's shape");
    }
    c->shm->attached.fetch_add(1);
    while (c->shm->attached.load() < (uint32_t)nranks) {
        if (now_s() - t0 > c->timeout_s) die(c, "only %u of %d ranks attached", c->shm->attached.load(), nranks);
        usleep(200);
    }
    if (rank == 0) shm_unlink(name);  // every rank holds its mapping: nothing is left behind whatever happens next
    if (hipHostMalloc((void **)&c->pin_in, c->slot_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->pin_out, c->slot_bytes, hipHostMallocDefault) != hipSuccess)
        die(c, "hipHostMalloc of the staging buffers failed");
    if (getenv("SYBL_STANDIN_VERBOSE")) fprintf(stderr, "[rccl_standin rank %d/%d] attached to %s (%s steps)\n", rank, nranks, name, c->sync ? "synchronous" : "stream-ordered");
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    if (getenv("SYBL_STANDIN_VERBOSE")) fprintf(stderr, "[rccl_standin rank %d/%d] %llu steps\n", c->rank, c->nranks, (unsigned long long)c->seq);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    if (c->shm) munmap(c->shm, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (rccl_standin)";
        case ncclUnhandledCudaError: return "unhandled HIP error (rccl_standin)";
        case ncclInvalidArgument: return "invalid argument (rccl_standin)";
        default: return "error (rccl_standin)";
    }
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (op != ncclSum && op != ncclMax && op != ncclMin) return ncclInvalidArgument;
    return collective((Comm *)comm, ALLREDUCE, send, recv, count, dtype, (int)op, st);
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (op != ncclSum && op != ncclMax && op != ncclMin) return ncclInvalidArgument;
    return collective((Comm *)comm, REDUCESCATTER, send, recv, recvcount, dtype, (int)op, st);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dtype, ncclComm_t comm, hipStream_t st) {
    return collective((Comm *)comm, ALLGATHER, send, recv, sendcount, dtype, -1, st);
}

// A marker the tests look for (dlsym) to be sure the stand-in, not librccl.so, answered.
int sybl_rccl_standin_marker(void) { return 1; }

}  // extern "C"
