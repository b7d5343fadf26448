// gobgpu.hip -- the varint walk of a column file on the GPU (round 6: int columns stored as `Values`).
//
// Reference: unpackIntCol (column_store_io.go:690-780) reads SavedIntColumn.Values through encoding/gob -- one varint after the
// other, each one's length known only from its first byte -- and then undoes the delta encoding.  The walk was the loader's
// host-CPU bound (~0.17 ms of a worker's 0.9 ms per block for each value-encoded column, the largest files
// of a block).  Here the file's bytes cross PCIe as they are and the GPU finds the value boundaries:
//
//   a gob unsigned integer is one byte below 128, or a byte 256 - n (n = 1..8) followed by n big-endian bytes; a signed one is
//   that with the sign in bit 0 (decode.go: decodeUint / decodeInt).  So L(b) = 1 for b < 128 and 1 + 256 - b otherwise tells how
//   far the next value is -- IF b is the first byte of a value.  Every thread takes one 64-byte chunk (sixteen registers) and
//   goes through it BACKWARDS: "a walk that stands on byte p leaves the chunk at offset x" follows from the same statement about
//   byte p + L(p), so nine nibbles of a 64-bit register (p + 1 .. p + 9, shifted along) carry it, and at the chunk's first byte
//   they hold the answer for each of the nine places a value can reach into the chunk from the one before (entry offset 0..8):
//   a map {0..8} -> {0..8} in 36 bits, no memory touched.  Maps compose.  An inclusive scan of the composition over the
//   workgroup's 256 chunks, and a look-back over the workgroups ahead (each publishes its total map in one 64-bit word as soon
//   as it has it; the file starts on a value), give every thread its true entry offset.  A second pass marks the bytes on which
//   values start (a 64-bit mask) and counts them; a scan + look-back of the counts gives every value its rank; the third pass
//   decodes the marked values (zig-zag) out of LDS and writes each one to its rank, as int64, for k_decode_delta to undo the
//   deltas (kernels.hip) -- the same kernel the host-parsed path feeds, which also holds the resulting column values against
//   the bounds the worker took from the block's info.db (IntInfoMap: what block_col_direct needs BEFORE the decode).
//
// A file of 65 536 four-byte values is 16 workgroups of 256 threads: the chain of a block on its stream grows by one short
// kernel.  (The first form of this kernel -- one workgroup of 1024 threads per file, each thread a run of chunks, counts carried
// in the maps -- took 365 us per block and made the open three times slower than the host parser: profiles/r06_gpu_varint.txt.)
//
// The status words say what the walk met: values found (which must cover the count the file's slice header announced), bytes
// that cannot start a value, a value cut off by the end of the region, a look-back that gave up.  loader.cpp looks at them when
// the load ends and takes a block that fails through the host parser again.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine.h"

namespace sybl {

namespace {

constexpr int kChunk = 64;                            // bytes per thread
constexpr unsigned long long kIdentity = 0x876543210ull;  // nibble j = j
constexpr unsigned long long kReady = 1ull << 63;
constexpr uint32_t kSpinLimit = 1u << 22;             // polls of one look-back word before the walk gives up (~seconds)

// how far the next value is from a first byte b, minus one: 0 for b < 128, n = 256 - b for 0xF8..0xFF
// (0x80..0xF7 never starts a value -- more than eight bytes: 0, and the last pass reports it)
__device__ __forceinline__ uint32_t gob_extra(uint32_t b) { return b >= 0xF8u ? 256u - b : 0u; }

// the map "a, then b"
__device__ __forceinline__ unsigned long long gob_compose(unsigned long long a, unsigned long long b) {
    unsigned long long o = 0;
#pragma unroll
    for (int e = 0; e < 9; e++) {
        const uint32_t mid = (uint32_t)(a >> (4 * e)) & 15u;
        o |= ((b >> (4u * mid)) & 15ull) << (4 * e);
    }
    return o;
}

__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int o) {
    const uint32_t lo = __shfl_up((uint32_t)v, o, 64), hi = __shfl_up((uint32_t)(v >> 32), o, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl_up64_xor(unsigned long long v, int o) {
    const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// waits for a look-back word to be published; 0 (not ready) when it gives up
__device__ __forceinline__ unsigned long long gob_wait(const unsigned long long *word) {
    for (uint32_t spin = 0; spin < kSpinLimit; spin++) {
        const unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v & kReady) return v;
        __builtin_amdgcn_s_sleep(2);
    }
    return 0;
}

}  // namespace

// blockIdx.y = job, blockIdx.x = workgroup of the job (256 chunks = 16 KB of the region each).
__global__ __launch_bounds__(kGobWgThreads) void k_gob_values(const GobValuesBatch B) {
    __shared__ uint32_t stage[19 * kGobWgThreads];   // pass 3: the chunks, [dword][thread]
    __shared__ unsigned long long wave_map[4];
    __shared__ unsigned long long wave_cnt[4];
    __shared__ unsigned long long ahead_map, ahead_cnt;
    __shared__ uint32_t gave_up;
    const GobValuesJob &J = B.job[blockIdx.y];
    const uint32_t g = blockIdx.x;
    if (g >= (uint32_t)J.n_wgs) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_bytes = J.n_bytes, want = J.n;
    const uint32_t n_chunks = (n_bytes + kChunk - 1) / kChunk;
    const uint32_t c = g * kGobWgThreads + tid;
    const bool has = c < n_chunks;
    unsigned long long *state = J.state;
    if (tid == 0) gave_up = 0;

    uint32_t w[19];
#pragma unroll
    for (int k = 0; k < 19; k++) w[k] = 0;
    if (has) {
        const uint4 *src = (const uint4 *)J.bytes + (size_t)c * 4;  // (16-byte aligned, readable up to the next multiple of 64 + 16: loader.cpp)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = src[q];
            w[4 * q] = x.x, w[4 * q + 1] = x.y, w[4 * q + 2] = x.z, w[4 * q + 3] = x.w;
        }
        const uint4 x = src[4];  // (a value may reach eight bytes into the next chunk)
        w[16] = x.x, w[17] = x.y, w[18] = x.z;
    }
    const uint32_t valid = has ? n_bytes - c * kChunk : 0u;  // (>= 64 but in the file's last chunk)

    // ---- pass 1, backwards: nibble j of R = where a walk standing on byte (p + 1 + j) leaves the chunk
    unsigned long long R = kIdentity;  // (beyond the chunk: it has left at offset j; a thread without a chunk keeps the identity)
    if (has) {
#pragma unroll
        for (int p = kChunk - 1; p >= 0; p--) {
            const uint32_t b = (w[p >> 2] >> (8 * (p & 3))) & 0xFFu;
            const uint32_t e = (uint32_t)(R >> (4u * gob_extra(b))) & 15u;
            R = (R << 4) | e;
        }
        R &= 0xFFFFFFFFFull;
    }
    // inclusive scan of the composition: lanes of a wave, then the waves ahead
    unsigned long long M = R;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long prev = shfl_up64(M, o);
        if ((int)lane >= o) M = gob_compose(prev, M);
    }
    if (lane == 63) wave_map[wave] = M;
    __syncthreads();
    unsigned long long before_wave = kIdentity;
    for (uint32_t k = 0; k < wave; k++) before_wave = gob_compose(before_wave, wave_map[k]);
    M = gob_compose(before_wave, M);  // chunks [this workgroup's first .. this thread's]
    if (tid == kGobWgThreads - 1) __hip_atomic_store(&state[kGobStateMaps + g], M | kReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look-back: the workgroups ahead, composed in order by the first wave
    if (wave == 0) {
        unsigned long long a = kIdentity;
        bool ok = true;
        if (lane < g) {
            a = gob_wait(&state[kGobStateMaps + lane]);
            ok = a != 0;
            a = ok ? a & 0xFFFFFFFFFull : kIdentity;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long prev = shfl_up64(a, o);
            if ((int)lane >= o) a = gob_compose(prev, a);
        }
        if (!ok) atomicOr(&gave_up, 1u);
        if (lane == 63) ahead_map = a;
    }
    __syncthreads();
    // (the region starts on a value: entry 0) where this thread's chunk is entered
    uint32_t cur;
    {
        unsigned long long ahead = shfl_up64(M, 1);  // this workgroup's chunks up to the previous thread's ...
        if (lane == 0) ahead = before_wave;          // ... which for a wave's first lane are the waves ahead
        const uint32_t entry_wg = (uint32_t)ahead_map & 15u;
        cur = (uint32_t)(ahead >> (4u * entry_wg)) & 15u;
    }

    // ---- pass 2: the bytes values start on, counted
    // (`Bins`: the values that are 0 -- single bytes 00 -- are counted beside them: their ranks go into a list of their own)
    const bool want_zeros = J.zpos != nullptr;
    unsigned long long starts = 0, zeros = 0;
    if (has) {
#pragma unroll
        for (int p = 0; p < kChunk; p++) {
            const uint32_t b = (w[p >> 2] >> (8 * (p & 3))) & 0xFFu;
            const bool hit = (uint32_t)p == cur;
            cur = hit ? cur + 1u + gob_extra(b) : cur;
            const bool start = hit && (uint32_t)p < valid;
            starts |= start ? 1ull << p : 0ull;
            zeros |= start && b == 0u ? 1ull << p : 0ull;
        }
    }
    // one scan for both counts: values in the low half, zeros in the high half (neither reaches 2^24 in a region of 1 MB)
    const unsigned long long mine = (unsigned long long)__popcll(starts) | (want_zeros ? (unsigned long long)__popcll(zeros) << 32 : 0ull);
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long prev = shfl_up64(incl, o);
        if ((int)lane >= o) incl += prev;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    unsigned long long kz = incl - mine;
    for (uint32_t q = 0; q < wave; q++) kz += wave_cnt[q];
    if (tid == kGobWgThreads - 1)
        __hip_atomic_store(&state[kGobStateCounts + g], (kz + mine) | kReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
        unsigned long long a = 0;
        bool ok = true;
        if (lane < g) {
            const unsigned long long v = gob_wait(&state[kGobStateCounts + lane]);
            ok = v != 0;
            a = v & ~kReady;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += shfl_up64_xor(a, o);
        if (!ok) atomicOr(&gave_up, 1u);
        if (lane == 0) ahead_cnt = a;
    }
    // ---- pass 3: the marked values, out of LDS, to their ranks
#pragma unroll
    for (int q = 0; q < 19; q++) stage[q * kGobWgThreads + tid] = w[q];
    __syncthreads();
    kz += ahead_cnt;
    uint32_t k = (uint32_t)kz, zk = (uint32_t)(kz >> 32);
    uint32_t bad = gave_up ? kGobGaveUp : 0u;
    long long *out = J.out;
    if (tid == kGobWgThreads - 1 && g + 1 == (uint32_t)J.n_wgs) {
        const unsigned long long all = kz + mine;
        state[kGobStateFound] = all & 0xFFFFFFFFull;
        state[kGobStateZeros] = all >> 32;
        if ((uint32_t)all < want && !want_zeros) bad |= kGobShort;  // (`Values`: the slice announced `want` of them)
    }
    const uint32_t n_zpos = J.n_zpos;
    uint32_t *zpos = J.zpos;
    while (starts != 0 && !gave_up) {
        const uint32_t p = (uint32_t)__builtin_ctzll(starts);
        starts &= starts - 1;
        const uint32_t i = p >> 2, sh = 8u * (p & 3u);
        const uint32_t d0 = stage[i * kGobWgThreads + tid], d1 = stage[(i + 1) * kGobWgThreads + tid], d2 = stage[(i + 2) * kGobWgThreads + tid];
        // nine bytes from p on: the first one, then eight with the first of them in the low byte
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), mid = __builtin_amdgcn_alignbit(d2, d1, sh), top = d2 >> sh;
        const uint32_t b = lo & 0xFFu;
        const uint32_t x_lo = __builtin_amdgcn_alignbyte(mid, lo, 1), x_hi = __builtin_amdgcn_alignbyte(top, mid, 1);
        const uint32_t nb = gob_extra(b);
        // big-endian: byte-swap the pair, keep the top n bytes
        const unsigned long long be = ((unsigned long long)__builtin_bswap32(x_lo) << 32) | (unsigned long long)__builtin_bswap32(x_hi);
        const unsigned long long u = nb ? be >> (8u * (8u - nb)) : (unsigned long long)b;
        if (k < want) {
            if (b >= 128u && b < 0xF8u) bad |= kGobBadByte;
            if (p + nb >= valid) bad |= kGobTruncated;  // (valid >= 64 + 8 unless the file ends here)
            out[k] = want_zeros ? (long long)u : (long long)((u >> 1) ^ (0ull - (u & 1ull)));
            if (want_zeros && b == 0u) {
                if (zk < n_zpos) zpos[zk] = k;
                zk++;
            }
        } else if (!want_zeros && k < want + 3u && p < valid) {
            // what stands behind the slice (loader.cpp holds it against a struct's end): values like the others
            if (b >= 128u && b < 0xF8u) bad |= kGobBadByte;
            if (p + nb >= valid) bad |= kGobTruncated;
            state[kGobStateTail + (k - want)] = u + 1ull;
        }
        k++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
    if (lane == 0 && bad) atomicOr(&state[kGobStateFlags], (unsigned long long)bad);
}

hipError_t launch_gob_values(const GobValuesBatch &B, hipStream_t st) {
    int max_wgs = 0;
    for (int i = 0; i < B.n; i++) max_wgs = std::max(max_wgs, (int)B.job[i].n_wgs);
    if (B.n <= 0 || max_wgs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gob_values, dim3((unsigned)max_wgs, (unsigned)B.n), dim3(kGobWgThreads), 0, st, B);
    return hipGetLastError();
}

}  // namespace sybl
