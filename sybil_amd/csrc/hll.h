// hll.h -- the count-distinct sketch of a Result (reference: Result.Distinct, query_spec.go:87,100,180-188; fed by
// FilterAndAggRecords, aggregate.go:205-243; read by the printers, printer.go:79-80,142-144,204-205).
//
// The reference's sketch is github.com/logv/loglogbeta -- not in the reference tree and without a pinned version -- so
// this follows the published algorithms the library implements, exactly as oracle/sybil_oracle.c restates them
// (PARITY UNPINNED, see sybil_oracle.h): LogLog-Beta (Qin, Kim, Tung 2016) at precision 14 over MetroHash64 (seed 1337).
// Host and device share the functions below; tests/test_hll_host.py runs them on the CPU against the oracle.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SYBL_HD __host__ __device__ __forceinline__
#else
#define SYBL_HD inline
#endif

namespace sybl {

constexpr int kHllBits = 14;
constexpr int kHllRegs = 1 << kHllBits;   // one byte each
constexpr uint64_t kHllSeed = 1337;       // loglogbeta.go: metro.Hash64(value, 1337)
constexpr int kMaxDistinct = 8;           // columns of a distinct list (== SYBL_MAX_GROUPS)

SYBL_HD uint64_t hll_rotr(uint64_t v, unsigned k) { return (v >> k) | (v << (64 - k)); }

// MetroHash64 of N little-endian 64-bit words (the int fast path hashes 8 bytes per column, aggregate.go:208-222): the
// general algorithm with its 32-, 16- and 8-byte steps; a whole number of words leaves no shorter tail.  N is a
// template parameter so that every index into w is a constant (an array indexed at run time lives in scratch memory
// on the GPU).
template <int N>
SYBL_HD uint64_t metro64_n(const uint64_t (&w)[kMaxDistinct], uint64_t seed) {
    static_assert(N >= 1 && N <= kMaxDistinct, "1..8 words");
    const uint64_t k0 = 0xD6D018F5ull, k1 = 0xA2AA033Bull, k2 = 0x62992FC1ull, k3 = 0x30BC5B29ull;
    uint64_t hash = (seed + k2) * k0;
    constexpr int B = N / 4;  // 32-byte blocks
    if (B > 0) {
        uint64_t v0 = hash, v1 = hash, v2 = hash, v3 = hash;
#pragma unroll
        for (int blk = 0; blk < B; blk++) {
            v0 += w[4 * blk] * k0;     v0 = hll_rotr(v0, 29) + v2;
            v1 += w[4 * blk + 1] * k1; v1 = hll_rotr(v1, 29) + v3;
            v2 += w[4 * blk + 2] * k2; v2 = hll_rotr(v2, 29) + v0;
            v3 += w[4 * blk + 3] * k3; v3 = hll_rotr(v3, 29) + v1;
        }
        v2 ^= hll_rotr(((v0 + v3) * k0) + v1, 37) * k1;
        v3 ^= hll_rotr(((v1 + v2) * k1) + v0, 37) * k0;
        v0 ^= hll_rotr(((v0 + v2) * k0) + v3, 37) * k1;
        v1 ^= hll_rotr(((v1 + v3) * k1) + v2, 37) * k0;
        hash += v0 ^ v1;
    }
    constexpr int T = 4 * B;  // first word of the tail
    if (N - T >= 2) {
        uint64_t v0 = hash + w[T < kMaxDistinct ? T : 0] * k2;             v0 = hll_rotr(v0, 29) * k3;
        uint64_t v1 = hash + w[T + 1 < kMaxDistinct ? T + 1 : 0] * k2;     v1 = hll_rotr(v1, 29) * k3;
        v0 ^= hll_rotr(v0 * k0, 21) + v1;
        v1 ^= hll_rotr(v1 * k3, 21) + v0;
        hash += v1;
    }
    constexpr int L = (N - T >= 2) ? T + 2 : T;  // the last, single word, if any
    if (N - L >= 1) {
        hash += w[L < kMaxDistinct ? L : 0] * k3;
        hash ^= hll_rotr(hash, 55) * k1;
    }
    hash ^= hll_rotr(hash, 28);
    hash *= k0;
    hash ^= hll_rotr(hash, 29);
    return hash;
}

SYBL_HD uint64_t metro64_words(const uint64_t (&w)[kMaxDistinct], int n, uint64_t seed) {
    switch (n) {
    case 1: return metro64_n<1>(w, seed);
    case 2: return metro64_n<2>(w, seed);
    case 3: return metro64_n<3>(w, seed);
    case 4: return metro64_n<4>(w, seed);
    case 5: return metro64_n<5>(w, seed);
    case 6: return metro64_n<6>(w, seed);
    case 7: return metro64_n<7>(w, seed);
    default: return metro64_n<8>(w, seed);
    }
}

// LogLogBeta.AddHash: the register a hash belongs to and the rank it carries there
SYBL_HD void hll_place(uint64_t x, uint32_t &reg, uint32_t &rank) {
    reg = (uint32_t)(x >> (64 - kHllBits));
    const uint64_t rest = (x << kHllBits) ^ (~(uint64_t)0 >> (64 - kHllBits));  // 14 guard bits: never zero
#ifdef __HIP_DEVICE_COMPILE__
    rank = (uint32_t)__clzll((long long)rest) + 1u;
#else
    rank = (uint32_t)__builtin_clzll(rest) + 1u;
#endif
}

// MetroHash64 of arbitrary bytes (the str slow path hashes strings, aggregate.go:225-239) -- host only: a str
// column's hashes are computed once per dictionary id at prepare time
inline uint64_t metro64_bytes(const uint8_t *p, size_t len, uint64_t seed) {
    const uint64_t k0 = 0xD6D018F5ull, k1 = 0xA2AA033Bull, k2 = 0x62992FC1ull, k3 = 0x30BC5B29ull;
    auto rd = [](const uint8_t *q, int n) {
        uint64_t v = 0;
        for (int i = 0; i < n; i++) v |= (uint64_t)q[i] << (8 * i);
        return v;
    };
    const uint8_t *end = p + len;
    uint64_t hash = (seed + k2) * k0;
    if (len >= 32) {
        uint64_t v0 = hash, v1 = hash, v2 = hash, v3 = hash;
        do {
            v0 += rd(p, 8) * k0; p += 8; v0 = hll_rotr(v0, 29) + v2;
            v1 += rd(p, 8) * k1; p += 8; v1 = hll_rotr(v1, 29) + v3;
            v2 += rd(p, 8) * k2; p += 8; v2 = hll_rotr(v2, 29) + v0;
            v3 += rd(p, 8) * k3; p += 8; v3 = hll_rotr(v3, 29) + v1;
        } while (p <= end - 32);
        v2 ^= hll_rotr(((v0 + v3) * k0) + v1, 37) * k1;
        v3 ^= hll_rotr(((v1 + v2) * k1) + v0, 37) * k0;
        v0 ^= hll_rotr(((v0 + v2) * k0) + v3, 37) * k1;
        v1 ^= hll_rotr(((v1 + v3) * k1) + v2, 37) * k0;
        hash += v0 ^ v1;
    }
    if (end - p >= 16) {
        uint64_t v0 = hash + rd(p, 8) * k2; p += 8; v0 = hll_rotr(v0, 29) * k3;
        uint64_t v1 = hash + rd(p, 8) * k2; p += 8; v1 = hll_rotr(v1, 29) * k3;
        v0 ^= hll_rotr(v0 * k0, 21) + v1;
        v1 ^= hll_rotr(v1 * k3, 21) + v0;
        hash += v1;
    }
    if (end - p >= 8) { hash += rd(p, 8) * k3; p += 8; hash ^= hll_rotr(hash, 55) * k1; }
    if (end - p >= 4) { hash += rd(p, 4) * k3; p += 4; hash ^= hll_rotr(hash, 26) * k1; }
    if (end - p >= 2) { hash += rd(p, 2) * k3; p += 2; hash ^= hll_rotr(hash, 48) * k1; }
    if (end - p >= 1) { hash += rd(p, 1) * k3; hash ^= hll_rotr(hash, 37) * k1; }
    hash ^= hll_rotr(hash, 28);
    hash *= k0;
    hash ^= hll_rotr(hash, 29);
    return hash;
}

// MetroHash64 of a byte string that arrives piece by piece (the slow path over several columns, aggregate.go:224-239: the
// decimal digits of an int, a dictionary string, a "\t" after each) -- host and device.  The algorithm consumes 32-byte
// blocks while at least 32 bytes remain, so a block can be folded in the moment the 32 pending bytes are complete; the
// pending words are four named registers (an array indexed at run time would live in scratch memory on the GPU).
struct Metro64Stream {
    static constexpr uint64_t k0 = 0xD6D018F5ull, k1 = 0xA2AA033Bull, k2 = 0x62992FC1ull, k3 = 0x30BC5B29ull;
    uint64_t hash, v0, v1, v2, v3, b0, b1, b2, b3;
    uint32_t fill;
    bool blocks;
    SYBL_HD void init(uint64_t seed) {
        hash = (seed + k2) * k0;
        v0 = v1 = v2 = v3 = hash;
        b0 = b1 = b2 = b3 = 0;
        fill = 0;
        blocks = false;
    }
    SYBL_HD void put(uint8_t c) {
        const uint64_t x = (uint64_t)c << ((fill & 7u) * 8u);
        const uint32_t w = fill >> 3;
        b0 |= w == 0 ? x : 0;
        b1 |= w == 1 ? x : 0;
        b2 |= w == 2 ? x : 0;
        b3 |= w == 3 ? x : 0;
        if (++fill == 32) {
            v0 += b0 * k0; v0 = hll_rotr(v0, 29) + v2;
            v1 += b1 * k1; v1 = hll_rotr(v1, 29) + v3;
            v2 += b2 * k2; v2 = hll_rotr(v2, 29) + v0;
            v3 += b3 * k3; v3 = hll_rotr(v3, 29) + v1;
            b0 = b1 = b2 = b3 = 0;
            fill = 0;
            blocks = true;
        }
    }
    SYBL_HD uint64_t word(uint32_t i) const { return i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3; }
    // n <= 8 pending bytes from byte offset `at` (a multiple of n: the tail steps halve)
    SYBL_HD uint64_t take(uint32_t at, uint32_t n) const {
        const uint64_t x = word(at >> 3) >> ((at & 7u) * 8u);
        return n == 8 ? x : x & (((uint64_t)1 << (n * 8u)) - 1);
    }
    SYBL_HD uint64_t finish() {
        uint64_t h = hash;
        if (blocks) {
            v2 ^= hll_rotr(((v0 + v3) * k0) + v1, 37) * k1;
            v3 ^= hll_rotr(((v1 + v2) * k1) + v0, 37) * k0;
            v0 ^= hll_rotr(((v0 + v2) * k0) + v3, 37) * k1;
            v1 ^= hll_rotr(((v1 + v3) * k1) + v2, 37) * k0;
            h += v0 ^ v1;
        }
        uint32_t at = 0;
        if (fill - at >= 16) {
            uint64_t x0 = h + take(at, 8) * k2; x0 = hll_rotr(x0, 29) * k3;
            uint64_t x1 = h + take(at + 8, 8) * k2; x1 = hll_rotr(x1, 29) * k3;
            x0 ^= hll_rotr(x0 * k0, 21) + x1;
            x1 ^= hll_rotr(x1 * k3, 21) + x0;
            h += x1;
            at += 16;
        }
        if (fill - at >= 8) { h += take(at, 8) * k3; at += 8; h ^= hll_rotr(h, 55) * k1; }
        if (fill - at >= 4) { h += take(at, 4) * k3; at += 4; h ^= hll_rotr(h, 26) * k1; }
        if (fill - at >= 2) { h += take(at, 2) * k3; at += 2; h ^= hll_rotr(h, 48) * k1; }
        if (fill - at >= 1) { h += take(at, 1) * k3; h ^= hll_rotr(h, 37) * k1; }
        h ^= hll_rotr(h, 28);
        h *= k0;
        h ^= hll_rotr(h, 29);
        return h;
    }
    // strconv.FormatInt(v, 10)
    SYBL_HD void put_decimal(int64_t v) {
        uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        if (v < 0) put((uint8_t)'-');
        uint64_t p = 1;
        while (u / p >= 10) p *= 10;  // (the highest power of ten not above u; u / p < 10 cannot overflow p)
        for (; p > 0; p /= 10) put((uint8_t)('0' + (u / p) % 10));
    }
};

// LogLogBeta.Cardinality: alpha m (m - ez) / (beta(ez) + sum 2^-reg), the registers summed in index order
uint64_t hll_cardinality(const uint8_t *regs);  // result.cpp

}  // namespace sybl
