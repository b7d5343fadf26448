// gob.cpp -- see gob.h.  Wire format as specified by Go's encoding/gob documentation
// ("Encoding Details"); sybil's use of it: column_store_io.go, table_io.go, file_decoder.go.
#include "gob.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <stdlib.h>
#include <new>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace sybl {
namespace gob {

namespace {

// predefined type ids (encoding/gob/type.go)
enum { tBool = 1, tInt = 2, tUint = 3, tFloat = 4, tBytes = 5, tString = 6, tComplex = 7, tInterface = 8 };

struct TypeDef {
    enum Kind { kArray, kSlice, kStruct, kMap, kOpaque } kind = kOpaque;
    std::string name;
    int64_t elem = 0, key = 0, len = 0;
    std::vector<std::pair<std::string, int64_t>> fields;
};


#if defined(__x86_64__)
// ---- the varint walk, 64 bytes at a time (AVX-512 VBMI: Zen 4 / Sapphire Rapids and later; checked at run time).
// A gob uint is one byte below 128, else a marker byte (256 - n) and n big-endian data bytes: where the next value starts
// is only known once this one's first byte has been looked at, and a data byte may look like a marker.  The scalar loops
// pay a load, a subtraction and an add in every value's critical path plus a branch the predictor cannot guess whenever
// the lengths are mixed: 7-10 cycles per value, and the record-id and value arrays of a block are 460 000 values -- that
// loop was most of a table load's parse time.  Here a window of 64 bytes is classified at once: J1[i] = i + (the length of
// a value that would start at byte i); J2 = J1 o J1 and J4 = J2 o J2 come from one byte permute each (vpermb), so the walk
// from value to value advances FOUR values per dependent table read and only collects where values start.  The values
// themselves are then gathered eight at a time: one two-table byte permute (vpermi2b) moves every value's data bytes,
// reversed, into its own 64-bit lane.  Two shortcuts in front of the walk: a leading run of one-byte values is copied, and a
// leading run of six or more values of three data bytes -- a value-encoded column's deltas -- is turned into numbers by one
// byte shuffle (every fourth byte is the marker 0xFD, so those are the starts).  (Measured on this round's build host, ns
// per value, scalar -> windows: a column of 1000 distinct values 4.4 -> 2.0, of 64 2.8 -> 1.0, of 16 0.9 -> 0.25,
// value-encoded 3.2 -> 1.6-1.9.  A variant that finds all 64 starts by doubling in registers -- J8 .. J32, six masked
// permutes -- came out the same as the walk: its window-to-window chain is longer than this one's table reads.)
// Returns the values decoded; *pp moves behind them.  Stops early -- the caller's checked reader takes that value, or
// reports it, and comes back -- at a marker with eight data bytes or a byte that is no marker.
// OUT: int64_t, or a narrower element (int32_t / uint16_t) -- *ovf then collects, as set bits, what did not fit.
template <bool SIGNED, typename OUT>
__attribute__((target("avx512f,avx512bw,avx512dq,avx512vbmi,avx512vl,bmi,bmi2,lzcnt,popcnt"))) static uint64_t ints_vbmi(const uint8_t **pp, const uint8_t *end,
                                                                                                                        OUT *dst, uint64_t n, uint64_t *ovf) {
    const uint8_t *q = *pp;
    uint64_t k = 0;
    __m512i misfit = _mm512_setzero_si512();
    alignas(64) uint8_t J1[128], J2[64], J4[64], S[128] = {};  // (S: lanes beyond a window's starts are read, masked out, never used)
    alignas(64) static const uint8_t kIota[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
                                                  22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43,
                                                  44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63};
    const __m512i iota = _mm512_load_si512((const void *)kIota);
    const __m512i c64 = _mm512_set1_epi8(64);
    // (a walk that left the window stays outside: J1[i] = i for i >= 64)
    _mm512_store_si512((void *)(J1 + 64), _mm512_add_epi8(iota, c64));
    const __m512i lane_b = _mm512_set1_epi64(0x0706050403020100ll);  // a byte's place in its 64-bit lane
    const __m512i rep0 = _mm512_broadcast_i32x4(_mm_set_epi8(8, 8, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 0, 0, 0, 0));  // a lane's low byte, eight times
// eight raw values in 64-bit lanes -> zig-zag (SIGNED) -> dst[0 .. 7] under a lane mask, as OUT; what does not fit OUT is noted
// (a macro, not a lambda: a lambda is a function of its own and does not inherit the target attribute)
#define SYBL_EMIT8(RAW, LANES, DST)                                                                                                              \
    do {                                                                                                                                         \
        __m512i v_ = (RAW);                                                                                                                      \
        const __mmask8 m_ = (LANES);                                                                                                             \
        if (SIGNED) v_ = _mm512_xor_si512(_mm512_srli_epi64(v_, 1), _mm512_sub_epi64(_mm512_setzero_si512(), _mm512_and_si512(v_, _mm512_set1_epi64(1)))); \
        if (sizeof(OUT) == 8) {                                                                                                                  \
            _mm512_mask_storeu_epi64((void *)(DST), m_, v_);                                                                                     \
        } else if (sizeof(OUT) == 4) { /* fits int32: v + 2^31 is below 2^32 */                                                                  \
            misfit = _mm512_mask_or_epi64(misfit, m_, misfit, _mm512_srli_epi64(_mm512_add_epi64(v_, _mm512_set1_epi64((int64_t)1 << 31)), 32));  \
            _mm512_mask_cvtepi64_storeu_epi32((void *)(DST), m_, v_);                                                                            \
        } else {                                                                                                                                 \
            misfit = _mm512_mask_or_epi64(misfit, m_, misfit, _mm512_srli_epi64(v_, 16));                                                         \
            _mm512_mask_cvtepi64_storeu_epi16((void *)(DST), m_, v_);                                                                            \
        }                                                                                                                                        \
    } while (0)
    // a dword "FD b2 b1 b0" (memory order) -> b2 b1 b0 as a little-endian number
    const __m512i be24 = _mm512_broadcast_i32x4(_mm_set_epi8((char)0x80, 13, 14, 15, (char)0x80, 9, 10, 11, (char)0x80, 5, 6, 7, (char)0x80, 1, 2, 3));
    while (k < n && (size_t)(end - q) >= 136) {
        const __m512i x = _mm512_loadu_si512((const void *)q), x2 = _mm512_loadu_si512((const void *)(q + 64));
        const __mmask64 hi = _mm512_movepi8_mask(x);  // bytes >= 128
        {
            // A run of values of three data bytes -- what a value-encoded column's deltas are, nineteen in twenty -- needs no
            // walk: the window starts on a value, so while every fourth byte is the marker 0xFD those ARE the starts.  One
            // byte shuffle turns up to sixteen of them into numbers (the general path below is ~5 cycles per value).
            const uint64_t fd = _mm512_cmpeq_epi8_mask(x, _mm512_set1_epi8((char)0xFD));
            const unsigned run = (unsigned)_tzcnt_u64(~(fd | ~0x1111111111111111ull)) >> 2;  // leading values "FD b b b"
            if (run >= 6) {
                const unsigned take = (unsigned)(n - k < run ? n - k : run);
                const __m512i w = _mm512_shuffle_epi8(x, be24);
                SYBL_EMIT8(_mm512_cvtepu32_epi64(_mm512_castsi512_si256(w)), take >= 8 ? (__mmask8)0xFF : (__mmask8)((1u << take) - 1), dst + k);
                if (take > 8)
                    SYBL_EMIT8(_mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(w, 1)), take >= 16 ? (__mmask8)0xFF : (__mmask8)((1u << (take - 8)) - 1),
                               dst + k + 8);
                k += take;
                q += 4 * take;
                continue;
            }
        }
        {
            // a leading run of one-byte values (the whole window for a column whose id deltas all fit seven bits) is copied, not walked
            const unsigned run = hi ? (unsigned)_tzcnt_u64(hi) : 64;
            if (run >= 8) {
                const unsigned take = (unsigned)(n - k < run ? n - k : run);
                for (unsigned g = 0; g < take; g += 8) {
                    const unsigned left = take - g;
                    SYBL_EMIT8(_mm512_cvtepu8_epi64(_mm_loadl_epi64((const __m128i *)(q + g))), left >= 8 ? (__mmask8)0xFF : (__mmask8)((1u << left) - 1), dst + k + g);
                }
                k += take;
                q += take;
                continue;
            }
        }
        // data bytes behind byte i if a value starts there: 0 below 128, else 256 - x = -x (mod 256)
        __m512i u = _mm512_maskz_sub_epi8(hi, _mm512_setzero_si512(), x);
        const __mmask64 badk = _mm512_cmpgt_epu8_mask(u, _mm512_set1_epi8(7));  // eight data bytes, or no marker at all
        u = _mm512_maskz_mov_epi8(~badk, u);  // (as one byte: the walk below stays ascending whatever the bytes are)
        const __m512i j1 = _mm512_add_epi8(_mm512_add_epi8(iota, u), _mm512_set1_epi8(1));
        const __m512i j2 = _mm512_mask_permutexvar_epi8(j1, _mm512_cmplt_epu8_mask(j1, c64), j1, j1);
        const __m512i j4 = _mm512_mask_permutexvar_epi8(j2, _mm512_cmplt_epu8_mask(j2, c64), j2, j2);
        _mm512_store_si512((void *)J1, j1);
        _mm512_store_si512((void *)J2, j2);
        _mm512_store_si512((void *)J4, j4);
        // where values start: S[0], S[1], ... ascending; the first one at or beyond 64 is where the next window begins
        unsigned c = 0, s = 0;
        do {
            const unsigned s2 = J2[s];
            S[c] = (uint8_t)s;
            S[c + 1] = J1[s];
            S[c + 2] = (uint8_t)s2;
            S[c + 3] = J1[s2];
            s = J4[s];
            c += 4;
        } while (s < 64);
        S[c] = (uint8_t)s;
        // (of the last four starts up to three lie beyond the window)
        unsigned cnt = c;
        while (S[cnt - 1] >= 64) cnt--;
        bool stop = false;
        if (badk) {
            // a start on a byte the walk cannot take ends the window there (rare: values of eight data bytes, damaged files)
            const __m512i sv = _mm512_load_si512((const void *)S);
            const uint64_t inside = cnt >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << cnt) - 1);
            const uint64_t bad_start = _mm512_movepi8_mask(_mm512_permutexvar_epi8(sv, _mm512_movm_epi8(badk))) & inside;
            if (bad_start) {
                cnt = (unsigned)_tzcnt_u64(bad_start);
                stop = true;
            }
        }
        const unsigned take = (unsigned)(n - k < cnt ? n - k : cnt);
        for (unsigned g = 0; g < take; g += 8) {
            const __m512i s8 = _mm512_cvtepu8_epi64(_mm_loadl_epi64((const __m128i *)(S + g)));
            const __m512i nx = _mm512_cvtepu8_epi64(_mm_loadl_epi64((const __m128i *)(S + g + 1)));  // where the value after it starts
            const __m512i last = _mm512_shuffle_epi8(_mm512_sub_epi64(nx, _mm512_set1_epi64(1)), rep0);  // the value's last byte
            const __m512i ub = _mm512_shuffle_epi8(_mm512_sub_epi64(_mm512_sub_epi64(nx, s8), _mm512_set1_epi64(1)), rep0);  // its data bytes
            // lane byte b <- window byte last - b for b < u (b = 0 always): the data bytes, least significant first
            const __mmask64 km = _mm512_cmplt_epu8_mask(lane_b, ub) | 0x0101010101010101ull;
            __m512i v = _mm512_maskz_permutex2var_epi8(km, x, _mm512_sub_epi8(last, lane_b), x2);
            const unsigned left = take - g;
            SYBL_EMIT8(v, left >= 8 ? (__mmask8)0xFF : (__mmask8)((1u << left) - 1), dst + k + g);
        }
        k += take;
        q += S[take];
        if (stop) break;
    }
    *pp = q;
    if (sizeof(OUT) != 8) *ovf |= (uint64_t)_mm512_reduce_or_epi64(misfit);
    return k;
#undef SYBL_EMIT8
}
static const bool g_have_vbmi = __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
                                __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("bmi2") && !env("SYBL_GOB_NO_VBMI");
#endif

struct Reader {
    const uint8_t *p, *end;
    std::string *err;
    bool ok = true;

    bool fail(const char *what) {
        if (ok) *err = std::string("gob: ") + what;
        ok = false;
        return false;
    }
    size_t left() const { return (size_t)(end - p); }
    uint64_t uvarint() {
        if (p >= end) {
            fail("truncated uint");
            return 0;
        }
        uint8_t b = *p++;
        if (b < 128) return b;
        int n = 256 - (int)b;  // byte count, stored negated
        if (n > 8 || (size_t)n > left()) {
            fail("bad uint length");
            return 0;
        }
        uint64_t v = 0;
        for (int i = 0; i < n; i++) v = (v << 8) | *p++;
        return v;
    }
    int64_t svarint() {
        uint64_t u = uvarint();
        return (int64_t)((u >> 1) ^ (0 - (u & 1)));
    }
    // n unsigned (or, SIGNED, zig-zag) integers into dst.  When the message certainly holds them (9 bytes each at most,
    // + 8 of slack for the unaligned load) every value is one byte test and, past 127, ONE unaligned big-endian 8-byte
    // load shifted down -- no per-byte loop, no per-value bounds checks: the record-id lists of bucket-encoded columns
    // are 65 536 values per column and block, and this loop was most of a table load's 3.3 s of parse CPU.
    // OUT: int64_t, or int32_t / uint16_t -- `misfit` then collects (as set bits) what did not fit the element.
    uint64_t misfit = 0;
    template <typename OUT>
    void put(OUT *dst, uint64_t k, int64_t v) {
        if (sizeof(OUT) == 4) misfit |= ((uint64_t)v + ((uint64_t)1 << 31)) >> 32;
        else if (sizeof(OUT) == 2) misfit |= (uint64_t)v >> 16;
        dst[k] = (OUT)v;
    }
    template <bool SIGNED, typename OUT = int64_t>
    bool ints(OUT *dst, uint64_t n) {
        uint64_t k = 0;
#if defined(__x86_64__)
        // (the 64-byte windows of ints_vbmi; a value it leaves alone goes through the checked reader, one at a time)
        while (g_have_vbmi && n - k >= 16 && left() >= 136) {
            k += ints_vbmi<SIGNED, OUT>(&p, end, dst + k, n - k, &misfit);
            if (k < n && left() >= 136) {
                put(dst, k++, SIGNED ? svarint() : (int64_t)uvarint());
                if (!ok) return false;
            }
        }
#endif
        if (k < n && left() >= 17) {
            // (a value is at most nine bytes and the unaligned load reads eight behind the length byte: while sixteen bytes
            // are left nothing is read past the end.  Round 3 asked for 9 n + 8 bytes up front -- never true for the value
            // arrays of value-encoded columns (three or four bytes per value), which therefore took the checked loop)
            const uint8_t *q = p, *const safe = end - 16;
            bool bad = false;
            for (; k < n && q <= safe; k++) {
                uint64_t v = *q++;
                if (v >= 128) {
                    const unsigned len = 256u - (unsigned)v;  // byte count, stored negated
                    if (len > 8 || len == 0) {                // malformed: the checked loop below reports it
                        bad = true;
                        break;
                    }
                    uint64_t be;
                    memcpy(&be, q, 8);
                    v = __builtin_bswap64(be) >> (64 - 8 * len);
                    q += len;
                }
                // (zig-zag without a branch: the sign bit of a value-encoded column's deltas is a coin toss, and as a
                // branch it cost more than the rest of the loop -- 8.5 against 3.0 ns per value)
                put(dst, k, SIGNED ? (int64_t)((v >> 1) ^ (0 - (v & 1))) : (int64_t)v);
            }
            p = bad ? q - 1 : q;  // (bad: back onto the length byte of the value that stopped the fast loop)
        }
        for (; k < n && ok; k++) put(dst, k, SIGNED ? svarint() : (int64_t)uvarint());
        return ok;
    }
    double float64() {
        uint64_t u = uvarint(), r = 0;  // IEEE bits, byte-reversed
        for (int i = 0; i < 8; i++) r |= ((u >> (8 * i)) & 0xFF) << (8 * (7 - i));
        double d;
        memcpy(&d, &r, 8);
        return d;
    }
    bool bytes(std::string &out) {
        uint64_t n = uvarint();
        if (!ok) return false;
        if (n > left()) return fail("truncated string");
        out.assign((const char *)p, (size_t)n);
        p += n;
        return true;
    }
};

struct Decoder {
    std::map<int64_t, TypeDef> types;
    std::string *err;
    DecodeOpts opts;
    bool stop = false;                      // DecodeOpts::raw_values was found: unwind
    const std::string *cur_field = nullptr; // name of the top-level struct's field being decoded
    bool cur_ends_plain = false;            // ... and what may follow it is at most ONE more field, `tail_delta` further on, of an
    int64_t cur_tail_delta = 0;             // integer type: [0] or [tail_delta][value][0] is then all the struct can end with

    bool fail(const std::string &m) {
        *err = "gob: " + m;
        return false;
    }

    // CommonType{Name string, Id typeId}
    bool common_type(Reader &r, std::string &name) {
        int64_t f = -1;
        for (;;) {
            uint64_t d = r.uvarint();
            if (!r.ok) return false;
            if (d == 0) return true;
            f += (int64_t)d;
            if (f == 0) {
                if (!r.bytes(name)) return false;
            } else if (f == 1) {
                r.svarint();
            } else {
                return fail("unexpected CommonType field");
            }
        }
    }

    // one of arrayType / sliceType / structType / mapType / gobEncoderType
    bool type_body(Reader &r, int which, TypeDef &td) {
        int64_t f = -1;
        for (;;) {
            uint64_t d = r.uvarint();
            if (!r.ok) return false;
            if (d == 0) return true;
            f += (int64_t)d;
            if (f == 0) {
                if (!common_type(r, td.name)) return false;
                continue;
            }
            switch (which) {
            case 0:  // ArrayT: Elem, Len
                if (f == 1) td.elem = r.svarint();
                else if (f == 2) td.len = r.svarint();
                else return fail("unexpected arrayType field");
                break;
            case 1:  // SliceT: Elem
                if (f == 1) td.elem = r.svarint();
                else return fail("unexpected sliceType field");
                break;
            case 2:  // StructT: Field []fieldType{Name string, Id typeId}
                if (f == 1) {
                    uint64_t n = r.uvarint();
                    for (uint64_t i = 0; i < n && r.ok; i++) {
                        std::string fname;
                        int64_t fid = 0, ff = -1;
                        for (;;) {
                            uint64_t dd = r.uvarint();
                            if (!r.ok) return false;
                            if (dd == 0) break;
                            ff += (int64_t)dd;
                            if (ff == 0) {
                                if (!r.bytes(fname)) return false;
                            } else if (ff == 1) {
                                fid = r.svarint();
                            } else {
                                return fail("unexpected fieldType field");
                            }
                        }
                        td.fields.emplace_back(fname, fid);
                    }
                } else {
                    return fail("unexpected structType field");
                }
                break;
            case 3:  // MapT: Key, Elem
                if (f == 1) td.key = r.svarint();
                else if (f == 2) td.elem = r.svarint();
                else return fail("unexpected mapType field");
                break;
            default:
                return fail("unexpected field in opaque type definition");
            }
            if (!r.ok) return false;
        }
    }

    // wireType{ArrayT, SliceT, StructT, MapT, GobEncoderT, BinaryMarshalerT, TextMarshalerT}
    bool wire_type(Reader &r, TypeDef &td) {
        int64_t f = -1;
        for (;;) {
            uint64_t d = r.uvarint();
            if (!r.ok) return false;
            if (d == 0) return true;
            f += (int64_t)d;
            if (f < 0 || f > 6) return fail("unexpected wireType field");
            static const TypeDef::Kind kinds[] = {TypeDef::kArray, TypeDef::kSlice, TypeDef::kStruct, TypeDef::kMap,
                                                  TypeDef::kOpaque, TypeDef::kOpaque, TypeDef::kOpaque};
            td.kind = kinds[f];
            if (!type_body(r, (int)f, td)) return false;
        }
    }

    bool is_int_kind(int64_t id) const { return id == tInt || id == tUint; }

    bool value(Reader &r, int64_t id, Value &out, int depth) {
        if (depth > 64) return fail("nesting too deep");
        switch (id) {
        case tBool:
            out.kind = Value::kBool;
            out.i = r.uvarint() != 0;
            return r.ok;
        case tInt:
            out.kind = Value::kInt;
            out.i = r.svarint();
            return r.ok;
        case tUint:
            out.kind = Value::kUint;
            out.u = r.uvarint();
            out.i = (int64_t)out.u;
            return r.ok;
        case tFloat:
            out.kind = Value::kFloat;
            out.f = r.float64();
            return r.ok;
        case tBytes:
        case tString:
            out.kind = Value::kString;
            return r.bytes(out.s);
        case tComplex:
            return fail("complex values are not supported");
        case tInterface:
            return fail("interface values are not supported (not used by sybil's column/info files)");
        default:
            break;
        }
        auto it = types.find(id);
        if (it == types.end()) return fail("value of undefined type id " + std::to_string((long long)id));
        const TypeDef &td = it->second;
        out.type_name = td.name;
        switch (td.kind) {
        case TypeDef::kStruct: {
            out.kind = Value::kStruct;
            int64_t f = -1;
            for (;;) {
                uint64_t d = r.uvarint();
                if (!r.ok) return false;
                if (d == 0) return true;
                f += (int64_t)d;
                if (f < 0 || f >= (int64_t)td.fields.size()) return fail("struct field index out of range in " + td.name);
                ValuePtr v = std::make_shared<Value>();
                if (depth == 0) {
                    cur_field = &td.fields[(size_t)f].first;
                    // (DecodeOpts::raw_*: the caller will hold what stands behind the slice against "[0] or [d][int][0]" -- that is
                    // only what THIS reader would accept there if every later field is an integer and the last one is d further on)
                    // (the field right behind a `Bins` slice -- `Values` -- may be anything: the bucket parser sends a file in which it
                    // shows up to the host parser)
                    cur_ends_plain = true;
                    const bool is_bins = *cur_field == "Bins";
                    for (size_t k = (size_t)f + (is_bins ? 2 : 1); k < td.fields.size(); k++) cur_ends_plain = cur_ends_plain && is_int_kind(td.fields[k].second);
                    cur_tail_delta = (int64_t)td.fields.size() - 1 - f;
                }
                if (!value(r, td.fields[(size_t)f].second, *v, depth + 1)) return false;
                out.fields.emplace_back(td.fields[(size_t)f].first, v);
                if (stop) return true;
            }
        }
        case TypeDef::kArray:
        case TypeDef::kSlice: {
            uint64_t n = r.uvarint();
            if (!r.ok) return false;
            if (n > r.left()) return fail("slice longer than the message");
            if (is_int_kind(td.elem)) {
                out.kind = Value::kIntVec;
                if (opts.raw_values && depth == 1 && td.elem == tInt && cur_field && *cur_field == "Values" && cur_ends_plain && cur_tail_delta == 1) {
                    opts.raw_values->p = r.p;
                    opts.raw_values->end = r.end;
                    opts.raw_values->n = n;
                    opts.raw_values->hit = true;
                    stop = true;
                    return true;
                }
                if (opts.narrow && depth <= 1) {
                    // as int32 when every value fits (DecodeOpts); the reader goes back and takes them as int64 when not
                    const Reader at = r;
                    out.ints.w = 4;
                    out.ints.resize((size_t)n);
                    r.misfit = 0;
                    const bool fine = td.elem == tInt ? r.ints<true, int32_t>((int32_t *)out.ints.data(), n) : r.ints<false, int32_t>((int32_t *)out.ints.data(), n);
                    if (!fine) return false;
                    if (r.misfit == 0) return true;
                    r = at;
                    out.ints.w = 8;
                }
                out.ints.resize((size_t)n);
                return td.elem == tInt ? r.ints<true>(out.ints.data(), n) : r.ints<false>(out.ints.data(), n);
            }
            {
                // []struct{Value int; Records []int}: flat arrays (see Value::kBinVec)
                auto et = types.find(td.elem);
                int fv = -1, fr = -1;
                bool vs = false, rs = false;
                if (et != types.end() && et->second.kind == TypeDef::kStruct && et->second.fields.size() == 2) {
                    for (int k = 0; k < 2; k++) {
                        const auto &fd = et->second.fields[(size_t)k];
                        if (fd.first == "Value" && is_int_kind(fd.second)) {
                            fv = k;
                            vs = fd.second == tInt;
                        } else if (fd.first == "Records") {
                            auto st = types.find(fd.second);
                            if (st != types.end() && st->second.kind == TypeDef::kSlice && is_int_kind(st->second.elem)) {
                                fr = k;
                                rs = st->second.elem == tInt;
                            }
                        }
                    }
                }
                if (fv >= 0 && fr >= 0) {
                    out.kind = Value::kBinVec;
                    if (opts.raw_bins && depth == 1 && fv == 0 && fr == 1 && vs && !rs && cur_field && *cur_field == "Bins" && cur_ends_plain &&
                        cur_tail_delta == 2) {
                        opts.raw_bins->p = r.p;
                        opts.raw_bins->end = r.end;
                        opts.raw_bins->n = n;
                        opts.raw_bins->hit = true;
                        out.bin_off.assign(1, 0);
                        stop = true;
                        return true;
                    }
                    out.bin_order = fr < fv ? 1 : 0;
                    // (DecodeOpts::narrow: the records as uint16 while they fit; one that does not sends the reader back
                    // to the first bin to take them all as int64 -- a block of more than 65536 rows, or a damaged file)
                    bool narrow = opts.narrow && !rs;
                    const Reader first_bin = r;
                again:
                    out.ints.w = narrow ? 2 : 8;
                    out.ints.resize(0);
                    out.bin_val.clear();
                    out.bin_has.clear();
                    out.bin_off.assign(1, 0);
                    out.bin_val.reserve((size_t)n);
                    out.bin_has.reserve((size_t)n);
                    r.misfit = 0;
                    for (uint64_t k = 0; k < n; k++) {
                        int64_t f = -1, val = 0;
                        uint8_t has = 0;
                        for (;;) {
                            const uint64_t d = r.uvarint();
                            if (!r.ok) return false;
                            if (d == 0) break;
                            f += (int64_t)d;
                            if (f == fv) {
                                val = vs ? r.svarint() : (int64_t)r.uvarint();
                                has |= 1;
                            } else if (f == fr) {
                                const uint64_t m = r.uvarint();
                                if (!r.ok) return false;
                                if (m > r.left()) return fail("slice longer than the message");
                                const size_t at = out.ints.size();
                                out.ints.resize(at + (size_t)m);
                                if (narrow) {
                                    r.ints<false, uint16_t>((uint16_t *)out.ints.data() + at, m);
                                    if (r.ok && r.misfit) {
                                        r = first_bin;
                                        narrow = false;
                                        goto again;
                                    }
                                } else {
                                    int64_t *dst = out.ints.data() + at;
                                    if (rs) r.ints<true>(dst, m);
                                    else r.ints<false>(dst, m);
                                }
                                has |= 2;
                            } else {
                                return fail("struct field index out of range in " + et->second.name);
                            }
                            if (!r.ok) return false;
                        }
                        out.bin_val.push_back(val);
                        out.bin_off.push_back((int64_t)out.ints.size());
                        out.bin_has.push_back(has);
                    }
                    return r.ok;
                }
            }
            if (td.elem == tFloat) {
                out.kind = Value::kFloatVec;
                out.floats.resize((size_t)n);
                for (uint64_t k = 0; k < n; k++) out.floats[(size_t)k] = r.float64();
                return r.ok;
            }
            out.kind = Value::kSlice;
            out.items.reserve((size_t)n);
            for (uint64_t k = 0; k < n; k++) {
                ValuePtr v = std::make_shared<Value>();
                if (!value(r, td.elem, *v, depth + 1)) return false;
                out.items.push_back(v);
            }
            return true;
        }
        case TypeDef::kMap: {
            uint64_t n = r.uvarint();
            if (!r.ok) return false;
            if (n > r.left()) return fail("map longer than the message");
            out.kind = Value::kMap;
            for (uint64_t k = 0; k < n; k++) {
                ValuePtr kv = std::make_shared<Value>(), vv = std::make_shared<Value>();
                if (!value(r, td.key, *kv, depth + 1)) return false;
                if (!value(r, td.elem, *vv, depth + 1)) return false;
                out.entries.emplace_back(kv, vv);
            }
            return true;
        }
        default:
            // GobEncoder / BinaryMarshaler / TextMarshaler: opaque bytes
            out.kind = Value::kString;
            return r.bytes(out.s);
        }
    }

    bool run(const uint8_t *data, size_t size, Value &out) {
        const uint8_t *p = data, *end = data + size;
        while (p < end) {
            Reader hdr{p, end, err};
            uint64_t mlen = hdr.uvarint();
            if (!hdr.ok) return false;
            if (mlen > hdr.left()) return fail("truncated message");
            Reader r{hdr.p, hdr.p + mlen, err};
            p = hdr.p + mlen;
            int64_t id = r.svarint();
            if (!r.ok) return false;
            if (id < 0) {
                TypeDef td;
                if (!wire_type(r, td)) return false;
                types[-id] = td;
                continue;
            }
            auto it = types.find(id);
            bool is_struct = it != types.end() && it->second.kind == TypeDef::kStruct;
            if (!is_struct) {
                // a top-level non-struct value is preceded by a zero field delta
                if (r.uvarint() != 0) return fail("missing singleton marker");
            }
            return value(r, id, out, 0);
        }
        return fail("no value in stream");
    }
};

}  // namespace

// ---- IntBuf: a per-thread stock of released buffers (a loader worker decodes one block's files at a time: a handful of
// arrays alive at once)
namespace {
struct IntStock {
    static constexpr int kKeep = 16;
    static constexpr size_t kKeepUnits = (size_t)1 << 23;  // int64 units a thread's stock holds in all (64 MB: ADVICE r4)
    int64_t *p[kKeep];
    size_t cap[kKeep];
    size_t total = 0;
    int n = 0;
    ~IntStock() {
        for (int i = 0; i < n; i++) free(p[i]);
    }
};
thread_local IntStock g_stock;
}  // namespace

void IntBuf::release() {
    if (!p) return;
    IntStock &S = g_stock;
    if (S.n < IntStock::kKeep && cap <= ((size_t)1 << 22) && S.total + cap <= IntStock::kKeepUnits) {
        S.p[S.n] = p;
        S.cap[S.n] = cap;
        S.total += cap;
        S.n++;
    } else {
        free(p);
    }
    p = nullptr;
    n = cap = 0;
}

void IntBuf::grow(size_t m) {
    const size_t need = (m * (size_t)w + 7) / 8;  // int64 units
    if (need > cap) {
        IntStock &S = g_stock;
        // the smallest stocked buffer that is large enough; else grow (doubling: a bucket-encoded file appends bin by bin)
        int best = -1;
        for (int i = 0; i < S.n; i++)
            if (S.cap[i] >= need && (best < 0 || S.cap[i] < S.cap[best])) best = i;
        if (best >= 0 && !p) {
            p = S.p[best];
            cap = S.cap[best];
            S.total -= cap;
            S.p[best] = S.p[S.n - 1];
            S.cap[best] = S.cap[S.n - 1];
            S.n--;
        } else {
            size_t want = cap * 2 > need ? cap * 2 : need;
            if (want < 4096) want = 4096;
            int64_t *q = (int64_t *)realloc(p, want * sizeof(int64_t));
            if (!q) throw std::bad_alloc();
            p = q;
            cap = want;
        }
    }
    n = m;
}

void IntBuf::assign(const int64_t *src, size_t m) {
    w = 8;
    resize(m);
    if (m) memcpy(p, src, m * sizeof(int64_t));
}

void IntBuf::copy_from(const IntBuf &o) {
    w = o.w;
    resize(o.n);
    if (o.n) memcpy(p, o.p, o.n * (size_t)w);
}

const Value *Value::field(const char *name) const {
    for (auto &f : fields)
        if (f.first == name) return f.second.get();
    return nullptr;
}

int64_t Value::as_int(int64_t dflt) const {
    switch (kind) {
    case kBool:
    case kInt:
    case kUint: return i;
    case kFloat: return (int64_t)f;
    default: return dflt;
    }
}

bool decode(const uint8_t *data, size_t size, Value &out, std::string &err, const DecodeOpts *opts) {
    Decoder d;
    d.err = &err;
    if (opts) d.opts = *opts;
    return d.run(data, size, out);
}

FileBuf::~FileBuf() { free(p); }

static bool filebuf_reserve(FileBuf &b, size_t want) {
    if (want <= b.cap) return true;
    size_t cap = b.cap + b.cap / 2;
    if (cap < want) cap = want;
    if (cap < 4096) cap = 4096;
    uint8_t *q = (uint8_t *)realloc(b.p, cap);
    if (!q) return false;
    b.p = q;
    b.cap = cap;
    return true;
}

bool read_file(const std::string &path, FileBuf &out, std::string &err) {
    // one open per candidate name, no existence probes: these are the syscalls of every column file of every block
    std::string use = path;
    int fd = open(use.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        use = path + ".gz";
        fd = open(use.c_str(), O_RDONLY | O_CLOEXEC);
    }
    if (fd < 0) {
        err = "cannot open " + path;
        return false;
    }
    const bool gz = use.size() > 3 && use.compare(use.size() - 3, 3, ".gz") == 0;
    out.n = 0;
    if (gz) {
        gzFile g = gzdopen(fd, "rb");  // (takes the descriptor over)
        if (!g) {
            close(fd);
            err = "cannot gzopen " + use;
            return false;
        }
        int n = 0;
        for (;;) {
            if (!filebuf_reserve(out, out.n + ((size_t)1 << 16))) {
                gzclose(g);
                err = "out of memory reading " + use;
                return false;
            }
            n = gzread(g, out.p + out.n, 1 << 16);
            if (n <= 0) break;
            out.n += (size_t)n;
        }
        gzclose(g);
        if (n < 0) {
            err = "gzip error in " + use;
            return false;
        }
        return true;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) {
        close(fd);
        err = "cannot stat " + use;
        return false;
    }
    // the size fstat reported, plus one byte: a read that fills it means the file grew under us ("short read" as before)
    const size_t want = (size_t)st.st_size;
    if (!filebuf_reserve(out, want + 1)) {
        close(fd);
        err = "out of memory reading " + use;
        return false;
    }
    size_t got = 0;
    while (got <= want) {
        const ssize_t n = read(fd, out.p + got, want + 1 - got);
        if (n < 0) {
            if (errno == EINTR) continue;
            close(fd);
            err = "read error on " + use;
            return false;
        }
        if (n == 0) break;
        got += (size_t)n;
    }
    close(fd);
    if (got != want) {
        err = "short read on " + use;
        return false;
    }
    out.n = got;
    return true;
}

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err) {
    FileBuf b;
    if (!read_file(path, b, err)) {
        out.clear();
        return false;
    }
    out.assign(b.p, b.p + b.n);
    return true;
}

static void json_str(const std::string &s, std::string &o) {
    o += '"';
    for (unsigned char c : s) {
        if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20) {
            char b[8];
            snprintf(b, sizeof(b), "\\u%04x", c);
            o += b;
        } else o += (char)c;
    }
    o += '"';
}

void to_json(const Value &v, std::string &o) {
    char b[64];
    switch (v.kind) {
    case Value::kNil: o += "null"; break;
    case Value::kBool: o += v.i ? "true" : "false"; break;
    case Value::kInt: snprintf(b, sizeof(b), "%lld", (long long)v.i); o += b; break;
    case Value::kUint: snprintf(b, sizeof(b), "%llu", (unsigned long long)v.u); o += b; break;
    case Value::kFloat: snprintf(b, sizeof(b), "%.17g", v.f); o += b; break;
    case Value::kString: json_str(v.s, o); break;
    case Value::kStruct:
        o += "{";
        for (size_t i = 0; i < v.fields.size(); i++) {
            if (i) o += ",";
            json_str(v.fields[i].first, o);
            o += ":";
            to_json(*v.fields[i].second, o);
        }
        o += "}";
        break;
    case Value::kSlice:
        o += "[";
        for (size_t i = 0; i < v.items.size(); i++) {
            if (i) o += ",";
            to_json(*v.items[i], o);
        }
        o += "]";
        break;
    case Value::kMap:
        o += "[";
        for (size_t i = 0; i < v.entries.size(); i++) {
            if (i) o += ",";
            o += "[";
            to_json(*v.entries[i].first, o);
            o += ",";
            to_json(*v.entries[i].second, o);
            o += "]";
        }
        o += "]";
        break;
    case Value::kBinVec:
        // rendered exactly as the generic tree would be: fields in struct order, zero-valued ones absent
        o += "[";
        for (size_t k = 0; k < v.bin_val.size(); k++) {
            if (k) o += ",";
            o += "{";
            bool first = true;
            for (int step = 0; step < 2; step++) {
                const bool records = (step == 0) == (v.bin_order == 1);
                if (records && (v.bin_has[k] & 2)) {
                    if (!first) o += ",";
                    first = false;
                    o += "\"Records\":[";
                    for (int64_t i = v.bin_off[k]; i < v.bin_off[k + 1]; i++) {
                        if (i > v.bin_off[k]) o += ",";
                        snprintf(b, sizeof(b), "%lld", (long long)v.ints.at((size_t)i));
                        o += b;
                    }
                    o += "]";
                } else if (!records && (v.bin_has[k] & 1)) {
                    if (!first) o += ",";
                    first = false;
                    snprintf(b, sizeof(b), "\"Value\":%lld", (long long)v.bin_val[k]);
                    o += b;
                }
            }
            o += "}";
        }
        o += "]";
        break;
    case Value::kIntVec:
        o += "[";
        for (size_t i = 0; i < v.ints.size(); i++) {
            if (i) o += ",";
            snprintf(b, sizeof(b), "%lld", (long long)v.ints.at(i));
            o += b;
        }
        o += "]";
        break;
    case Value::kFloatVec:
        o += "[";
        for (size_t i = 0; i < v.floats.size(); i++) {
            if (i) o += ",";
            snprintf(b, sizeof(b), "%.17g", v.floats[i]);
            o += b;
        }
        o += "]";
        break;
    }
}

}  // namespace gob
}  // namespace sybl
