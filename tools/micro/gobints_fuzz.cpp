// Reader::ints (csrc/gob.cpp) with the AVX-512 VBMI windows against the checked value-by-value reader, on random byte
// strings -- valid streams, marker soup, bytes that are no marker, values of eight data bytes: the same values, the same
// position behind them, the same verdict.  Build and run (any host; without VBMI both sides are the scalar loops):
//   g++ -O2 -std=c++17 -I../../sybil_amd/csrc -I../../include gobints_fuzz.cpp -lz -o gobints_fuzz && ./gobints_fuzz [trials = 400000]
// (tests/test_gob.py builds and runs it with 60 000)
#include "../../sybil_amd/csrc/gob.cpp"
#include <random>
namespace sybl {
const char *env(const char *name) { return getenv(name); }
}  // namespace sybl
using namespace sybl::gob;

// a valid stream of values whose byte counts (0 = a one-byte value, else the data bytes behind a marker) come from `next`
template <class F>
static void fill_valid(std::vector<uint8_t> &buf, std::mt19937_64 &rng, F next) {
    size_t i = 0;
    const size_t len = buf.size();
    while (i + 10 < len) {
        const unsigned nb = next();
        if (nb == 0) buf[i++] = (uint8_t)(rng() % 128);
        else {
            buf[i++] = (uint8_t)(256 - nb);
            for (unsigned j = 0; j < nb; j++) buf[i++] = (uint8_t)rng();
        }
    }
    for (; i < len; i++) buf[i] = 0;
}

// the narrow element types (DecodeOpts::narrow): the same values where they fit, `misfit` set exactly when one does not
template <bool SIGNED, typename OUT>
static bool narrow_agrees(const std::vector<uint8_t> &buf, uint64_t n, const std::vector<int64_t> &want, bool ok_want, const uint8_t *p_want) {
    std::string e;
    Reader r{buf.data(), buf.data() + buf.size(), &e};
    std::vector<OUT> out(n, (OUT)-7);
    const bool ok = r.template ints<SIGNED, OUT>(out.data(), n);
    if (ok != ok_want) return false;
    if (!ok) return true;
    if (r.p != p_want) return false;
    bool any = false;
    for (uint64_t k = 0; k < n; k++) {
        const bool fits = sizeof(OUT) == 4 ? (want[k] >= INT32_MIN && want[k] <= INT32_MAX) : (want[k] >= 0 && want[k] <= 65535);
        any = any || !fits;
        if (fits && (int64_t)out[k] != want[k]) return false;
    }
    return any == (r.misfit != 0);
}

int main(int argc, char **argv) {
    std::mt19937_64 rng(99);
    long trials = 0, bad = 0;
    const int n_trials = argc > 1 ? atoi(argv[1]) : 400000;
    for (int t = 0; t < n_trials; t++) {
        size_t len = 136 + rng() % 900;
        std::vector<uint8_t> buf(len);
        int mode = rng() % 11;
        if (mode >= 5) {
            // the shapes the run front end is made for, and the ones it must give up on
            unsigned left = 0, shape = 0;
            const unsigned pa = 2 + rng() % 30;  // a change of shape every pa values or so
            auto pick = [&](std::initializer_list<unsigned> of) { return *(of.begin() + rng() % of.size()); };
            fill_valid(buf, rng, [&]() -> unsigned {
                switch (mode) {
                case 5: return rng() % 8 == 0 ? 0 : 2;                      // id deltas of a column of 1000 values: "FE b b", now and then one byte
                case 6: return rng() % 8 == 0 ? pick({1u, 2u}) : 0;          // ... of 64 values: one byte, now and then "FF b" / "FE b b"
                case 7: return rng() % 15 == 0 ? pick({2u, 4u}) : 3;         // value deltas: "FD b b b", now and then shorter / longer
                case 8: return (left++ & 1) ? 0 : pick({1u, 2u, 3u});       // alternating: every run is one value long
                case 9:                                                       // runs of random shapes and lengths
                    if (left == 0) {
                        left = 1 + rng() % pa;
                        shape = pick({0u, 0u, 1u, 2u, 2u, 3u, 3u, 4u, 8u});
                    }
                    left--;
                    return shape;
                default: return rng() % 40 == 0 ? (unsigned)(rng() % 9) : pick({0u, 1u, 2u, 3u});
                }
            });
            if (rng() % 4 == 0) buf[rng() % len] = (uint8_t)rng();  // ... and damage
        }
        if (mode == 4) {
            // valid values, nearly all of three data bytes (a value-encoded column's deltas), now and then another length
            size_t i = 0;
            while (i + 10 < len) {
                const unsigned nb = rng() % 12 == 0 ? (unsigned)(rng() % 9) : 3;
                if (nb == 0) buf[i++] = (uint8_t)(rng() % 128);
                else {
                    buf[i++] = (uint8_t)(256 - nb);
                    for (unsigned j = 0; j < nb; j++) buf[i++] = (uint8_t)rng();
                }
            }
            for (; i < len; i++) buf[i] = 0;
        }
        if (mode < 4) for (auto &b : buf) {
            uint64_t r = rng();
            if (mode == 0) b = (uint8_t)r;                                   // soup
            else if (mode == 1) b = (r % 10 < 7) ? (uint8_t)(r >> 8) % 128 : (uint8_t)(0xF8 + (r >> 8) % 8);   // markers and small
            else if (mode == 2) b = (r % 16 == 0) ? (uint8_t)(0x80 + (r >> 8) % 0x78) : (uint8_t)((r >> 8) % 128);  // some invalid markers
            else b = (r % 3 == 0) ? 0xF7 + (r >> 8) % 9 : (uint8_t)(r >> 16);
        }
        uint64_t n = 1 + rng() % 400;
        std::vector<int64_t> a(n, -7), b(n, -7);
        std::string e1, e2;
        Reader r1{buf.data(), buf.data() + len, &e1}, r2{buf.data(), buf.data() + len, &e2};
        bool sg = rng() & 1;
        bool ok1 = sg ? r1.ints<true>(a.data(), n) : r1.ints<false>(a.data(), n);
        // scalar reference: the checked reader only
        uint64_t k = 0;
        for (; k < n && r2.ok; k++) b[k] = sg ? r2.svarint() : (int64_t)r2.uvarint();
        bool ok2 = r2.ok;
        trials++;
        if (ok1 != ok2) { bad++; printf("ok mismatch t=%d mode=%d n=%zu\n", t, mode, (size_t)n); if (bad > 5) return 1; continue; }
        if (ok1) {
            if (a != b || r1.p != r2.p) { bad++; printf("value/pos mismatch t=%d mode=%d n=%zu\n", t, mode, (size_t)n); if (bad > 5) return 1; }
        }
        const bool nar = sg ? narrow_agrees<true, int32_t>(buf, n, b, ok2, r2.p) : narrow_agrees<false, uint16_t>(buf, n, b, ok2, r2.p);
        const bool nar2 = sg ? narrow_agrees<true, uint16_t>(buf, n, b, ok2, r2.p) : narrow_agrees<false, int32_t>(buf, n, b, ok2, r2.p);
        if (!nar || !nar2) { bad++; printf("narrow mismatch t=%d mode=%d n=%zu\n", t, mode, (size_t)n); if (bad > 5) return 1; }
    }
    printf("%ld trials, %ld mismatches (vbmi %d)\n", trials, bad, (int)g_have_vbmi);
    return bad != 0;
}
