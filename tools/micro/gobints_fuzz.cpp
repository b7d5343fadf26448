// Reader::ints (csrc/gob.cpp) with the AVX-512 VBMI windows against the checked value-by-value reader, on random byte
// strings -- valid streams, marker soup, bytes that are no marker, values of eight data bytes: the same values, the same
// position behind them, the same verdict.  Build and run (any host; without VBMI both sides are the scalar loops):
//   hipcc -O2 -std=c++17 -x c++ -I../../sybil_amd/csrc -I../../include gobints_fuzz.cpp -lz -o gobints_fuzz && ./gobints_fuzz
#include "../../sybil_amd/csrc/gob.cpp"
#include <random>
using namespace sybl::gob;
int main() {
    std::mt19937_64 rng(99);
    long trials = 0, bad = 0;
    for (int t = 0; t < 200000; t++) {
        size_t len = 136 + rng() % 600;
        std::vector<uint8_t> buf(len);
        int mode = rng() % 5;
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
        if (mode != 4) for (auto &b : buf) {
            uint64_t r = rng();
            if (mode == 0) b = (uint8_t)r;                                   // soup
            else if (mode == 1) b = (r % 10 < 7) ? (uint8_t)(r >> 8) % 128 : (uint8_t)(0xF8 + (r >> 8) % 8);   // markers and small
            else if (mode == 2) b = (r % 16 == 0) ? (uint8_t)(0x80 + (r >> 8) % 0x78) : (uint8_t)((r >> 8) % 128);  // some invalid markers
            else b = (r % 3 == 0) ? 0xF7 + (r >> 8) % 9 : (uint8_t)(r >> 16);
        }
        uint64_t n = 1 + rng() % 300;
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
    }
    printf("%ld trials, %ld mismatches (vbmi %d)\n", trials, bad, (int)g_have_vbmi);
    return bad != 0;
}
