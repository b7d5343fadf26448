// gob::decode (csrc/gob.cpp) on damaged column files: every file given on the command line is decoded as it is, then
// with random bytes overwritten, bytes inserted / removed, the tail cut off and huge lengths planted, with and without the
// narrow slices the loader asks for.  Nothing is checked but that decode returns (true or false) -- build with the
// sanitizers so that a read or write outside a buffer, or undefined arithmetic on a hostile length, stops the run:
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -I../../sybil_amd/csrc -I../../include gobfile_fuzz.cpp -lz -o gobfile_fuzz
//   python loader_parse_blocks.py /tmp/lp && ./gobfile_fuzz 3000 /tmp/lp/t/block000000001/*.db /tmp/lp/t/info.db
#include "../../sybil_amd/csrc/gob.cpp"
#include <random>
namespace sybl {
const char *env(const char *name) { return getenv(name); }
}  // namespace sybl
using namespace sybl::gob;

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: gobfile_fuzz <mutations per file> <file> ...\n");
        return 2;
    }
    const int per_file = atoi(argv[1]);
    std::mt19937_64 rng(4242);
    long decoded = 0, refused = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> orig;
        std::string err;
        if (!read_file(argv[a], orig, err) || orig.empty()) {
            fprintf(stderr, "cannot read %s: %s\n", argv[a], err.c_str());
            return 2;
        }
        for (int t = 0; t <= per_file; t++) {
            std::vector<uint8_t> b = orig;
            if (t > 0) {
                const int kind = (int)(rng() % 5);
                const size_t n = b.size();
                // (most damage near the head: the type definitions and the first lengths decide everything behind them)
                auto place = [&]() { return (size_t)(rng() % 3 == 0 ? rng() % n : rng() % std::min<size_t>(n, 400)); };
                if (kind == 0) {
                    for (int k = 0, m = 1 + (int)(rng() % 4); k < m; k++) b[place()] = (uint8_t)rng();
                } else if (kind == 1) {
                    b.resize(rng() % n);  // cut
                } else if (kind == 2) {
                    const size_t at = place();
                    b.insert(b.begin() + (long)at, (size_t)(1 + rng() % 9), (uint8_t)rng());
                } else if (kind == 3) {
                    const size_t at = place();
                    b.erase(b.begin() + (long)at, b.begin() + (long)std::min(n, at + 1 + rng() % 9));
                } else {
                    // a huge length where a small one stood: 0xF8 + eight bytes
                    const size_t at = place();
                    b[at] = 0xF8;
                    for (size_t k = 1; k <= 8 && at + k < n; k++) b[at + k] = (uint8_t)(rng() % 4 == 0 ? 0xFF : rng());
                }
            }
            // (an exact-size heap copy: the sanitizer sees a read one byte past the message)
            std::unique_ptr<uint8_t[]> exact(new uint8_t[b.size() ? b.size() : 1]);
            if (!b.empty()) memcpy(exact.get(), b.data(), b.size());
            for (int narrow = 0; narrow < 2; narrow++) {
                Value v;
                DecodeOpts o;
                o.narrow = narrow != 0;
                std::string e;
                bool ok = false;
                try {
                    ok = decode(exact.get(), b.size(), v, e, &o);
                } catch (const std::exception &) {  // (bad_alloc / length_error from a hostile length: the loader's workers catch it too)
                    ok = false;
                }
                (ok ? decoded : refused)++;
            }
        }
    }
    printf("%ld decoded, %ld refused\n", decoded, refused);
    return 0;
}
