// Re2Lite (csrc/re2lite.cpp: the str filters' and -str-replace's regular expressions) on random patterns -- pieces of RE2
// syntax glued together at random, and plain byte soup -- and random texts (ASCII, multi-byte runes, invalid UTF-8): compile,
// search, replace_all with random templates.  Nothing is checked but that every call returns; build with the sanitizers:
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -I../../sybil_amd/csrc re2lite_fuzz.cpp ../../sybil_amd/csrc/re2lite.cpp -o re2lite_fuzz
//   ./re2lite_fuzz [trials = 200000]
// (tests/test_regex.py holds the engine against Python's re on the semantics; this is about memory.)
#include "re2lite.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
using namespace sybl;

int main(int argc, char **argv) {
    const long trials = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(2718);
    static const char *piece[] = {"a", "b", "ab", ".", "\\d", "\\D", "\\w", "\\W", "\\s", "\\S", "\\b", "\\B", "\\A", "\\z", "^", "$", "[a-c]", "[^a-c]", "[[:alpha:]]", "[[:^digit:]x]",
                                  "[\\d-z]", "[]a]", "[^]a]", "(", ")", "(?:", "(?i)", "(?s)", "(?m)", "(?U)", "(?-i)", "(?i:", "(?P<n>", "(?<m>", "|", "*", "+", "?", "*?", "+?", "??",
                                  "{2}", "{1,3}", "{2,}", "{0}", "{1000}", "{1001}", "{3,1}", "{", "}", "\\x41", "\\x{1F600}", "\\x{110000}", "\\Q.*\\E", "\\Q", "\\E", "\\", "\\pL",
                                  "\\p{Greek}", "\xc3\xa9", "\xe2\x82\xac", "\xf0\x9f\x98\x80", "\xff", "\xc3", "[\xc3\xa9-\xe2\x82\xac]", "\\n", "\\t", "\\1", "(?", "(?P<", "(?P<>", "[", "[a-",
                                  "[z-a]", "[[:nope:]]", "\\_", "\\-", "x{,3}", "((((", "))))", "a**", "(a*)*", "(a|b)*c", "(?i)straSSe"};
    static const char *tpiece[] = {"$1", "${1}", "$n", "${n}", "$$", "$", "${", "${1", "$99", "$0", "x", "-", "\xc3\xa9", "$m$1", "${nope}"};
    const size_t np = sizeof piece / sizeof *piece, nt = sizeof tpiece / sizeof *tpiece;
    long compiled = 0, refused = 0, matched = 0;
    for (long t = 0; t < trials; t++) {
        std::string pat;
        if (rng() % 5 == 0) {
            for (int k = 0, m = (int)(rng() % 24); k < m; k++) pat += (char)(rng() % 3 ? " ()[]{}|*+?\\^$.-:<>PQEidswbAz01,"[rng() % 33] : (char)rng());
        } else {
            for (int k = 0, m = 1 + (int)(rng() % 10); k < m; k++) pat += piece[rng() % np];
        }
        Re2Lite re;
        std::string err;
        if (!re.compile(pat, &err)) {
            refused++;
            continue;
        }
        compiled++;
        for (int q = 0; q < 3; q++) {
            std::string text;
            for (int k = 0, m = (int)(rng() % 40); k < m; k++) {
                const unsigned r = (unsigned)(rng() % 16);
                if (r < 9) text += "abcAB 019_-\n\tz"[rng() % 14];
                else if (r < 11) text += "\xc3\xa9";
                else if (r < 12) text += "\xe2\x82\xac";
                else if (r < 13) text += "\xf0\x9f\x98\x80";
                else if (r < 14) text += "straSSe";
                else text += (char)rng();  // (invalid UTF-8 now and then)
            }
            // (an exact-size heap copy: a read past the text is seen)
            char *exact = (char *)malloc(text.size() ? text.size() : 1);
            if (!text.empty()) memcpy(exact, text.data(), text.size());
            matched += re.search(exact, text.size());
            free(exact);
            std::string templ;
            for (int k = 0, m = (int)(rng() % 5); k < m; k++) templ += tpiece[rng() % nt];
            const std::string out = re.replace_all(text, templ);
            matched += out.size() & 1;
        }
    }
    printf("%ld compiled, %ld refused, %ld (matches + odd replacements)\n", compiled, refused, matched);
    return 0;
}
