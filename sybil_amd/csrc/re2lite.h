// re2lite.h -- the regular expressions of sybil's str filters (`-str-filter col:re:...`), for hosts that do
// not pre-evaluate them.
//
// The reference compiles the filter value with Go's regexp package (RE2 syntax) and tests every
// dictionary entry with MatchString -- an unanchored search (filter.go:213-236, filter.go:301-318).  Round
// 1 used std::regex (ECMAScript): different syntax ((?i), (?P<name>), \z, [[:alpha:]] inside sets), a
// backtracking matcher that is exponential where RE2 is linear, and recursion on the input that can
// overflow the stack on long dictionary strings.  This is a small RE2-style engine instead: the pattern is
// parsed into an NFA program and run as a Pike VM over the runes of the text -- O(text x program), no
// backtracking, no recursion on the input.
//
// Supported (Go regexp/syntax, https://pkg.go.dev/regexp/syntax): literals, . [set] [^set] ranges, escapes
// \d \D \s \S \w \W \b \B \A \z \n \t \r \f \v \a \xHH \x{H..} \Q..\E and escaped punctuation, POSIX classes
// [[:alpha:]] ..., groups (...) (?:...) (?P<name>...) (?<name>...), flags (?i) (?s) (?m) (?U) and their
// negations, scoped (?i:...), alternation, * + ? {n} {n,} {n,m} with optional lazy ?, ^ $.  Case folding
// under (?i) is ASCII only.  Unicode classes (\pL, \p{Greek}) are rejected with an error instead of being
// matched wrongly; so is a repeat count above 1000 (RE2's limit).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace sybl {

class Re2Lite {
public:
    // false + *err on a syntax error or an unsupported construct
    bool compile(const std::string &pattern, std::string *err);
    // regexp.MatchString: does the pattern match anywhere in `text`?
    bool search(const char *text, size_t n) const;
    bool search(const std::string &s) const { return search(s.data(), s.size()); }
    // regexp.ReplaceAllString (-str-replace, column_store_io.go:517-530): every non-overlapping leftmost-first match
    // replaced by `templ` with $1 / ${1} / $name / ${name} / $$ expanded as regexp.Expand does
    std::string replace_all(const std::string &text, const std::string &templ) const;
    int num_captures() const { return n_cap_; }

private:
    enum Op : uint8_t { kChar, kAny, kAnyNotNl, kSplit, kJmp, kMatch, kAssert, kSave };
    enum Assert : uint8_t { kBol, kEol, kBot, kEot, kWordB, kNotWordB, kBolM, kEolM };
    struct Range {
        int32_t lo, hi;
    };
    struct Inst {
        Op op;
        uint8_t arg;    // kAssert: which; kChar: 1 = negated set
        int32_t x, y;   // kSplit: both targets; kJmp: x; kChar: ranges [x, y) into ranges_
    };
    std::vector<Inst> prog_;
    std::vector<Range> ranges_;
    int n_cap_ = 0;                       // capture groups (group 0 = the whole match, not counted)
    std::vector<std::string> cap_names_;  // [group] name, "" = unnamed
    // leftmost-first match starting at or after rune `start`: cap[2g], cap[2g+1] = rune range of group g (-1: unset)
    bool match_from(const std::vector<int32_t> &runes, size_t start, std::vector<int> &cap) const;
    friend struct Re2Parser;
};

}  // namespace sybl
