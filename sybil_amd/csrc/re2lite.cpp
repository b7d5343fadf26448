// re2lite.cpp -- see re2lite.h.  Parser (Go regexp/syntax subset) -> NFA program -> Pike VM.
#include "re2lite.h"

#include <string.h>

#include <algorithm>
#include <memory>
#include <unordered_map>

namespace sybl {

namespace {

constexpr int kMaxRepeat = 1000;     // RE2's limit on {n,m}
constexpr size_t kMaxProg = 100000;  // instructions after expanding counted repetitions
constexpr int32_t kMaxRune = 0x10FFFF;

struct Node {
    enum Kind { kEmpty, kSet, kAny, kAnyNotNl, kCat, kAlt, kStar, kPlus, kQuest, kRepeat, kAssert, kCapture } kind = kEmpty;
    bool lazy = false;  // repetitions: prefer fewer (x*?); only submatch extraction can tell
    int cap = 0;        // kCapture: group number (1-based)
    std::vector<std::unique_ptr<Node>> kids;
    std::vector<std::pair<int32_t, int32_t>> ranges;  // kSet
    bool negated = false;                             // kSet
    int min = 0, max = 0;                             // kRepeat (max < 0: unbounded)
    int which = 0;                                    // kAssert
    bool quoted = false;                              // kCat from \Q..\E: a following quantifier binds to its last rune only
};
typedef std::unique_ptr<Node> NodeP;

bool is_word(int32_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }

// UTF-8 decoding the way Go ranges over a string: an invalid byte is U+FFFD and one byte wide
int32_t decode_rune(const unsigned char *s, size_t n, size_t *width) {
    const unsigned c = s[0];
    *width = 1;
    if (c < 0x80) return (int32_t)c;
    int need = c >= 0xF0 && c <= 0xF4 ? 3 : c >= 0xE0 ? 2 : c >= 0xC2 && c < 0xE0 ? 1 : 0;
    if (c >= 0xF5 || need == 0 || (size_t)need >= n) return 0xFFFD;
    int32_t r = need == 3 ? (c & 0x07) : need == 2 ? (c & 0x0F) : (c & 0x1F);
    for (int k = 1; k <= need; k++) {
        if ((s[k] & 0xC0) != 0x80) return 0xFFFD;
        r = (r << 6) | (s[k] & 0x3F);
    }
    if ((need == 2 && r < 0x800) || (need == 3 && r < 0x10000) || r > kMaxRune || (r >= 0xD800 && r <= 0xDFFF)) return 0xFFFD;
    *width = (size_t)need + 1;
    return r;
}

}  // namespace

struct Re2Parser {
    std::vector<int32_t> pat;  // runes of the pattern
    size_t at = 0;
    std::string err;
    bool fold = false, dot_nl = false, multi = false, ungreedy = false;
    int depth = 0;
    int n_cap = 0;
    std::vector<std::string> cap_names = {""};  // [0] = the whole match

    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }
    bool more() const { return at < pat.size(); }
    int32_t peek(size_t k = 0) const { return at + k < pat.size() ? pat[at + k] : -1; }

    static void add_range(Node *n, int32_t lo, int32_t hi) { n->ranges.emplace_back(lo, hi); }
    static void add_complement(Node *n, const std::vector<std::pair<int32_t, int32_t>> &rs) {
        std::vector<std::pair<int32_t, int32_t>> s = rs;
        std::sort(s.begin(), s.end());
        int32_t next = 0;
        for (auto &r : s) {
            if (r.first > next) n->ranges.emplace_back(next, r.first - 1);
            next = std::max(next, r.second + 1);
        }
        if (next <= kMaxRune) n->ranges.emplace_back(next, kMaxRune);
    }
    // Perl classes: \d \w \s and their negations (regexp/syntax: \s is [\t\n\f\r ])
    static bool perl_class(int32_t c, Node *n) {
        std::vector<std::pair<int32_t, int32_t>> rs;
        switch (c | 0x20) {
        case 'd': rs = {{'0', '9'}}; break;
        case 'w': rs = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}; break;
        case 's': rs = {{'\t', '\n'}, {'\f', '\r'}, {' ', ' '}}; break;
        default: return false;
        }
        if (c & 0x20) {
            for (auto &r : rs) n->ranges.push_back(r);
        } else {
            add_complement(n, rs);
        }
        return true;
    }
    static bool posix_class(const std::string &name, bool neg, Node *n) {
        std::vector<std::pair<int32_t, int32_t>> rs;
        if (name == "alnum") rs = {{'0', '9'}, {'A', 'Z'}, {'a', 'z'}};
        else if (name == "alpha") rs = {{'A', 'Z'}, {'a', 'z'}};
        else if (name == "ascii") rs = {{0, 0x7F}};
        else if (name == "blank") rs = {{'\t', '\t'}, {' ', ' '}};
        else if (name == "cntrl") rs = {{0, 0x1F}, {0x7F, 0x7F}};
        else if (name == "digit") rs = {{'0', '9'}};
        else if (name == "graph") rs = {{'!', '~'}};
        else if (name == "lower") rs = {{'a', 'z'}};
        else if (name == "print") rs = {{' ', '~'}};
        else if (name == "punct") rs = {{'!', '/'}, {':', '@'}, {'[', '`'}, {'{', '~'}};
        else if (name == "space") rs = {{'\t', '\r'}, {' ', ' '}};
        else if (name == "upper") rs = {{'A', 'Z'}};
        else if (name == "word") rs = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
        else if (name == "xdigit") rs = {{'0', '9'}, {'A', 'F'}, {'a', 'f'}};
        else return false;
        if (neg) add_complement(n, rs);
        else for (auto &r : rs) n->ranges.push_back(r);
        return true;
    }
    void fold_ranges(Node *n) const {
        if (!fold) return;
        const size_t k = n->ranges.size();
        for (size_t i = 0; i < k; i++) {
            const int32_t lo = n->ranges[i].first, hi = n->ranges[i].second;
            const int32_t ulo = std::max<int32_t>(lo, 'A'), uhi = std::min<int32_t>(hi, 'Z');
            if (ulo <= uhi) n->ranges.emplace_back(ulo + 32, uhi + 32);
            const int32_t llo = std::max<int32_t>(lo, 'a'), lhi = std::min<int32_t>(hi, 'z');
            if (llo <= lhi) n->ranges.emplace_back(llo - 32, lhi - 32);
        }
    }
    NodeP literal(int32_t c) const {
        NodeP n(new Node());
        n->kind = Node::kSet;
        add_range(n.get(), c, c);
        fold_ranges(n.get());
        return n;
    }
    static int hexval(int32_t c) { return c >= '0' && c <= '9' ? c - '0' : (c | 0x20) >= 'a' && (c | 0x20) <= 'f' ? (c | 0x20) - 'a' + 10 : -1; }

    // one escaped rune that stands for itself (after the backslash); false: not such an escape
    bool escape_rune(int32_t *out) {
        const int32_t c = peek();
        if (c < 0) return fail("trailing backslash at end of expression");
        switch (c) {
        case 'n': *out = '\n'; at++; return true;
        case 't': *out = '\t'; at++; return true;
        case 'r': *out = '\r'; at++; return true;
        case 'f': *out = '\f'; at++; return true;
        case 'v': *out = '\v'; at++; return true;
        case 'a': *out = 7; at++; return true;
        case 'x': {
            at++;
            int32_t v = 0;
            if (peek() == '{') {
                at++;
                int nd = 0;
                while (more() && peek() != '}') {
                    const int h = hexval(peek());
                    if (h < 0 || ++nd > 8) return fail("invalid escape sequence: \\x{");
                    v = v * 16 + h;
                    if (v > kMaxRune) return fail("invalid escape sequence: \\x{ beyond U+10FFFF");
                    at++;
                }
                if (!more() || nd == 0) return fail("invalid escape sequence: \\x{");
                at++;
            } else {
                const int a = hexval(peek()), b = hexval(peek(1));
                if (a < 0 || b < 0) return fail("invalid escape sequence: \\x");
                v = a * 16 + b;
                at += 2;
            }
            *out = v;
            return true;
        }
        default:
            if (c < 0x80 && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) {  // punctuation
                *out = c;
                at++;
                return true;
            }
            if (c >= '0' && c <= '7' && c != '0') return fail("backreferences are not supported by RE2");
            if (c == '0') {  // octal \0, \012
                int32_t v = 0;
                int nd = 0;
                while (nd < 3 && peek() >= '0' && peek() <= '7') {
                    v = v * 8 + (peek() - '0');
                    at++;
                    nd++;
                }
                *out = v;
                return true;
            }
            return false;
        }
    }

    NodeP parse_set() {  // after '['
        NodeP n(new Node());
        n->kind = Node::kSet;
        bool neg = false;
        if (peek() == '^') {
            neg = true;
            at++;
        }
        bool first = true;
        while (true) {
            if (!more()) return fail("missing closing ]"), nullptr;
            int32_t c = peek();
            if (c == ']' && !first) {
                at++;
                break;
            }
            first = false;
            if (c == '[' && peek(1) == ':') {
                size_t e = at + 2;
                std::string name;
                bool pneg = false;
                if (e < pat.size() && pat[e] == '^') {
                    pneg = true;
                    e++;
                }
                while (e < pat.size() && pat[e] != ':' && pat[e] < 0x80) name.push_back((char)pat[e++]);
                if (e + 1 < pat.size() && pat[e] == ':' && pat[e + 1] == ']') {
                    if (!posix_class(name, pneg, n.get())) return fail("invalid character class range: [:" + name + ":]"), nullptr;
                    at = e + 2;
                    continue;
                }
            }
            int32_t lo;
            if (c == '\\') {
                at++;
                const int32_t e = peek();
                if (e == 'd' || e == 'D' || e == 'w' || e == 'W' || e == 's' || e == 'S') {
                    perl_class(e, n.get());
                    at++;
                    continue;
                }
                if (e == 'p' || e == 'P') return fail("Unicode classes (\\p) are not supported"), nullptr;
                if (!escape_rune(&lo)) return fail("invalid escape sequence in character class"), nullptr;
            } else {
                lo = c;
                at++;
            }
            int32_t hi = lo;
            if (peek() == '-' && peek(1) != ']' && peek(1) >= 0) {
                at++;
                if (peek() == '\\') {
                    at++;
                    if (!escape_rune(&hi)) return fail("invalid escape sequence in character class"), nullptr;
                } else {
                    hi = peek();
                    at++;
                }
                if (hi < lo) return fail("invalid character class range"), nullptr;
            }
            add_range(n.get(), lo, hi);
        }
        fold_ranges(n.get());
        n->negated = neg;
        return n;
    }

    // group flags after "(?": returns 0 on error, 1 = flags only "(?i)", 2 = scoped group "(?i:" / "(?:" / named
    std::string last_name;  // of the named group parse_group_header just read
    // 3 = named capturing group
    int parse_group_header(bool *sfold, bool *sdot, bool *smulti) {
        if (peek() == 'P' && peek(1) == '<') at += 1;  // (?P<name>
        if (peek() == '<') {                           // (?<name>
            at++;
            size_t n = 0;
            last_name.clear();
            while (more() && peek() != '>') {
                if (!is_word(peek())) return fail("invalid named capture"), 0;
                last_name.push_back((char)peek());
                at++;
                n++;
            }
            if (!more() || n == 0) return fail("invalid named capture"), 0;
            at++;
            return 3;
        }
        bool f = fold, d = dot_nl, m = multi, neg = false, any = false;
        while (more()) {
            const int32_t c = peek();
            at++;
            switch (c) {
            case 'i': f = !neg; any = true; break;
            case 's': d = !neg; any = true; break;
            case 'm': m = !neg; any = true; break;
            case 'U': ungreedy = !neg; any = true; break;  // swaps greedy / lazy (only submatch extraction can tell)
            case '-':
                if (neg) return fail("invalid or unsupported Perl syntax"), 0;
                neg = true;
                any = false;
                break;
            case ':':
            case ')':
                if (neg && !any) return fail("invalid or unsupported Perl syntax"), 0;
                *sfold = f;
                *sdot = d;
                *smulti = m;
                return c == ':' ? 2 : 1;
            default: return fail("invalid or unsupported Perl syntax"), 0;
            }
        }
        return fail("missing closing )"), 0;
    }

    NodeP parse_atom() {
        const int32_t c = peek();
        NodeP n(new Node());
        switch (c) {
        case '(': {
            at++;
            if (++depth > 200) return fail("expression nests too deeply"), nullptr;
            const bool of = fold, od = dot_nl, om = multi, ou = ungreedy;
            int cap = 0;  // > 0: a capturing group
            if (peek() == '?') {
                at++;
                bool f, d, m;
                f = fold, d = dot_nl, m = multi;
                const bool u_before = ungreedy;
                const int kind = parse_group_header(&f, &d, &m);
                if (!kind) return nullptr;
                fold = f, dot_nl = d, multi = m;
                if (kind == 1) {  // (?i): flags for the rest of the enclosing group
                    depth--;
                    n->kind = Node::kEmpty;
                    return n;
                }
                if (kind == 3) {
                    ungreedy = u_before;
                    cap = ++n_cap;
                    cap_names.push_back(last_name);
                }
            } else {
                cap = ++n_cap;
                cap_names.push_back("");
            }
            NodeP inner = parse_alt();
            if (!inner) return nullptr;
            if (peek() != ')') return fail("missing closing )"), nullptr;
            at++;
            depth--;
            fold = of, dot_nl = od, multi = om, ungreedy = ou;
            if (cap > 0) {
                n->kind = Node::kCapture;
                n->cap = cap;
                n->kids.push_back(std::move(inner));
                return n;
            }
            return inner;
        }
        case '[': at++; return parse_set();
        case '.':
            at++;
            n->kind = dot_nl ? Node::kAny : Node::kAnyNotNl;
            return n;
        case '^':
            at++;
            n->kind = Node::kAssert;
            n->which = multi ? 6 : 0;  // kBolM : kBol
            return n;
        case '$':
            at++;
            n->kind = Node::kAssert;
            n->which = multi ? 7 : 1;  // kEolM : kEol
            return n;
        case '*':
        case '+':
        case '?': return fail("missing argument to repetition operator"), nullptr;
        case '\\': {
            at++;
            const int32_t e = peek();
            if (e == 'd' || e == 'D' || e == 'w' || e == 'W' || e == 's' || e == 'S') {
                at++;
                n->kind = Node::kSet;
                perl_class(e, n.get());
                return n;
            }
            if (e == 'b' || e == 'B' || e == 'A' || e == 'z') {
                at++;
                n->kind = Node::kAssert;
                n->which = e == 'b' ? 4 : e == 'B' ? 5 : e == 'A' ? 2 : 3;
                return n;
            }
            if (e == 'Q') {  // literal text up to \E
                at++;
                n->kind = Node::kCat;
                n->quoted = true;
                while (more() && !(peek() == '\\' && peek(1) == 'E')) {
                    n->kids.push_back(literal(peek()));
                    at++;
                }
                if (more()) at += 2;
                return n;
            }
            if (e == 'p' || e == 'P') return fail("Unicode classes (\\p) are not supported"), nullptr;
            if (e == 'C') return fail("\\C is not supported"), nullptr;
            int32_t r;
            if (!escape_rune(&r)) return fail("invalid escape sequence"), nullptr;
            return literal(r);
        }
        default: at++; return literal(c);
        }
    }

    // {n}, {n,}, {n,m} at `at` (pointing at '{'); false: not a repetition (the brace is a literal)
    bool parse_counts(int *mn, int *mx) {
        size_t p = at + 1;
        auto num = [&](int *v) {
            size_t s = p;
            long x = 0;
            // (every digit: Go reads the whole number and rejects a count above 1000 -- `x{1234567}` is an error, not a literal)
            while (p < pat.size() && pat[p] >= '0' && pat[p] <= '9') x = std::min<long>(x * 10 + (pat[p++] - '0'), 1000000);
            *v = (int)x;
            return p > s;
        };
        if (!num(mn)) return false;
        *mx = *mn;
        if (p < pat.size() && pat[p] == ',') {
            p++;
            if (p < pat.size() && pat[p] == '}') *mx = -1;
            else if (!num(mx)) return false;
        }
        if (p >= pat.size() || pat[p] != '}') return false;
        at = p + 1;
        return true;
    }

    NodeP parse_repeat() {
        NodeP a = parse_atom();
        if (!a) return nullptr;
        bool repeated = false;
        while (more()) {
            const int32_t c = peek();
            NodeP r(new Node());
            if (c == '*' || c == '+' || c == '?') {
                at++;
                r->kind = c == '*' ? Node::kStar : c == '+' ? Node::kPlus : Node::kQuest;
            } else if (c == '{') {
                int mn, mx;
                if (!parse_counts(&mn, &mx)) break;
                if (mn > kMaxRepeat || mx > kMaxRepeat || (mx >= 0 && mx < mn)) return fail("invalid repeat count"), nullptr;
                r->kind = Node::kRepeat;
                r->min = mn;
                r->max = mx;
            } else {
                break;
            }
            if (repeated) return fail("invalid nested repetition operator"), nullptr;
            r->lazy = ungreedy;
            if (peek() == '?') {  // x*?: prefer fewer (under (?U): prefer more)
                at++;
                r->lazy = !ungreedy;
            }
            repeated = true;
            if (a->kind == Node::kCat && a->quoted && a->kids.size() > 1) {
                // \Qab\E+ is a(b+) in Go: the quoted runes are ordinary literals of the enclosing concatenation
                r->kids.push_back(std::move(a->kids.back()));
                a->kids.back() = std::move(r);
                continue;
            }
            r->kids.push_back(std::move(a));
            a = std::move(r);
        }
        return a;
    }

    NodeP parse_cat() {
        NodeP n(new Node());
        n->kind = Node::kCat;
        while (more() && peek() != '|' && peek() != ')') {
            NodeP k = parse_repeat();
            if (!k) return nullptr;
            n->kids.push_back(std::move(k));
        }
        return n;
    }

    NodeP parse_alt() {
        NodeP first = parse_cat();
        if (!first) return nullptr;
        if (peek() != '|') return first;
        NodeP n(new Node());
        n->kind = Node::kAlt;
        n->kids.push_back(std::move(first));
        // flags set by (?i) inside one branch stay in force for the following branches (Go keeps them to the group's end)
        while (peek() == '|') {
            at++;
            NodeP k = parse_cat();
            if (!k) return nullptr;
            n->kids.push_back(std::move(k));
        }
        return n;
    }

    // ---- code generation
    Re2Lite *re = nullptr;
    std::unordered_map<const Node *, std::pair<int32_t, int32_t>> set_ranges;
    bool emit(const Node *n) {
        auto &P = re->prog_;
        if (P.size() > kMaxProg) return fail("expression too large");
        auto add = [&](Re2Lite::Op op, uint8_t arg, int32_t x, int32_t y) {
            P.push_back(Re2Lite::Inst{op, arg, x, y});
            return (int32_t)P.size() - 1;
        };
        switch (n->kind) {
        case Node::kEmpty: return true;
        case Node::kSet: {
            // (one range table per class, shared by the copies a counted repetition makes of it: [..]{1000} inside
            // another {1000} would otherwise hold a million copies of the table before kMaxProg trips)
            auto it = set_ranges.find(n);
            if (it == set_ranges.end()) {
                const int32_t r0 = (int32_t)re->ranges_.size();
                for (auto &r : n->ranges) re->ranges_.push_back(Re2Lite::Range{r.first, r.second});
                it = set_ranges.emplace(n, std::make_pair(r0, (int32_t)re->ranges_.size())).first;
            }
            add(Re2Lite::kChar, n->negated ? 1 : 0, it->second.first, it->second.second);
            return true;
        }
        case Node::kAny: add(Re2Lite::kAny, 0, 0, 0); return true;
        case Node::kAnyNotNl: add(Re2Lite::kAnyNotNl, 0, 0, 0); return true;
        case Node::kAssert: add(Re2Lite::kAssert, (uint8_t)n->which, 0, 0); return true;
        case Node::kCat:
            for (auto &k : n->kids)
                if (!emit(k.get())) return false;
            return true;
        case Node::kAlt: {
            std::vector<int32_t> jumps;
            for (size_t i = 0; i < n->kids.size(); i++) {
                int32_t split = -1;
                if (i + 1 < n->kids.size()) split = add(Re2Lite::kSplit, 0, 0, 0);
                if (split >= 0) P[(size_t)split].x = (int32_t)P.size();
                if (!emit(n->kids[i].get())) return false;
                if (i + 1 < n->kids.size()) {
                    jumps.push_back(add(Re2Lite::kJmp, 0, 0, 0));
                    P[(size_t)split].y = (int32_t)P.size();
                }
            }
            for (int32_t j : jumps) P[(size_t)j].x = (int32_t)P.size();
            return true;
        }
        // kSplit tries x before y (thread priority): greedy repetitions prefer the body, lazy ones the way out
        case Node::kCapture: {
            add(Re2Lite::kSave, 0, 2 * n->cap, 0);
            if (!emit(n->kids[0].get())) return false;
            add(Re2Lite::kSave, 0, 2 * n->cap + 1, 0);
            return true;
        }
        case Node::kStar: {
            const int32_t split = add(Re2Lite::kSplit, 0, 0, 0);
            const int32_t body = (int32_t)P.size();
            if (!emit(n->kids[0].get())) return false;
            add(Re2Lite::kJmp, 0, split, 0);
            const int32_t out = (int32_t)P.size();
            P[(size_t)split].x = n->lazy ? out : body;
            P[(size_t)split].y = n->lazy ? body : out;
            return true;
        }
        case Node::kPlus: {
            const int32_t start = (int32_t)P.size();
            if (!emit(n->kids[0].get())) return false;
            const int32_t split = add(Re2Lite::kSplit, 0, 0, 0);
            const int32_t out = (int32_t)P.size();
            P[(size_t)split].x = n->lazy ? out : start;
            P[(size_t)split].y = n->lazy ? start : out;
            return true;
        }
        case Node::kQuest: {
            const int32_t split = add(Re2Lite::kSplit, 0, 0, 0);
            const int32_t body = (int32_t)P.size();
            if (!emit(n->kids[0].get())) return false;
            const int32_t out = (int32_t)P.size();
            P[(size_t)split].x = n->lazy ? out : body;
            P[(size_t)split].y = n->lazy ? body : out;
            return true;
        }
        case Node::kRepeat: {
            for (int i = 0; i < n->min; i++)
                if (!emit(n->kids[0].get())) return false;
            if (n->max < 0) {
                Node star;
                star.kind = Node::kStar;
                star.lazy = n->lazy;
                // (borrow the child for the duration of the call)
                star.kids.emplace_back(const_cast<Node *>(n->kids[0].get()));
                const bool ok = emit(&star);
                star.kids[0].release();
                return ok;
            }
            // x{n,m}: n copies, then m - n nested optional copies  (x(x(x)?)?)?
            std::vector<int32_t> splits;
            for (int i = n->min; i < n->max; i++) {
                splits.push_back(add(Re2Lite::kSplit, 0, 0, 0));
                P[(size_t)splits.back()].x = (int32_t)P.size();
                if (!emit(n->kids[0].get())) return false;
            }
            for (int32_t s : splits) {
                P[(size_t)s].y = (int32_t)P.size();
                if (n->lazy) std::swap(P[(size_t)s].x, P[(size_t)s].y);
            }
            return true;
        }
        }
        return true;
    }
};

bool Re2Lite::compile(const std::string &pattern, std::string *err) {
    prog_.clear();
    ranges_.clear();
    Re2Parser ps;
    ps.re = this;
    const unsigned char *s = (const unsigned char *)pattern.data();
    for (size_t i = 0; i < pattern.size();) {
        size_t w;
        const int32_t r = decode_rune(s + i, pattern.size() - i, &w);
        if (r == 0xFFFD && !(w == 3 && s[i] == 0xEF && s[i + 1] == 0xBF && s[i + 2] == 0xBD)) {
            if (err) *err = "invalid UTF-8";
            return false;
        }
        ps.pat.push_back(r);
        i += w;
    }
    NodeP root = ps.parse_alt();
    if (root && ps.more()) {
        ps.fail(ps.peek() == ')' ? "unexpected )" : "unexpected character");
        root.reset();
    }
    if (!root || !ps.emit(root.get()) || !ps.err.empty()) {
        if (err) *err = ps.err.empty() ? "invalid regular expression" : ps.err;
        prog_.clear();
        return false;
    }
    prog_.push_back(Inst{kMatch, 0, 0, 0});
    n_cap_ = ps.n_cap;
    cap_names_ = ps.cap_names;
    return true;
}

bool Re2Lite::search(const char *text, size_t n) const {
    if (prog_.empty()) return false;
    std::vector<int32_t> runes;
    runes.reserve(n);
    for (size_t i = 0; i < n;) {
        size_t w;
        runes.push_back(decode_rune((const unsigned char *)text + i, n - i, &w));
        i += w;
    }
    const size_t len = runes.size(), np = prog_.size();
    std::vector<int32_t> clist, nlist, stack;
    std::vector<uint32_t> mark(np, 0);
    uint32_t gen = 0;
    clist.reserve(np);
    nlist.reserve(np);
    // follows the empty transitions from pc at text position i into `list`; true if kMatch was reached
    auto add_thread = [&](std::vector<int32_t> &list, int32_t pc0, size_t i) -> bool {
        stack.clear();
        stack.push_back(pc0);
        while (!stack.empty()) {
            const int32_t pc = stack.back();
            stack.pop_back();
            if (mark[(size_t)pc] == gen) continue;
            mark[(size_t)pc] = gen;
            const Inst &in = prog_[(size_t)pc];
            switch (in.op) {
            case kJmp: stack.push_back(in.x); break;
            case kSave: stack.push_back(pc + 1); break;  // (submatches: match_from)
            case kSplit:
                stack.push_back(in.y);
                stack.push_back(in.x);
                break;
            case kAssert: {
                const int32_t prev = i > 0 ? runes[i - 1] : -1, next = i < len ? runes[i] : -1;
                bool ok = false;
                switch (in.arg) {
                case kBol:
                case kBot: ok = i == 0; break;
                case kEol:
                case kEot: ok = i == len; break;
                case kBolM: ok = i == 0 || prev == '\n'; break;
                case kEolM: ok = i == len || next == '\n'; break;
                case kWordB: ok = (prev >= 0 && is_word(prev)) != (next >= 0 && is_word(next)); break;
                case kNotWordB: ok = (prev >= 0 && is_word(prev)) == (next >= 0 && is_word(next)); break;
                }
                if (ok) stack.push_back(pc + 1);
                break;
            }
            case kMatch: return true;
            default: list.push_back(pc); break;
            }
        }
        return false;
    };
    for (size_t i = 0; i <= len; i++) {
        // an unanchored search starts a new thread at every position (clist keeps the threads that
        // consumed rune i - 1; they were added under generation `gen`)
        if (i == 0) gen++;
        if (add_thread(clist, 0, i)) return true;
        if (i == len) break;
        const int32_t c = runes[i];
        gen++;
        nlist.clear();
        for (int32_t pc : clist) {
            const Inst &in = prog_[(size_t)pc];
            bool hit = false;
            switch (in.op) {
            case kAny: hit = true; break;
            case kAnyNotNl: hit = c != '\n'; break;
            case kChar: {
                bool inset = false;
                for (int32_t r = in.x; r < in.y; r++)
                    if (c >= ranges_[(size_t)r].lo && c <= ranges_[(size_t)r].hi) {
                        inset = true;
                        break;
                    }
                hit = inset != (in.arg != 0);
                break;
            }
            default: break;
            }
            if (hit && add_thread(nlist, pc + 1, i + 1)) return true;
        }
        clist.swap(nlist);
    }
    return false;
}

// Pike VM with submatch tracking, leftmost-first (Go's default, non-POSIX semantics): threads are kept in priority
// order, a thread that reaches kMatch cuts off every thread of lower priority, and new start positions are only tried
// while nothing has matched.
bool Re2Lite::match_from(const std::vector<int32_t> &runes, size_t start, std::vector<int> &cap) const {
    if (prog_.empty()) return false;
    const size_t len = runes.size(), np = prog_.size(), nslot = 2 * (size_t)(n_cap_ + 1);
    struct Thread {
        int32_t pc;
        std::vector<int> cap;
    };
    std::vector<Thread> clist, nlist;
    std::vector<uint32_t> mark(np, 0);
    uint32_t gen = 0;
    bool matched = false;
    // follows the empty transitions in priority order (recursion depth is bounded by the program size)
    struct Frame {
        int32_t pc;
        std::vector<int> cap;
    };
    auto add_thread = [&](std::vector<Thread> &list, int32_t pc0, size_t i, const std::vector<int> &cap0) {
        std::vector<Frame> stack;
        stack.push_back(Frame{pc0, cap0});
        while (!stack.empty()) {
            Frame f = std::move(stack.back());
            stack.pop_back();
            if (mark[(size_t)f.pc] == gen) continue;
            mark[(size_t)f.pc] = gen;
            const Inst &in = prog_[(size_t)f.pc];
            switch (in.op) {
            case kJmp: stack.push_back(Frame{in.x, std::move(f.cap)}); break;
            case kSplit:
                stack.push_back(Frame{in.y, f.cap});
                stack.push_back(Frame{in.x, std::move(f.cap)});
                break;
            case kSave:
                if ((size_t)in.x < nslot) f.cap[(size_t)in.x] = (int)i;
                stack.push_back(Frame{f.pc + 1, std::move(f.cap)});
                break;
            case kAssert: {
                const int32_t prev = i > 0 ? runes[i - 1] : -1, next = i < len ? runes[i] : -1;
                bool ok = false;
                switch (in.arg) {
                case kBol:
                case kBot: ok = i == 0; break;
                case kEol:
                case kEot: ok = i == len; break;
                case kBolM: ok = i == 0 || prev == '\n'; break;
                case kEolM: ok = i == len || next == '\n'; break;
                case kWordB: ok = (prev >= 0 && is_word(prev)) != (next >= 0 && is_word(next)); break;
                case kNotWordB: ok = (prev >= 0 && is_word(prev)) == (next >= 0 && is_word(next)); break;
                }
                if (ok) stack.push_back(Frame{f.pc + 1, std::move(f.cap)});
                break;
            }
            default: list.push_back(Thread{f.pc, std::move(f.cap)}); break;  // kChar / kAny / kAnyNotNl / kMatch
            }
        }
    };
    std::vector<int> fresh(nslot, -1);
    for (size_t i = start; i <= len; i++) {
        if (i == start) gen++;
        if (!matched) {  // a later start has the lowest priority, and none once something matched
            fresh[0] = (int)i;
            add_thread(clist, 0, i, fresh);
        }
        if (clist.empty()) {
            if (matched) break;
            gen++;     // (the start thread died on an assertion: the next position gets its own, under fresh marks)
            continue;
        }
        gen++;
        nlist.clear();
        const int32_t c = i < len ? runes[i] : -1;
        for (size_t t = 0; t < clist.size(); t++) {
            const Inst &in = prog_[(size_t)clist[t].pc];
            if (in.op == kMatch) {
                cap = clist[t].cap;
                cap[1] = (int)i;
                matched = true;
                break;  // lower-priority threads are cut off
            }
            if (c < 0) continue;
            bool hit = false;
            switch (in.op) {
            case kAny: hit = true; break;
            case kAnyNotNl: hit = c != '\n'; break;
            case kChar: {
                bool inset = false;
                for (int32_t r = in.x; r < in.y; r++)
                    if (c >= ranges_[(size_t)r].lo && c <= ranges_[(size_t)r].hi) {
                        inset = true;
                        break;
                    }
                hit = inset != (in.arg != 0);
                break;
            }
            default: break;
            }
            if (hit) add_thread(nlist, clist[t].pc + 1, i + 1, clist[t].cap);
        }
        clist.swap(nlist);
    }
    return matched;
}

std::string Re2Lite::replace_all(const std::string &text, const std::string &templ) const {
    // runes of the text with their byte offsets (Go ranges over the string the same way)
    std::vector<int32_t> runes;
    std::vector<size_t> off;
    for (size_t i = 0; i < text.size();) {
        size_t w;
        runes.push_back(decode_rune((const unsigned char *)text.data() + i, text.size() - i, &w));
        off.push_back(i);
        i += w;
    }
    off.push_back(text.size());
    // regexp.Expand: $name / ${name} / $$; a name is the longest run of letters, digits and _; all digits = a group number
    auto expand = [&](const std::vector<int> &cap, std::string &out) {
        const std::string &t = templ;
        for (size_t i = 0; i < t.size();) {
            if (t[i] != '$' || i + 1 >= t.size()) {
                out.push_back(t[i++]);
                continue;
            }
            if (t[i + 1] == '$') {
                out.push_back('$');
                i += 2;
                continue;
            }
            size_t j = i + 1;
            const bool brace = t[j] == '{';
            if (brace) j++;
            const size_t n0 = j;
            while (j < t.size() && is_word((unsigned char)t[j])) j++;
            if (j == n0 || (brace && (j >= t.size() || t[j] != '}'))) {  // malformed: the $ is literal text
                out.push_back(t[i++]);
                continue;
            }
            const std::string name = t.substr(n0, j - n0);
            if (brace) j++;
            int g = -1;
            if (name.find_first_not_of("0123456789") == std::string::npos) {
                // (Go: a number with a leading zero is not a group number -- and no group has such a name: empty)
                g = name.size() <= 6 && !(name.size() > 1 && name[0] == '0') ? atoi(name.c_str()) : -1;
            } else {
                for (size_t k = 1; k < cap_names_.size(); k++)
                    if (cap_names_[k] == name) g = (int)k;
            }
            if (g >= 0 && g <= n_cap_ && cap[2 * (size_t)g] >= 0 && cap[2 * (size_t)g + 1] >= 0)
                out.append(text, off[(size_t)cap[2 * (size_t)g]], off[(size_t)cap[2 * (size_t)g + 1]] - off[(size_t)cap[2 * (size_t)g]]);
            i = j;
        }
    };
    std::string out;
    size_t last_end = 0, pos = 0;  // rune indices
    std::vector<int> cap;
    while (pos <= runes.size()) {
        if (!match_from(runes, pos, cap)) break;
        const size_t m0 = (size_t)cap[0], m1 = (size_t)cap[1];
        out.append(text, off[last_end], off[m0] - off[last_end]);
        // no replacement for an empty match right behind another match
        if (m1 > last_end || m0 == 0) expand(cap, out);
        last_end = m1;
        if (pos + 1 > m1) pos += 1;  // always advance at least one rune
        else pos = m1;
    }
    out.append(text, off[last_end], text.size() - off[last_end]);
    return out;
}

}  // namespace sybl
