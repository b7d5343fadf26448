"""The library's regular-expression engine for re / nre str filters (csrc/re2lite.cpp): Go regexp (RE2) syntax,
unanchored search like regexp.MatchString (filter.go:213-236).  Checked against Python's `re` on the syntax the
two share, and on hand-derived cases for what is specific to Go's syntax.  No GPU needed."""
import re

import pytest

from sybil_amd import _native as N


def match(pattern, text):
    t = text.encode("utf-8")
    return N.lib().sybl_debug_regex_match(pattern.encode("utf-8"), t, len(t))


CORPUS = ["", "a", "ab", "abc", "aaa", "abab", "2", "20", "21", "29", "12", "102", "x2", "hello world", "Hello", "HELLO",
          "foo.bar", "foo_bar", "foo-bar", "a\nb", "line1\nline2", "tab\there", "  padded  ", "user@example.com",
          "10.0.0.1", "192.168.1.254", "2024-01-15", "abc123", "123abc", "___", "a+b", "a*b", "(x)", "[y]", "{z}", "a|b",
          "café", "é", "naïve", "中文", "edge", "gecko", "desktop", "tablet", "aXb", "a.b", "ab" * 40]

SHARED = [r"^2", r"2$", r"^2\d$", r"a", r"^$", r"ab*", r"ab+c?", r"(ab)+", r"(?:ab){2}", r"a{2,3}", r"a{2,}", r"^a{0,1}$",
          r"a|b", r"^(foo|bar)", r"[abc]", r"[^abc]", r"[a-c0-2]+", r"^[^a-z]*$", r"\d+", r"\D+", r"\w+@\w+\.\w+", r"\s", r"\S+\s\S+",
          r"^\s+\S", r"\bfoo\b", r"\Bo", r"foo\.bar", r"foo.bar", r"a\+b", r"a\*b", r"\(x\)", r"\[y\]", r"a\|b", r".", r"^.$",
          r"^..$", r"a.b", r"(?i)hello", r"(?i)^HELLO$", r"(?i:h)ello", r"(?s)a.b", r"(?m)^line2$", r"(?m)^b",
          r"^(\d{1,3}\.){3}\d{1,3}$", r"^\d{4}-\d{2}-\d{2}$", r"(a|ab)(c|bcd)?", r"(a*)*b", r"(a+)+$", r"(a|a)*c", r"x*", r"(?:)",
          r"[a-]", r"[]a]", r"[\d_]+$", r"[^\W\d]", r"caf.", r"^.{4}$", r"\x61", r"\t", r"[\t ]+$", r"e$|^g", r"(?P<n>ab)c",
          r"a??b", r"a*?b", r"a+?", r"a{2}?", r"\A2", r"^$|^a$", r"[[]", r"[a\]]", r"\.\d+\."]


@pytest.mark.parametrize("pattern", SHARED)
def test_agrees_with_python_re(pattern):
    rx = re.compile(pattern, re.ASCII)   # Go: \w \d \s \b are ASCII classes
    for text in CORPUS:
        assert match(pattern, text) == (1 if rx.search(text) else 0), (pattern, text)


def test_go_specific_syntax():
    # \z (end of text; Python spells it \Z), $ without (?m) is end of TEXT (a trailing newline does not count)
    assert match(r"b\z", "ab") == 1 and match(r"a\z", "ab") == 0
    assert match(r"a$", "a\n") == 0 and match(r"(?m)a$", "a\nb") == 1
    # POSIX classes inside sets
    assert match(r"^[[:alpha:]]+$", "Hello") == 1 and match(r"^[[:alpha:]]+$", "abc123") == 0
    assert match(r"[[:digit:][:space:]]", "x y") == 1 and match(r"^[[:^digit:]]+$", "abc") == 1 and match(r"^[[:^digit:]]+$", "ab1") == 0
    assert match(r"[[:word:]]", "_") == 1 and match(r"[[:punct:]]", "a-b") == 1 and match(r"[[:upper:]]", "abc") == 0
    # \s is [\t\n\f\r ] -- no vertical tab (unlike Perl / Python)
    assert match(r"\s", "\v") == 0 and match(r"[[:space:]]", "\v") == 1
    # flags: (?U) swaps greedy / lazy (immaterial for a yes / no match); (?-i) switches folding off again
    assert match(r"(?U)a+b", "aab") == 1 and match(r"(?i)a(?-i)b", "AB") == 0 and match(r"(?i)a(?-i)b", "Ab") == 1
    # a flag group in mid-pattern applies from there on (Python would apply it to the whole pattern)
    assert match(r"h(?i)ELLO", "hello") == 1 and match(r"h(?i)ELLO", "Hello") == 0
    assert match(r"\x{e9}", "caf\u00e9") == 1 and match(r"\x{4e2d}", "\u4e2d\u6587") == 1 and match(r"\x{e9}", "cafe") == 0
    # named groups, both spellings; \Q...\E quotes
    assert match(r"(?P<x>a)(?<y>b)", "ab") == 1 and match(r"\Qa.b\E", "a.b") == 1 and match(r"\Qa.b\E", "axb") == 0
    # a quantifier behind \Q..\E binds to the LAST quoted rune only (the quoted runes are ordinary literals)
    assert match(r"^\Qab\E+$", "abbb") == 1 and match(r"^\Qab\E+$", "abab") == 0 and match(r"^\Qa.\E{2}$", "a..") == 1
    # a brace that is no repetition is a literal
    assert match(r"a{", "a{") == 1 and match(r"a{x}", "a{x}") == 1 and match(r"{z}", "{z}") == 1
    # . matches a whole rune, not a byte
    assert match(r"^.$", "é") == 1 and match(r"^..$", "中文") == 1
    # flags of one alternation branch stay in force to the end of the group
    assert match(r"(?:a(?i)b|c)", "C") == 1


@pytest.mark.parametrize("pattern", [r"a**", r"a++", r"*a", r"(ab", r"ab)", r"[a", r"a{1001}", r"a{1234567}", r"a{2,12345678}", r"a{3,2}", r"\pL", r"[\p{Greek}]",
                                     r"\1", r"(?<=a)b", r"(?=a)", r"\8", r"\C", r"(?z)", "\\"])
def test_rejects_what_go_rejects_or_what_is_not_supported(pattern):
    assert match(pattern, "abc") == -1
    assert N.lib().sybl_last_error()


def test_linear_time_on_pathological_patterns():
    # exponential for a backtracking matcher (round 1 used std::regex), linear for a Thompson NFA
    import time
    t0 = time.perf_counter()
    assert match(r"(a*)*b", "a" * 5000) == 0
    assert match(r"(a|aa)+$", "a" * 5000 + "b") == 0
    assert match(r"(x+x+)+y", "x" * 3000) == 0
    assert time.perf_counter() - t0 < 10
    # long dictionary strings: no recursion on the input
    assert match(r"^(ab)*$", "ab" * 200_000) == 1


# regexp.ReplaceAllString / Expand known answers (Go's regexp documentation and package tests; "$1W" names group "1W",
# which does not exist; an empty match right behind another match gets no replacement)
REPLACE_KATS = [
    ("a(x*)b", "-ab-axxb-", "T", "-T-T-"), ("a(x*)b", "-ab-axxb-", "$1", "--xx-"), ("a(x*)b", "-ab-axxb-", "$1W", "---"),
    ("a(x*)b", "-ab-axxb-", "${1}W", "-W-xxW-"), ("a+", "baaab", "x", "bxb"), ("a*", "baaac", "x", "xbxcx"), ("", "abc", "-", "-a-b-c-"),
    (r"(?P<first>\w+) (?P<last>\w+)", "john smith", "$last, $first", "smith, john"), (r"(\d+)\.(\d+)", "v1.22 and 3.4", "${2}_$1", "v22_1 and 4_3"),
    (r"(\d+)\.(\d+)", "v1.22 and 3.4", "$2_$1", "v1 and 3"), (r"^www\.", "www.example.www.com", "", "example.www.com"), ("b*?", "abb", "x", "xaxbxbx"),
    ("a.*?c", "abcabc", "X", "XX"), ("(?U)a.*c", "abcabc", "X", "XX"), ("a.*c", "abcabc", "X", "X"), (r"\$", "cost $5", "$$", "cost $5"),
    ("(a)|(b)", "ab", "[$1|$2]", "[a|][|b]"), ("x*", "\u00e9x", "-", "-\u00e9-"), ("[aeiou]", "education", "$0$0", "eeduucaatiioon"),
    ("(?i)HOST", "host1.Host2", "<$0>", "<host>1.<Host>2"), ("[0-9]+$", "user123", "", "user"), ("$", "ab", "!", "ab!"),
    ("(a)", "a", "[$01]", "[]"), ("(a)", "a", "[${01}]", "[]"), ("(a)", "a", "[$0][$1]", "[a][a]"),   # a leading zero is no group number
]


@pytest.mark.parametrize("pattern,text,templ,want", REPLACE_KATS)
def test_replace_all_known_answers(pattern, text, templ, want):
    from sybil_amd import _native as N
    got = N.lib().sybl_debug_regex_replace(pattern.encode(), text.encode(), templ.encode())
    assert got is not None and got.decode() == want


def test_replace_all_agrees_with_python_where_the_semantics_coincide():
    """Patterns that cannot match the empty string, templates with numbered groups: Perl-style leftmost-first
    backtracking (Python's re) and Go's leftmost-first RE2 agree."""
    import random
    import re as pyre
    from sybil_amd import _native as N
    rng = random.Random(5)
    pats = [r"(\d+)", r"([a-c]+)(\d)", r"user(\d\d?)", r"\.(com|org)$", r"^(\w)\w*", r"(a|ab)(c|bcd)", r"x+?y", r"[^a-z]+", r"(\w+)@(\w+)\.com"]
    for pat in pats:
        for _ in range(200):
            text = "".join(rng.choice("abcxy019.@ user_com org") for _ in range(rng.randint(0, 24)))
            for templ, py in (("<$1>", r"<\1>"), ("", ""), ("[$0]", r"[\g<0>]")):
                if "$1" in templ and "(" not in pat:
                    continue
                got = N.lib().sybl_debug_regex_replace(pat.encode(), text.encode(), templ.encode()).decode()
                assert got == pyre.sub(pat, py, text), (pat, text, templ)



def _random_pattern(rng, depth=0):
    """A random pattern in the syntax Go's regexp and Python's re share (ASCII classes; no look-around, no backrefs)."""
    atoms = ["a", "b", "c", "2", ".", r"\d", r"\w", r"\s", "[ab]", "[^a]", "[a-c2]", r"\.", "x"]
    parts = []
    for _ in range(int(rng.integers(1, 4))):
        r = rng.random()
        if r < 0.2 and depth < 2:
            inner = _random_pattern(rng, depth + 1)
            a = ("(%s)" if rng.random() < 0.5 else "(?:%s)") % inner
        else:
            a = atoms[int(rng.integers(0, len(atoms)))]
        r = rng.random()
        if r < 0.15:
            a += "*"
        elif r < 0.3:
            a += "+"
        elif r < 0.4:
            a += "?"
        elif r < 0.48:
            a += "{%d,%d}" % (int(rng.integers(0, 2)), int(rng.integers(2, 4)))
        elif r < 0.53:
            a += "*?"
        parts.append(a)
    p = "".join(parts)
    if depth == 0:
        if rng.random() < 0.2:
            p = "^" + p
        if rng.random() < 0.2:
            p = p + "$"
    if rng.random() < 0.25 and depth < 2:
        p = p + "|" + _random_pattern(rng, depth + 1)
    return p


def test_random_patterns_agree_with_python_re():
    """Differential check of MatchString (unanchored search, a boolean) on 1500 random patterns x 40 random texts."""
    import numpy as np
    rng = np.random.default_rng(2024)
    alphabet = "abc2x. \t"
    texts = ["".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 12)))) for _ in range(40)]
    checked = 0
    for _ in range(1500):
        pat = _random_pattern(rng)
        try:
            rx = re.compile(pat, re.ASCII)
        except re.error:
            continue
        for text in texts:
            got = match(pat, text)
            assert got == (1 if rx.search(text) else 0), (pat, text, got)
            checked += 1
    assert checked > 40_000


def test_random_replacements_agree_with_python_re_sub():
    """regexp.ReplaceAllString (what -str-replace applies to every dictionary string, column_store_io.go:517-530) against
    re.sub on random patterns that cannot match the empty string (the two differ on empty matches next to a match) --
    both pick the leftmost match and, among those, the first alternative."""
    import numpy as np
    rng = np.random.default_rng(4048)
    alphabet = "abc2x. "
    texts = ["".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 16)))) for _ in range(30)]
    checked = 0
    for _ in range(800):
        pat = "abc"[int(rng.integers(0, 3))] + _random_pattern(rng, depth=1)   # a mandatory first character: never empty
        if "|" in pat:
            pat = "(?:%s)" % pat.replace("|", "|x")                               # (every alternative keeps one too)
        try:
            rx = re.compile(pat, re.ASCII)
        except re.error:
            continue
        if rx.search("") is not None:
            continue
        # a repeated group that can match nothing keeps different text in a backtracker and in RE2's automaton
        # ((x*)* on "bb": Python's last, empty iteration empties the group; a Pike VM never takes it) -- whole matches
        # agree, captures of such groups are left out of the comparison
        if re.search(r"\)[*+?{]", pat):
            continue
        has_group = rx.groups >= 1
        go_t, py_t = ("[${1}]", r"[\1]") if has_group else ("[X]", "[X]")
        for text in texts:
            out = N.lib().sybl_debug_regex_replace(pat.encode(), text.encode(), go_t.encode())
            assert out is not None, pat
            assert out.decode() == rx.sub(py_t, text), (pat, text, go_t)
            checked += 1
    assert checked > 5_000


def test_random_patterns_and_texts_under_the_sanitizers(tmp_path):
    """tools/micro/re2lite_fuzz.cpp built with AddressSanitizer + UndefinedBehaviorSanitizer: pieces of RE2 syntax glued
    together at random and byte soup as patterns, texts with multi-byte runes and invalid UTF-8 in exact-size buffers,
    random replacement templates -- compile / search / replace_all must return, nothing more."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "re2lite_fuzz")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-I", os.path.join(root, "sybil_amd", "csrc"),
                        os.path.join(root, "tools", "micro", "re2lite_fuzz.cpp"), os.path.join(root, "sybil_amd", "csrc", "re2lite.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, "40000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "compiled" in r.stdout, (r.stdout[-300:], r.stderr[-3000:])
