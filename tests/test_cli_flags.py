"""The flag surface of sybil-gpu-query (tools/sybil_gpu_query.cpp) against `sybil query`'s (src/cmd/cmd_query.go:19-74,
src/lib/config.go:147-152): every flag of the reference is either served, accepted without effect, or refused BY NAME -- never
"provided but not defined".  No GPU: the binary stops at the missing -table, after its flags were parsed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sybil_amd", "sybil-gpu-query")

# (flag, takes a value)
SERVED = [("dir", True), ("op", True), ("limit", True), ("print", False), ("json", False), ("sort", True), ("sort-asc", False), ("time", False),
          ("time-col", True), ("time-bucket", True), ("weight-col", True), ("loghist", False), ("encode-results", False), ("int-filter", True),
          ("int-bucket", True), ("str-replace", True), ("str-filter", True), ("set-filter", True), ("int", True), ("str", True), ("set", True),
          ("group", True), ("distinct", True), ("field-separator", True), ("filter-separator", True)]
NO_EFFECT = [("prune-sort", True), ("distinct-limit", True), ("recycle-mem", False), ("fast-recycle", False), ("shorten-key-table", False),
             ("cache-queries", False), ("debug", False)]
ELSEWHERE = ["samples", "sample-cols", "export", "read-log", "tables", "info", "update-info", "encode-flags", "decode-flags", "tdigest"]


def _run(*args):
    if not os.path.exists(EXE):
        pytest.skip("sybil-gpu-query not built (python -c 'import __graft_entry__ as g; g.build()')")
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, timeout=60)


@pytest.mark.parametrize("flag,valued", SERVED + NO_EFFECT)
def test_reference_flags_are_accepted(flag, valued):
    r = _run("-" + flag, "x") if valued else _run("-" + flag)
    assert r.returncode != 0 and "no table specified" in r.stderr, (flag, r.stderr)  # parsed; stopped at the missing -table
    r = _run("--%s=%s" % (flag, "x" if valued else "true"))
    assert "no table specified" in r.stderr, (flag, r.stderr)


@pytest.mark.parametrize("flag", ELSEWHERE)
def test_flags_of_other_paths_are_refused_by_name(flag):
    r = _run("-" + flag)
    assert r.returncode == 2 and "does not serve" in r.stderr and flag in r.stderr, r.stderr


def test_an_unknown_flag_is_undefined():
    r = _run("-no-such-flag")
    assert r.returncode == 2 and "flag provided but not defined: -no-such-flag" in r.stderr
