"""GPU: the append path's copy guard (SYBL_VERIFY_COPIES=1, csrc/table.cpp: host_to_device) -- every copy of caller memory into
HBM (sybl_table_append_block's columns, validity words, dictionaries' look-up tables, set CSR, group dictionaries) goes through
the context's pinned staging buffer and, under the switch, is digested on the device (k_copy_digest) and compared with the host
bytes.  tests/conftest.py turns it on for the whole suite.  Here: it is on, it notices a corrupted destination (a test hook
overwrites one word behind the copy), and the table / query sequence behind round 5's one unexplained wrong-keys event
(tools/repro_rank.py) loops for a minute under it without a mismatch."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guard_is_on_and_notices_a_corrupted_copy(monkeypatch):
    import sybil_amd
    from sybil_amd import _native as N
    assert os.environ.get("SYBL_VERIFY_COPIES") == "1"
    ctx = sybil_amd.Context(0)
    try:
        rng = np.random.default_rng(5)
        vals = rng.integers(-(1 << 40), 1 << 40, size=70_000).astype(np.int64)
        pop = (rng.random(vals.size) > 0.1).astype(np.uint8)
        names = dict(ids=rng.integers(0, 40, size=vals.size).astype(np.int32), strings=["s%d" % k for k in range(40)])
        for compact in (False, True):
            tb = ctx.create_table("g")
            tb.add_column("v", "int")
            tb.add_column("s", "str")
            if compact:
                tb.compact()
            tb.append_block(vals.size, {"v": (vals, pop), "s": names})       # verified copies: green
            assert np.array_equal(tb.read_int("v", 0, vals.size)[pop != 0], vals[pop != 0])
            monkeypatch.setenv("SYBL_VERIFY_COPIES_FAULT", "1")
            with pytest.raises(N.SyblError) as e:
                tb.append_block(vals.size, {"v": (vals, pop), "s": names})
            assert "SYBL_VERIFY_COPIES" in str(e.value) and "differ in HBM" in str(e.value)
            monkeypatch.delenv("SYBL_VERIFY_COPIES_FAULT")
            tb.free()
    finally:
        ctx.close()


def test_the_unexplained_sequence_loops_clean_under_the_guard():
    env = dict(os.environ, PYTHONPATH=ROOT, SYBL_VERIFY_COPIES="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "repro_rank.py"), "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "mismatches 0" in out, out[-3000:] + p.stderr.decode(errors="replace")[-2000:]
