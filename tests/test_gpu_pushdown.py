"""GPU: -limit pushed into the scan (csrc/pushdown.hip, sybl_query_desc.printed_only = 2): a printer's histogram query sorted by
$COUNT counts the groups from the key column alone, chooses the printed groups on the device and fills only their bucket arrays
and Cumulative's.  What the printers show -- the first `limit` rows of SortResults' order (aggregate.go:497-525) and TOTAL -- must
be what the full path (printed_only = 1, strategy 5) shows, byte for byte, and must agree with the CPU oracle."""
import numpy as np
import pytest

import sybil_amd
from sybil_amd import synth
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _both(tb, q):
    out = []
    for level in (1, 2):
        qy = tb.query(**dict(q, printed_only=level))
        r = qy.run()
        st = qy.stats()
        out.append((r, qy, st))
    return out


@pytest.mark.parametrize("rows,limit,aggs", [(6_000_000, 50, ["c07"]), (3_000_000, 7, ["c07", "c08"]), (2_500_000, 100_000, ["c07"])])
def test_pushed_down_printer_prints_what_the_full_path_prints(ctx, oracle, rows, limit, aggs):
    cols = ["c03"] + aggs
    tb = ctx.synth_table("pd", synth.SEED, rows, 0, rows, synth.synth_cols(cols))
    if len(aggs) == 2:
        # (two aggregations: 65 536 cells would be 2048 (cell, agg) partitions, more than strategy 5 takes -- a key of 30 000
        # values, from the host, beside the generator's value columns)
        tb.free()
        rng = np.random.default_rng(9)
        key = rng.integers(0, 30_000, rows)
        cdata = {n: oracle.synth_fill(*(synth.COLUMNS[n][0], synth.COLUMNS[n][2], synth.COLUMNS[n][3]), synth.SEED, synth.COLUMNS[n][1], 0, rows, rows) for n in aggs}
        tb = ctx.create_table("pd")
        tb.add_column("c03", "int", 0, 29_999)
        for n in aggs:
            tb.add_column(n, "int", synth.COLUMNS[n][4], synth.COLUMNS[n][5])
        for r0 in range(0, rows, 65536):
            tb.append_block(min(65536, rows - r0), dict({"c03": key[r0:r0 + 65536]}, **{n: cdata[n][r0:r0 + 65536] for n in aggs}))
    tb.compact()
    q = dict(groups=["c03"], aggs=aggs, op="hist", want_percentiles=True, limit=limit, order_by="$COUNT")
    (r1, q1, s1), (r2, q2, s2) = _both(tb, q)
    assert s1["strategy"] == 5 and s2["strategy"] == 8, (s1["strategy"], s2["strategy"])
    # both printers, byte for byte
    assert r2.render("text") == r1.render("text")
    assert r2.render("json") == r1.render("json")
    assert r2.matched == r1.matched == rows
    # ... and against the oracle: every group's Count, the printed rows and Cumulative in full
    info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in cols}
    ocols = parity.oracle_synth_cols(oracle, cols, rows, 0, rows)
    if len(aggs) == 2:
        ocols[0] = {"type": "int", "data": key}
    ores = oracle.run_query(ocols, n_threads=8, **parity.oracle_query_kwargs(cols, info, q))
    omap = {r["key"]: r for r in ores["results"]}
    rows2 = r2.rows(0)
    assert len(rows2) == len(omap)
    counts = [r["count"] for r in rows2]
    assert counts == sorted(counts, reverse=True)
    for i, g in enumerate(rows2):
        o = omap[g["key"]]
        assert g["count"] == o["count"]
        if i < limit:
            for a in range(len(aggs)):
                parity.compare_hist(g["hists"][a], o["hists"][a], "hist", True, ctx=(i, a))
    for a in range(len(aggs)):
        parity.compare_hist(r2.cumulative["hists"][a], ores["cumulative"]["hists"][a], "hist", True, ctx=("cumulative", a), cumulative=True)
    # the order among equal counts is the full path's (stable over the cell order)
    assert [g["key"] for g in rows2[:min(limit, len(rows2))]] == [g["key"] for g in r1.rows(0)[:min(limit, len(rows2))]]
    # a rescan gives the same
    r3 = q2.run()
    assert r3.render("text") == r1.render("text")
    for r in (r1, r2, r3):
        r.free()
    q1.free()
    q2.free()
    tb.free()


def test_a_hot_group_fills_its_counter_field_many_times(ctx, oracle):
    """Skew: one key holds a third of the rows, so its 15-bit LDS counter wraps dozens of times per workgroup (the guard bit
    and the device-side carries); a range of keys never occurs."""
    rng = np.random.default_rng(3)
    n = 4_000_000
    key = rng.integers(0, 40_000, n)
    key[rng.random(n) < 0.33] = 12_345
    key[rng.random(n) < 0.10] = 39_999
    val = rng.integers(0, 1_000_000, n)
    tb = ctx.create_table("skew")
    tb.add_column("k", "int", 0, 49_999)
    tb.add_column("v", "int", 0, 999_999)
    for r0 in range(0, n, 65536):
        tb.append_block(min(65536, n - r0), {"k": key[r0:r0 + 65536], "v": val[r0:r0 + 65536]})
    tb.set_bounds("k", 0, 49_999)
    tb.compact()
    q = dict(groups=["k"], aggs=["v"], op="hist", want_percentiles=True, limit=20, order_by="$COUNT")
    (r1, q1, s1), (r2, q2, s2) = _both(tb, q)
    assert s2["strategy"] == 8
    assert r2.render("text") == r1.render("text") and r2.render("json") == r1.render("json")
    got = {g["key_vals"][0]: g["count"] for g in r2.rows(0)}
    vals, cnts = np.unique(key, return_counts=True)
    assert got == dict(zip(vals.tolist(), cnts.tolist()))
    for r in (r1, r2):
        r.free()
    q1.free()
    q2.free()
    tb.free()


def test_queries_the_pushdown_does_not_take_answer_as_a_printer(ctx):
    rows = 1_000_000
    tb = ctx.synth_table("pd2", synth.SEED, rows, 0, rows, synth.synth_cols(["c03", "c07", "c04"]))
    tb.compact()
    base = dict(groups=["c03"], aggs=["c07"], op="hist", want_percentiles=True, limit=10, printed_only=2)
    for extra in (dict(order_by="c07"), dict(order_by="$COUNT", order_asc=True), dict(order_by="$COUNT", filters=[("c04", "gt", 500)])):
        qy = tb.query(**dict(base, **extra))
        r = qy.run()
        assert qy.stats()["strategy"] == 5
        r.free()
        qy.free()
    tb.free()


def test_nothing_to_scan_and_odd_shapes(ctx):
    """An empty table, a table whose key space is odd in size, and a limit of one: the pushed-down path and the full path print
    the same (an empty table scans nothing: the ordinary zeroed tables answer)."""
    rng = np.random.default_rng(17)
    for n, card in ((0, 3001), (400_000, 3001), (300_000, 2049)):
        tb = ctx.create_table("odd")
        tb.add_column("k", "int", 0, card - 1)
        tb.add_column("v", "int", 0, 999)   # (BucketSize 1, 1002 buckets: no value can be an outlier -- what the pushdown asks for)
        key, val = rng.integers(0, card, n), rng.integers(0, 1000, n)
        for r0 in range(0, n, 65536):
            tb.append_block(min(65536, n - r0), {"k": key[r0:r0 + 65536], "v": val[r0:r0 + 65536]})
        tb.set_bounds("k", 0, card - 1)
        tb.set_bounds("v", 0, 999)
        tb.compact()
        for limit in (1, 10):
            q = dict(groups=["k"], aggs=["v"], op="hist", want_percentiles=True, limit=limit, order_by="$COUNT")
            (r1, q1, s1), (r2, q2, s2) = _both(tb, q)
            if n:
                assert s2["strategy"] == 8, s2["strategy"]
            assert r2.render("text") == r1.render("text") and r2.render("json") == r1.render("json"), (n, card, limit)
            assert r2.matched == r1.matched == n
            for r in (r1, r2):
                r.free()
            q1.free()
            q2.free()
        tb.free()
