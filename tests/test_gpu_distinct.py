"""GPU: count distinct (csrc/distinct.hip, hll.h) against the oracle's sketches -- register by register.

First run on an MI355X in round 3 (all green on the first run: gpurun_out/r03_distinct.log); part of the suite since.
The functions the kernel is built from are also covered on the CPU (tests/test_hll_host.py), the algorithm by
tests/test_oracle_distinct.py.  The sketch itself stays "parity unpinned" (the reference's dependency is absent)."""
import numpy as np
import pytest

import sybil_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _int_table(ctx, cols, pops=None, info=None, block_rows=65536, compact=False):
    names = list(cols)
    n = len(cols[names[0]])
    tb = ctx.create_table("d")
    for c in names:
        lo, hi = (info or {}).get(c, (1, 0))
        tb.add_column(c, "int", lo, hi)
    for r0 in range(0, n, block_rows):
        r1 = min(r0 + block_rows, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if pops and c in pops else cols[c][r0:r1]) for c in names})
    if compact:
        tb.compact()
    return tb


def _compare(gres, ores, time_mode=False):
    assert gres.matched == ores["matched"]
    for which, name in ((0, "results"), (1, "time_results")):
        grows = gres.rows(which)
        omap = {(r["time_bucket"], r["key"]): r for r in ores[name]}
        assert len(grows) == len(omap), name
        for i, g in enumerate(grows):
            o = omap[(g["time_bucket"], g["key"])]
            card, regs = gres.distinct(which, i, registers=True)
            assert np.array_equal(regs, o["registers"]), (name, g["key_vals"])
            assert card == o["distinct"] == g["distinct"], (name, g["key_vals"], card, o["distinct"])
            assert g["count"] == o["count"]
    card, regs = gres.distinct(2, 0, registers=True)
    assert np.array_equal(regs, ores["cumulative"]["registers"])
    assert card == ores["cumulative"]["distinct"]


@pytest.mark.parametrize("compact", [False, True])
def test_one_int_column_grouped(ctx, oracle, compact):
    rng = np.random.default_rng(1)
    n = 1_000_000
    cols = {"g": rng.integers(0, 50, n), "user": rng.integers(0, 200_000, n), "f": rng.integers(0, 1000, n)}
    pops = {"user": (rng.random(n) > 0.05).astype(np.uint8)}
    tb = _int_table(ctx, cols, pops, compact=compact)
    q = tb.query(filters=[("f", "gt", 99), ("f", "lt", 900)], groups=["g"], distincts=["user"], order_by="$COUNT")
    gres = q.run()
    ores = oracle.run_query([{"type": "int", "data": cols["g"]}, {"type": "int", "data": cols["user"], "populated": pops["user"]},
                             {"type": "int", "data": cols["f"]}],
                            filters=[(2, "gt", 99), (2, "lt", 900)], groups=[0], distincts=[1], n_threads=4, want_registers=True)
    _compare(gres, ores)
    # the estimate itself: within the sketch's error of the truth
    for g in gres.rows(0):
        sel = (cols["g"] == g["key_vals"][0]) & (cols["f"] > 99) & (cols["f"] < 900)
        true = len(np.unique(np.where(pops["user"][sel] != 0, cols["user"][sel], -1)))
        assert abs(g["distinct"] - true) <= 0.03 * true
    gres.free()
    q.free()
    tb.free()


@pytest.mark.parametrize("ncols", [2, 3, 5, 8])
def test_several_int_columns_no_groups(ctx, oracle, ncols):
    rng = np.random.default_rng(ncols)
    n = 300_000
    cols = {"c%d" % i: rng.integers(-3, 4 + i, n) for i in range(ncols)}
    tb = _int_table(ctx, cols)
    q = tb.query(distincts=list(cols))
    gres = q.run()
    ores = oracle.run_query([{"type": "int", "data": cols[c]} for c in cols], distincts=list(range(ncols)), n_threads=4, want_registers=True)
    _compare(gres, ores)
    gres.free()
    q.free()
    tb.free()


def test_distinct_column_is_also_the_group_and_an_aggregation(ctx, oracle):
    rng = np.random.default_rng(7)
    n = 400_000
    cols = {"a": rng.integers(0, 300, n), "b": rng.integers(0, 20, n)}
    tb = _int_table(ctx, cols, info={"a": (0, 299)})
    q = tb.query(groups=["b"], aggs=["a"], distincts=["a", "b"], op="avg")
    gres = q.run()
    ores = oracle.run_query([{"type": "int", "data": cols["a"]}, {"type": "int", "data": cols["b"]}], groups=[1], aggs=[(0, 0, 299)],
                            distincts=[0, 1], n_threads=4, want_registers=True)
    _compare(gres, ores)
    gres.free()
    q.free()
    tb.free()


def test_time_series_sketches_live_in_the_time_results(ctx, oracle):
    rng = np.random.default_rng(11)
    n = 500_000
    cols = {"g": rng.integers(0, 8, n), "user": rng.integers(0, 30_000, n),
            "t": np.sort(rng.integers(1_700_000_000, 1_700_000_000 + 6 * 3600, n))}
    tb = _int_table(ctx, cols)
    q = tb.query(groups=["g"], distincts=["user"], time_col="t", time_bucket=3600)
    gres = q.run()
    ores = oracle.run_query([{"type": "int", "data": cols[c]} for c in cols], groups=[0], distincts=[1], time_col=2, time_bucket=3600,
                            n_threads=4, want_registers=True)
    _compare(gres, ores, time_mode=True)
    assert all(r["distinct"] == 0 for r in gres.rows(0))  # the all-time Results only count (aggregate.go:156-169)
    gres.free()
    q.free()
    tb.free()


def test_one_str_column(ctx, oracle):
    rng = np.random.default_rng(13)
    n = 200_000
    vocab = ["agent/%d.%d" % (i, i * 7 % 13) for i in range(4000)]
    ids = rng.integers(0, len(vocab), n).astype(np.int32)
    g = rng.integers(0, 5, n)
    tb = ctx.create_table("s")
    tb.add_column("ua", "str")
    tb.add_column("g", "int")
    tb.append_block(n, {"ua": {"ids": ids, "strings": vocab}, "g": g})
    q = tb.query(groups=["g"], distincts=["ua"])
    gres = q.run()
    strs = tb.column_dict("ua")  # table-global ids may be numbered differently from the block's
    back = {s: i for i, s in enumerate(strs)}
    gids = np.array([back[vocab[i]] for i in range(len(vocab))], dtype=np.int32)[ids]
    ores = oracle.run_query([{"type": "str", "data": gids}, {"type": "int", "data": g}], groups=[1], distincts=[0],
                            distinct_dicts={0: strs}, n_threads=4, want_registers=True)
    _compare(gres, ores)
    gres.free()
    q.free()
    tb.free()


def test_refusals(ctx):
    rng = np.random.default_rng(17)
    n = 1000
    tb = ctx.create_table("r")
    tb.add_column("ua", "str")
    tb.add_column("g", "int")
    tb.append_block(n, {"ua": {"ids": rng.integers(0, 3, n).astype(np.int32), "strings": ["a", "b", "c"]}, "g": rng.integers(0, 5, n)})
    with pytest.raises(sybil_amd.SyblError):
        tb.query(distincts=["nope"])
    tb.free()


@pytest.mark.parametrize("compact", [False, True])
def test_str_column_together_with_other_columns(ctx, oracle, compact):
    """aggregate.go:224-239, the slow path (refused until round 5): one buffer per row -- the decimal digits of an int, the
    dictionary string of a str id, nothing for a row without the column, a tab after each -- hashed as a whole.  The engine
    assembles and hashes it per row (hll.h: Metro64Stream); register by register against the oracle: str + int, int + str +
    str, negative ints, rows without one of the columns, buffers on both sides of MetroHash's 32-byte block."""
    rng = np.random.default_rng(23)
    n = 150_000
    vocab = ["agent/%d.%d" % (i, i * 7 % 13) for i in range(300)] + ["", "x" * 40]
    host = ["h%d" % i for i in range(40)]
    ua = rng.integers(0, len(vocab), n).astype(np.int32)
    hs = rng.integers(0, len(host), n).astype(np.int32)
    ua_pop = (rng.random(n) > 0.1).astype(np.uint8)
    code = rng.integers(-500, 500, n).astype(np.int64) * 1_000_003
    code_pop = (rng.random(n) > 0.15).astype(np.uint8)
    g = rng.integers(0, 6, n)
    tb = ctx.create_table("mx")
    tb.add_column("ua", "str")
    tb.add_column("host", "str")
    tb.add_column("code", "int")
    tb.add_column("g", "int")
    for r0 in range(0, n, 50_000):
        r1 = r0 + 50_000
        tb.append_block(50_000, {"ua": {"ids": ua[r0:r1], "strings": vocab, "populated": ua_pop[r0:r1]}, "host": {"ids": hs[r0:r1], "strings": host},
                                 "code": (code[r0:r1], code_pop[r0:r1]), "g": g[r0:r1]})
    if compact:
        tb.compact()
    dicts = {}
    gids = {}
    for name, voc, ids in (("ua", vocab, ua), ("host", host, hs)):
        strs = tb.column_dict(name)  # table-global ids may be numbered differently from the block's
        back = {s_: i for i, s_ in enumerate(strs)}
        dicts[name] = strs
        gids[name] = np.array([back[v] for v in voc], dtype=np.int32)[ids]
    ocols = [{"type": "str", "data": gids["ua"], "populated": ua_pop}, {"type": "str", "data": gids["host"]},
             {"type": "int", "data": code, "populated": code_pop}, {"type": "int", "data": g}]
    for names in (["ua", "code"], ["code", "ua", "host"], ["host", "ua"]):
        ix = [["ua", "host", "code"].index(c) for c in names]
        q = tb.query(groups=["g"], distincts=names, filters=[("g", "lt", 5)])
        gres = q.run()
        ores = oracle.run_query(ocols, groups=[3], distincts=ix, filters=[(3, "lt", 5)],
                                distinct_dicts={i: dicts[["ua", "host"][i]] for i in ix if i < 2}, n_threads=4, want_registers=True)
        _compare(gres, ores)
        gres.free()
        q.free()
    tb.free()


def test_cli_text_and_json(ctx, oracle, tmp_path):
    """printer.go:142-144,204-205 through sybl_result_render."""
    import json
    rng = np.random.default_rng(19)
    n = 100_000
    cols = {"g": rng.integers(0, 3, n), "user": rng.integers(0, 5000, n)}
    tb = _int_table(ctx, cols)
    q = tb.query(groups=["g"], distincts=["user"], order_by="$COUNT")
    gres = q.run()
    rows = gres.rows(0)
    js = json.loads(gres.render("json"))
    assert [r["Distinct"] for r in js] == [r["distinct"] for r in rows]
    assert all(r["Count"] == r["Distinct"] and "Samples" not in r for r in js)
    text = gres.render("text").splitlines()
    assert any(line.endswith(" Distinct: %d" % rows[0]["distinct"]) for line in text)
    gres.free()
    q.free()
    tb.free()


@pytest.mark.parametrize("forced", [True, False])
def test_hashed_group_by_keeps_a_sketch_per_key_it_found(ctx, oracle, monkeypatch, forced):
    """Count distinct over a group-by that goes through the hash table (round 6; refused before): the sketch pass runs once
    the dense key list is final, every row finding its key's place by binary search.  forced: a small key space sent
    through the table (SYBL_FORCE_HASH); else two keys whose product of ranges does not direct-map."""
    rng = np.random.default_rng(23)
    n = 600_000
    if forced:
        monkeypatch.setenv("SYBL_FORCE_HASH", "1")
        cols = {"g": rng.integers(0, 40, n), "h": rng.integers(-5, 5, n), "user": rng.integers(0, 50_000, n), "f": rng.integers(0, 1000, n)}
    else:
        # 2^31 x 2^31 possible cells, a few thousand live ones
        pool_g, pool_h = rng.integers(0, 1 << 31, 60), rng.integers(0, 1 << 31, 50)
        cols = {"g": pool_g[rng.integers(0, 60, n)], "h": pool_h[rng.integers(0, 50, n)], "user": rng.integers(0, 50_000, n), "f": rng.integers(0, 1000, n)}
        monkeypatch.setenv("SYBL_NO_GDICT", "1")  # (the keys stay offsets in their ranges: 2^62 cells, hashed)
    pops = {"user": (rng.random(n) > 0.05).astype(np.uint8)}
    tb = _int_table(ctx, cols, pops)
    q = tb.query(filters=[("f", "gt", 99), ("f", "lt", 900)], groups=["g", "h"], distincts=["user"], aggs=["f"], op="avg")
    gres = q.run()
    assert q.stats()["strategy"] == 7
    ores = oracle.run_query([{"type": "int", "data": cols["g"]}, {"type": "int", "data": cols["h"]},
                             {"type": "int", "data": cols["user"], "populated": pops["user"]}, {"type": "int", "data": cols["f"]}],
                            filters=[(3, "gt", 99), (3, "lt", 900)], groups=[0, 1], aggs=[(3, 0, 999)], distincts=[2], n_threads=4, want_registers=True)
    _compare(gres, ores)
    gres.free()
    # a rescan of the same query: the pass runs again over the (same) key list
    gres = q.run()
    _compare(gres, ores)
    gres.free()
    q.free()
    tb.free()


def test_encode_results_carries_the_sketches(ctx, oracle):
    """-encode-results of a count-distinct result (printer.go:284-289; refused before round 6): Result.Distinct travels as a
    self-marshalling value -- a precision byte and the 16384 registers.  PARITY UNPINNED (the reference's dependency is absent):
    what is pinned here is that the blob holds exactly the registers sybl_result_distinct hands out, row by row."""
    from tests import gobfmt
    rng = np.random.default_rng(31)
    n = 200_000
    cols = {"g": rng.integers(0, 6, n), "user": rng.integers(0, 20_000, n)}
    tb = _int_table(ctx, cols)
    q = tb.query(groups=["g"], distincts=["user"], order_by="$COUNT")
    r = q.run()
    v, types = gobfmt.decode(r.encode(), want_types=True)
    assert ("opaque", "LogLogBeta") in [(k, nm) for _, k, nm in types]
    res = v["QuerySpec"]["QueryResults"]
    rows = r.rows(0)
    assert len(res["Results"]) == len(rows) == 6
    for i, row in enumerate(rows):
        blob = res["Results"][row["group_by_key"]]["Distinct"]
        card, regs = r.distinct(0, i, registers=True)
        assert len(blob) == 1 + 16384 and blob[0] == 14 and np.array_equal(np.frombuffer(blob[1:], dtype=np.uint8), regs)
    card, regs = r.distinct(2, 0, registers=True)
    assert np.array_equal(np.frombuffer(res["Cumulative"]["Distinct"][1:], dtype=np.uint8), regs)
    assert [s["GroupByKey"] for s in res["Sorted"]] == [row["group_by_key"] for row in rows]
    r.free()
    q.free()
    tb.free()
