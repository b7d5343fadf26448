"""GPU: compact storage (sybl_table_compact).  Columns re-encoded as 1/2/4-byte offsets from the
column minimum must give the same answers as canonical int64 / int32 storage -- against the CPU
oracle, bit-exact -- while the scan streams fewer bytes."""
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


def test_widths_and_read_back(ctx, oracle):
    names = ["c00", "c01", "c02", "c03", "c04", "c07", "c08", "c09"]
    n = 300_000
    t = ctx.synth_table("w", synth.SEED, n, 0, n, synth.synth_cols(names))
    before = {c: t.read_int(c, 1000, 5000) for c in names}
    assert all(t.column_storage(c) == (8, 0) for c in names)
    bytes_before = t.hbm_bytes
    t.compact()
    widths = {c: t.column_storage(c)[0] for c in names}
    assert widths == {"c00": 4, "c01": 1, "c02": 1, "c03": 2, "c04": 2, "c07": 4, "c08": 4, "c09": 2}, widths
    assert t.column_storage("c00")[1] >= 1_700_000_000      # the base is the column minimum
    assert t.hbm_bytes < bytes_before / 2
    for c in names:
        assert np.array_equal(t.read_int(c, 1000, 5000), before[c]), c
    t.compact()  # idempotent
    assert {c: t.column_storage(c)[0] for c in names} == widths
    t.free()


@pytest.mark.parametrize("name,rows", [
    ("cfg1_count_range", 600_003),
    ("cfg2_group1_avg2", 800_000),
    ("cfg3_filter3_group2_stddev", 1_000_000),
    ("cfg4_hist_highcard", 300_000),
    ("cfg5_time_rollup", 900_000),
])
def test_baseline_workloads_compact(ctx, oracle, name, rows):
    wl = synth.WORKLOADS[name]
    q = wl["query"]
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], rows, 0, rows, q, compact=True)
    parity.compare(gres, ores, op=q.get("op", "avg"), full=q.get("want_percentiles", True) and q.get("op") == "hist",
                   n_aggs=len(q.get("aggs", [])), time_mode=bool(q.get("time_col")))
    assert stats["algorithmic_bytes"] < stats["canonical_bytes"]
    if name != "cfg5_time_rollup":  # (too few blocks at this size for per-workgroup time windows: global atomics)
        assert stats["packed_kernel"] == 1, stats   # the offset-domain kernels, not the any-width fallback
    if name == "cfg4_hist_highcard":
        assert stats["strategy"] == 5
    assert stats["canonical_bytes"] == rows * 8 * len(wl["columns"])
    gres.free()


@pytest.mark.parametrize("q", [
    dict(aggs=["c07"], op="hist"),
    dict(groups=["c01"], aggs=["c07", "c08"], op="hist", hist_bucket=5000),
    dict(filters=[("c04", "neq", 500), ("c04", "neq", 7), ("c05", "gt", 10)], groups=["c01"]),
    dict(filters=[("c04", "gt", 2000)], groups=["c01"], aggs=["c07"]),
    dict(groups=["c01", "c02"], aggs=["c07"], op="avg", order_by="c07"),
    dict(groups=["c04"], aggs=["c04"], op="hist"),
    dict(filters=[("c04", "gt", 99), ("c04", "lt", 900)], groups=["c01", "c02"], aggs=["c07", "c08"], op="hist"),
])
def test_query_shapes_compact(ctx, oracle, q):
    names = ["c01", "c02", "c04", "c05", "c07", "c08"]
    gres, ores, _ = parity.run_both(ctx, oracle, names, 500_001, 0, 500_001, q, compact=True)
    parity.compare(gres, ores, op=q.get("op", "avg"), full=True, n_aggs=len(q.get("aggs", [])))
    gres.free()


def test_negative_wide_and_missing_values(ctx, oracle):
    rng = np.random.default_rng(77)
    n = 90_000
    cols = {
        "neg": rng.integers(-300, -100, size=n),                       # 1 byte, negative base
        "mid": rng.integers(-40_000, 20_000, size=n),                  # 2 bytes
        "big": rng.integers(-(1 << 40), 1 << 40, size=n),              # does not fit 4 bytes: stays int64
        "edge": rng.integers((1 << 62) - 200, (1 << 62), size=n),      # 1 byte at a huge base
        "v": rng.integers(0, 70_000, size=n),
    }
    cols = {k: v.astype(np.int64) for k, v in cols.items()}
    pop = (rng.random(n) > 0.25).astype(np.uint8)
    info = {k: (int(v.min()), int(v.max())) for k, v in cols.items()}
    tb = ctx.create_table("nw")
    for c in cols:
        tb.add_column(c, "int", *info[c])
    for r0 in range(0, n, 20_000):
        r1 = min(n, r0 + 20_000)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pop[r0:r1]) if c == "mid" else cols[c][r0:r1]) for c in cols})
    tb.compact()
    assert [tb.column_storage(c)[0] for c in cols] == [1, 2, 8, 1, 4]
    assert tb.column_storage("edge")[1] == info["edge"][0]
    names = list(cols)
    ocols = [{"type": "int", "data": cols[c], **({"populated": pop} if c == "mid" else {})} for c in names]
    for q in (dict(groups=["neg"], aggs=["mid", "v"], op="hist"),
              dict(filters=[("mid", "gt", -10_000), ("big", "lt", 0)], groups=["edge"], aggs=["v"], op="avg"),
              dict(filters=[("edge", "gt", (1 << 62) - 100)], aggs=["big", "mid"], op="avg"),
              dict(groups=["mid"], aggs=["v"], op="avg")):
        query = tb.query(**q)
        r = query.run()
        o = oracle.run_query(ocols, block_rows=20_000, **parity.oracle_query_kwargs(names, info, q))
        parity.compare(r, o, op=q["op"], full=True, n_aggs=len(q["aggs"]))
        r.free()
        query.free()
    tb.free()


def test_append_to_compact_table_packs_in_place_and_widens(ctx, oracle):
    n = 50_000
    rng = np.random.default_rng(5)
    g = rng.integers(0, 10, size=n).astype(np.int64)
    v = rng.integers(0, 1000, size=n).astype(np.int64)
    tb = ctx.create_table("ap")
    tb.add_column("g", "int")
    tb.add_column("v", "int", 0, 5_000_000)
    tb.append_block(n, {"g": g, "v": v})
    tb.compact()
    assert tb.column_storage("v") == (2, int(v.min())) and tb.column_storage("g")[0] == 1
    query = tb.query(groups=["g"], aggs=["v"])
    r = query.run()
    r.free()
    # a block that fits the compact layout is packed into place ...
    g1 = rng.integers(0, 10, size=n).astype(np.int64)
    v1 = rng.integers(int(v.min()), 1000, size=n).astype(np.int64)
    tb.append_block(n, {"g": g1, "v": v1})
    assert tb.column_storage("v") == (2, int(v.min()))
    with pytest.raises(sybil_amd.SyblError):
        query.run()          # the plan predates the append
    query.free()
    # ... one that does not makes the column wider (resident rows re-encoded once), never canonical
    g2 = rng.integers(0, 10, size=n).astype(np.int64)
    v2 = rng.integers(1_000_000, 5_000_000, size=n).astype(np.int64)
    pop2 = (rng.random(n) > 0.3).astype(np.uint8)
    tb.append_block(n, {"g": g2, "v": (v2, pop2)})
    assert tb.column_storage("v")[0] == 4 and tb.column_storage("g")[0] == 1
    assert np.array_equal(tb.read_int("v", 0, 2 * n), np.concatenate([v, v1]))
    query = tb.query(groups=["g"], aggs=["v"], op="hist")
    r = query.run()
    assert query.stats()["algorithmic_bytes"] == 3 * n * 5
    o = oracle.run_query([{"type": "int", "data": np.concatenate([g, g1, g2])},
                          {"type": "int", "data": np.concatenate([v, v1, v2]),
                           "populated": np.concatenate([np.ones(2 * n, dtype=np.uint8), pop2])}],
                         groups=[0], aggs=[(1, 0, 5_000_000)], op="hist", block_rows=n)
    parity.compare(r, o, op="hist", full=True, n_aggs=1)
    r.free()
    query.free()
    tb.free()


def test_compact_from_the_first_block(ctx, oracle):
    """compact() on an empty table only switches the mode on: every block is packed as it arrives,
    the canonical form of the table never exists in HBM."""
    rng = np.random.default_rng(11)
    n, nb = 30_000, 6
    tb = ctx.create_table("cf")
    tb.add_column("k", "int")
    tb.add_column("s", "str")
    tb.add_column("v", "int", 0, 100_000)
    tb.compact()
    vocab = ["w%03d" % i for i in range(300)]
    ks, ss, vs = [], [], []
    for b in range(nb):
        k = rng.integers(-5 - b, 20 + 400 * b, size=n).astype(np.int64)      # the range grows block by block
        s = rng.integers(0, 50 * (b + 1), size=n).astype(np.int32)          # ... and so does the dictionary
        v = rng.integers(0, 1 + 20_000 * b, size=n).astype(np.int64)
        tb.append_block(n, {"k": k, "s": {"ids": s, "strings": vocab}, "v": v})
        ks.append(k), ss.append(s), vs.append(v)
    assert [tb.column_storage(c)[0] for c in ("k", "s", "v")] == [2, 2, 4]
    assert tb.column_storage("k")[1] == -5 - (nb - 1)
    k, s, v = np.concatenate(ks), np.concatenate(ss), np.concatenate(vs)
    assert np.array_equal(tb.read_int("k", 0, n * nb), k)
    for q, okw in ((dict(groups=["k"], aggs=["v"], op="hist"), dict(groups=[0], aggs=[(2, 0, 100_000)], op="hist")),
                   (dict(filters=[("v", "gt", 5_000), ("k", "lt", 100)], groups=["s"], aggs=["v"]),
                    dict(filters=[(2, "gt", 5_000), (0, "lt", 100)], groups=[1], aggs=[(2, 0, 100_000)]))):
        query = tb.query(**q)
        r = query.run()
        o = oracle.run_query([{"type": "int", "data": k}, {"type": "str", "data": s}, {"type": "int", "data": v}],
                             block_rows=n, **okw)
        if "s" in q["groups"]:
            gmap = {g["group_by_key"]: g for g in r.results}
            omap = {vocab[x["key_vals"][0]] + "\t": x for x in o["results"]}
            assert set(gmap) == set(omap) and r.matched == o["matched"]
            for key, x in omap.items():
                assert gmap[key]["count"] == x["count"]
                parity.compare_hist(gmap[key]["hists"][0], x["hists"][0], "avg", True, ctx=key)
        else:
            parity.compare(r, o, op=q["op"], full=True, n_aggs=1)
        r.free()
        query.free()
    tb.free()


def test_nullable_and_str_columns_run_the_packed_kernel(ctx, oracle):
    """Missing rows in filter / group / aggregation columns, a str group column and a str filter over a
    compact table: k_scan_packed<NUL> (validity bitmaps next to the offsets), not the 2-row GEN kernel."""
    rng = np.random.default_rng(23)
    n, nb = 40_000, 5
    vocab = ["s%02d" % i for i in range(40)]
    tb = ctx.create_table("nul")
    for c in ("f1", "g1", "v1", "v2"):
        tb.add_column(c, "int", 0, 999_999)
    tb.add_column("gs", "str")
    tb.compact()
    cols = {k: [] for k in ("f1", "g1", "v1", "v2", "gs", "pf", "pg", "pv")}
    for b in range(nb):
        f1 = rng.integers(0, 1000, size=n)
        g1 = rng.integers(0, 12, size=n)
        v1 = rng.integers(0, 1_000_000, size=n)
        v2 = rng.integers(0, 60_000, size=n)
        gs = rng.integers(0, len(vocab), size=n).astype(np.int32)
        pf, pg, pv = [(rng.random(n) > p).astype(np.uint8) for p in (0.1, 0.2, 0.15)]
        tb.append_block(n, {"f1": (f1, pf), "g1": (g1, pg), "v1": (v1, pv), "v2": v2, "gs": {"ids": gs, "strings": vocab}})
        for k, x in (("f1", f1), ("g1", g1), ("v1", v1), ("v2", v2), ("gs", gs), ("pf", pf), ("pg", pg), ("pv", pv)):
            cols[k].append(x)
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    assert [tb.column_storage(c)[0] for c in ("f1", "g1", "v1", "v2", "gs")] == [2, 1, 4, 2, 1]
    ocols = [{"type": "int", "data": cat["f1"].astype(np.int64), "populated": cat["pf"]},
             {"type": "int", "data": cat["g1"].astype(np.int64), "populated": cat["pg"]},
             {"type": "int", "data": cat["v1"].astype(np.int64), "populated": cat["pv"]},
             {"type": "int", "data": cat["v2"].astype(np.int64)},
             {"type": "str", "data": cat["gs"]}]
    import re
    for q, okw in (
        (dict(filters=[("f1", "gt", 99), ("f1", "lt", 900)], groups=["g1"], aggs=["v1", "v2"], op="hist", want_percentiles=False),
         dict(filters=[(0, "gt", 99), (0, "lt", 900)], groups=[1], aggs=[(2, 0, 999_999), (3, 0, 999_999)], op="hist")),
        (dict(filters=[("gs", "re", "^s[01]")], groups=["g1"], aggs=["v1"], op="avg"),
         dict(filters=[(4, "re", 0, np.array([bool(re.search("^s[01]", s)) for s in vocab], dtype=np.uint8))], groups=[1],
              aggs=[(2, 0, 999_999)], op="avg")),
        (dict(filters=[("f1", "lt", 500)], groups=["g1"], aggs=["v1"], op="hist", time_col="v2", time_bucket=20_000, want_percentiles=False),
         dict(filters=[(0, "lt", 500)], groups=[1], aggs=[(2, 0, 999_999)], op="hist", time_col=3, time_bucket=20_000)),
    ):
        query = tb.query(**q)
        r = query.run()
        st = query.stats()
        assert st["packed_kernel"] == 1 and st["strategy"] == 2, st
        o = oracle.run_query(ocols, block_rows=n, **okw)
        parity.compare(r, o, op=q["op"], full=False, n_aggs=len(q["aggs"]), time_mode=bool(q.get("time_col")))
        r.free()
        query.free()
    tb.free()


@pytest.mark.parametrize("compact", [False, True])
def test_int_neq_runs_the_specialised_kernels(ctx, oracle, compact):
    """Config 3 with an int `neq` (two constants on one filter column, one of them next to the range; a third on a
    column of its own): the role-specialised kernels take it (k_scan_fast<GEN> / k_scan_packed<NUL>), not the
    plan-interpreting k_scan (filter.go:171-195)."""
    wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    q = dict(wl["query"])
    q["filters"] = list(q["filters"]) + [("c04", "neq", 500), ("c04", "neq", 123), ("c02", "neq", 7)]
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 900_000, 0, 900_000, q, compact=compact)
    assert stats["strategy"] == 2 and stats["packed_kernel"] == (1 if compact else 0), stats
    parity.compare(gres, ores, op="hist", full=False, n_aggs=2)
    assert len(gres.results) == 16 * 63     # c02 = 7 is gone
    gres.free()


def test_three_and_four_group_columns_run_the_packed_body(ctx, oracle):
    """More group columns than k_scan_packed is instantiated for: the same offset-domain row body with run-time column
    counts (k_scan_hash_packed<.., HASH = false>) over compact storage -- a missing-value key digit, a filter with a
    neq, moments and avg modes (aggregate.go:125-143)."""
    rng = np.random.default_rng(314)
    n = 700_000
    cols = {"g1": rng.integers(0, 4, n), "g2": rng.integers(10, 15, n), "g3": rng.integers(-3, 3, n), "g4": rng.integers(0, 3, n),
            "f": rng.integers(0, 1000, n), "v": rng.integers(0, 1000, n), "u": rng.integers(100, 900, n)}
    cols = {k: v.astype(np.int64) for k, v in cols.items()}
    pops = {"g2": (rng.random(n) > 0.1).astype(np.uint8)}
    info = {"v": (0, 999), "u": (100, 899)}   # (bucket geometries without outliers: those run the GEN body)
    tb = ctx.create_table("g4")
    for c in cols:
        lo, hi = info.get(c, (1, 0))
        tb.add_column(c, "int", lo, hi)
    for r0 in range(0, n, 65536):
        r1 = min(r0 + 65536, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if c in pops else cols[c][r0:r1]) for c in cols})
    tb.compact()
    names = list(cols)
    ocols = [{"type": "int", "data": cols[c], **({"populated": pops[c]} if c in pops else {})} for c in names]
    for q, okw in ((dict(filters=[("f", "gt", 99), ("f", "neq", 500)], groups=["g1", "g2", "g3"], aggs=["v", "u"], op="hist", want_percentiles=False),
                    dict(filters=[(4, "gt", 99), (4, "neq", 500)], groups=[0, 1, 2], aggs=[(5, 0, 999), (6, 100, 899)], op="hist")),
                   (dict(groups=["g1", "g2", "g3", "g4"], aggs=["v"], op="avg"),
                    dict(groups=[0, 1, 2, 3], aggs=[(5, 0, 999)], op="avg"))):
        query = tb.query(**q)
        gres = query.run()
        st = query.stats()
        assert st["strategy"] == 2 and st["packed_kernel"] == 1, st
        ores = oracle.run_query(ocols, n_threads=4, **okw)
        parity.compare(gres, ores, op=q["op"], full=False, n_aggs=len(q["aggs"]))
        gres.free()
        query.free()
    tb.free()


def test_three_and_four_aggregation_columns_run_the_packed_body(ctx, oracle, monkeypatch):
    """More aggregation columns than k_scan_packed is instantiated for (`-int a,b,c[,d]`): the run-time-count packed body
    takes them over compact storage -- moments and avg modes, a column with missing rows (its own count / populated-count
    fields), and the same through the hash table (aggregate.go:246-261)."""
    rng = np.random.default_rng(2718)
    n = 600_000
    cols = {"g1": rng.integers(0, 6, n), "g2": rng.integers(20, 27, n), "f": rng.integers(0, 1000, n),
            "a": rng.integers(0, 1000, n), "b": rng.integers(100, 900, n), "c": rng.integers(0, 1000, n), "d": rng.integers(5, 250, n)}
    cols = {k: v.astype(np.int64) for k, v in cols.items()}
    pops = {"c": (rng.random(n) > 0.2).astype(np.uint8)}
    info = {"a": (0, 999), "b": (100, 899), "c": (0, 999), "d": (5, 249)}   # (bucket geometries without outliers: those run the GEN body)
    tb = ctx.create_table("a4")
    for c in cols:
        lo, hi = info.get(c, (1, 0))
        tb.add_column(c, "int", lo, hi)
    for r0 in range(0, n, 65536):
        r1 = min(r0 + 65536, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if c in pops else cols[c][r0:r1]) for c in cols})
    tb.compact()
    names = list(cols)
    ocols = [{"type": "int", "data": cols[c], **({"populated": pops[c]} if c in pops else {})} for c in names]
    ix = {c: i for i, c in enumerate(names)}
    agg = lambda c: (ix[c],) + info[c]
    cases = ((dict(filters=[("f", "gt", 99), ("f", "lt", 900)], groups=["g1", "g2"], aggs=["a", "b", "d"], op="hist", want_percentiles=False),
              dict(filters=[(ix["f"], "gt", 99), (ix["f"], "lt", 900)], groups=[ix["g1"], ix["g2"]], aggs=[agg("a"), agg("b"), agg("d")], op="hist")),
             (dict(groups=["g1"], aggs=["a", "b", "c", "d"], op="avg"),
              dict(groups=[ix["g1"]], aggs=[agg("a"), agg("b"), agg("c"), agg("d")], op="avg")),
             (dict(filters=[("f", "neq", 7)], groups=["g2"], aggs=["c", "a", "b"], op="hist", want_percentiles=False),
              dict(filters=[(ix["f"], "neq", 7)], groups=[ix["g2"]], aggs=[agg("c"), agg("a"), agg("b")], op="hist")))
    for hashed in (False, True):
        if hashed:
            monkeypatch.setenv("SYBL_FORCE_HASH", "1")
        for q, okw in cases:
            query = tb.query(**q)
            gres = query.run()
            st = query.stats()
            assert st["strategy"] == (7 if hashed else 2) and (hashed or st["packed_kernel"] == 1), st
            ores = oracle.run_query(ocols, n_threads=4, **okw)
            parity.compare(gres, ores, op=q["op"], full=False, n_aggs=len(q["aggs"]))
            gres.free()
            query.free()
    tb.free()


def test_set_filters_and_a_fifth_filter_column_run_the_packed_body(ctx, oracle, monkeypatch):
    """Filters the packed row bodies do not evaluate -- set members (filter.go:252-285), a fifth and sixth filter column --
    used to send the whole query to the plan-interpreting kernel.  They now run first (k_prefilter writes a row bitmap) and
    the scan proper stays on the packed body, reading the bitmap like a validity word: same results as the oracle and as
    the plan interpreter (SYBL_NO_PREFILTER)."""
    rng = np.random.default_rng(91)
    n = 600_000
    ints = {k: rng.integers(0, 1000, n, dtype=np.int64) for k in ("a", "b", "c", "d", "e", "f")}
    g = rng.integers(0, 40, n, dtype=np.int64)
    v = rng.integers(0, 1_000_000, n, dtype=np.int64)
    v_pop = (rng.random(n) > 0.1).astype(np.uint8)
    # set column: 0-3 members per row out of six tags
    cnt = rng.integers(0, 4, n)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(cnt)
    members = rng.integers(0, 6, int(off[-1])).astype(np.int32)
    tags = ["t%d" % i for i in range(6)]
    tb = ctx.create_table("pre")
    for k in ints:
        tb.add_column(k, "int", 0, 999)
    tb.add_column("g", "int", 0, 39)
    tb.add_column("v", "int", 0, 999_999)
    tb.add_column("tags", "set")
    for r0 in range(0, n, 65536):
        r1 = min(r0 + 65536, n)
        blk = {k: x[r0:r1] for k, x in ints.items()}
        blk["g"] = g[r0:r1]
        blk["v"] = (v[r0:r1], v_pop[r0:r1])
        blk["tags"] = {"ids": members[off[r0]:off[r1]], "offsets": off[r0:r1 + 1] - off[r0], "strings": tags}
        tb.append_block(r1 - r0, blk)
    tb.compact()
    names = list(ints) + ["g", "v", "tags"]
    ocols = [{"type": "int", "data": ints[k]} for k in ints] + [{"type": "int", "data": g}, {"type": "int", "data": v, "populated": v_pop},
                                                               {"type": "set", "data": members, "offsets": off}]
    info = dict({k: (0, 999) for k in ints}, g=(0, 39), v=(0, 999_999))
    six = [(k, "gt", 150) for k in ints]
    cases = [
        dict(filters=[("tags", "in", "t2")], groups=["g"], aggs=["v"], op="avg"),
        dict(filters=[("tags", "in", "t1"), ("tags", "nin", "t4"), ("a", "lt", 700), ("b", "gt", 100)], groups=["g"], aggs=["v"], op="hist", want_percentiles=False),
        dict(filters=six, groups=["g"], aggs=["v"], op="avg"),
        dict(filters=six + [("tags", "nin", "t0")], groups=["g"], aggs=["v"], op="hist", want_percentiles=False),
        dict(filters=[("tags", "in", "nope")], groups=["g"], aggs=["v"], op="avg"),   # a tag no row has
    ]
    for q in cases:
        okw = parity.oracle_query_kwargs(names, info, q)
        okw["filters"] = [(f[0], f[1], tags.index(f[2]) if f[2] in tags else -1) if isinstance(f[2], str) else f for f in okw["filters"]]
        ores = oracle.run_query(ocols, **okw)
        digests = {}
        for pre in (True, False):
            if pre:
                monkeypatch.delenv("SYBL_NO_PREFILTER", raising=False)
            else:
                monkeypatch.setenv("SYBL_NO_PREFILTER", "1")
            query = tb.query(**q)
            gres = query.run()
            st = query.stats()
            assert st["packed_kernel"] == (1 if pre else 0), (q, st)
            parity.compare(gres, ores, op=q.get("op", "avg"), full=False, n_aggs=1)
            digests[pre] = (gres.matched, sorted((r["key"], r["count"], r["hists"][0]["sum"]) for r in gres.results))
            gres.free()
            query.free()
        assert digests[True] == digests[False], q
    monkeypatch.delenv("SYBL_NO_PREFILTER", raising=False)
    # the same through the hash table (the packed hash body reads the bitmap too), and with a pre-pass filter that leaves
    # whole tiles without a passing row (a wave then loads nothing of such a tile)
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    for q in (cases[1], cases[3], dict(filters=six + [("tags", "in", "t5"), ("f", "gt", 990)], groups=["g"], aggs=["v"], op="avg")):
        okw = parity.oracle_query_kwargs(names, info, q)
        okw["filters"] = [(f[0], f[1], tags.index(f[2]) if f[2] in tags else -1) if isinstance(f[2], str) else f for f in okw["filters"]]
        ores = oracle.run_query(ocols, **okw)
        query = tb.query(**q)
        gres = query.run()
        assert query.stats()["strategy"] == 7
        parity.compare(gres, ores, op=q.get("op", "avg"), full=False, n_aggs=1)
        gres.free()
        query.free()
    monkeypatch.delenv("SYBL_FORCE_HASH")
    q = dict(filters=six + [("tags", "in", "t5"), ("f", "gt", 990)], groups=["g"], aggs=["v"], op="avg")
    okw = parity.oracle_query_kwargs(names, info, q)
    okw["filters"] = [(f[0], f[1], tags.index(f[2]) if f[2] in tags else -1) if isinstance(f[2], str) else f for f in okw["filters"]]
    gres = tb.query(**q).run()
    parity.compare(gres, oracle.run_query(ocols, **okw), op="avg", full=False, n_aggs=1)
    gres.free()
    tb.free()
