"""Oracle full-query checks: numpy brute force + the property assertions the reference's
own tests make (src/lib/aggregate_test.go, filter_test.go) on the same data shapes."""
import numpy as np
import pytest


def _people(n, seed=1):
    rng = np.random.default_rng(seed)
    age = rng.integers(10, 30, size=n)          # filter_test.go: age in [10,29]
    t = 1_700_000_000 + rng.integers(0, 86400 * 3, size=n)
    f1 = rng.integers(0, 1000, size=n)
    return age.astype(np.int64), np.sort(t).astype(np.int64), f1.astype(np.int64)


def test_group_by_int_mean_is_key(oracle):
    # aggregate_test.go:13-56: grouping by age, the mean of age in each group is the key
    age, t, f1 = _people(5000)
    cols = [{"type": "int", "data": age}, {"type": "int", "data": f1}]
    r = oracle.run_query(cols, groups=[0], aggs=[(0, 10, 29), (1, 0, 999)], op="avg", block_rows=100)
    assert r["matched"] == 5000 and len(r["results"]) == 20
    for res in r["results"]:
        k = res["key_vals"][0]
        assert res["hists"][0]["avg"] == pytest.approx(k, abs=1e-9)
        sel = age == k
        assert res["count"] == res["samples"] == int(sel.sum())
        assert res["hists"][1]["sum_exact"] == int(f1[sel].sum())
        assert res["hists"][1]["avg"] == pytest.approx(f1[sel].mean(), rel=1e-12)
    c = r["cumulative"]
    assert c["count"] == 5000
    assert c["hists"][0]["avg"] == pytest.approx(age.mean(), rel=1e-12)  # aggregate_test.go:59-99


@pytest.mark.parametrize("op,val,expect_groups", [("neq", 20, 19), ("eq", 20, 1), ("lt", 20, 10), ("gt", 20, 9)])
def test_int_filters(oracle, op, val, expect_groups):
    # filter_test.go:42-310
    age, t, f1 = _people(4000)
    cols = [{"type": "int", "data": age}]
    r = oracle.run_query(cols, filters=[(0, op, val)], groups=[0], aggs=[(0, 10, 29)], block_rows=100)
    assert len(r["results"]) == expect_groups
    fn = {"neq": np.not_equal, "eq": np.equal, "lt": np.less, "gt": np.greater}[op]
    assert r["matched"] == int(fn(age, val).sum())


def test_str_and_set_filters(oracle):
    age, t, f1 = _people(3000)
    ids = (age - 10).astype(np.int32)                  # dictionary id i <-> str(10+i)
    regex_table = np.array([str(10 + i).startswith("2") for i in range(20)], dtype=np.uint8)  # re ^2
    # set column: each row holds {age id, (age id + 1) % 20}
    off = np.arange(0, 2 * age.size + 1, 2, dtype=np.int64)
    sv = np.stack([ids, (ids + 1) % 20], axis=1).reshape(-1).astype(np.int32)
    cols = [{"type": "int", "data": age}, {"type": "str", "data": ids},
            {"type": "set", "data": sv, "offsets": off}]
    r = oracle.run_query(cols, filters=[(1, "re", 0, regex_table)], groups=[1], block_rows=100)
    assert len(r["results"]) == 10 and r["matched"] == int((age >= 20).sum())
    r = oracle.run_query(cols, filters=[(1, "nre", 0, regex_table)], groups=[1], block_rows=100)
    assert len(r["results"]) == 10 and r["matched"] == int((age < 20).sum())
    r = oracle.run_query(cols, filters=[(1, "eq", 10)], groups=[0], block_rows=100)
    assert len(r["results"]) == 1 and r["results"][0]["key_vals"] == (20,)
    r = oracle.run_query(cols, filters=[(2, "in", 10)], groups=[0], block_rows=100)
    assert sorted(x["key_vals"][0] for x in r["results"]) == [19, 20]
    r = oracle.run_query(cols, filters=[(2, "nin", 10)], groups=[0], block_rows=100)
    assert len(r["results"]) == 18


def test_hist_percentiles_single_valued_groups(oracle):
    # aggregate_test.go:140-154: single-valued group => p25 = p50 = p75 = key
    age, t, f1 = _people(4000)
    cols = [{"type": "int", "data": age}]
    r = oracle.run_query(cols, groups=[0], aggs=[(0, 10, 29)], op="hist", block_rows=100)
    for res in r["results"]:
        p = res["hists"][0]["percentiles"]
        assert int(p[25]) == int(p[50]) == int(p[75]) == res["key_vals"][0]
        assert int(res["hists"][0]["values"].sum()) == res["count"]
    # ungrouped: percentiles within +-1 of the sorted sample (aggregate_test.go:175-184)
    r = oracle.run_query(cols, aggs=[(0, 10, 29)], op="hist", block_rows=100)
    s = np.sort(age)
    p = r["results"][0]["hists"][0]["percentiles"]
    for k in (25, 50, 75, 99):
        assert abs(int(p[k]) - int(s[k * s.size // 100])) <= 1
    assert r["results"][0]["key"] == b"" and len(r["results"]) == 1


def test_time_series(oracle):
    age, t, f1 = _people(6000)
    cols = [{"type": "int", "data": age}, {"type": "int", "data": t}, {"type": "int", "data": f1}]
    r = oracle.run_query(cols, groups=[0], aggs=[(2, 0, 999)], time_col=1, time_bucket=3600, block_rows=100)
    tb = t // 3600 * 3600
    assert len({x["time_bucket"] for x in r["time_results"]}) == len(np.unique(tb))
    assert sum(x["count"] for x in r["time_results"]) == 6000
    # Results carry only Count/Samples in time-series mode (aggregate.go:156-183)
    assert all(not h["present"] for x in r["results"] for h in x["hists"])
    assert sum(x["count"] for x in r["results"]) == 6000
    for x in r["time_results"][:50]:
        sel = (tb == x["time_bucket"]) & (age == x["key_vals"][0])
        assert x["count"] == int(sel.sum()) and x["hists"][0]["sum_exact"] == int(f1[sel].sum())


def test_missing_values_and_weights(oracle):
    n = 1000
    rng = np.random.default_rng(3)
    g = rng.integers(0, 4, size=n).astype(np.int64)
    v = rng.integers(0, 100, size=n).astype(np.int64)
    w = rng.integers(1, 6, size=n).astype(np.int64)
    gp = (rng.random(n) > 0.2).astype(np.uint8)
    vp = (rng.random(n) > 0.3).astype(np.uint8)
    cols = [{"type": "int", "data": g, "populated": gp}, {"type": "int", "data": v, "populated": vp},
            {"type": "int", "data": w}]
    r = oracle.run_query(cols, groups=[0], aggs=[(1, 0, 99)], weight_col=2, block_rows=64)
    keys = {x["key_vals"][0] for x in r["results"]}
    assert 0xFFFFFFFFFFFFFFFF in keys and len(keys) == 5          # MISSING_VALUE group
    for x in r["results"]:
        k = x["key_vals"][0]
        sel = (gp == 0) if k == 0xFFFFFFFFFFFFFFFF else ((gp == 1) & (g == k))
        assert x["samples"] == int(sel.sum()) and x["count"] == int(w[sel].sum())
        s2 = sel & (vp == 1)
        assert x["hists"][0]["count"] == int(w[s2].sum())
        assert x["hists"][0]["sum_exact"] == int((v[s2] * w[s2]).sum())
        assert x["hists"][0]["avg"] == pytest.approx((v[s2] * w[s2]).sum() / w[s2].sum(), rel=1e-12)
    # an int filter on an unpopulated value fails (filter.go:172-174)
    r = oracle.run_query(cols, filters=[(1, "neq", -1)], block_rows=64)
    assert r["matched"] == int(vp.sum())


def test_weight_carries_over_rows_without_one_inside_a_block(oracle):
    """aggregate.go:68,100-102: `weight` is declared before the row loop of FilterAndAggRecords (one call per block) as 1 and
    only assigned when the row's weight column is populated -- a row without one aggregates with the last weight seen in
    its block, whether or not the row that carried it passed the filters.  Checked against a numpy restatement."""
    n, block = 5000, 700
    rng = np.random.default_rng(11)
    g = rng.integers(0, 5, size=n).astype(np.int64)
    v = rng.integers(0, 100, size=n).astype(np.int64)
    f = rng.integers(0, 10, size=n).astype(np.int64)
    w = rng.integers(2, 9, size=n).astype(np.int64)
    wp = (rng.random(n) > 0.6).astype(np.uint8)
    wp[block * 2:block * 2 + 50] = 0  # a block that starts without weights: 1 until the first one
    eff = np.empty(n, dtype=np.int64)
    for b0 in range(0, n, block):
        cur = 1
        for i in range(b0, min(b0 + block, n)):
            if wp[i]:
                cur = w[i]
            eff[i] = cur
    cols = [{"type": "int", "data": g}, {"type": "int", "data": v}, {"type": "int", "data": w, "populated": wp}, {"type": "int", "data": f}]
    r = oracle.run_query(cols, filters=[(3, "gt", 2)], groups=[0], aggs=[(1, 0, 99)], weight_col=2, block_rows=block)
    assert len(r["results"]) == 5
    for x in r["results"]:
        sel = (g == x["key_vals"][0]) & (f > 2)
        assert x["samples"] == int(sel.sum()) and x["count"] == int(eff[sel].sum())
        assert x["hists"][0]["count"] == int(eff[sel].sum()) and x["hists"][0]["sum_exact"] == int((v[sel] * eff[sel]).sum())


def test_block_skip_is_result_neutral(oracle):
    age, t, f1 = _people(5000)
    cols = [{"type": "int", "data": t}, {"type": "int", "data": f1}]
    lo = int(t[1500])
    a = oracle.run_query(cols, filters=[(0, "gt", lo)], aggs=[(1, 0, 999)], block_rows=100, block_skip=False)
    b = oracle.run_query(cols, filters=[(0, "gt", lo)], aggs=[(1, 0, 999)], block_rows=100, block_skip=True)
    assert b["blocks_skipped"] >= 14 and a["blocks_skipped"] == 0
    assert a["matched"] == b["matched"] == int((t > lo).sum())
    assert a["results"][0]["hists"][0]["sum_exact"] == b["results"][0]["hists"][0]["sum_exact"]


def test_threads_do_not_change_results(oracle):
    age, t, f1 = _people(20000)
    cols = [{"type": "int", "data": age}, {"type": "int", "data": f1}]
    a = oracle.run_query(cols, groups=[0], aggs=[(1, 0, 999)], op="hist", block_rows=256, n_threads=1)
    b = oracle.run_query(cols, groups=[0], aggs=[(1, 0, 999)], op="hist", block_rows=256, n_threads=4)
    for x, y in zip(a["results"], b["results"]):
        assert x["key"] == y["key"] and x["count"] == y["count"]
        assert x["hists"][0]["avg"] == y["hists"][0]["avg"]  # merge is always in block order
        assert (x["hists"][0]["values"] == y["hists"][0]["values"]).all()


def test_synth_generator_is_counter_based(oracle):
    a = oracle.synth_fill(oracle.SYN_UNIFORM, 0, 1000, 20241022, 4, 0, 1000, 10_000)
    b = oracle.synth_fill(oracle.SYN_UNIFORM, 0, 1000, 20241022, 4, 500, 500, 10_000)
    assert (a[500:] == b).all() and a.min() >= 0 and a.max() < 1000
    t = oracle.synth_fill(oracle.SYN_TIME, 1_700_000_000, 2_592_000, 20241022, 0, 0, 10_000, 10_000)
    assert (np.diff(t) >= 0).all() and t[0] == 1_700_000_000 and t[-1] < 1_700_000_000 + 2_592_000
    bell = oracle.synth_fill(oracle.SYN_BELL, 0, 250_000, 20241022, 8, 0, 10_000, 10_000)
    assert 0 <= bell.min() and bell.max() <= 999_996
    assert oracle.splitmix64(0) == 0xE220A8397B1DCDAF


def test_columnar_baseline_matches_the_reference_shaped_oracle(oracle):
    """orc_columnar_scan (the plain columnar CPU baseline bench.py times beside the reference-shaped one) must
    produce the same integers as the per-block hash-map restatement on the config-3 query."""
    from sybil_amd import synth
    from tests import parity
    wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    names = wl["columns"]
    n = 300_000
    cols = parity.oracle_synth_cols(oracle, names, 1_000_000_000, 0, n)
    d = {nm: c["data"] for nm, c in zip(names, cols)}
    m, tab = oracle.columnar_scan([d["c04"], d["c05"], d["c06"]], [(100, 899)] * 3, [d["c01"], d["c02"]], [(0, 16), (0, 64)],
                                  [d["c07"], d["c08"]], [(0, 999), (0, 999)], n_threads=3)
    info = {x: (synth.COLUMNS[x][4], synth.COLUMNS[x][5]) for x in names}
    o = oracle.run_query(cols, n_threads=2, want_values=True, **parity.oracle_query_kwargs(names, info, dict(wl["query"], want_percentiles=True)))
    assert m == o["matched"] == int(tab[0].sum())
    import numpy as np
    for r in o["results"]:
        cell = r["key_vals"][0] * 64 + r["key_vals"][1]
        assert tab[0][cell] == r["count"]
        for a in range(2):
            h = r["hists"][a]
            assert tab[1 + 3 * a][cell] == h["sum_exact"]
            b = np.arange(len(h["values"]), dtype=np.int64)
            assert tab[2 + 3 * a][cell] == int((b * h["values"]).sum())
            assert tab[3 + 3 * a][cell] == int((b * b * h["values"]).sum())
    # a key outside the declared bounds is reported, not silently dropped
    assert oracle.columnar_scan([], [], [d["c02"]], [(0, 8)], [], [], n_threads=1)[0] == -1


@pytest.mark.parametrize("wl_name", ["cfg3_filter3_group2_stddev", "cfg4_hist_highcard", "cfg5_time_rollup"])
def test_full_size_checker_matches_the_reference_shaped_oracle(oracle, wl_name):
    """orc_synth_scan (generator fused with the direct-mapped row loop: what the 1e9-row GPU tests are checked
    against) must produce the same integers as the per-block hash-map restatement on a materialised slice of
    the same virtual table -- every Count, exact sum, bucket and bucket moment, for the config 3 / 4 / 5 shapes."""
    from sybil_amd import synth
    from tests import parity
    wl = synth.WORKLOADS[wl_name]
    names = wl["columns"]
    total, row0, n = 1_000_000_000, 777 * 65536, 250_000 if wl_name != "cfg4_hist_highcard" else 120_000
    q = dict(wl["query"])
    hist = q.get("op") == "hist"
    s = oracle.synth_scan(synth.COLUMNS, synth.SEED, total, row0, n, filters=q.get("filters", ()), groups=q.get("groups", ()),
                          aggs=q.get("aggs", ()), time_col=q.get("time_col"), time_bucket=q.get("time_bucket", 0),
                          want_buckets=hist, n_threads=3)
    cols = parity.oracle_synth_cols(oracle, names, total, row0, n)
    info = {x: (synth.COLUMNS[x][4], synth.COLUMNS[x][5]) for x in names}
    o = oracle.run_query(cols, n_threads=2, want_values=True, **parity.oracle_query_kwargs(names, info, dict(q, want_percentiles=True)))
    assert s["matched"] == o["matched"] == int(s["count"].sum())
    rows = o["time_results"] if q.get("time_col") else o["results"]
    assert len(rows) == int((s["count"] != 0).sum())
    na = len(q.get("aggs", ()))
    for r in rows:
        cell = r["time_bucket"] // q["time_bucket"] - s["tb_min"] if q.get("time_col") else 0
        for g, card in enumerate(s["cells"][1:]):
            cell = cell * card + (r["key_vals"][g] - s["gmin"][g])
        assert s["count"][cell] == r["count"]
        for a in range(na):
            h = r["hists"][a]
            assert s["sum"][a][cell] == h["sum_exact"]
            if hist:
                assert np.array_equal(s["buckets"][cell][a][:h["n_values"]], h["values"])
                b = np.arange(len(h["values"]), dtype=np.int64)
                assert s["sb"][a][cell] == int((b * h["values"]).sum())
                assert s["sb2"][a][cell] == int((b * b * h["values"]).sum())
