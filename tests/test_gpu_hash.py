"""GPU: hash group-by (strategy 7, csrc/hashgroup.hip) -- group keys that do not direct-map (the reference's
map[string]*Result, aggregate.go:186-200) -- against the CPU oracle: forced on the BASELINE shapes, wide single keys,
wide composite keys with few live combinations (LDS staging), missing keys, bucket arrays, 10^7 distinct keys, and the
multi-rank union protocol."""
import numpy as np
import pytest

import sybil_amd
from sybil_amd import synth
from tests import parity

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rows_mode")]  # (each test also with SYBL_LAZY_ROWS=1: conftest.py)


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _table(ctx, cols, info=None, pops=None, block_rows=65536, compact=False):
    names = list(cols)
    n = len(cols[names[0]])
    tb = ctx.create_table("h")
    for c in names:
        lo, hi = (info or {}).get(c, (1, 0))
        tb.add_column(c, "int", lo, hi)
    for r0 in range(0, n, block_rows):
        r1 = min(r0 + block_rows, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if pops and c in pops else cols[c][r0:r1]) for c in names})
    if compact:
        tb.compact()
    return tb


def _ocols(cols, pops=None):
    return [{"type": "int", "data": cols[c], **({"populated": pops[c]} if pops and c in pops else {})} for c in cols]


@pytest.mark.parametrize("staging", [True, False])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("specialised", ["packed", "fast", "generic"])
def test_forced_hash_equals_direct_mapped_and_oracle(ctx, oracle, monkeypatch, staging, full, specialised):
    """Config 3 (1024 groups) through the hash table: every row hits the LDS staging table (or, without it, the
    global table with device-scope atomics); bucket arrays (full) always live in the global table.  Both row bodies:
    k_scan_hash_packed (compact storage, no bucket arrays), k_scan_hash_fast (SYBL_NO_HASH_PACKED) and the plan-interpreting k_scan_hash (SYBL_NO_HASH_FAST)."""
    wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    if specialised == "generic":
        monkeypatch.setenv("SYBL_NO_HASH_FAST", "1")
    elif specialised == "fast":
        monkeypatch.setenv("SYBL_NO_HASH_PACKED", "1")
    if not staging:
        monkeypatch.setenv("SYBL_NO_HASH_LDS", "1")
    q = dict(wl["query"], want_percentiles=full)
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 1_200_000, 0, 1_200_000, q, oracle_threads=4, compact=not full)
    assert stats["strategy"] == 7, stats
    assert (stats["lds_bytes"] > 0) == (staging and not full), stats
    parity.compare(gres, ores, op="hist", full=full, n_aggs=2)
    gres.free()


def test_wide_keys_through_a_dictionary_digit(ctx, oracle):
    """A key column spread over 2^40 values (its digit is the rank in the column's distinct values, ~300 000 of them)
    times a 1000-value column: 3e8 possible cells, more than direct mapping takes."""
    rng = np.random.default_rng(7)
    n = 300_000
    cols = {"k": rng.integers(-(1 << 39), 1 << 39, size=n).astype(np.int64),
            "j": rng.integers(0, 1000, size=n).astype(np.int64),
            "v": rng.integers(0, 1_000_000, size=n).astype(np.int64)}
    tb = _table(ctx, cols, info={"v": (0, 999_999)})
    for op, full in (("avg", False), ("hist", False)):
        q = dict(groups=["k", "j"], aggs=["v"], op=op, want_percentiles=full)
        query = tb.query(**q)
        gres = query.run()
        assert query.stats()["strategy"] == 7, query.stats()
        ores = oracle.run_query(_ocols(cols), groups=[0, 1], aggs=[(2, 0, 999_999)], op=op, n_threads=4)
        parity.compare(gres, ores, op=op, full=full, n_aggs=1)
        gres.free()
        query.free()
    tb.free()


def test_composite_key_with_few_live_groups_is_staged_in_lds(ctx, oracle):
    """Three key columns of 2000 values each = 8e9 possible cells, 600 live combinations: the LDS staging table
    absorbs every row; negative values, a filter, two aggregations, missing values in a key and in an aggregation."""
    rng = np.random.default_rng(11)
    n = 400_000
    combos = rng.integers(-1000, 1000, size=(600, 3))
    pick = rng.integers(0, 600, size=n)
    cols = {"a": combos[pick, 0].astype(np.int64), "b": combos[pick, 1].astype(np.int64), "c": combos[pick, 2].astype(np.int64),
            "f": rng.integers(0, 100, size=n).astype(np.int64),
            "v": rng.integers(-5000, 5000, size=n).astype(np.int64), "u": rng.integers(0, 70_000, size=n).astype(np.int64)}
    # widen the ranges so that the product of the three exceeds 2^27 cells whatever the sample drew
    cols["a"][0], cols["a"][1] = -1000, 999
    cols["b"][0], cols["b"][1] = -1000, 999
    cols["c"][0], cols["c"][1] = -1000, 999
    pops = {"b": (rng.random(n) > 0.1).astype(np.uint8), "v": (rng.random(n) > 0.2).astype(np.uint8)}
    info = {"v": (-4000, 4000), "u": (0, 69_999)}
    for compact in (False, True):
        tb = _table(ctx, cols, info=info, pops=pops, block_rows=50_000, compact=compact)
        for op in ("avg", "hist"):
            q = dict(filters=[("f", "gt", 9)], groups=["a", "b", "c"], aggs=["v", "u"], op=op, want_percentiles=False)
            query = tb.query(**q)
            gres = query.run()
            st = query.stats()
            assert st["strategy"] == 7 and st["lds_bytes"] > 0, st
            names = list(cols)
            ores = oracle.run_query(_ocols(cols, pops), filters=[(names.index("f"), "gt", 9)], groups=[0, 1, 2],
                                    aggs=[(names.index("v"), -4000, 4000), (names.index("u"), 0, 69_999)], op=op, block_rows=50_000, n_threads=4)
            parity.compare(gres, ores, op=op, full=False, n_aggs=2)
            gres.free()
            query.free()
        tb.free()


def test_five_group_columns(ctx, oracle):
    """More group columns than the direct-mapped kernels specialise for (the reference takes any number)."""
    rng = np.random.default_rng(13)
    n = 200_000
    cols = {"g%d" % i: rng.integers(0, 45 + i, size=n).astype(np.int64) for i in range(5)}
    cols["v"] = rng.integers(0, 1000, size=n).astype(np.int64)
    tb = _table(ctx, cols, info={"v": (0, 999)})
    q = dict(groups=["g0", "g1", "g2", "g3", "g4"], aggs=["v"], op="avg")
    query = tb.query(**q)
    gres = query.run()
    assert query.stats()["strategy"] == 7, query.stats()  # 45 * 46 * 47 * 48 * 49 > 2^27 cells
    ores = oracle.run_query(_ocols(cols), groups=[0, 1, 2, 3, 4], aggs=[(5, 0, 999)], op="avg", n_threads=4)
    parity.compare(gres, ores, op="avg", full=False, n_aggs=1)
    gres.free()
    query.free()
    tb.free()


def test_bucket_arrays_through_the_hash_table(ctx, oracle, monkeypatch):
    rng = np.random.default_rng(17)
    n = 60_000
    cols = {"a": rng.integers(0, 50, size=n).astype(np.int64) * 2000, "b": rng.integers(-3, 4, size=n).astype(np.int64),
            "v": rng.integers(0, 500, size=n).astype(np.int64)}
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    tb = _table(ctx, cols, info={"v": (0, 499)})
    q = dict(groups=["a", "b"], aggs=["v"], op="hist", want_percentiles=True)
    query = tb.query(**q)
    gres = query.run()
    assert query.stats()["strategy"] == 7, query.stats()
    ores = oracle.run_query(_ocols(cols), groups=[0, 1], aggs=[(2, 0, 499)], op="hist", n_threads=4)
    parity.compare(gres, ores, op="hist", full=True, n_aggs=1)
    gres.free()
    query.free()
    tb.free()


def test_ten_million_distinct_keys(ctx, oracle):
    """12 M rows, a key drawn from 2^40 values (~12 M distinct: far beyond the 2^22-entry group dictionary): keys, Count
    and exact sum(v) of every group bit for bit against the oracle's exact group-by."""
    n = 12_000_000
    t = ctx.synth_table("wide", synth.SEED, n, 0, n, [
        {"name": "k", "kind": synth.UNIFORM, "col_index": 40, "a": -(1 << 39), "b": 1 << 40},
        {"name": "v", "kind": synth.UNIFORM, "col_index": 41, "a": 0, "b": 1_000_000, "info_min": 0, "info_max": 999_999}])
    query = t.query(groups=["k"], aggs=["v"], op="avg", order_by=None)
    gres = query.run()
    st = query.stats()
    assert st["strategy"] == 7, st
    k = oracle.synth_fill(synth.UNIFORM, -(1 << 39), 1 << 40, synth.SEED, 40, 0, n, n)
    v = oracle.synth_fill(synth.UNIFORM, 0, 1_000_000, synth.SEED, 41, 0, n, n)
    okeys, ocount, osum = oracle.group_count_sum([k], v)
    assert len(okeys) > 10_000_000
    assert gres.matched == n
    gkeys = query.hash_keys()
    # composite key = value - min over the resident rows (single group column)
    assert np.array_equal(gkeys.astype(np.int64) + int(k.min()), okeys[:, 0])
    assert np.array_equal(query.debug_cells("count"), ocount)
    assert np.array_equal(query.debug_cells("sum", 0), osum)
    rows = gres.rows(2)
    assert rows[0]["count"] == n and rows[0]["hists"][0]["sum"] == int(v.sum())
    gres.free()
    query.free()
    t.free()


def test_union_of_two_ranks_keys(ctx, oracle, monkeypatch):
    """Two shards scanned separately find different key sets; after installing the union their dense partial tables
    line up: SUM / MAX of the two equals the table of the whole, and rank 0 finalized from the merged table equals
    the oracle on the whole table."""
    import torch
    rng = np.random.default_rng(23)
    n = 200_000
    cols = {"k": rng.integers(0, 500_000, size=n).astype(np.int64), "j": rng.integers(0, 40, size=n).astype(np.int64),
            "v": rng.integers(-300, 90_000, size=n).astype(np.int64)}
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")  # (20 M possible cells would still direct-map)
    halves = [{c: a[:n // 2] for c, a in cols.items()}, {c: a[n // 2:] for c, a in cols.items()}]
    bounds = {c: (int(a.min()), int(a.max())) for c, a in cols.items()}
    q = dict(groups=["k", "j"], aggs=["v"], op="avg")  # avg over negative values: minima AND maxima are tracked
    tabs, queries = [], []
    for h in halves:
        tb = _table(ctx, h, info={"v": (-300, 89_999)})
        for c, (lo, hi) in bounds.items():
            tb.set_bounds(c, lo, hi)
        qq = tb.query(**q).scan()
        assert qq.stats()["strategy"] == 7
        tabs.append(tb)
        queries.append(qq)
    keys = [qq.hash_keys() for qq in queries]
    union = np.unique(np.concatenate(keys))
    assert len(union) > max(len(k) for k in keys)
    parts = []
    for qq in queries:
        qq.hash_install_union(union)
        parts.append(qq.partial_sizes())
    assert parts[0] == parts[1], parts
    s0, m0 = queries[0].partials_torch("cuda:0")
    s1, m1 = queries[1].partials_torch("cuda:0")
    assert (s0.numel(), m0.numel()) == parts[0]
    s0 += s1
    torch.maximum(m0, m1, out=m0)
    torch.cuda.synchronize()
    gres = queries[0].finalize()
    ores = oracle.run_query(_ocols(cols), groups=[0, 1], aggs=[(2, -300, 89_999)], op="avg", n_threads=4)
    parity.compare(gres, ores, op="avg", full=False, n_aggs=1)
    gres.free()
    for x in queries:
        x.free()
    for tb in tabs:
        tb.free()


def test_inlibrary_allreduce_of_a_hash_query_single_rank(ctx, oracle, monkeypatch):
    """sybl_query_allreduce on a hash group-by runs the whole protocol (key counts, key lists, union, re-layout,
    SUM / MAX all-reduce); with one rank the result must not change."""
    rng = np.random.default_rng(29)
    n = 150_000
    cols = {"k": rng.integers(0, 300_000, size=n).astype(np.int64), "j": rng.integers(0, 50, size=n).astype(np.int64),
            "v": rng.integers(0, 1000, size=n).astype(np.int64)}
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    tb = _table(ctx, cols, info={"v": (0, 999)})
    ctx.comm_init(ctx.comm_unique_id(), 1, 0)
    try:
        query = tb.query(groups=["k", "j"], aggs=["v"], op="hist", want_percentiles=False)
        query.scan()
        assert query.stats()["strategy"] == 7
        query.allreduce()
        gres = query.finalize()
        ores = oracle.run_query(_ocols(cols), groups=[0, 1], aggs=[(2, 0, 999)], op="hist", n_threads=4)
        parity.compare(gres, ores, op="hist", full=False, n_aggs=1)
        gres.free()
        query.free()
    finally:
        ctx.comm_free()
        tb.free()


def test_table_full_is_reported(ctx, monkeypatch):
    rng = np.random.default_rng(31)
    n = 100_000
    cols = {"k": rng.integers(0, 1 << 20, size=n).astype(np.int64), "j": rng.integers(0, 9, size=n).astype(np.int64)}
    tb = _table(ctx, cols)
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    monkeypatch.setenv("SYBL_HASH_SLOTS", "4096")
    query = tb.query(groups=["k", "j"], aggs=[], op="avg")
    with pytest.raises(sybil_amd.SyblError) as e:
        query.run()
    assert "more distinct group keys" in str(e.value)
    query.free()
    tb.free()


def test_table_full_at_default_sizing_fails_fast(ctx):
    """150 M rows keyed on a column of 2^40 values: ~150 M distinct keys against the 2^27-slot ceiling of the table
    (planner default sizing, no SYBL_HASH_SLOTS).  Probing is bounded and the first row that gives up tells the others
    (csrc/hashgroup.hip: hash_find_or_insert), so the query fails with SYBL_E_NOMEM in about the time of a scan instead
    of walking a full table for every remaining row."""
    import time
    n = 150_000_000
    t = ctx.synth_table("toomany", synth.SEED, n, 0, n, [
        {"name": "k", "kind": synth.UNIFORM, "col_index": 40, "a": 0, "b": 1 << 40}])
    query = t.query(groups=["k"], aggs=[], op="avg", order_by=None)
    assert query.stats()["strategy"] == 7 and query.stats()["n_cells"] == 1 << 27
    t0 = time.perf_counter()
    with pytest.raises(sybil_amd.SyblError) as e:
        query.run()
    assert "more distinct group keys" in str(e.value)
    assert time.perf_counter() - t0 < 20.0
    query.free()
    t.free()


@pytest.mark.eager_only
def test_time_series_through_the_hash_table(ctx, oracle, monkeypatch):
    """The reference groups arbitrary keys inside every time bucket (aggregate.go:146-200).  (a) config 5's 721 x 500
    cells forced through the table; (b) 721 hourly buckets x a key of 2^20 values = 7.6e8 cells, more than direct mapping
    takes: [time bucket || key] composite keys, TimeResults in (bucket, key) order, all-time Results per key."""
    wl = synth.WORKLOADS["cfg5_time_rollup"]
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 900_000, 0, 900_000, wl["query"], oracle_threads=4)
    assert stats["strategy"] == 7, stats
    parity.compare(gres, ores, op="avg", n_aggs=1, time_mode=True)
    gres.free()
    monkeypatch.delenv("SYBL_FORCE_HASH")
    n = 1_500_000
    cols = [dict(name="c00", kind=synth.TIME, col_index=0, a=1_700_000_000, b=2_592_000, info_min=1_700_000_000, info_max=1_702_591_999),
            dict(name="k", kind=synth.UNIFORM, col_index=43, a=-5, b=1 << 20, info_min=-5, info_max=(1 << 20) - 6),
            dict(name="v", kind=synth.UNIFORM, col_index=44, a=0, b=1_000_000, info_min=0, info_max=999_999)]
    t = ctx.synth_table("tsh", synth.SEED, n, 0, n, cols)
    q = dict(groups=["k"], aggs=["v"], op="hist", want_percentiles=False, time_col="c00", time_bucket=3600)
    query = t.query(**q)
    gres = query.run()
    assert query.stats()["strategy"] == 7, query.stats()
    ocols = [{"type": "int", "data": oracle.synth_fill(c["kind"], c["a"], c["b"], synth.SEED, c["col_index"], 0, n, n)} for c in cols]
    ores = oracle.run_query(ocols, groups=[1], aggs=[(2, 0, 999_999)], op="hist", time_col=0, time_bucket=3600, n_threads=4)
    parity.compare(gres, ores, op="hist", full=False, n_aggs=1, time_mode=True)
    tr = gres.time_results
    assert len(tr) > 1_400_000 and [x["time_bucket"] for x in tr[:2000]] == sorted(x["time_bucket"] for x in tr[:2000])
    gres.free()
    query.free()
    t.free()


@pytest.mark.parametrize("first", ["query", "table", "rescan"])
def test_lazy_rows_of_a_hashed_result_outlive_query_and_table(ctx, oracle, monkeypatch, first):
    """A hash group-by's result of >= 2048 groups builds its rows when first asked for, from the query's group columns and
    the table's dictionaries: freeing the query (or the table, or scanning the query again) before anybody looked at the
    rows must build them then -- the result stays valid, and equals the oracle's."""
    rng = np.random.default_rng(3)
    n = 200_000
    cols = {"k": rng.integers(-(1 << 38), 1 << 38, n, dtype=np.int64) // 7 * 7, "v": rng.integers(0, 1000, n, dtype=np.int64)}
    cols["k"][: n // 2] = cols["k"][n // 2:]  # (every key twice)
    tb = _table(ctx, cols, info={"v": (0, 999)})
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")  # (100 000 distinct keys would get a dictionary digit otherwise)
    monkeypatch.setenv("SYBL_NO_GDICT", "1")
    q = tb.query(groups=["k"], aggs=["v"], op="avg")
    monkeypatch.delenv("SYBL_FORCE_HASH")
    monkeypatch.delenv("SYBL_NO_GDICT")
    r = q.scan().finalize()
    assert q.stats()["strategy"] == 7
    if first == "query":
        q.free()
        tb.free()
    elif first == "table":
        tb.free()
        q.free()
    else:
        q.scan()
        ctx.sync()
    o = oracle.run_query(_ocols(cols), groups=[0], aggs=[(1, 0, 999)], op="avg")
    parity.compare(r, o, op="avg", full=False, n_aggs=1)
    assert len(r.results) == len(np.unique(cols["k"]))  # (every key at least twice: about n / 2 groups)
    r.free()
    if first == "rescan":
        q.free()
        tb.free()
