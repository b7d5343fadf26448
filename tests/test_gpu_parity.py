"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs -- bit-exact for counts/sums/keys/buckets/percentiles/extrema, 1e-6
relative for avg/stddev (north_star).  Edge cases follow the reference's own tests
(src/lib/aggregate_test.go, filter_test.go, table_query_test.go, column_store_test.go)."""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sybil_amd
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _wl(name):
    from sybil_amd import synth
    return synth.WORKLOADS[name]


# ---------------------------------------------------------------- BASELINE workloads (reduced rows)

@pytest.mark.parametrize("name,rows", [
    ("cfg1_count_range", 1_000_003),
    ("cfg2_group1_avg2", 1_500_000),
    ("cfg3_filter3_group2_stddev", 2_000_000),
    ("cfg5_time_rollup", 1_200_000),
])
def test_baseline_workloads(ctx, oracle, name, rows):
    wl = _wl(name)
    q = wl["query"]
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], rows, 0, rows, q)
    parity.compare(gres, ores, op=q.get("op", "avg"), full=q.get("want_percentiles", True) and q.get("op") == "hist",
                   n_aggs=len(q.get("aggs", [])), time_mode=bool(q.get("time_col")))
    assert stats["rows_scanned"] == rows
    gres.free()


def test_cfg3_with_full_histograms(ctx, oracle):
    wl = _wl("cfg3_filter3_group2_stddev")
    q = dict(wl["query"], want_percentiles=True)
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 600_000, 0, 600_000, q)
    parity.compare(gres, ores, op="hist", full=True, n_aggs=2)
    gres.free()


def test_cfg4_high_cardinality_histograms(ctx, oracle):
    wl = _wl("cfg4_hist_highcard")
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 400_000, 0, 400_000, wl["query"])
    assert stats["strategy"] == 5  # 65536 groups x 1002 buckets: partitioned histograms
    parity.compare(gres, ores, op="hist", full=True, n_aggs=1)
    gres.free()


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("aggs", [["c07", "c08", "c09"], ["c07", "c08", "c09", "c04"]])
def test_three_and_four_aggregations_with_full_histograms(ctx, oracle, compact, aggs):
    """`-op hist -int a,b,c[,d]` over 1024 groups: the kernels of the partitioned histograms take one or two aggregations,
    so the third / fourth go through the count -> emit -> k_part_hist sequence a second time (same rows, same record
    buffers, their own bucket arrays; matched rows and Result.Count are counted once) -- every bucket and percentile
    against the oracle (hist_basic.go:101-183)."""
    wl = _wl("cfg3_filter3_group2_stddev")
    cols = wl["columns"] + ["c09"]
    q = dict(wl["query"], aggs=aggs, want_percentiles=True)
    gres, ores, stats = parity.run_both(ctx, oracle, cols, 700_000, 0, 700_000, q, compact=compact)
    assert stats["strategy"] == 5, stats
    parity.compare(gres, ores, op="hist", full=True, n_aggs=len(aggs))
    gres.free()


@pytest.mark.parametrize("compact", [False, True])
def test_partitioned_histograms_with_outliers(ctx, oracle, compact):
    """`-hist-bucket 990` puts the top ~1 % of c07 beyond the last bucket (as BucketSize = size / 1000 does for many an
    ordinary column): k_part_hist clips them into the last bucket and remembers them -- exact count / sum / sum of squares in
    the cell's outlier fields, the values themselves in the log -- instead of the query falling back to one device-scope
    atomic per value (hist_basic.go:132-135, 221-257).  65 536 groups (one pass), then config 3's 1024 groups with three
    aggregations (two passes, several workgroups per partition)."""
    wl = _wl("cfg4_hist_highcard")
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 400_000, 0, 400_000, dict(wl["query"], hist_bucket=990), compact=compact)
    assert stats["strategy"] == 5, stats
    parity.compare(gres, ores, op="hist", full=True, n_aggs=1)  # includes the outlier values
    assert sum(r["hists"][0]["n_outliers"] for r in gres.results) > 1000
    gres.free()
    wl = _wl("cfg3_filter3_group2_stddev")
    q = dict(wl["query"], aggs=["c07", "c08", "c09"], want_percentiles=True, hist_bucket=990)
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"] + ["c09"], 700_000, 0, 700_000, q, compact=compact)
    assert stats["strategy"] == 5, stats
    parity.compare(gres, ores, op="hist", full=True, n_aggs=3)
    assert sum(r["hists"][0]["n_outliers"] for r in gres.results) > 1000
    gres.free()


def test_cfg4_global_atomic_strategy(ctx, oracle, monkeypatch):
    # the fallback when a query is not eligible for partitioned histograms
    monkeypatch.setenv("SYBL_NO_PARTHIST", "1")
    wl = _wl("cfg4_hist_highcard")
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 300_000, 0, 300_000, wl["query"])
    assert stats["strategy"] == 1
    parity.compare(gres, ores, op="hist", full=True, n_aggs=1)
    gres.free()


def test_all_rows_in_one_partition(ctx, oracle):
    """All rows land in ONE of 2048 partitions (declared bounds far wider than the data).  The partition
    buffers are sized by the counting pass, so badly skewed keys need no overflow handling: the query
    stays on the partitioned-histogram strategy and the results are unchanged."""
    from sybil_amd import synth
    n = 400_000
    cols = [dict(name="g", kind=synth.UNIFORM, col_index=1, a=7, b=1, info_min=0, info_max=65535),
            dict(name="v", kind=synth.UNIFORM, col_index=7, a=0, b=1_000_000, info_min=0, info_max=999_999)]
    t = ctx.synth_table("skew", synth.SEED, n, 0, n, cols)
    t.set_bounds("g", 0, 65535)
    q = t.query(groups=["g"], aggs=["v"], op="hist")
    assert q.stats()["strategy"] == 5
    r = q.run()
    assert q.stats()["strategy"] == 5
    v = oracle.synth_fill(synth.UNIFORM, 0, 1_000_000, synth.SEED, 7, 0, n, n)
    g = np.full(n, 7, dtype=np.int64)
    o = oracle.run_query([{"type": "int", "data": g}, {"type": "int", "data": v}], groups=[0], aggs=[(1, 0, 999_999)],
                         op="hist")
    parity.compare(r, o, op="hist", full=True, n_aggs=1)
    r.free()
    q.free()
    t.free()


def test_shard_of_a_larger_table(ctx, oracle):
    # a rank's shard: rows [row0, row0+n) of a larger virtual table (time column depends on N)
    wl = _wl("cfg5_time_rollup")
    gres, ores, _ = parity.run_both(ctx, oracle, wl["columns"], 10_000_000, 3_276_800, 655_360, wl["query"])
    parity.compare(gres, ores, op="avg", n_aggs=1, time_mode=True)
    gres.free()


@pytest.mark.parametrize("q", [
    dict(aggs=["c07"], op="hist"),                                           # ungrouped percentiles
    dict(groups=["c01"], aggs=["c07", "c08"], op="hist", hist_bucket=5000),  # -int-bucket override
    dict(filters=[("c04", "eq", 500)], groups=["c02"], aggs=["c08"]),
    dict(filters=[("c04", "neq", 500), ("c04", "neq", 7), ("c05", "gt", 10)], groups=["c01"]),
    dict(filters=[("c04", "gt", 2000)], groups=["c01"], aggs=["c07"]),       # matches nothing
    dict(groups=["c01", "c02"], aggs=["c07"], op="avg", order_by="c07"),
    dict(groups=["c04"], aggs=["c04"], op="hist"),                           # same column grouped and aggregated
])
def test_query_shapes(ctx, oracle, q):
    names = ["c01", "c02", "c04", "c05", "c07", "c08"]
    gres, ores, _ = parity.run_both(ctx, oracle, names, 700_001, 0, 700_001, q)
    parity.compare(gres, ores, op=q.get("op", "avg"), full=True, n_aggs=len(q.get("aggs", [])))
    gres.free()


def test_sort_order_and_limit(ctx, oracle):
    names = ["c02", "c07"]
    q = dict(groups=["c02"], aggs=["c07"], op="avg", order_by="$COUNT", limit=5)
    gres, ores, _ = parity.run_both(ctx, oracle, names, 300_000, 0, 300_000, q)
    counts = [r["count"] for r in gres.results]
    assert counts == sorted(counts, reverse=True)          # aggregate.go:43-54
    assert len(counts) == 64                               # the limit applies when printing
    means = None
    q2 = dict(q, order_by="c07", order_asc=True)
    gres2, _, _ = parity.run_both(ctx, oracle, names, 300_000, 0, 300_000, q2)
    means = [r["hists"][0]["avg"] for r in gres2.results]
    assert means == sorted(means)                          # aggregate_test.go:281-413
    gres.free()
    gres2.free()


# ---------------------------------------------------------------- host-decoded blocks (append path)

def _people(n, seed=1):
    rng = np.random.default_rng(seed)
    age = rng.integers(10, 30, size=n).astype(np.int64)
    t = np.sort(1_700_000_000 + rng.integers(0, 86400 * 3, size=n)).astype(np.int64)
    f1 = rng.integers(-500, 1000, size=n).astype(np.int64)
    w = rng.integers(1, 6, size=n).astype(np.int64)
    return age, t, f1, w


def _append_in_blocks(table, nrows, block_rows, cols):
    for r0 in range(0, nrows, block_rows):
        r1 = min(r0 + block_rows, nrows)
        blk = {}
        for name, spec in cols.items():
            if isinstance(spec, dict):
                b = {"ids": spec["ids"][r0:r1], "strings": spec["strings"]}
                if "populated" in spec:
                    b["populated"] = spec["populated"][r0:r1]
                blk[name] = b
            elif isinstance(spec, tuple):
                blk[name] = (spec[0][r0:r1], spec[1][r0:r1])
            else:
                blk[name] = spec[r0:r1]
        table.append_block(r1 - r0, blk)


def test_blocks_missing_values_weights_and_strings(ctx, oracle):
    n = 50_007                       # ragged: blocks of 1000 rows + a 7-row tail
    age, t, f1, w = _people(n)
    rng = np.random.default_rng(5)
    age_pop = (rng.random(n) > 0.15).astype(np.uint8)
    f1_pop = (rng.random(n) > 0.25).astype(np.uint8)
    strings = [str(10 + i) for i in range(20)]
    sid = (age - 10).astype(np.int32)
    tb = ctx.create_table("people")
    tb.add_column("age", "int", 10, 29)
    tb.add_column("time", "int")
    tb.add_column("f1", "int", -100, 120)    # IntInfo narrower than the data: rejects + outliers
    tb.add_column("w", "int")
    tb.add_column("age_str", "str")
    _append_in_blocks(tb, n, 1000, {"age": (age, age_pop), "time": t, "f1": (f1, f1_pop), "w": w,
                                     "age_str": {"ids": sid, "strings": strings}})
    assert tb.rows == n and tb.blocks == 51
    ocols = [{"type": "int", "data": age, "populated": age_pop}, {"type": "int", "data": t},
             {"type": "int", "data": f1, "populated": f1_pop}, {"type": "int", "data": w},
             {"type": "str", "data": sid}]
    names = ["age", "time", "f1", "w", "age_str"]
    info = {"age": (10, 29), "f1": (-100, 120), "time": (int(t.min()), int(t.max())), "w": (1, 5)}
    re2 = np.array([s.startswith("2") for s in strings], dtype=np.uint8)
    cases = [
        dict(groups=["age"], aggs=["f1"], op="hist"),                               # MISSING group + outliers
        dict(groups=["age"], aggs=["f1"], op="hist", want_percentiles=False),       # moments form of the same
        dict(groups=["age_str"], aggs=["f1", "age"], op="avg", weight_col="w"),     # weights
        dict(filters=[("f1", "gt", 0)], groups=["age"], aggs=["f1"]),               # filter on a nullable column
        dict(filters=[("age_str", "re", "^2", re2)], groups=["age_str"]),            # regex via id table
        dict(filters=[("age_str", "nre", "^2", re2)], groups=["age_str"]),
        dict(filters=[("age_str", "eq", "20")], groups=["age"]),
        dict(filters=[("age_str", "neq", "20")], groups=["age_str"]),
        dict(filters=[("age_str", "eq", "nope")], groups=["age_str"]),              # value not in the dictionary
        dict(groups=["age"], aggs=["f1"], time_col="time", time_bucket=3600),
        dict(filters=[("time", "gt", int(t[20_000]))], aggs=["f1"], block_skip=True),
    ]
    for q in cases:
        query = tb.query(**q)
        gres = query.run()
        stats = query.stats()
        okw = parity.oracle_query_kwargs(names, info, q)
        # oracle str filters take ids: translate literal values
        okw["filters"] = [(f[0], f[1], strings.index(f[2]) if f[2] in strings else -1) + tuple(f[3:])
                          if isinstance(f[2], str) else f for f in okw["filters"]]
        ores = oracle.run_query(ocols, block_rows=1000, **okw)
        parity.compare(gres, ores, op=q.get("op", "avg"), full=q.get("want_percentiles", True),
                       n_aggs=len(q.get("aggs", [])), time_mode=bool(q.get("time_col")))
        if q.get("block_skip"):
            assert stats["blocks_skipped"] == ores["blocks_skipped"] >= 19
        if q.get("groups") == ["age_str"] and not q.get("filters"):
            assert {r["group_by_key"] for r in gres.results} == {s + "\t" for s in strings}
        gres.free()
        query.free()
    # filter_test.go: re ^2 => 10 groups, neq 20 => 19 groups
    qq = tb.query(filters=[("age_str", "re", "^2")], groups=["age_str"])   # library-side regex
    r = qq.run()
    assert len(r.results) == 10
    r.free()
    qq.free()
    tb.free()


def test_large_values_round_trip(ctx, oracle):
    # column_store_test.go:143-211: 2^50-scale ints survive
    n = 20_000
    rng = np.random.default_rng(9)
    big = (rng.integers(0, 1 << 20, size=n).astype(np.int64) << 30) - (1 << 49)
    g = rng.integers(0, 8, size=n).astype(np.int64)
    tb = ctx.create_table("big")
    tb.add_column("g", "int")
    tb.add_column("v", "int", int(big.min()), int(big.max()))
    _append_in_blocks(tb, n, 4096, {"g": g, "v": big})
    assert np.array_equal(tb.read_int("v", 100, 5000), big[100:5100])
    for op in ("avg", "hist"):
        q = dict(groups=["g"], aggs=["v"], op=op)
        query = tb.query(**q)
        gres = query.run()
        ores = oracle.run_query([{"type": "int", "data": g}, {"type": "int", "data": big}], groups=[0],
                                aggs=[(1, int(big.min()), int(big.max()))], op=op, block_rows=4096)
        parity.compare(gres, ores, op=op, full=True, n_aggs=1)
        gres.free()
        query.free()
    tb.free()


def test_empty_table_and_empty_blocks(ctx):
    tb = ctx.create_table("empty")
    tb.add_column("a", "int")
    q = tb.query(groups=["a"], aggs=["a"])
    r = q.run()
    assert r.matched == 0 and r.results == [] and r.cumulative["count"] == 0
    r.free()
    q.free()
    tb.append_block(0, {"a": np.zeros(0, dtype=np.int64)})
    tb.append_block(3, {"a": np.array([5, 5, 7], dtype=np.int64)})
    q = tb.query(groups=["a"])
    r = q.run()
    assert r.matched == 3 and [(x["key_vals"][0], x["count"]) for x in r.results] == [(5, 2), (7, 1)]
    assert r.results[0]["group_by_key"] == "5\t" and r.cumulative["group_by_key"] == "TOTAL"
    r.free()
    q.free()
    tb.free()


def test_error_paths(ctx):
    import sybil_amd
    tb = ctx.create_table("err")
    tb.add_column("a", "int")
    tb.append_block(2, {"a": np.array([1, 2], dtype=np.int64)})
    with pytest.raises(sybil_amd.SyblError):
        tb.query(groups=["nope"])
    with pytest.raises(sybil_amd.SyblError):
        tb.add_column("late", "int")
    with pytest.raises(sybil_amd.SyblError):
        tb.append_block(2, {"zzz": np.array([1, 2], dtype=np.int64)})   # block rejected, table intact
    q = tb.query()
    with pytest.raises(sybil_amd.SyblError):
        q.finalize()                                                     # finalize before scan
    r = q.run()
    assert r.matched == 2
    r.free()
    tb.append_block(1, {"a": np.array([3], dtype=np.int64)})
    with pytest.raises(sybil_amd.SyblError):
        q.scan()                                                         # stale plan: table changed after prepare
    q.free()
    q = tb.query()
    r = q.run()
    assert r.matched == 3
    r.free()
    q.free()
    tb.free()


# ---------------------------------------------------------------- multi-rank merge on one GPU

def test_partials_add_up_like_ranks(ctx, oracle):
    """Two shards with identical declared bounds produce partial tables whose SUM / MAX is the
    table of the whole -- the all-reduce contract of sybl_query_partials, checked without RCCL."""
    import torch
    from sybil_amd import synth
    wl = _wl("cfg3_filter3_group2_stddev")
    names, q = wl["columns"], dict(wl["query"], want_percentiles=True)
    total = 1_310_720
    whole = ctx.synth_table("w", synth.SEED, total, 0, total, synth.synth_cols(names))
    parts = []
    for rank in range(2):
        row0, nrows = synth.shard(total, rank, 2)
        parts.append(ctx.synth_table("p%d" % rank, synth.SEED, total, row0, nrows, synth.synth_cols(names)))
    for t in parts + [whole]:
        for n in names:
            _, _, a, b, _, _ = synth.COLUMNS[n]
            t.set_bounds(n, a, 4 * (b - 1) if n == "c08" else a + b - 1)
    qw = whole.query(**q)
    sw, mw = qw.bind_torch("cuda:0")
    qw.scan()
    acc_s = acc_m = None
    queries = []
    for t in parts:
        qq = t.query(**q)
        s, m = qq.bind_torch("cuda:0")
        qq.scan()
        ctx.sync()
        acc_s = s.clone() if acc_s is None else acc_s + s
        acc_m = m.clone() if acc_m is None else torch.maximum(acc_m, m)
        queries.append(qq)
    ctx.sync()
    assert torch.equal(acc_s, sw) and torch.equal(acc_m, mw)
    # finalize rank 0's query from the reduced tables: must equal the oracle on the whole table
    s0, m0 = queries[0]._bound
    s0.copy_(acc_s)
    m0.copy_(acc_m)
    torch.cuda.synchronize()
    gres = queries[0].finalize()
    info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in names}
    ores = oracle.run_query(parity.oracle_synth_cols(oracle, names, total, 0, total), n_threads=4,
                            **parity.oracle_query_kwargs(names, info, q))
    parity.compare(gres, ores, op="hist", full=True, n_aggs=2)
    gres.free()
    for x in queries + [qw]:
        x.free()
    for t in parts + [whole]:
        t.free()


def test_inlibrary_rccl_single_rank(ctx):
    """sybl_comm_* / sybl_query_allreduce with a 1-rank communicator: exercises the RCCL link."""
    from sybil_amd import synth
    uid = ctx.comm_unique_id()
    ctx.comm_init(uid, 1, 0)
    t = ctx.synth_table("r", synth.SEED, 200_000, 0, 200_000, synth.synth_cols(["c01", "c07"]))
    q = t.query(groups=["c01"], aggs=["c07"])
    q.scan()
    q.allreduce()
    r = q.finalize()
    assert r.matched == 200_000 and sum(x["count"] for x in r.results) == 200_000
    r.free()
    q.free()
    t.free()
    ctx.comm_free()


@pytest.mark.parametrize("slices32", [True, False])
def test_inlibrary_scatter_and_outlier_gather_single_rank(ctx, oracle, monkeypatch, slices32):
    """The N > 1 forms of the in-library merge with a 1-rank communicator (SYBL_FORCE_SCATTER): the bucket table leaves
    through the reduce-scatter -- as int32 slices (k_pack32 / k_unpack32) or, SYBL_NO_SCATTER32, as int64 -- with the
    collective finalize behind it, and a query with outliers has its outlier log gathered (rccl.cpp)."""
    from sybil_amd import synth
    monkeypatch.setenv("SYBL_FORCE_SCATTER", "1")
    if not slices32:
        monkeypatch.setenv("SYBL_NO_SCATTER32", "1")
    ctx.comm_init(ctx.comm_unique_id(), 1, 0)
    try:
        n = 500_000
        wl = _wl("cfg4_hist_highcard")
        t = ctx.synth_table("rs", synth.SEED, n, 0, n, synth.synth_cols(wl["columns"]))
        q = t.query(**dict(wl["query"], limit=40))
        q.scan()
        q.allreduce()
        assert q.collective_finalize()
        r = q.finalize()
        ocols = parity.oracle_synth_cols(oracle, wl["columns"], n, 0, n)
        info = {c: (synth.COLUMNS[c][4], synth.COLUMNS[c][5]) for c in wl["columns"]}
        o = oracle.run_query(ocols, n_threads=4, **parity.oracle_query_kwargs(wl["columns"], info, wl["query"]))
        omap = {x["key"]: x for x in o["results"]}
        rows = r.results
        assert r.matched == o["matched"] and len(rows) == len(omap)
        for i, g in enumerate(rows):
            oh, h = omap[g["key"]]["hists"][0], g["hists"][0]
            assert (g["count"], h["sum"]) == (omap[g["key"]]["count"], oh["sum_exact"])
            assert np.array_equal(h["percentiles"], oh["percentiles"])
            if i < 40:
                assert np.array_equal(h["values"], oh["values"])
        r.free()
        q.free()
        t.free()
        # outliers (the table of test_outlier_values_are_logged...): the log is gathered, the values stay available
        rng = np.random.default_rng(77)
        m = 150_000
        g = rng.integers(0, 5, size=m).astype(np.int64)
        v = rng.integers(0, 4000, size=m).astype(np.int64)
        v[rng.random(m) < 0.01] += 50_000
        tb = ctx.create_table("ro")
        tb.add_column("g", "int")
        tb.add_column("v", "int", 0, 60_000)
        for r0 in range(0, m, 40_000):
            tb.append_block(min(40_000, m - r0), {"g": g[r0:r0 + 40_000], "v": v[r0:r0 + 40_000]})
        qd = dict(groups=["g"], aggs=["v"], op="hist", hist_bucket=3, want_percentiles=True)
        q = tb.query(**qd)
        q.scan()
        q.allreduce()
        r = q.finalize()
        o = oracle.run_query([{"type": "int", "data": g}, {"type": "int", "data": v}], groups=[0], aggs=[(1, 0, 60_000)], op="hist",
                             hist_bucket=3, block_rows=40_000, n_threads=2)
        parity.compare(r, o, op="hist", full=True, n_aggs=1)  # includes the outlier values
        assert all(x["hists"][0]["n_outliers"] > 0 and x["hists"][0]["n_outlier_values"] == x["hists"][0]["n_outliers"] for x in r.results)
        r.free()
        q.free()
        tb.free()
    finally:
        ctx.comm_free()


# ---------------------------------------------------------------- full BASELINE size: properties

def test_full_size_properties(ctx):
    """1B rows x the 7 referenced columns of config 3 (56 GB in HBM): size-independent checks --
    matched = sum of group counts = an independent count(*) query; group sums add up to the
    ungrouped sum; two half-table scans add up to the whole; sampled rows equal the oracle's
    generator (checked in test_generator_matches_oracle)."""
    from sybil_amd import synth
    info = ctx.device_info()
    wl = _wl("cfg3_filter3_group2_stddev")
    rows = wl["rows"]
    if info["hbm_bytes"] < 80e9:
        rows = 100_000_000
    t = ctx.synth_table("full", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    q = t.query(**wl["query"])
    r = q.run()
    st = q.stats()
    assert st["rows_scanned"] == rows and st["algorithmic_bytes"] == rows * 56
    groups = r.results
    assert len(groups) == 1024
    assert sum(g["count"] for g in groups) == r.matched == r.cumulative["count"]
    assert abs(r.matched / rows - 0.512) < 0.001            # 0.8^3 selectivity
    for a in range(2):
        assert sum(g["hists"][a]["sum"] for g in groups) == r.cumulative["hists"][a]["sum"]
    # independent count(*) with the same predicates, no grouping
    qc = t.query(filters=wl["query"]["filters"], aggs=["c07", "c08"], op="avg")
    rc = qc.run()
    assert rc.matched == r.matched
    for a in range(2):
        assert rc.results[0]["hists"][a]["sum"] == r.cumulative["hists"][a]["sum"]
    total_sums = [r.cumulative["hists"][a]["sum"] for a in range(2)]
    matched = r.matched
    key_counts = {g["key"]: g["count"] for g in groups}
    full = {g["key"]: (g["count"],) + tuple((h["sum"], h["count"], h["stddev"], h["avg"]) for h in g["hists"]) for g in groups}
    for x in (r, rc):
        x.free()
    for x in (q, qc):
        x.free()
    # the same table in compact storage (16 stored bytes per row, k_scan_packed): identical integers,
    # hence identical floats
    t.compact()
    qk = t.query(**wl["query"])
    rk = qk.run()
    st = qk.stats()
    assert st["packed_kernel"] == 1 and st["algorithmic_bytes"] == rows * 16 and st["canonical_bytes"] == rows * 56
    assert rk.matched == matched
    assert {g["key"]: (g["count"],) + tuple((h["sum"], h["count"], h["stddev"], h["avg"]) for h in g["hists"])
            for g in rk.results} == full
    rk.free()
    qk.free()
    t.free()
    # halves add up (what the multi-GPU merge relies on)
    acc = {}
    m2 = 0
    s2 = [0, 0]
    for rank in range(2):
        row0, n = synth.shard(rows, rank, 2)
        th = ctx.synth_table("half", synth.SEED, rows, row0, n, synth.synth_cols(wl["columns"]))
        if rank == 1:
            th.compact()  # ranks may use different storage: the partial tables still add up
        qh = th.query(**wl["query"])
        rh = qh.run()
        m2 += rh.matched
        for g in rh.results:
            acc[g["key"]] = acc.get(g["key"], 0) + g["count"]
        for a in range(2):
            s2[a] += rh.cumulative["hists"][a]["sum"]
        rh.free()
        qh.free()
        th.free()
    assert m2 == matched and s2 == total_sums and acc == key_counts


def test_generator_matches_oracle(ctx, oracle):
    from sybil_amd import synth
    names = ["c00", "c01", "c03", "c07", "c08", "c09", "c20"]
    total = 1_000_000_000
    row0, n = 999_000_000, 1_000_000
    t = ctx.synth_table("gen", synth.SEED, total, row0, n, synth.synth_cols(names))
    ocols = parity.oracle_synth_cols(oracle, names, total, row0, n)
    for name, oc in zip(names, ocols):
        got = t.read_int(name, 0, n)
        assert np.array_equal(got, oc["data"]), name
        ci = t.column_info(name)
        assert ci["exact_min"] == int(oc["data"].min()) and ci["exact_max"] == int(oc["data"].max())
    t.free()


# ---------------------------------------------------------------- sparse group keys (dictionary)

def test_group_by_sparse_keys(ctx, oracle):
    """Key ranges far beyond direct mapping (40-bit ids): digits are ranks among the distinct values."""
    rng = np.random.default_rng(11)
    n = 120_000
    pool = np.unique(rng.integers(-(1 << 40), 1 << 40, size=3000))
    pool = np.concatenate([pool, [-1]])
    uid = pool[rng.integers(0, pool.size, size=n)].astype(np.int64)
    upop = (rng.random(n) > 0.1).astype(np.uint8)
    g2 = rng.integers(0, 5, size=n).astype(np.int64)
    v = rng.integers(0, 100_000, size=n).astype(np.int64)
    tb = ctx.create_table("sparse")
    tb.add_column("uid", "int")
    tb.add_column("g2", "int")
    tb.add_column("v", "int", 0, 99_999)
    _append_in_blocks(tb, n, 8192, {"uid": (uid, upop), "g2": g2, "v": v})
    ocols = [{"type": "int", "data": uid, "populated": upop}, {"type": "int", "data": g2}, {"type": "int", "data": v}]
    names = ["uid", "g2", "v"]
    info = {"v": (0, 99_999)}
    assert np.array_equal(tb.column_distinct("uid"), np.unique(uid[upop == 1]))
    for q in (dict(groups=["uid"], aggs=["v"], op="avg"),
              dict(groups=["g2", "uid"], aggs=["v"], op="hist", want_percentiles=False),
              dict(filters=[("v", "gt", 50_000)], groups=["uid", "g2"])):
        query = tb.query(**q)
        gres = query.run()
        ores = oracle.run_query(ocols, block_rows=8192, **parity.oracle_query_kwargs(names, info, q))
        parity.compare(gres, ores, op=q.get("op", "avg"), full=False, n_aggs=len(q.get("aggs", [])))
        # canonical order = key order; -1 and missing share the MISSING_VALUE group and print as ""
        if q["groups"] == ["uid"]:
            keys = [int.from_bytes(r["key"], "little", signed=True) for r in sorted(gres.results, key=lambda r: -r["count"])]
            assert len(keys) == len(set(keys))
            assert any(r["group_by_key"] == "\t" for r in gres.results)
        gres.free()
        query.free()
    tb.free()


@pytest.mark.parametrize("compact", [False, True])
def test_dictionary_digit_keys_run_the_specialised_bodies_through_a_rank_column(ctx, oracle, monkeypatch, compact):
    """Round 5: a sparse int key's digits (ranks in the column's distinct values) used to be found by probing the dictionary
    per row in the plan-interpreting k_scan (0.07 of peak).  The ranks are now laid out once as a narrow derived column
    (k_rank_column, Column::rank_col) that the group-by direct-maps through: the query takes the same row bodies as any
    dense key -- k_scan_packed over a compact table -- and SYBL_NO_RANKCOL=1 (the probing path) must agree with it.  The
    key is also filtered and aggregated in one of the queries (its own values stay in a slot of their own), has missing
    rows and the value -1 (the MISSING_VALUE group), and the table grows between two queries (the ranks are rebuilt)."""
    rng = np.random.default_rng(21)
    n = 400_000
    pool = np.unique(rng.integers(-(1 << 30), 1 << 30, size=900))
    pool = np.concatenate([pool, [-1]])
    uid = pool[rng.integers(0, pool.size, size=n)].astype(np.int64)
    upop = (rng.random(n) > 0.05).astype(np.uint8)
    f = rng.integers(0, 1000, size=n).astype(np.int64)
    v = rng.integers(0, 100_000, size=n).astype(np.int64)
    tb = ctx.create_table("rk")
    tb.add_column("uid", "int")
    tb.add_column("f", "int")
    tb.add_column("v", "int", 0, 99_999)
    half = n // 2
    _append_in_blocks(tb, half, 65536, {"uid": (uid[:half], upop[:half]), "f": f[:half], "v": v[:half]})
    if compact:
        tb.compact()
    names = ["uid", "f", "v"]
    info = {"v": (0, 99_999), "uid": (-(1 << 30), 1 << 30), "f": (0, 999)}
    queries = (dict(filters=[("f", "gt", 99), ("f", "lt", 900)], groups=["uid"], aggs=["v"], op="hist", want_percentiles=False),
               dict(groups=["uid"], aggs=["v"], op="avg"),
               # (the key's own values are filtered on -- they keep a slot of their own next to the rank column's)
               dict(filters=[("uid", "gt", 0)], groups=["uid"], aggs=["f", "v"], op="avg"),
               # ... and aggregated: negative values in avg mode track a minimum, which only the plan interpreter does
               dict(filters=[("uid", "lt", 1 << 29)], groups=["uid"], aggs=["uid", "v"], op="avg"))

    def check(rows):
        ocols = [{"type": "int", "data": uid[:rows], "populated": upop[:rows]}, {"type": "int", "data": f[:rows]}, {"type": "int", "data": v[:rows]}]
        for q in queries:
            ores = oracle.run_query(ocols, block_rows=65536, **parity.oracle_query_kwargs(names, info, q))
            seen = {}
            for off in (False, True):
                if off:
                    monkeypatch.setenv("SYBL_NO_RANKCOL", "1")
                query = tb.query(**q)
                if off:
                    monkeypatch.delenv("SYBL_NO_RANKCOL")
                gres = query.run()
                st = query.stats()
                seen[off] = (st["strategy"], st["packed_kernel"])
                parity.compare(gres, ores, op=q["op"], full=False, n_aggs=len(q["aggs"]))
                gres.free()
                query.free()
            assert seen[True][0] in (0, 1), seen      # the probing path: the plan interpreter
            if "uid" not in q["aggs"]:
                assert seen[False][0] == 2, seen      # the rank column: a role-specialised body ...
                if compact:
                    assert seen[False][1] == 1, seen  # ... the packed one when every column it reads is narrow

    check(half)
    _append_in_blocks(tb, n - half, 65536, {"uid": (uid[half:], upop[half:]), "f": f[half:], "v": v[half:]})
    check(n)
    tb.free()


@pytest.mark.parametrize("compact", [False, True])
def test_avg_over_negative_values_tracks_the_minimum_in_the_specialised_bodies(ctx, oracle, monkeypatch, compact):
    """`-op avg` starts BasicHist.Min / Max at Go's zero value (hist_basic.go:72-85): a column with negative values needs its
    minimum tracked, one that never exceeds 0 needs no maximum.  Round 5: the role-specialised bodies do both (as max(-v) in
    the MAX section, FastPlan::ext_general) -- such queries used to fall to the plan interpreter.  Three aggregations: mixed
    signs (min and max), all negative (min only), all positive (max only); with SYBL_NO_FAST_MIN=1 the old path must agree."""
    rng = np.random.default_rng(33)
    n = 300_000
    g = rng.integers(0, 40, size=n).astype(np.int64)
    f = rng.integers(0, 1000, size=n).astype(np.int64)
    a = rng.integers(-50_000, 50_000, size=n).astype(np.int64)
    b = rng.integers(-90_000, -10, size=n).astype(np.int64)
    c = rng.integers(5, 70_000, size=n).astype(np.int64)
    bpop = (rng.random(n) > 0.1).astype(np.uint8)
    tb = ctx.create_table("mn")
    for name in ("g", "f", "a", "b", "c"):
        tb.add_column(name, "int")
    _append_in_blocks(tb, n, 65536, {"g": g, "f": f, "a": a, "b": (b, bpop), "c": c})
    if compact:
        tb.compact()
    ocols = [{"type": "int", "data": g}, {"type": "int", "data": f}, {"type": "int", "data": a}, {"type": "int", "data": b, "populated": bpop},
             {"type": "int", "data": c}]
    names = ["g", "f", "a", "b", "c"]
    info = {"a": (-50_000, 49_999), "b": (-90_000, -11), "c": (5, 69_999)}
    for q in (dict(filters=[("f", "gt", 99), ("f", "lt", 900)], groups=["g"], aggs=["a", "c"], op="avg"),
              dict(groups=["g"], aggs=["a", "b"], op="avg"),
              dict(groups=["g"], aggs=["b"], op="avg")):
        ores = oracle.run_query(ocols, block_rows=65536, **parity.oracle_query_kwargs(names, info, q))
        for off in (False, True):
            if off:
                monkeypatch.setenv("SYBL_NO_FAST_MIN", "1")
            query = tb.query(**q)
            if off:
                monkeypatch.delenv("SYBL_NO_FAST_MIN")
            gres = query.run()
            assert query.stats()["strategy"] == (0 if off else 2), (q, off, query.stats())
            parity.compare(gres, ores, op="avg", n_aggs=len(q["aggs"]))
            assert all(h["min"] < 0 for r in gres.results for h in r["hists"][:1])
            gres.free()
            query.free()
    tb.free()


def test_sparse_keys_across_ranks(ctx, oracle):
    """Two shards see different subsets of the ids: with the union dictionary installed on both,
    their partial tables add up to the table of the whole."""
    import torch
    rng = np.random.default_rng(12)
    n = 60_000
    pool = np.unique(rng.integers(0, 1 << 45, size=500))
    uid = pool[rng.integers(0, pool.size, size=n)].astype(np.int64)
    uid[: n // 2] = np.where(uid[: n // 2] % 3 == 0, uid[: n // 2], pool[0])   # shard 0 misses most ids
    v = rng.integers(0, 1000, size=n).astype(np.int64)
    parts, acc = [], None
    union = None
    for rank in range(2):
        sl = slice(rank * n // 2, (rank + 1) * n // 2)
        tb = ctx.create_table("p%d" % rank)
        tb.add_column("uid", "int")
        tb.add_column("v", "int", 0, 999)
        tb.append_block(n // 2, {"uid": uid[sl], "v": v[sl]})
        parts.append(tb)
    union = np.unique(np.concatenate([p.column_distinct("uid") for p in parts]))
    queries = []
    for tb in parts:
        tb.set_group_dict("uid", union)
        tb.set_bounds("v", 0, 999)
        q = tb.query(groups=["uid"], aggs=["v"], op="avg")
        s, m = q.bind_torch("cuda:0")
        q.scan()
        ctx.sync()
        acc = (s.clone(), m.clone()) if acc is None else (acc[0] + s, torch.maximum(acc[1], m))
        queries.append(q)
    s0, m0 = queries[0]._bound
    s0.copy_(acc[0])
    m0.copy_(acc[1])
    torch.cuda.synchronize()
    gres = queries[0].finalize()
    ores = oracle.run_query([{"type": "int", "data": uid}, {"type": "int", "data": v}], groups=[0], aggs=[(1, 0, 999)])
    parity.compare(gres, ores, op="avg", n_aggs=1)
    gres.free()
    for q in queries:
        q.free()
    for tb in parts:
        tb.free()


def test_str_dictionaries_across_ranks(ctx, oracle):
    """Two shards built their str dictionaries in different first-seen orders; after installing
    the union on both (sybil_amd.dist.agree_str_dict's protocol) their partial tables add up."""
    import torch
    rng = np.random.default_rng(21)
    n = 40_000
    vocab = ["host%02d" % i for i in range(30)]
    ids = rng.integers(0, 30, size=n)
    ids[: n // 2] = ids[: n // 2] % 11            # shard 0 never sees most hosts
    v = rng.integers(0, 1000, size=n).astype(np.int64)
    tags_off = np.arange(0, n + 1, dtype=np.int64)
    tag_ids = (ids % 4).astype(np.int32)
    tag_names = ["t0", "t1", "t2", "t3"]
    parts = []
    for rank in range(2):
        sl = slice(rank * n // 2, (rank + 1) * n // 2)
        perm = rng.permutation(30)
        inv = np.argsort(perm)
        tperm = rng.permutation(4)
        tinv = np.argsort(tperm)
        tb = ctx.create_table("r%d" % rank)
        tb.add_column("host", "str")
        tb.add_column("v", "int", 0, 999)
        tb.add_column("tags", "set")
        tb.append_block(n // 2, {"host": {"ids": inv[ids[sl]].astype(np.int32), "strings": [vocab[i] for i in perm]},
                                 "v": v[sl],
                                 "tags": {"ids": tinv[tag_ids[sl]].astype(np.int32), "offsets": tags_off[: n // 2 + 1],
                                          "strings": [tag_names[i] for i in tperm]}})
        parts.append(tb)
    union = sorted(set(s for p in parts for s in p.column_dict("host")))
    assert union == vocab
    tunion = sorted(set(s for p in parts for s in p.column_dict("tags")))
    queries, acc = [], None
    for tb in parts:
        tb.set_dict("host", union)
        tb.set_dict("tags", tunion)
        tb.set_bounds("v", 0, 999)
        q = tb.query(filters=[("tags", "nin", "t1"), ("host", "neq", "host03")], groups=["host"], aggs=["v"], op="avg")
        s, m = q.bind_torch("cuda:0")
        q.scan()
        ctx.sync()
        acc = (s.clone(), m.clone()) if acc is None else (acc[0] + s, torch.maximum(acc[1], m))
        queries.append(q)
    s0, m0 = queries[0]._bound
    s0.copy_(acc[0])
    m0.copy_(acc[1])
    torch.cuda.synchronize()
    gres = queries[0].finalize()
    ores = oracle.run_query([{"type": "str", "data": ids.astype(np.int32)}, {"type": "int", "data": v},
                             {"type": "set", "data": tag_ids, "offsets": tags_off}],
                            filters=[(2, "nin", 1), (0, "neq", 3)], groups=[0], aggs=[(1, 0, 999)])
    gmap = {r["group_by_key"]: r for r in gres.results}
    omap = {vocab[r["key_vals"][0]] + "\t": r for r in ores["results"]}
    assert gres.matched == ores["matched"] and set(gmap) == set(omap)
    for k, o in omap.items():
        assert gmap[k]["count"] == o["count"] and gmap[k]["hists"][0]["sum"] == o["hists"][0]["sum_exact"]
    import sybil_amd
    with pytest.raises(sybil_amd.SyblError):
        parts[0].set_dict("host", ["host00"])       # must contain every resident value
    gres.free()
    for q in queries:
        q.free()
    for tb in parts:
        tb.free()


def test_weighted_and_nullable_take_the_fast_kernels(ctx, oracle):
    """-weight-col, missing rows, the Info.Min/Max reject gate and str group keys are all inside the
    role-specialised (GEN) kernels: same results as the oracle, strategy 2."""
    n = 90_000
    rng = np.random.default_rng(31)
    g = rng.integers(0, 12, size=n).astype(np.int64)
    gpop = (rng.random(n) > 0.1).astype(np.uint8)
    sid = rng.integers(0, 7, size=n).astype(np.int32)
    strings = ["k%d" % i for i in range(7)]
    v = rng.integers(0, 50_000, size=n).astype(np.int64)
    vpop = (rng.random(n) > 0.2).astype(np.uint8)
    f = rng.integers(0, 1000, size=n).astype(np.int64)
    w = rng.integers(1, 9, size=n).astype(np.int64)
    tb = ctx.create_table("wn")
    tb.add_column("g", "int")
    tb.add_column("s", "str")
    tb.add_column("v", "int", 100, 40_000)        # IntInfo narrower than the data: rejects on both sides
    tb.add_column("f", "int")
    tb.add_column("w", "int")
    tb.add_column("v2", "int", 0, 59_999)         # nullable, IntInfo covers the data: hist without outliers
    _append_in_blocks(tb, n, 7000, {"g": (g, gpop), "s": {"ids": sid, "strings": strings}, "v": (v, vpop), "f": f, "w": w,
                                     "v2": (v, vpop)})
    ocols = [{"type": "int", "data": g, "populated": gpop}, {"type": "str", "data": sid},
             {"type": "int", "data": v, "populated": vpop}, {"type": "int", "data": f}, {"type": "int", "data": w},
             {"type": "int", "data": v, "populated": vpop}]
    names = ["g", "s", "v", "f", "w", "v2"]
    info = {"v": (100, 40_000), "v2": (0, 59_999)}
    for q in (dict(filters=[("f", "gt", 100), ("f", "lt", 800)], groups=["g", "s"], aggs=["v"], op="avg", weight_col="w"),
              dict(groups=["s"], aggs=["v2"], op="hist", want_percentiles=False, weight_col="w"),
              dict(filters=[("f", "lt", 500)], groups=["g"], aggs=["v2"], op="hist", want_percentiles=False),
              dict(groups=["s", "g"], aggs=["v"], op="avg"),
              dict(filters=[("s", "neq", "k3"), ("f", "gt", 50)], groups=["g"], aggs=["v"], op="avg"),      # str filter as id mask
              dict(filters=[("s", "re", "k[0-2]")], groups=["s"], aggs=["v2"], op="hist", want_percentiles=False),
              # IntInfo [100, 40000] vs data [0, 50000): rejects below, h.Max above, outliers beyond the last bucket
              dict(groups=["s"], aggs=["v"], op="hist", want_percentiles=False),
              dict(groups=["g"], aggs=["v"], op="hist", want_percentiles=True)):
        query = tb.query(**q)
        assert query.stats()["strategy"] == (6 if q.get("want_percentiles", False) else 2), q
        gres = query.run()
        okw = parity.oracle_query_kwargs(names, info, q)
        import re as _re
        okw["filters"] = [(f[0], f[1], strings.index(f[2])) if f[1] in ("eq", "neq") and isinstance(f[2], str) else
                          ((f[0], f[1], 0, np.array([bool(_re.search(f[2], x)) for x in strings], dtype=np.uint8))
                           if f[1] in ("re", "nre") else f) for f in okw["filters"]]
        ores = oracle.run_query(ocols, block_rows=7000, **okw)
        # str group ids are engine-private but here the block dictionary order == id order
        parity.compare(gres, ores, op=q["op"], full=q.get("want_percentiles", False), n_aggs=1)
        gres.free()
        query.free()
    tb.free()


def test_snapshot_lets_finalize_overlap_the_next_scan(ctx, oracle):
    """sybl_query_snapshot: two prepared queries alternate; finalize(i) is called only after scan(i+1)
    was launched and must still see step i's table (and wait for nothing but its own snapshot)."""
    wl = _wl("cfg3_filter3_group2_stddev")
    from sybil_amd import synth
    rows = 700_000
    t = ctx.synth_table("snap", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    qa = t.query(**wl["query"])
    qb = t.query(**dict(wl["query"], filters=[("c04", "gt", 499)]))  # a different query: results must not mix
    ra = qa.run()
    rb = qb.run()
    want_a = sorted((g["key"], g["count"], g["hists"][0]["sum"]) for g in ra.results)
    want_b = sorted((g["key"], g["count"], g["hists"][0]["sum"]) for g in rb.results)
    assert want_a != want_b
    ra.free()
    rb.free()
    pending = None
    for i in range(6):
        q, want = (qa, want_a) if i % 2 == 0 else (qb, want_b)
        q.scan().snapshot()
        if pending is not None:
            pq, pwant = pending
            r = pq.finalize()
            assert sorted((g["key"], g["count"], g["hists"][0]["sum"]) for g in r.results) == pwant
            r.free()
        pending = (q, want)
    r = pending[0].finalize()
    assert sorted((g["key"], g["count"], g["hists"][0]["sum"]) for g in r.results) == pending[1]
    r.free()
    with pytest.raises(Exception):
        t.query(**wl["query"]).snapshot()  # before any scan
    qa.free()
    qb.free()
    t.free()


def test_many_groups_with_limit_fetches_only_printed_bucket_arrays(ctx, oracle):
    """65 536 groups x 1002 buckets with -limit 25: percentiles / stddev of EVERY row come from the GPU
    summary kernels, bucket arrays cross PCIe only for the 25 rows a printer shows."""
    wl = _wl("cfg4_hist_highcard")
    q = dict(wl["query"], limit=25, order_by="$COUNT")
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"], 500_000, 0, 500_000, q, compact=True)
    rows = gres.results
    omap = {r["key"]: r for r in ores["results"]}
    assert len(rows) == len(omap) == 65536 or len(rows) == len(omap)
    counts = [r["count"] for r in rows]
    assert counts == sorted(counts, reverse=True)
    for i, g in enumerate(rows):
        o = omap[g["key"]]
        assert g["count"] == o["count"]
        gh, oh = g["hists"][0], o["hists"][0]
        assert gh["sum"] == oh["sum_exact"] and gh["count"] == oh["count"]
        assert np.array_equal(gh.get("percentiles", np.zeros(0, dtype=np.int64)), oh["percentiles"]), g["key"]
        assert parity._close(gh["stddev"], oh["stddev_exact"], 1e-9, max(abs(oh["avg"]), oh["bucket_size"], 1.0))
        if i < 25:
            assert np.array_equal(gh["values"], oh["values"])
        else:
            assert "values" not in gh
    gc, oc = gres.cumulative, ores["cumulative"]
    parity.compare_hist(gc["hists"][0], oc["hists"][0], "hist", True, ctx="cumulative")
    # the text / JSON renderers only touch the printed rows
    assert gres.render("json").count('"buckets"') >= 25
    gres.free()
    # ordered by the mean of the aggregated column (radix path, float keys), ascending
    q2 = dict(wl["query"], limit=10, order_by="c07", order_asc=True)
    gres, ores, _ = parity.run_both(ctx, oracle, wl["columns"], 500_000, 0, 500_000, q2, compact=False)
    avgs = [r["hists"][0]["avg"] for r in gres.results]
    assert avgs == sorted(avgs) and len(avgs) == len(ores["results"])
    assert sorted(r["key"] for r in gres.results) == sorted(r["key"] for r in ores["results"])
    gres.free()


@pytest.mark.parametrize("compact", [False, True])
def test_weight_column_with_unpopulated_rows_carries_the_last_weight(ctx, oracle, compact):
    """aggregate.go:68,100-102 (refused until round 5): a row without a weight aggregates with the last weight seen in its
    block, 1 before the first.  k_weight_carry lays the weights in force out as a dense column of the query's own; every
    row body that takes a weight column then runs unchanged -- the GEN kernels (avg, moments), the LDS histograms, the
    generic kernel (weighted bucket arrays, the weight column also filtered / aggregated), a time series, the hash table."""
    n, block = 200_000, 7000
    rng = np.random.default_rng(41)
    g = rng.integers(0, 12, size=n).astype(np.int64)
    v = rng.integers(0, 50_000, size=n).astype(np.int64)
    vpop = (rng.random(n) > 0.2).astype(np.uint8)
    f = rng.integers(0, 1000, size=n).astype(np.int64)
    t = (1_700_000_000 + np.sort(rng.integers(0, 86_400 * 3, size=n))).astype(np.int64)
    w = rng.integers(1, 9, size=n).astype(np.int64)
    wpop = (rng.random(n) > 0.7).astype(np.uint8)   # most rows inherit
    wpop[block * 3:block * 3 + 900] = 0             # a block that starts without a weight
    wpop[block * 5:block * 6] = 0                   # a block without any
    tb = ctx.create_table("wc")
    tb.add_column("g", "int")
    tb.add_column("v", "int", 0, 49_999)
    tb.add_column("f", "int")
    tb.add_column("t", "int")
    tb.add_column("w", "int")
    _append_in_blocks(tb, n, block, {"g": g, "v": (v, vpop), "f": f, "t": t, "w": (w, wpop)})
    if compact:
        tb.compact()
    ocols = [{"type": "int", "data": g}, {"type": "int", "data": v, "populated": vpop}, {"type": "int", "data": f}, {"type": "int", "data": t},
             {"type": "int", "data": w, "populated": wpop}]
    names = ["g", "v", "f", "t", "w"]
    info = {"v": (0, 49_999), "w": (1, 8)}
    import os
    for q, env in ((dict(filters=[("f", "gt", 100), ("f", "lt", 800)], groups=["g"], aggs=["v"], op="avg", weight_col="w"), {}),
                   (dict(groups=["g"], aggs=["v"], op="hist", want_percentiles=False, weight_col="w"), {}),
                   (dict(groups=["g"], aggs=["v"], op="hist", want_percentiles=True, weight_col="w"), {}),
                   (dict(filters=[("w", "lt", 7)], groups=["g"], aggs=["v", "w"], op="avg", weight_col="w"), {}),   # the column in three roles
                   (dict(groups=["g"], aggs=["v"], op="avg", weight_col="w", time_col="t", time_bucket=3600), {}),
                   (dict(groups=["g", "f"], aggs=["v"], op="avg", weight_col="w"), {"SYBL_FORCE_HASH": "1"})):
        os.environ.update(env)
        try:
            query = tb.query(**q)
        finally:
            for k in env:
                del os.environ[k]
        gres = query.run()
        ores = oracle.run_query(ocols, block_rows=block, **parity.oracle_query_kwargs(names, info, q))
        parity.compare(gres, ores, op=q["op"], full=q.get("want_percentiles", False), n_aggs=len(q["aggs"]), time_mode=bool(q.get("time_col")))
        gres.free()
        query.free()
    tb.free()


@pytest.mark.parametrize("order", [dict(order_by="$COUNT"), dict(order_by="c07", order_asc=True)])
def test_printers_query_summarises_only_the_printed_rows(ctx, oracle, order):
    """sybl_query_desc.printed_only (ABI 4): the reference calls GetPercentiles / GetStdDev at print time, for the `Limit`
    rows it prints (printer.go:60-76,291-308).  A printer's query over 65 536 groups x 1002 buckets therefore runs no
    summary pass over every group: the printed rows' percentiles and stddev come from their own bucket arrays, Cumulative's
    from k_hist_total; the other rows carry count / sum / avg / extrema only.  Both printers must not notice."""
    from sybil_amd import synth
    wl = _wl("cfg4_hist_highcard")
    rows = 500_000
    t = ctx.synth_table("pr", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    q = dict(wl["query"], limit=25, **order)
    full_q = t.query(**q)
    full = full_q.run()
    pr_q = t.query(**dict(q, printed_only=True))
    pr = pr_q.run()
    assert pr_q.stats()["strategy"] == 5
    info = {c: (synth.COLUMNS[c][4], synth.COLUMNS[c][5]) for c in wl["columns"]}
    o = oracle.run_query(parity.oracle_synth_cols(oracle, wl["columns"], rows, 0, rows), n_threads=4, **parity.oracle_query_kwargs(wl["columns"], info, q))
    omap = {r["key"]: r for r in o["results"]}
    got, ref = pr.results, full.results
    assert [g["key"] for g in got] == [g["key"] for g in ref] and len(got) == len(omap)
    for i, g in enumerate(got):
        oh, h = omap[g["key"]]["hists"][0], g["hists"][0]
        assert (g["count"], h["count"], h["sum"], h["min"], h["max"]) == (omap[g["key"]]["count"], oh["count"], oh["sum_exact"], oh["min"], oh["max"])
        assert h["avg"] == ref[i]["hists"][0]["avg"]
        if i < 25:
            assert np.array_equal(h["values"], oh["values"]) and np.array_equal(h["percentiles"], oh["percentiles"])
            assert parity._close(h["stddev"], oh["stddev_exact"], 1e-9, max(abs(oh["avg"]), oh["bucket_size"], 1.0))
            assert parity._close(h["stddev"], ref[i]["hists"][0]["stddev"], 1e-9, max(abs(oh["avg"]), oh["bucket_size"], 1.0))
        else:
            assert "values" not in h and "percentiles" not in h and h["stddev"] != h["stddev"]
    parity.compare_hist(pr.cumulative["hists"][0], o["cumulative"]["hists"][0], "hist", True, ctx="cumulative", cumulative=True)
    # (the summary path derives stddev from exact bucket moments in long double, the printed rows' own arrays go through
    # GetStdDev's float loop: the last digits may differ, so the printers are compared on everything but that field)
    import json
    import re
    def strip(s):
        return re.sub(r'"(stddev|std)":\s*[-0-9.eE+]+', '"stddev": 0', s)
    assert json.loads(strip(pr.render("json"))) == json.loads(strip(full.render("json")))
    pr.free()
    full.free()
    pr_q.free()
    full_q.free()
    t.free()


def test_outlier_values_are_kept_and_printed(ctx, oracle, monkeypatch):
    """-hist-bucket leaves NumBuckets at 1000, so values beyond Min + 1001 * bucket are Outliers: clipped into the last
    bucket AND remembered (hist_basic.go:132-142).  GetStrBuckets prints each under its own value (:239-257) and the gob
    HistCompat carries the list.  The engine keeps every block's outliers (the reference: one block's, Combine does
    not merge them)."""
    import json
    import sybil_amd
    rng = np.random.default_rng(77)
    n = 150_000
    g = rng.integers(0, 5, size=n).astype(np.int64)
    v = rng.integers(0, 4000, size=n).astype(np.int64)
    v[rng.random(n) < 0.01] += 50_000  # far outliers
    cols = [{"type": "int", "data": g}, {"type": "int", "data": v}]
    for compact, generic in ((False, False), (True, False), (False, True)):
        if generic:
            monkeypatch.setenv("SYBL_NO_FAST", "1")  # the plan-interpreting kernel instead of the role-specialised one
        tb = ctx.create_table("o")
        tb.add_column("g", "int")
        tb.add_column("v", "int", 0, 60_000)
        for r0 in range(0, n, 40_000):
            tb.append_block(min(40_000, n - r0), {"g": g[r0:r0 + 40_000], "v": v[r0:r0 + 40_000]})
        if compact:
            tb.compact()
        q = dict(groups=["g"], aggs=["v"], op="hist", hist_bucket=3, want_percentiles=True)
        query = tb.query(**q)
        gres = query.run()
        ores = oracle.run_query(cols, groups=[0], aggs=[(1, 0, 60_000)], op="hist", hist_bucket=3, block_rows=40_000, n_threads=2)
        parity.compare(gres, ores, op="hist", full=True, n_aggs=1)  # includes the outlier values
        rows = {r["key_vals"][0]: r for r in gres.results}
        assert all(r["hists"][0]["n_outliers"] > 0 and r["hists"][0]["n_outlier_values"] == r["hists"][0]["n_outliers"] for r in rows.values())
        out = json.loads(gres.render("json"))
        for jr in out:
            h = rows[int(jr["g"])]["hists"][0]
            want = {}
            for b, c in enumerate(h["values"]):
                if c > 0:
                    want[str(b * 3)] = want.get(str(b * 3), 0) + int(c)
            for x in h["outlier_values"]:
                want[str(int(x))] = want.get(str(int(x)), 0) + 1
            assert jr["v"]["buckets"] == want
        from tests import gobfmt
        enc = gobfmt.decode(gres.encode())["QuerySpec"]["QueryResults"]["Results"]
        for key, e in enc.items():
            ci = e["Hists"]["v"]["value"]["BasicHist"]["BasicHistCachedInfo"]
            assert ci["Outliers"] == rows[int(key.strip())]["hists"][0]["outlier_values"].tolist()
        gres.free()
        query.free()
        # a log too small for the outliers: counts and sigma stay exact, the values are reported as unavailable and the
        # printers refuse instead of showing buckets without them
        monkeypatch.setenv("SYBL_OUTLIER_LOG_CAP", "16")
        query = tb.query(**q)
        gres = query.run()
        monkeypatch.delenv("SYBL_OUTLIER_LOG_CAP")
        parity.compare(gres, ores, op="hist", full=True, n_aggs=1)
        assert all(r["hists"][0]["n_outlier_values"] == -1 for r in gres.results)
        with pytest.raises(sybil_amd.SyblError):
            gres.render("json")
        with pytest.raises(sybil_amd.SyblError):
            gres.encode()
        assert "g" not in gres.render("text")[:1]  # (the text table prints no buckets: it still renders)
        gres.free()
        query.free()
        tb.free()


def test_str_replace_rewrites_the_dictionary_for_filters_and_groups(ctx, oracle):
    """-str-replace col:pattern:replacement (column_store_io.go:517-545): str filters and group-by see the rewritten
    strings, and strings that become equal share a group.  Oracle: the same rows with the rewritten strings' ids."""
    import re
    rng = np.random.default_rng(61)
    n = 90_000
    hosts = ["web%02d.%s.example.com" % (i, dc) for i in range(40) for dc in ("ams", "sfo", "nrt")]
    ids = rng.integers(0, len(hosts), size=n).astype(np.int32)
    pop = (rng.random(n) > 0.05).astype(np.uint8)
    v = rng.integers(0, 5000, size=n).astype(np.int64)
    g = rng.integers(0, 3, size=n).astype(np.int64)
    for compact in (False, True):
        tb = ctx.create_table("sr")
        tb.add_column("host", "str")
        tb.add_column("g", "int")
        tb.add_column("v", "int", 0, 4999)
        for r0 in range(0, n, 32_768):
            r1 = min(r0 + 32_768, n)
            perm = rng.permutation(len(hosts))  # block-local dictionaries in scrambled order
            inv = np.argsort(perm)
            tb.append_block(r1 - r0, {"host": {"ids": inv[ids[r0:r1]].astype(np.int32), "strings": [hosts[i] for i in perm], "populated": pop[r0:r1]},
                                      "g": g[r0:r1], "v": v[r0:r1]})
        if compact:
            tb.compact()
        for pattern, templ, py in ((r"^web\d+\.", "", ""), (r"^(web\d)\d\.(\w+)\..*$", "$2-$1", r"\2-\1"), (r"\d", "#", "#")):
            rewritten = [re.sub(pattern, py, h) for h in hosts]
            uniq = sorted(set(rewritten))
            new_ids = np.array([uniq.index(rewritten[i]) for i in ids], dtype=np.int32)
            ocols = [{"type": "str", "data": new_ids, "populated": pop}, {"type": "int", "data": g}, {"type": "int", "data": v}]
            target = rewritten[7]
            cases = [
                (dict(groups=["host"], aggs=["v"], op="hist"), dict(groups=[0], aggs=[(2, 0, 4999)], op="hist")),
                (dict(filters=[("host", "eq", target)], groups=["g", "host"], aggs=["v"]),
                 dict(filters=[(0, "eq", uniq.index(target))], groups=[1, 0], aggs=[(2, 0, 4999)])),
                (dict(filters=[("host", "neq", target), ("host", "re", "s")], groups=["host"]),
                 dict(filters=[(0, "neq", uniq.index(target)), (0, "re", 0, np.array([bool(re.search("s", u)) for u in uniq], dtype=np.uint8))],
                      groups=[0])),
            ]
            for q, okw in cases:
                for sr in ([("host", pattern, templ)], [("host", rewritten_for(tb, pattern, py))]):
                    query = tb.query(str_replace=sr, **q)
                    gres = query.run()
                    ores = oracle.run_query(ocols, block_rows=32_768, n_threads=2, **okw)
                    assert gres.matched == ores["matched"]
                    hpos = q["groups"].index("host")

                    def tr(r, names):
                        kv = r["key_vals"][hpos]
                        return tuple("" if (i == hpos and kv == 0xFFFFFFFFFFFFFFFF) else (names[kv] if i == hpos else x)
                                     for i, x in enumerate(r["key_vals"]))
                    omap = {tr(r, uniq): r for r in ores["results"]}
                    gmap = {}
                    for r in gres.results:
                        parts = r["group_by_key"].split("\t")[:-1]
                        gmap[tuple(parts[i] if i == hpos else r["key_vals"][i] for i in range(len(parts)))] = r
                    assert set(gmap) == set(omap), (pattern, q)
                    for k, o in omap.items():
                        assert gmap[k]["count"] == o["count"]
                        for a in range(len(q.get("aggs", []))):
                            parity.compare_hist(gmap[k]["hists"][a], o["hists"][a], q.get("op", "avg"), True, ctx=(pattern, k))
                    gres.free()
                    query.free()
        tb.free()


def rewritten_for(tb, pattern, py):
    """the host-evaluated form of -str-replace: one rewritten string per table-global dictionary id"""
    import re
    return [re.sub(pattern, py, s) for s in tb.column_dict("host")]


@pytest.mark.parametrize("compact", [False, True])
def test_bucket_counters_that_wrap(ctx, oracle, compact):
    """k_part_hist keeps its bucket arrays as 16-bit LDS counters, two to a word, and logs every wrap for k_part_fix
    (csrc/kernels.hip).  One group holds most rows and two NEIGHBOURING buckets of it -- an even one and the odd one
    sharing its word -- take > 450 000 values each: both fields wrap several times and every wrap of the low field
    carries into the high one.  Buckets, Count and the exact sum must come out as the oracle's."""
    rng = np.random.default_rng(20260926)
    n = 1_300_000
    g = rng.integers(0, 4096, n, dtype=np.int64)
    g[rng.random(n) < 0.85] = 5
    v = rng.integers(0, 1_000_000, n, dtype=np.int64)
    u = rng.random(n)
    heavy = g == 5
    v[heavy & (u < 0.45)] = rng.integers(0, 999, int((heavy & (u < 0.45)).sum()))      # bucket 0 (BucketSize 999)
    v[heavy & (u >= 0.45) & (u < 0.9)] = 1000                                          # bucket 1
    tb = ctx.create_table("wrap")
    tb.add_column("g", "int", 0, 4095)
    tb.add_column("v", "int", 0, 999_999)
    for r0 in range(0, n, 65536):
        r1 = min(r0 + 65536, n)
        tb.append_block(r1 - r0, {"g": g[r0:r1], "v": v[r0:r1]})
    if compact:
        tb.compact()
    q = tb.query(groups=["g"], aggs=["v"], op="hist")
    r = q.run()
    assert q.stats()["strategy"] == 5
    o = oracle.run_query([{"type": "int", "data": g}, {"type": "int", "data": v}], groups=[0], aggs=[(1, 0, 999_999)], op="hist")
    parity.compare(r, o, op="hist", full=True, n_aggs=1)
    big = [x for x in o["results"] if x["count"] > 1_000_000]
    assert len(big) == 1 and big[0]["hists"][0]["values"][0] > 7 * 65536 and big[0]["hists"][0]["values"][1] > 7 * 65536
    r.free()
    q.free()
    tb.free()


def _append_all(tb, n, cols):
    for r0 in range(0, n, 65536):
        r1 = min(r0 + 65536, n)
        tb.append_block(r1 - r0, {k: v[r0:r1] for k, v in cols.items()})


@pytest.mark.parametrize("compact", [False, True])
def test_three_aggregations_rescanned_with_unequal_splits(ctx, oracle, compact):
    """Three aggregations over ~6000 groups: the first partitioned-histogram pass (two aggregations, > 128 partitions) gives
    every partition to one workgroup, the second (one aggregation, half the partitions) shares its partitions between
    workgroups and accumulates with atomics -- into a table that must start from zero on EVERY scan of the prepared query,
    not only on the first (the allocation is memset once)."""
    rng = np.random.default_rng(41)
    n = 900_000
    g = rng.integers(0, 6000, n, dtype=np.int64)
    a = rng.integers(0, 1_000_000, n, dtype=np.int64)
    b = rng.integers(0, 50_000, n, dtype=np.int64)
    c = rng.integers(0, 1_000_000, n, dtype=np.int64)
    tb = ctx.create_table("resplit")
    tb.add_column("g", "int", 0, 5999)
    tb.add_column("a", "int", 0, 999_999)
    tb.add_column("b", "int", 0, 49_999)
    tb.add_column("c", "int", 0, 999_999)
    _append_all(tb, n, {"g": g, "a": a, "b": b, "c": c})
    if compact:
        tb.compact()
    q = tb.query(groups=["g"], aggs=["a", "b", "c"], op="hist")
    o = oracle.run_query([{"type": "int", "data": x} for x in (g, a, b, c)], groups=[0],
                         aggs=[(1, 0, 999_999), (2, 0, 49_999), (3, 0, 999_999)], op="hist")
    for _ in range(3):
        r = q.run()
        assert q.stats()["strategy"] == 5
        parity.compare(r, o, op="hist", full=True, n_aggs=3)
        r.free()
    q.free()
    tb.free()


@pytest.mark.parametrize("two", [False, True])
def test_wrapping_last_bucket_of_an_odd_bucket_count(ctx, oracle, two):
    """Info = [0, 99] gives 101 buckets (hist_basic.go:34-70): the last one, index 100, is the LOW half of a counter word
    whose high half belongs to no bucket, and it is where values beyond the range are clipped to (outliers, accepted up
    to Info.Max x 10).  > 65 536 of them in one group wrap the 16-bit field and carry into that unused half: the carry must
    neither be counted nor logged as a -1 for a bucket that does not exist (with a second, 1002-bucket aggregation the
    words are laid out for 1002: the -1 would land in the neighbour's bucket 0)."""
    rng = np.random.default_rng(43)
    n = 600_000
    g = rng.integers(0, 3000, n, dtype=np.int64)
    g[rng.random(n) < 0.6] = 7
    v = rng.integers(0, 100, n, dtype=np.int64)
    hot = (g == 7) & (rng.random(n) < 0.5)
    v[hot] = rng.integers(200, 990, int(hot.sum()))  # outliers: beyond bucket 100's lower edge, within Info.Max x 10
    w = rng.integers(0, 1_000_000, n, dtype=np.int64)
    tb = ctx.create_table("oddwrap")
    tb.add_column("g", "int", 0, 2999)
    tb.add_column("v", "int", 0, 99)
    tb.add_column("w", "int", 0, 999_999)
    _append_all(tb, n, {"g": g, "v": v, "w": w})
    aggs = ["v", "w"] if two else ["v"]
    q = tb.query(groups=["g"], aggs=aggs, op="hist")
    r = q.run()
    assert q.stats()["strategy"] == 5, q.stats()
    o = oracle.run_query([{"type": "int", "data": x} for x in (g, v, w)], groups=[0],
                         aggs=[(1, 0, 99), (2, 0, 999_999)][:len(aggs)], op="hist")
    big = [x for x in o["results"] if x["count"] > 300_000]
    assert len(big) == 1 and big[0]["hists"][0]["n_values"] == 101 and big[0]["hists"][0]["values"][100] > 2 * 65536
    parity.compare(r, o, op="hist", full=True, n_aggs=len(aggs))
    r.free()
    q.free()
    tb.free()


@pytest.mark.parametrize("late", ["0", "1"])
@pytest.mark.parametrize("compact", [False, True])
def test_filtered_partitioned_histograms_either_order_of_loads(ctx, oracle, monkeypatch, late, compact):
    """Strategy 5 with filters: the counting and emitting kernels as they were (all columns of a tile requested together) and
    their LATE forms (filters a tile ahead, key and value columns only for waves with a passing row: k_count_packed /
    k_emit_packed<.., LATE>, which the planner takes at an estimated selectivity below 0.5 % -- SYBL_LATE_PATH forces either):
    config 4 with one, two and three range filters at ~10 %, ~1 % and ~0.1 %, config 3's two keys and three aggregations in two
    passes, a filter nothing passes, and one that everything passes."""
    monkeypatch.setenv("SYBL_LATE_PATH", late)
    wl = _wl("cfg4_hist_highcard")
    cols = wl["columns"] + ["c04", "c05", "c06"]
    for filters in ([("c04", "gt", 899)], [("c04", "gt", 899), ("c05", "lt", 100)], [("c04", "gt", 899), ("c05", "gt", 899), ("c06", "gt", 899)],
                    [("c04", "gt", 999)], [("c04", "gt", -1)]):
        q = dict(wl["query"], filters=filters)
        gres, ores, stats = parity.run_both(ctx, oracle, cols, 500_000, 0, 500_000, q, compact=compact)
        assert stats["strategy"] == 5, stats
        parity.compare(gres, ores, op="hist", full=True, n_aggs=1)
        gres.free()
    wl = _wl("cfg3_filter3_group2_stddev")
    q = dict(wl["query"], aggs=["c07", "c08", "c09"], want_percentiles=True, hist_bucket=990,
             filters=[("c04", "gt", 949), ("c05", "lt", 500)])
    gres, ores, stats = parity.run_both(ctx, oracle, wl["columns"] + ["c09"], 700_000, 0, 700_000, q, compact=compact)
    assert stats["strategy"] == 5, stats
    parity.compare(gres, ores, op="hist", full=True, n_aggs=3)
    gres.free()
