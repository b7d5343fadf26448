"""GPU: the BASELINE.json configurations at their FULL sizes.  First through size-independent properties (counts add
up, group sums equal ungrouped sums, bucket arrays sum to counts, percentiles are monotone, compact and canonical
storage agree); then, further down, bit for bit against the CPU oracle (orc_synth_scan regenerates the synthetic table
on every host thread: 10^9 rows in a few seconds) -- configs 2, 3, 4 and 5."""
import numpy as np
import pytest

import sybil_amd
from sybil_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _rows(ctx, wl):
    # a 1-GPU box with less HBM than the full table scans a tenth of it
    need = wl["rows"] * 8 * len(wl["columns"])
    return wl["rows"] if ctx.device_info()["hbm_bytes"] > 1.5 * need else wl["rows"] // 10


def test_config2_group_sums_add_up(ctx):
    wl = synth.WORKLOADS["cfg2_group1_avg2"]
    rows = _rows(ctx, wl)
    t = ctx.synth_table("c2", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    out = {}
    for storage in ("canonical", "compact"):
        if storage == "compact":
            t.compact()
        q = t.query(**wl["query"])
        r = q.run()
        assert q.stats()["packed_kernel"] == (1 if storage == "compact" else 0)
        groups = r.results
        assert len(groups) == 16 and r.matched == rows == sum(g["count"] for g in groups)
        # uniform keys: every group holds 1/16 of the rows to within 1 %
        assert all(abs(g["count"] * 16 / rows - 1) < 0.01 for g in groups)
        qt = t.query(aggs=wl["query"]["aggs"], op="avg")
        rt = qt.run()
        for a in range(2):
            assert sum(g["hists"][a]["sum"] for g in groups) == rt.results[0]["hists"][a]["sum"] == r.cumulative["hists"][a]["sum"]
            assert all(0 <= g["hists"][a]["avg"] <= 1_000_000 for g in groups)
        out[storage] = sorted((g["key"], g["count"], g["hists"][0]["sum"], g["hists"][1]["sum"]) for g in groups)
        for x in (r, rt):
            x.free()
        for x in (q, qt):
            x.free()
    assert out["canonical"] == out["compact"]
    t.free()


def test_config4_histograms_by_65536_groups(ctx):
    wl = synth.WORKLOADS["cfg4_hist_highcard"]
    rows = _rows(ctx, wl)
    t = ctx.synth_table("c4", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    q = t.query(**dict(wl["query"], limit=100, order_by="$COUNT"))
    r = q.run()
    st = q.stats()
    assert st["strategy"] == 5 and st["packed_kernel"] == 1 and st["rows_scanned"] == rows
    groups = r.results
    assert len(groups) == 65536 and r.matched == rows == sum(g["count"] for g in groups)
    counts = [g["count"] for g in groups]
    assert counts == sorted(counts, reverse=True)
    for i, g in enumerate(groups):
        h = g["hists"][0]
        p = h["percentiles"]
        assert h["count"] == g["count"] and np.all(np.diff(p) >= 0) and p[0] >= 0 and p[99] <= 999_999
        # uniform values: the median of ~15 000 samples sits near the middle of the range
        assert abs(int(p[50]) - 509_000) < 45_000
        if i < 100:
            assert int(h["values"].sum()) == g["count"]      # the printed rows carry their bucket arrays
        else:
            assert "values" not in h
    c = r.cumulative["hists"][0]
    assert int(c["values"].sum()) == rows == c["count"]
    assert c["sum"] == sum(g["hists"][0]["sum"] for g in groups)
    # GetPercentiles reports slot i at the bucket where the cumulative share first exceeds i % (the known
    # answers of SURVEY.md 8c: uniform 0..999 000 gives pct[50] = 508 491, pct[25] = 258 741)
    assert 500_000 <= int(c["percentiles"][50]) <= 520_000 and 250_000 <= int(c["percentiles"][25]) <= 270_000
    r.free()
    q.free()
    t.free()


def test_config5_time_rollup(ctx):
    wl = synth.WORKLOADS["cfg5_time_rollup"]
    rows = _rows(ctx, wl)
    t = ctx.synth_table("c5", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    out = {}
    for storage in ("canonical", "compact"):
        if storage == "compact":
            t.compact()
        q = t.query(**wl["query"])
        r = q.run()
        assert q.stats()["packed_kernel"] == (1 if storage == "compact" else 0)
        tr = r.time_results
        assert r.matched == rows == sum(x["count"] for x in tr) == sum(g["count"] for g in r.results)
        per_bucket = {}
        for x in tr:
            per_bucket[x["time_bucket"]] = per_bucket.get(x["time_bucket"], 0) + x["count"]
        # 30 days starting inside an hour: 721 hourly buckets, the first and the last partial
        assert len(per_bucket) == 721 and min(per_bucket) == 1_700_000_000 // 3600 * 3600
        # the time column advances linearly: every full hour holds rows / 720 rows
        full = [per_bucket[k] for k in sorted(per_bucket)[1:-1]]
        assert max(full) - min(full) <= 2
        qs = t.query(aggs=wl["query"]["aggs"], op="avg")
        rs = qs.run()
        assert sum(x["hists"][0]["sum"] for x in tr) == rs.results[0]["hists"][0]["sum"]
        out[storage] = sorted((x["time_bucket"], x["key"], x["count"], x["hists"][0]["sum"]) for x in tr)
        for x in (r, rs):
            x.free()
        for x in (q, qs):
            x.free()
    assert out["canonical"] == out["compact"]
    t.free()


# ---------------------------------------------------------------------------------------------------
# Bit-exact parity with the CPU oracle AT THE BASELINE SIZE.  orc_synth_scan regenerates the synthetic table
# block by block on every host thread and runs the reference's row loop in its direct-mapped form
# (oracle/sybil_oracle.h; pinned against the per-block hash-map restatement by
# tests/test_oracle_query.py::test_full_size_checker_matches_the_reference_shaped_oracle); the engine's exact
# per-cell integers (sybl_debug_query_cells) and every result row are compared with it: matched rows, Count,
# sum(v), sum(b), sum(b^2), every bucket, every percentile.  10^9 rows exercise what the 2 M-row parity tests
# cannot: many tiles per workgroup, the 2^28-row chunk boundary of the packed kernels, 32-bit per-thread
# counters, partition buffers of hundreds of thousands of records.
def _oracle_scan(orc, wl, rows, want_buckets):
    import os
    q = wl["query"]
    return orc.synth_scan(synth.COLUMNS, synth.SEED, rows, 0, rows, filters=q.get("filters", ()), groups=q.get("groups", ()),
                          aggs=q.get("aggs", ()), time_col=q.get("time_col"), time_bucket=q.get("time_bucket", 0),
                          want_buckets=want_buckets, n_threads=os.cpu_count() or 8)


def _check_cells(q, o, n_aggs, moments):
    assert np.array_equal(q.debug_cells("count"), o["count"])
    for a in range(n_aggs):
        assert np.array_equal(q.debug_cells("sum", a), o["sum"][a])
        if moments:
            assert np.array_equal(q.debug_cells("sb", a), o["sb"][a])
            assert np.array_equal(q.debug_cells("sb2", a), o["sb2"][a])


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def test_config2_full_size_matches_the_oracle_bit_for_bit(ctx, orc):
    wl = synth.WORKLOADS["cfg2_group1_avg2"]
    rows = wl["rows"]  # 10^8 rows x 3 resident columns: fits any box
    o = _oracle_scan(orc, wl, rows, want_buckets=False)
    assert o["matched"] == rows == int(o["count"].sum())
    t = ctx.synth_table("c2o", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    for storage in ("canonical", "compact"):
        if storage == "compact":
            t.compact()
        q = t.query(**wl["query"])
        r = q.run()
        st = q.stats()
        assert st["strategy"] == 2 and st["packed_kernel"] == (1 if storage == "compact" else 0) and st["rows_scanned"] == rows
        assert r.matched == o["matched"]
        _check_cells(q, o, 2, moments=False)
        groups = r.results
        assert len(groups) == 16
        for g in groups:
            cell = g["key_vals"][0]
            assert g["count"] == o["count"][cell]
            for a in range(2):
                h = g["hists"][a]
                assert h["count"] == o["count"][cell] and h["sum"] == o["sum"][a][cell]
                assert abs(h["avg"] - int(o["sum"][a][cell]) / int(o["count"][cell])) <= 1e-9 * h["avg"]
        r.free()
        q.free()
    t.free()


def test_config3_full_size_matches_the_oracle_bit_for_bit(ctx, orc):
    wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    rows = _rows(ctx, wl)
    o = _oracle_scan(orc, wl, rows, want_buckets=True)
    assert o["matched"] == int(o["count"].sum()) > 0
    t = ctx.synth_table("c3", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    for storage in ("canonical", "compact"):
        if storage == "compact":
            t.compact()
        # moments mode: k_scan_fast / k_scan_packed (the headline kernels)
        q = t.query(**wl["query"])
        r = q.run()
        st = q.stats()
        assert st["strategy"] == 2 and st["packed_kernel"] == (1 if storage == "compact" else 0) and st["rows_scanned"] == rows
        assert r.matched == o["matched"]
        _check_cells(q, o, 2, moments=True)
        groups = r.results
        assert len(groups) == 1024 == int((o["count"] != 0).sum())
        for g in groups:
            cell = g["key_vals"][0] * 64 + g["key_vals"][1]
            assert g["count"] == o["count"][cell]
            for a in range(2):
                h = g["hists"][a]
                assert h["count"] == o["count"][cell] and h["sum"] == o["sum"][a][cell]
                # GetStdDev (hist_basic.go:192-219) over the oracle's bucket array, avg = exact sum / count
                want = orc.stddev_from_values(o["buckets"][cell][a][:o["n_values"][a]], o["bucket_size"][a], o["hmin"][a], int(o["count"][cell]),
                                              int(o["sum"][a][cell]) / int(o["count"][cell]))
                assert abs(h["stddev"] - want) <= 1e-9 * want
        r.free()
        q.free()
    # the same query with every bucket array: k_emit_packed + k_part_hist over 64 partitions
    q = t.query(**dict(wl["query"], want_percentiles=True))
    r = q.run()
    assert q.stats()["strategy"] == 5 and r.matched == o["matched"]
    for g in r.results:
        cell = g["key_vals"][0] * 64 + g["key_vals"][1]
        for a in range(2):
            h = g["hists"][a]
            nv = o["n_values"][a]
            assert h["sum"] == o["sum"][a][cell] and np.array_equal(h["values"], o["buckets"][cell][a][:nv])
            assert np.array_equal(h["percentiles"], orc.percentiles_from_values(o["buckets"][cell][a][:nv], o["bucket_size"][a], o["hmin"][a],
                                                                                 int(o["count"][cell])))
    r.free()
    q.free()
    t.free()


def test_config4_full_size_matches_the_oracle_bit_for_bit(ctx, orc):
    wl = synth.WORKLOADS["cfg4_hist_highcard"]
    rows = _rows(ctx, wl)
    o = _oracle_scan(orc, wl, rows, want_buckets=True)
    t = ctx.synth_table("c4", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    q = t.query(**wl["query"])   # no limit: every group's bucket array comes back
    r = q.run()
    st = q.stats()
    assert st["strategy"] == 5 and st["packed_kernel"] == 1 and st["rows_scanned"] == rows == r.matched == o["matched"]
    _check_cells(q, o, 1, moments=True)    # sum(b), sum(b^2): k_hist_summary's bucket moments
    groups = r.results
    assert len(groups) == 65536
    nv, bs, hmin = o["n_values"][0], o["bucket_size"][0], o["hmin"][0]
    total = np.zeros(nv, dtype=np.int64)
    for g in groups:
        cell = g["key_vals"][0]
        h = g["hists"][0]
        want = o["buckets"][cell][0][:nv]
        assert g["count"] == h["count"] == o["count"][cell] and h["sum"] == o["sum"][0][cell]
        assert np.array_equal(h["values"], want)
        assert np.array_equal(h["percentiles"], orc.percentiles_from_values(want, bs, hmin, int(o["count"][cell])))
        total += h["values"]
    c = r.cumulative["hists"][0]
    assert np.array_equal(c["values"], total) and np.array_equal(total, o["buckets"][:, 0, :nv].sum(axis=0))
    assert c["count"] == rows and c["sum"] == int(o["sum"][0].sum())
    r.free()
    q.free()
    t.free()


def test_config5_full_size_matches_the_oracle_bit_for_bit(ctx, orc):
    wl = synth.WORKLOADS["cfg5_time_rollup"]
    rows = _rows(ctx, wl)
    o = _oracle_scan(orc, wl, rows, want_buckets=False)
    t = ctx.synth_table("c5", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    for storage in ("canonical", "compact"):
        if storage == "compact":
            t.compact()
        q = t.query(**wl["query"])
        r = q.run()
        st = q.stats()
        assert st["strategy"] == 4 and st["packed_kernel"] == (1 if storage == "compact" else 0)   # lds-window
        assert r.matched == rows == o["matched"]
        _check_cells(q, o, 1, moments=False)
        n_tb, card = o["cells"]
        tr = r.time_results
        assert len(tr) == int((o["count"] != 0).sum())
        for x in tr[::97]:   # (every cell was compared above; spot-check the row <-> cell mapping)
            cell = (x["time_bucket"] // 3600 - o["tb_min"]) * card + x["key_vals"][0]
            assert x["count"] == o["count"][cell] and x["hists"][0]["sum"] == o["sum"][0][cell]
        r.free()
        q.free()
    t.free()
