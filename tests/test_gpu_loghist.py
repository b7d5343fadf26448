"""GPU: -loghist (MultiHist, reference src/lib/hist_multi.go) against the CPU oracle: sub-histogram geometry, every
bucket, the sub-histograms' outliers, percentiles (bit-exact), stddev / avg (1e-6), text / -json rendering."""
import json

import numpy as np
import pytest

import sybil_amd
from tests import parity

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rows_mode")]  # (each test also with SYBL_LAZY_ROWS=1: conftest.py)


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def compare(gres, ores, n_aggs, time_mode=False):
    parity.compare(gres, ores, n_aggs=n_aggs, time_mode=time_mode, loghist=True)


def _table(ctx, cols, info, pops=None, block_rows=20_000, compact=False):
    names = list(cols)
    n = len(cols[names[0]])
    tb = ctx.create_table("lh")
    for c in names:
        lo, hi = info.get(c, (1, 0))
        tb.add_column(c, "int", lo, hi)
    for r0 in range(0, n, block_rows):
        r1 = min(r0 + block_rows, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if pops and c in pops else cols[c][r0:r1]) for c in names})
    if compact:
        tb.compact()
    return tb


@pytest.mark.parametrize("lo,hi,bucket", [(0, 999_999, 0), (10, 29, 0), (-5000, 70_000, 0), (-90_000, -100, 0), (0, 30_000, 7)])
def test_loghist_matches_the_oracle(ctx, oracle, lo, hi, bucket):
    rng = np.random.default_rng(abs(hi) + bucket)
    n = 70_000
    cols = {"g": rng.integers(0, 6, size=n).astype(np.int64),
            "v": rng.integers(lo - 20, hi + 100, size=n).astype(np.int64),
            "u": rng.integers(0, 5000, size=n).astype(np.int64),
            "w": rng.integers(1, 4, size=n).astype(np.int64)}
    cols["v"][:3] = [lo, hi, hi * 10 if hi > 0 else hi]
    pops = {"v": (rng.random(n) > 0.1).astype(np.uint8)}
    info = {"v": (lo, hi), "u": (0, 4999)}
    ocols = [{"type": "int", "data": cols[c], **({"populated": pops[c]} if c in pops else {})} for c in cols]
    for compact in (False, True):
        tb = _table(ctx, cols, info, pops, compact=compact)
        for q, okw in ((dict(groups=["g"], aggs=["v", "u"], op="hist", hist_bucket=bucket, loghist=True),
                        dict(groups=[0], aggs=[(1, lo, hi), (2, 0, 4999)], op="hist", hist_bucket=bucket)),
                       (dict(filters=[("u", "gt", 999)], aggs=["v"], op="hist", hist_bucket=bucket, loghist=True, weight_col="w"),
                        dict(filters=[(2, "gt", 999)], aggs=[(1, lo, hi)], op="hist", hist_bucket=bucket, weight_col=3)),
                       (dict(groups=["g"], aggs=["v"], op="avg", loghist=True), dict(groups=[0], aggs=[(1, lo, hi)], op="avg"))):
            query = tb.query(**q)
            gres = query.run()
            ores = oracle.run_query(ocols, block_rows=20_000, n_threads=2, loghist=True, **okw)
            compare(gres, ores, len(q["aggs"]))
            gres.free()
            query.free()
        tb.free()


def test_loghist_time_series_hash_and_rendering(ctx, oracle, monkeypatch):
    rng = np.random.default_rng(41)
    n = 50_000
    cols = {"g": rng.integers(0, 4, size=n).astype(np.int64), "t": np.sort(1_700_000_000 + rng.integers(0, 4 * 3600, size=n)).astype(np.int64),
            "v": rng.integers(0, 200_000, size=n).astype(np.int64)}
    info = {"v": (0, 199_999)}
    ocols = [{"type": "int", "data": cols[c]} for c in cols]
    tb = _table(ctx, cols, info)
    # time series
    q = dict(groups=["g"], aggs=["v"], op="hist", loghist=True, time_col="t", time_bucket=3600)
    query = tb.query(**q)
    gres = query.run()
    ores = oracle.run_query(ocols, groups=[0], aggs=[(2, 0, 199_999)], op="hist", loghist=True, time_col=1, time_bucket=3600, block_rows=20_000)
    compare(gres, ores, 1, time_mode=True)
    gres.free()
    query.free()
    # through the hash table (bucket words follow the key's slot, then the dense order)
    monkeypatch.setenv("SYBL_FORCE_HASH", "1")
    q = dict(groups=["g"], aggs=["v"], op="hist", loghist=True)
    query = tb.query(**q)
    gres = query.run()
    assert query.stats()["strategy"] == 7
    monkeypatch.delenv("SYBL_FORCE_HASH")
    ores = oracle.run_query(ocols, groups=[0], aggs=[(2, 0, 199_999)], op="hist", loghist=True, block_rows=20_000)
    compare(gres, ores, 1)
    # -json: percentiles, stddev, and GetStrBuckets' union (a later sub-histogram REPLACES an equal key, hist_multi.go:175-188)
    out = json.loads(gres.render("json"))
    rows = {r["key_vals"][0]: r for r in gres.results}
    omap = {r["key_vals"][0]: r for r in ores["results"]}
    subs = gres.subhists(0)
    for jr in out:
        h, o = rows[int(jr["g"])]["hists"][0], omap[int(jr["g"])]["hists"][0]
        assert jr["v"]["percentiles"] == o["percentiles"].tolist() and jr["Count"] == rows[int(jr["g"])]["count"]
        assert abs(jr["v"]["stddev"] - o["stddev_exact"]) <= 1e-6 * o["stddev_exact"]
        want = {}
        for s in subs:
            sub = {str(k * s["bucket_size"] + s["info_min"]): int(h["values"][s["offset"] + k]) for k in range(s["n_values"])}
            for k in range(s["n_ext"]):
                c = int(h["values"][s["ext_offset"] + k])
                if c:
                    sub[str(s["ext_first"] + k)] = sub.get(str(s["ext_first"] + k), 0) + c
            want.update(sub)
        assert jr["v"]["buckets"] == {k: c for k, c in want.items() if c > 0}
    assert gres.render("text")
    # -encode-results: *sybil.MultiHistCompat{MultiHist, Histogram} (both the same histogram: gob flattens pointers),
    # each sub-histogram a HistCompat with its buckets and its outliers written out
    from tests import gobfmt
    enc = gobfmt.decode(gres.encode())["QuerySpec"]
    assert enc["QueryParams"]["Aggregations"] == [{"Op": "hist", "Name": "v", "HistType": "multi"}]
    for key, e in enc["QueryResults"]["Results"].items():
        h = rows[int(key.strip())]["hists"][0]
        iv = e["Hists"]["v"]
        assert iv["@type"] == "*sybil.MultiHistCompat" and iv["value"]["MultiHist"] == iv["value"]["Histogram"]
        mh = iv["value"]["MultiHist"]
        assert (mh["Count"], mh["Max"], mh.get("Min", 0), mh["Avg"], mh["PercentileMode"]) == (h["count"], h["max"], h["min"], h["avg"], True)
        assert mh["Info"] == {"Max": 199_999} and len(mh["Subhists"]) == len(subs)
        for s, sh in zip(subs, mh["Subhists"]):
            ci = sh["BasicHist"]["BasicHistCachedInfo"]
            vals = h["values"][s["offset"]:s["offset"] + s["n_values"]]
            assert ci.get("Values", [0] * s["n_values"]) == vals.tolist() or (not vals.any() and "Values" in ci)
            assert ci["BucketSize"] == s["bucket_size"] and ci["NumBuckets"] == s["num_buckets"] and ci.get("Count", 0) == int(vals.sum())
            want_out = [s["ext_first"] + k for k in range(s["n_ext"]) for _ in range(int(h["values"][s["ext_offset"] + k]))]
            assert ci.get("Outliers", []) == want_out
            assert ci["Info"] == {k: v for k, v in (("Min", s["info_min"]), ("Max", s["info_max"])) if v != 0}
    gres.free()
    query.free()
    tb.free()
