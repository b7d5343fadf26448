"""CPU: the oracle's MultiHist (-loghist, reference src/lib/hist_multi.go) against an independent, literal pure-Python
restatement of the same file, and the property the reference's own test asserts for it (query_cache_test.go:152-232:
a positive standard deviation that two runs agree on)."""
import math

import numpy as np
import pytest


def _setup(mn, mx, hist_bucket):
    # BasicHist.SetupBuckets, hist_basic.go:34-70
    size = mx - mn
    nb, bs = 1000, size // 1000
    if hist_bucket > 0:
        bs = hist_bucket
    if bs == 0:
        if size < 100:
            bs, nb = 1, size
        else:
            bs = size // 100
            nb = size // bs
    nb += 1
    return {"min": mn, "max": mx, "bs": bs, "nb": nb, "values": [0] * (nb + 1), "outliers": []}


def py_multihist(info_min, info_max, values, weights=None, hist_bucket=0):
    # TrackPercentiles, hist_multi.go:223-257
    subs, width, n = [], info_max - info_min, 0
    t = width
    while t > 1000:
        n += 1
        t >>= 1
    right = info_max
    for _ in range(n):
        width >>= 1
        subs.append(_setup(right - width, right, hist_bucket))
        right -= width
    subs.append(_setup(info_min, right, hist_bucket))
    count, samples, avg, vmin, vmax = 0, 0, 0.0, info_min, info_max
    for i, v in enumerate(values):  # AddWeightedValue, :48-91
        w = 1 if weights is None else weights[i]
        if v > info_max * 10 or v < info_min:
            continue
        if weights is not None or w > 1:
            samples += 1
            count += w
        else:
            count += 1
        avg = avg + ((v - avg) / count) * w
        vmax, vmin = max(vmax, v), min(vmin, v)
        for s in subs:
            if s["min"] <= v <= s["max"]:
                if not (v > s["max"] * 10 or v < s["min"]):  # the sub-histogram's own AddWeightedValue gate
                    b = (v - s["min"]) // s["bs"]
                    if b >= len(s["values"]):
                        s["outliers"].append(v)
                        b = len(s["values"]) - 1
                    s["values"][b] += w
                break
    sparse = {}  # GetSparseBuckets, :190-207
    for s in subs:
        for k, c in enumerate(s["values"]):
            if c > 0:
                sparse[k * s["bs"] + s["min"]] = sparse.get(k * s["bs"] + s["min"], 0) + c
        for o in s["outliers"]:
            sparse[o] = sparse.get(o, 0) + 1
    pct, prev, c, total = [0] * 101, 0, 0, sum(sparse.values())  # GetPercentiles, :93-128
    for k in sorted(sparse):
        c += sparse[k]
        p = (100 * c) // total
        for ip in range(prev, p + 1):
            if ip <= 100:
                pct[ip] = k
        if p <= 100:
            pct[p] = k
        prev = p
    sd = math.sqrt(sum((k - avg) ** 2 * (n_ / count) for k, n_ in sparse.items())) if count else 0.0  # GetStdDev, :140-155
    return {"subs": subs, "sparse": sparse, "pct": pct[:100] if count else [], "stddev": sd, "count": count, "samples": samples,
            "avg": avg, "min": vmin, "max": vmax}


@pytest.mark.parametrize("lo,hi,bucket", [(0, 999_999, 0), (10, 29, 0), (-5000, 70_000, 0), (0, 1500, 0), (100, 100 + 2 ** 20, 0),
                                          (-90_000, -100, 0), (0, 40_000, 7), (0, 1000, 0), (0, 1001, 0)])
def test_multihist_matches_the_literal_restatement(oracle, lo, hi, bucket):
    rng = np.random.default_rng(abs(lo) + hi + bucket)
    vals = rng.integers(lo - 10, hi + 50, size=20_000).tolist() + [lo, hi, hi * 10 if hi > 0 else hi]
    h = oracle.Hist(lo, hi, "hist", hist_bucket=bucket, loghist=True)
    for v in vals:
        h.add(v)
    want = py_multihist(lo, hi, vals, hist_bucket=bucket)
    info = h.info()
    assert (info["count"], info["min"], info["max"]) == (want["count"], want["min"], want["max"])
    assert abs(info["avg"] - want["avg"]) <= 1e-9 * max(abs(want["avg"]), 1)
    assert [(s[0], s[1], s[2], s[3], s[4]) for s in h.subhists()] == [(s["min"], s["max"], s["bs"], s["nb"], len(s["values"])) for s in want["subs"]]
    assert h.values().tolist() == [c for s in want["subs"] for c in s["values"]]
    assert h.sparse() == want["sparse"]
    assert h.percentiles().tolist() == want["pct"]
    assert abs(info["stddev_ref"] - want["stddev"]) <= 1e-9 * max(want["stddev"], 1)
    assert info["n_outliers"] == sum(len(s["outliers"]) for s in want["subs"])


def test_weighted_multihist(oracle):
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 50_000, size=5000).tolist()
    ws = rng.integers(1, 6, size=5000).tolist()
    h = oracle.Hist(0, 49_999, "hist", weight_col=True, loghist=True)
    for v, w in zip(vals, ws):
        h.add(v, w)
    want = py_multihist(0, 49_999, vals, weights=ws)
    info = h.info()
    assert (info["count"], info["samples"]) == (want["count"], want["samples"]) == (sum(ws), 5000)
    assert h.sparse() == want["sparse"] and h.percentiles().tolist() == want["pct"]


def test_query_with_loghist_combines_like_one_histogram(oracle):
    """The reference's check (query_cache_test.go:152-232): StdDev > 0 and equal between two runs; plus: blocks combined
    (MultiHist.Combine, :209-221) give the buckets of one histogram fed every value."""
    rng = np.random.default_rng(8)
    n = 30_000
    age = rng.integers(10, 80, size=n).astype(np.int64)
    v = rng.integers(0, 300_000, size=n).astype(np.int64)
    cols = [{"type": "int", "data": age}, {"type": "int", "data": v}]
    kw = dict(filters=[(0, "lt", 20)], aggs=[(1, 0, 299_999)], op="hist", loghist=True, block_rows=4096)
    a, b = oracle.run_query(cols, **kw), oracle.run_query(cols, n_threads=3, **kw)
    ha, hb = a["cumulative"]["hists"][0], b["cumulative"]["hists"][0]
    assert ha["stddev_ref"] > 0 and abs(ha["stddev_ref"] - hb["stddev_ref"]) <= 0.1
    one = oracle.Hist(0, 299_999, "hist", loghist=True)
    for x in v[age < 20]:
        one.add(int(x))
    assert ha["values"].tolist() == one.values().tolist() and ha["sparse"] == one.sparse()
    assert ha["percentiles"].tolist() == one.percentiles().tolist() and ha["count"] == one.info()["count"]
    # avg mode: no sub-histograms, Min / Max still start at Info (hist_multi.go:31-32), stddev 0
    c = oracle.run_query(cols, aggs=[(1, 1000, 299_999)], op="avg", loghist=True)["cumulative"]["hists"][0]
    assert c["min"] == 1000 and c["max"] == 299_999 and c["stddev_ref"] == 0 and c["count"] == int((v >= 1000).sum())
