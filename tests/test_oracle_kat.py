"""Pins the CPU oracle: known-answer tests derived from the reference source
 cross-checked by an independent pure-Python restatement of
hist_basic.go:101-219 kept in this file, and the reference's own golden
NodeResults file (tests/golden/node_results_hists.json, made by make_golden.py)."""
import json
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "node_results_hists.json")


# ---- independent pure-Python BasicHist (second derivation; tiny inputs only) ----
class PyHist:
    def __init__(self, imin, imax, hist, bucket=0, weight_col=False):
        self.imin, self.imax, self.hist, self.wc = imin, imax, hist, weight_col
        self.avg, self.count, self.samples = 0.0, 0, 0
        self.min = self.max = 0
        self.outliers, self.underliers = [], []
        if hist:
            self.min, self.max = imin, imax
            size = imax - imin
            nb = 1000
            bs = int(size / nb) if size >= 0 else -int(-size / nb)
            if bucket > 0:
                bs = bucket
            if bs == 0:
                if size < 100:
                    bs, nb = 1, size
                else:
                    bs = size // 100
                    nb = size // bs
            nb += 1
            self.bs, self.nb = bs, nb
            self.values = [0] * (nb + 1)

    def add(self, v, w=1):
        if v > self.imax * 10 or v < self.imin:
            return
        if self.wc or w > 1:
            self.samples += 1
            self.count += w
        else:
            self.count += 1
        self.avg = self.avg + ((float(v) - self.avg) / float(self.count)) * float(w)
        self.max = max(self.max, v)
        self.min = min(self.min, v)
        if not self.hist:
            return
        b = int((v - self.min) / self.bs) if (v - self.min) >= 0 else -int(-(v - self.min) / self.bs)
        if b >= len(self.values):
            self.outliers.append(v)
            b = len(self.values) - 1
        if b < 0:
            self.underliers.append(v)
            b = 0
        self.values[b] += w

    def percentiles(self):
        if self.count == 0:
            return []
        p = [0] * 101
        p[0] = self.min
        c = prev = 0
        for k, n in enumerate(self.values):
            c += n
            q = (100 * c) // self.count
            for ip in range(prev, q + 1):
                p[ip] = k * self.bs + self.min
            p[q] = k
            prev = q
        return p[:100]

    def stddev(self):
        s = 0.0
        for b, n in enumerate(self.values if self.hist else []):
            d = float(b * self.bs + self.min) - self.avg
            s += (d * d) * (float(n) / float(self.count))
        for o in self.outliers + self.underliers:
            s += ((float(o) - self.avg) ** 2) * (1 / float(self.count))
        return math.sqrt(s)


def _feed(orc, imin, imax, op, vals, weights=None, bucket=0, wc=False):
    h = orc.Hist(imin, imax, op, bucket, wc)
    p = PyHist(imin, imax, op == "hist", bucket, wc)
    for i, v in enumerate(vals):
        w = 1 if weights is None else weights[i]
        h.add(v, w)
        p.add(v, w)
    return h, p


def test_kat_hist_million(oracle):
    h, p = _feed(oracle, 0, 999999, "hist", range(0, 1000000, 1000))
    i = h.info()
    assert (i["bucket_size"], i["n_values"], i["count"]) == (999, 1002, 1000)
    assert i["avg"] == 499500.0
    pc = h.percentiles()
    assert [int(pc[k]) for k in (0, 25, 50, 75, 99)] == [8991, 258741, 508491, 758241, 999000]
    assert i["stddev_ref"] == pytest.approx(288388.4764280622, rel=1e-12)
    assert list(pc) == p.percentiles()
    assert i["stddev_ref"] == p.stddev()
    assert i["sum_exact"] == sum(range(0, 1000000, 1000))


def test_kat_hist_small_range(oracle):
    h, p = _feed(oracle, 10, 29, "hist", range(10, 30))
    i = h.info()
    assert (i["bucket_size"], i["n_values"]) == (1, 21)
    assert i["avg"] == 19.5
    pc = h.percentiles()
    assert [int(pc[k]) for k in (0, 25, 50, 75, 99)] == [10, 15, 20, 25, 29]
    assert i["stddev_ref"] == pytest.approx(5.766281297335398, rel=1e-14)
    assert list(pc) == p.percentiles() and i["stddev_ref"] == p.stddev()


def test_kat_avg_mode_quirks(oracle):
    h, p = _feed(oracle, 10, 29, "avg", [12, 15, 29, 10])
    i = h.info()
    assert i["avg"] == 16.5 and i["count"] == 4
    assert i["min"] == 0 and i["max"] == 29  # Min starts at the Go zero value in avg mode
    assert i["stddev_ref"] == 0.0
    assert i["true_min"] == 10 and i["true_max"] == 29
    assert h.percentiles().size == 100 or h.percentiles().size == 0


def test_kat_outliers(oracle):
    h, p = _feed(oracle, 0, 1999, "hist", [0, 1001, 1002, 1500, 1999])
    i = h.info()
    v = h.values()
    assert i["bucket_size"] == 1 and v[0] == 1 and v[1001] == 4 and v.size == 1002
    assert list(h.outliers()) == [1002, 1500, 1999]
    assert int(h.percentiles()[50]) == 1001
    assert i["stddev_ref"] == pytest.approx(667.4198498696304, rel=1e-14)
    assert i["stddev_ref"] == p.stddev()
    assert i["n_outliers"] == 3


def test_kat_weights(oracle):
    h, p = _feed(oracle, 0, 100, "avg", [10, 20, 30], [1, 10, 100], wc=True)
    i = h.info()
    assert i["count"] == 111 and i["samples"] == 3
    assert i["avg"] == pytest.approx(28.91891891891892, rel=1e-15)
    assert i["avg"] == p.avg
    assert i["sum_exact"] == 10 + 200 + 3000


def test_kat_rejects_out_of_info_range(oracle):
    # hist_basic.go:104: v > Info.Max*10 or v < Info.Min is dropped from the hist
    h, _ = _feed(oracle, 10, 20, "avg", [9, 10, 200, 201, 15])
    i = h.info()
    assert i["count"] == 3 and i["sum_exact"] == 10 + 200 + 15


def test_kat_time_bucket(oracle):
    assert oracle.time_bucket(1700003599, 3600) == 1700002800
    assert oracle.time_bucket(-1, 3600) == 0
    assert oracle.time_bucket(-3601, 3600) == -3600


def test_setup_buckets_geometry(oracle):
    assert oracle.setup_buckets(30, 23500) == {"bucket_size": 23, "num_buckets": 1001, "n_values": 1002}
    assert oracle.setup_buckets(0, 999999) == {"bucket_size": 999, "num_buckets": 1001, "n_values": 1002}
    assert oracle.setup_buckets(10, 29) == {"bucket_size": 1, "num_buckets": 20, "n_values": 21}
    assert oracle.setup_buckets(0, 500) == {"bucket_size": 5, "num_buckets": 101, "n_values": 102}
    assert oracle.setup_buckets(0, 999999, 5000)["bucket_size"] == 5000


def test_random_against_python_restatement(oracle):
    rng = np.random.default_rng(7)
    for trial in range(20):
        imin = int(rng.integers(-50, 50))
        imax = imin + int(rng.integers(1, 5000))
        vals = rng.integers(imin - 5, imax * 3 + 10, size=300).tolist()
        wts = rng.integers(1, 5, size=300).tolist() if trial % 2 else None
        h, p = _feed(oracle, imin, imax, "hist", vals, wts, wc=bool(trial % 2))
        i = h.info()
        assert i["count"] == p.count and i["avg"] == p.avg
        assert list(h.values()) == p.values
        assert list(h.percentiles()) == p.percentiles()
        assert i["stddev_ref"] == p.stddev()
        assert (i["min"], i["max"]) == (p.min, p.max)


# ---- the reference's golden NodeResults ----
@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def test_golden_geometry_and_keys(oracle, gold):
    cum = gold["Cumulative"]["Hists"]["pageload"]
    g = oracle.setup_buckets(cum["InfoMin"], cum["InfoMax"])
    assert g["bucket_size"] == cum["BucketSize"] and g["num_buckets"] == cum["NumBuckets"]
    assert g["n_values"] == len(cum["Values"]) == 1002
    # BinaryByKey: 8 little-endian bytes per group column (aggregate.go:16,125-143)
    for k, r in gold["Results"].items():
        assert len(bytes.fromhex(r["BinaryByKeyHex"])) == 8 * len(gold["Groups"])
        assert k == r["GroupByKey"] and k.endswith("\t") and k.count("\t") == 2


def test_golden_combine_identity(oracle, gold):
    """Feeding the 12 per-group hists (real reference output) through the oracle's
    Result.Combine/BasicHist.Combine must reproduce the reference's Cumulative."""
    cum = gold["Cumulative"]
    ch = cum["Hists"]["pageload"]
    total = np.zeros(1002, dtype=np.int64)
    count = 0
    avg = 0.0
    # Go map iteration order is random; the count-weighted float merge is order dependent
    # in the last bits, so compare at 1e-12 and check every bucket exactly.
    for k in sorted(gold["Results"]):
        h = gold["Results"][k]["Hists"]["pageload"]
        total += np.asarray(h["Values"], dtype=np.int64)
        avg = oracle.combine_avg(avg, count, h["Avg"], h["Count"])
        count += h["Count"]
        assert sum(h["Values"]) == h["Count"] == gold["Results"][k]["Count"]
    assert count == ch["Count"] == cum["Count"] == gold["MatchedCount"] == 20000
    assert total.tolist() == ch["Values"]
    assert avg == pytest.approx(ch["Avg"], rel=1e-12)
    assert ch["Outliers"] == []  # Combine never merges outlier lists (hist_basic.go:259-279)


GOLD_KATS = {  # SURVEY.md 8c "golden-derived KATs"
    "TOTAL\t": (20000, 2930.9991000000014, [76, 536, 1479, 3733, 23053], 3854.750176573751),
    "edge\tdesktop\t": (1643, 1167.1381618989653, [53, 191, 582, 1640, 6585], 1377.784713975415),
    "gecko\ttablet\t": (1707, 4466.12595196252, [168, 858, 2215, 6401, 23053], 5141.204759365595),
}


def test_golden_percentiles_and_stddev(oracle, gold):
    res = dict(gold["Results"])
    res["TOTAL\t"] = gold["Cumulative"]
    for key, (count, avg, pcts, sd) in GOLD_KATS.items():
        h = res[key]["Hists"]["pageload"]
        assert h["Count"] == count and h["Avg"] == avg
        p = oracle.percentiles_from_values(h["Values"], h["BucketSize"], h["Min"], h["Count"])
        assert [int(p[i]) for i in (0, 25, 50, 75, 99)] == pcts
        s = oracle.stddev_from_values(h["Values"], h["BucketSize"], h["Min"], h["Count"], h["Avg"],
                                      h["Outliers"], h["Underliers"])
        assert s == pytest.approx(sd, rel=1e-13)
    # outlier index rule: (23500-30)/23 = 1020 >= 1002
    assert res["gecko\ttablet\t"]["Hists"]["pageload"]["Outliers"] == [23500]


def test_golden_sort_order(gold):
    counts = [gold["Results"][k]["Count"] for k in gold["SortedKeys"]]
    assert counts == sorted(counts, reverse=True)
    assert counts == [1740, 1724, 1707, 1700, 1682, 1670, 1657, 1653, 1643, 1622, 1621, 1581]
