"""Shared helpers for the GPU-vs-oracle parity tests (test infrastructure)."""
import numpy as np

from sybil_amd import synth

REL = 1e-6  # north_star: avg / stddev / percentile-derived floats within 1e-6 relative


def oracle_synth_cols(orc, names, total_rows, row0, nrows):
    cols = []
    for n in names:
        kind, idx, a, b, _, _ = synth.COLUMNS[n]
        cols.append({"type": "int", "data": orc.synth_fill(kind, a, b, synth.SEED, idx, row0, nrows, total_rows)})
    return cols


def oracle_query_kwargs(names, info, q):
    """Translates sybil_amd query kwargs (column names) into oracle.run_query kwargs (indices).
    info: {name: (info_min, info_max)}"""
    ix = {n: i for i, n in enumerate(names)}
    kw = {}
    kw["filters"] = [(ix[f[0]],) + tuple(f[1:]) for f in q.get("filters", [])]
    kw["groups"] = [ix[g] for g in q.get("groups", [])]
    kw["aggs"] = [(ix[a], info[a][0], info[a][1]) for a in q.get("aggs", [])]
    kw["op"] = q.get("op", "avg")
    kw["hist_bucket"] = q.get("hist_bucket", 0)
    if q.get("time_col"):
        kw["time_col"] = ix[q["time_col"]]
        kw["time_bucket"] = q.get("time_bucket", 0)
    if q.get("weight_col"):
        kw["weight_col"] = ix[q["weight_col"]]
    kw["block_skip"] = q.get("block_skip", False)
    if q.get("loghist"):
        kw["loghist"] = True
    return kw


def _close(a, b, rel=REL, scale=0.0):
    """|a-b| <= rel * max(|a|,|b|,scale): `scale` is the magnitude of the data the quantity was
    derived from (a stddev of 0 vs 2e-15 on values ~1e3 is agreement, not a 100% error)."""
    if a == b or (a != a and b != b):  # both NaN: an existing hist without accepted values (0/0 in Go too)
        return True
    return abs(a - b) <= rel * max(abs(a), abs(b), scale)


def compare_hist(g, o, op, full, ctx="", cumulative=False):
    assert bool(g["present"]) == bool(o["present"]), ctx
    if not o["present"]:
        return
    assert g["count"] == o["count"], (ctx, g["count"], o["count"])
    assert g["samples"] == o["samples"], ctx
    assert g["sum"] == o["sum_exact"], (ctx, g["sum"], o["sum_exact"])
    assert (g["min"], g["max"]) == (o["min"], o["max"]), (ctx, g["min"], g["max"], o["min"], o["max"])
    # avg: exact sum/count here vs the reference-order running mean of the oracle.  The reference's
    # mean turns into NaN for good once a hist with Count == 0 (every value of the group in that
    # block rejected by the Info.Min/Max gate) is combined (0/0 in hist_basic.go:264-265) -- and
    # whether that happens depends on block completion order.  The engine returns the exact mean.
    o_avg = o["avg"]
    if o_avg != o_avg:
        o_avg = o["sum_exact"] / o["count"] if o["count"] else 0.0
    # (relative to the magnitude of the values when the exact mean is (nearly) zero -- sums of signed values that cancel:
    # the reference-order running mean then holds its rounding residue, ~1e-12, against an exact 0.0)
    vscale = 1e-6 * max(abs(o["true_min"]), abs(o["true_max"])) if o["count"] else 0.0
    assert _close(g["avg"], o_avg, REL, vscale), (ctx, g["avg"], o["avg"])
    if op == "hist":
        assert g["bucket_size"] == o["bucket_size"] and g["n_values"] == o["n_values"], ctx
        assert g["num_buckets"] == o["num_buckets"], ctx
        assert g["n_outliers"] == o["n_outliers"] + o["n_underliers"], ctx
        scale = max(abs(o["avg"]), abs(o["bucket_size"]), 1.0)
        assert _close(g["stddev"], o["stddev_exact"], 1e-9, scale), (ctx, g["stddev"], o["stddev_exact"])
        if o["n_outliers"] + o["n_underliers"] == 0 and o["avg"] == o["avg"]:
            assert _close(g["stddev"], o["stddev_ref"], REL, scale), (ctx, g["stddev"], o["stddev_ref"])
        if full:
            assert np.array_equal(g["values"], o["values"]), ctx
            # GetPercentiles returns an empty slice while Count == 0 (hist_basic.go:154-156)
            assert np.array_equal(g.get("percentiles", np.zeros(0, dtype=np.int64)), o["percentiles"]), ctx
            # the outliers' values (every block's, where the reference keeps one block's: hist_basic.go:259-279)
            if g["n_outlier_values"] >= 0 and not cumulative:
                assert np.array_equal(g.get("outlier_values", np.zeros(0, dtype=np.int64)), o["outlier_values"]), ctx
    else:
        assert g["stddev"] == 0.0, ctx


def compare_loghist(g, o, subs, ctx=""):
    """-loghist (MultiHist): outer fields, sub-histogram geometry, every bucket, the sub-histograms' outliers (exact
    counters on the engine side), percentiles bit-exact, stddev against the oracle's exact variant."""
    assert bool(g["present"]) == bool(o["present"]), ctx
    if not o["present"]:
        return
    assert (g["count"], g["samples"], g["sum"]) == (o["count"], o["samples"], o["sum_exact"]), ctx
    assert (g["min"], g["max"]) == (o["min"], o["max"]), (ctx, g["min"], g["max"], o["min"], o["max"])
    o_avg = o["avg"] if o["avg"] == o["avg"] else (o["sum_exact"] / o["count"] if o["count"] else 0.0)
    vscale = 1e-6 * max(abs(o["true_min"]), abs(o["true_max"])) if o["count"] else 0.0
    assert _close(g["avg"], o_avg, REL, vscale), (ctx, g["avg"], o["avg"])
    if "subhists" not in o:  # avg mode: no sub-histograms
        assert g["stddev"] == 0.0
        return
    assert [(s["info_min"], s["info_max"], s["bucket_size"], s["num_buckets"], s["n_values"]) for s in subs] == \
           [s[:5] for s in o["subhists"]], ctx
    outliers = []
    for s, os_ in zip(subs, o["subhists"]):
        assert np.array_equal(g["values"][s["offset"]:s["offset"] + s["n_values"]], o["values"][os_[5]:os_[5] + os_[4]]), (ctx, s)
        ext = g["values"][s["ext_offset"]:s["ext_offset"] + s["n_ext"]]
        for k in np.flatnonzero(ext):
            outliers += [s["ext_first"] + int(k)] * int(ext[k])
    assert sorted(outliers) == o["outlier_values"].tolist(), ctx
    assert g["n_outliers"] == len(outliers)
    assert np.array_equal(g.get("percentiles", np.zeros(0, dtype=np.int64)), o["percentiles"]), ctx
    assert _close(g["stddev"], o["stddev_exact"], 1e-9, max(abs(o_avg), 1.0)), (ctx, g["stddev"], o["stddev_exact"])


def compare(gres, ores, op="avg", full=True, n_aggs=0, time_mode=False, loghist=False):
    """gres: sybil_amd.Result; ores: dict from oracle.run_query.  Bit-exact on counts, sums, keys,
    buckets, percentiles, extrema; REL on avg/stddev."""
    assert gres.matched == ores["matched"], (gres.matched, ores["matched"])
    if loghist:
        subs = [gres.subhists(a) for a in range(n_aggs)]
        for which, name in ((0, "results"), (1, "time_results")):
            gmap = {(r["time_bucket"], r["key"]): r for r in gres.rows(which)}
            omap = {(r["time_bucket"], r["key"]): r for r in ores[name]}
            assert set(gmap) == set(omap), name
            for k, o in omap.items():
                assert gmap[k]["count"] == o["count"] and gmap[k]["samples"] == o["samples"]
                for a in range(n_aggs):
                    compare_loghist(gmap[k]["hists"][a], o["hists"][a], subs[a], ctx=(name, k, a))
        if not time_mode:
            for a in range(n_aggs):
                compare_loghist(gres.cumulative["hists"][a], ores["cumulative"]["hists"][a], subs[a], ctx=("cumulative", a))
        return
    for which, name in ((0, "results"), (1, "time_results")):
        grows = gres.rows(which)
        orows = ores[name]
        gmap = {(r["time_bucket"], r["key"]): r for r in grows}
        omap = {(r["time_bucket"], r["key"]): r for r in orows}
        assert len(gmap) == len(grows), "duplicate keys from the GPU path"
        assert set(gmap) == set(omap), (name, len(gmap), len(omap), sorted(set(gmap) ^ set(omap))[:5])
        for k, o in omap.items():
            g = gmap[k]
            assert g["count"] == o["count"] and g["samples"] == o["samples"], (name, k, g["count"], o["count"])
            for a in range(n_aggs):
                compare_hist(g["hists"][a], o["hists"][a], op, full, ctx=(name, k, a))
    gc, oc = gres.cumulative, ores["cumulative"]
    assert gc["count"] == oc["count"] and gc["samples"] == oc["samples"]
    if not time_mode:
        for a in range(n_aggs):
            compare_hist(gc["hists"][a], oc["hists"][a], op, full, ctx=("cumulative", a), cumulative=True)


def run_both(ctx, orc, names, total_rows, row0, nrows, q, block_rows=65536, oracle_threads=4, compact=False):
    """Synthetic table on the GPU and in host memory; same query through both."""
    t = ctx.synth_table("synth", synth.SEED, total_rows, row0, nrows, synth.synth_cols(names))
    try:
        if compact:
            t.compact()
        query = t.query(**q)
        try:
            gres = query.run()
            stats = query.stats()
        finally:
            query.free()
        info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in names}
        ocols = oracle_synth_cols(orc, names, total_rows, row0, nrows)
        ores = orc.run_query(ocols, block_rows=block_rows, n_threads=oracle_threads,
                             **oracle_query_kwargs(names, info, q))
        return gres, ores, stats
    finally:
        t.free()
