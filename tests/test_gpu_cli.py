"""GPU: the output surface (`sybil query` text / -json formats, src/lib/printer.go:25-308) through
the sybil-gpu-query CLI (tools/sybil_gpu_query.cpp), which accepts the reference's query flags
(src/cmd/cmd_query.go:19-74).  Expected strings are derived by hand from printer.go's format
verbs; numeric content is checked against the CPU oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import sybil_fixture as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rows_mode")]  # (each test also with SYBL_LAZY_ROWS=1: conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "sybil_amd", "sybil-gpu-query")


def _run(*args, tz="UTC"):
    env = dict(os.environ, TZ=tz)
    p = subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("clidb"))
    # two blocks of a tiny, fully known table
    b1 = {"browser": ("str", ["edge", "edge", "gecko", "webkit", "gecko", "edge"]),
          "load": ("int", np.array([100, 300, 50, 1000, 150, 200])),
          "time": ("int", np.array([1700000000, 1700000100, 1700003700, 1700003800, 1700007300, 1700007400])),
          "tags": ("set", [["a"], ["a", "b"], None, ["b"], ["c"], ["a", "c"]])}
    b2 = {"browser": ("str", ["webkit", "gecko", "edge", "edge"]),
          "load": ("int", np.array([400, 250, 500, 600])),
          "time": ("int", np.array([1700007500, 1700010900, 1700011000, 1700011100])),
          "tags": ("set", [["b"], ["a"], None, ["c"]])}
    F.write_table(root, "pages", [b1, b2], int_info={"load": (0, 1000)})
    return root


def test_text_avg(db):
    out = _run("-dir", db, "-table", "pages", "-group", "browser", "-int", "load")
    # printSortedResults: TOTAL first (more than one result), groups by Count desc; printResult:
    # "%-20s"[:20] key, "%.0d" count, then "  %5s" name and "%.2f" mean (printer.go:183-232)
    assert out == ("TOTAL               10\n"
                   "   load 355.00\n"
                   "edge                5\n"
                   "   load 340.00\n"
                   "gecko               3\n"
                   "   load 150.00\n"
                   "webkit              2\n"
                   "   load 700.00\n")


def test_text_hist_and_filters(db):
    out = _run("-dir", db, "-table", "pages", "-group", "browser", "-int", "load", "-op", "hist",
               "-int-filter", "load:gt:99,load:lt:601", "-set-filter", "tags:nin:b")
    lines = out.splitlines()
    # edge rows passing: 100, 200 (tags a / a,c) + 600 (c); 300 has tag b, 500 has no tags -> nin fails
    # gecko: 150 (c), 250 (a);  webkit: 1000 filtered by lt, 400 has b
    assert lines[0].startswith("TOTAL               5")
    assert lines[2] == "edge                3" and lines[4] == "gecko               2"
    # Info [0,1000] -> BucketSize 1: percentiles are exact values; "  load | p0 p99 | avg | p0 p25 p50 p75 p99 | std"
    assert lines[3] == "   load | 100 600 | 300.00 | 100 100 200 600 600 | 216.02"
    assert lines[5] == "   load | 150 250 | 200.00 | 150 150 250 250 250 | 50.00"


def test_json_hist(db, oracle):
    out = _run("-dir", db, "-table", "pages", "-group", "browser", "-int", "load", "-op", "hist", "-json", "-limit", "2")
    rows = json.loads(out)
    assert [r["browser"] for r in rows] == ["edge", "gecko"]        # sorted by count, limited to 2
    e = rows[0]
    assert e["Count"] == 5 and e["Samples"] == 5
    h = e["load"]
    assert sorted(h) == ["avg", "buckets", "percentiles", "samples", "stddev", "sum"]   # toResultJSON, printer.go:109-125
    assert h["avg"] == 340.0 and h["sum"] == 1700.0 and h["samples"] == 5
    assert h["buckets"] == {"100": 1, "200": 1, "300": 1, "500": 1, "600": 1}
    assert len(h["percentiles"]) == 100 and h["percentiles"][0] == 100 and h["percentiles"][99] == 600
    ref = oracle.Hist(0, 1000, "hist")
    for v in (100, 300, 200, 500, 600):
        ref.add(v)
    assert h["stddev"] == pytest.approx(ref.info()["stddev_ref"], rel=1e-9)
    assert list(ref.percentiles()) == h["percentiles"]
    # encoding/json writes map keys sorted and floats in shortest form
    assert out.startswith('[{"Count":5,"Samples":5,"browser":"edge","load":{"avg":340,"buckets":{"100":1,')


def test_json_avg_and_no_groups(db):
    rows = json.loads(_run("-dir", db, "-table", "pages", "-int", "load", "-json"))
    assert rows == [{"Count": 10, "Samples": 10, "load": 355}]
    rows = json.loads(_run("-dir", db, "-table", "pages", "-group", "browser", "-json", "-str-filter", "browser:re:^e"))
    assert rows == [{"Count": 5, "Samples": 5, "browser": "edge"}]
    rows = json.loads(_run("-dir", db, "-table", "pages", "-group", "browser", "-json", "-str-filter", "browser:neq:edge",
                           "-sort-asc"))
    assert [(r["browser"], r["Count"]) for r in rows] == [("webkit", 2), ("gecko", 3)]


def test_time_series_json_and_text(db):
    out = _run("-dir", db, "-table", "pages", "-time", "-time-col", "time", "-time-bucket", "3600", "-group", "browser",
               "-int", "load", "-json")
    got = json.loads(out)
    # buckets: truncating division of the timestamp (aggregate.go:174)
    assert sorted(got) == ["1699999200", "1700002800", "1700006400", "1700010000"]
    first = {r["browser"]: r for r in got["1699999200"]}
    assert first == {"edge": {"Count": 2, "Samples": 2, "browser": "edge", "load": 200}}
    assert {r["browser"]: r["Count"] for r in got["1700006400"]} == {"gecko": 1, "edge": 1, "webkit": 1}
    text = _run("-dir", db, "-table", "pages", "-time", "-time-bucket", "3600", "-group", "browser", "-int", "load")
    lines = text.splitlines()
    assert len(lines) == 8
    # Fprintln gives "<time> \t <count> \t <key>\t \t <agg> \t <avg> \t"; tabwriter (AlignRight, padding 0)
    # right-aligns each tab-terminated cell to its column's widest cell: " edge" -> 7 (" webkit"),
    # " 200.00 " -> 9 (" 1000.00 ")
    assert lines[0] == "2023-11-14 22:00:00 +0000 UTC  2    edge  load   200.00 "
    assert "2023-11-14 23:00:00 +0000 UTC  1  webkit  load  1000.00 " in lines
    assert "2023-11-14 23:00:00 +0000 UTC  1   gecko  load    50.00 " in lines
    assert all(l.startswith("2023-11-1") for l in lines)
    # a time filter is aligned down to its bucket in a time-series query (filter.go:86-95)
    got2 = json.loads(_run("-dir", db, "-table", "pages", "-time", "-time-bucket", "3600", "-group", "browser", "-json",
                           "-int-filter", "time:gt:1700006500"))
    assert sorted(got2) == ["1700006400", "1700010000"]


def test_cli_errors(db):
    p = subprocess.run([CLI, "-dir", db, "-table", "nope"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"open table" in p.stderr
    p = subprocess.run([CLI, "-dir", db, "-table", "pages", "-bogus"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 2 and b"flag provided but not defined" in p.stderr
    p = subprocess.run([CLI, "-dir", db, "-table", "pages", "-group", "tags"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"set column" in p.stderr


# ---------------------------------------------------------------- -encode-results (gob NodeResults)

def _wire_kinds(types):
    builtin = {1: "bool", 2: "int", 3: "uint", 4: "float", 5: "bytes", 6: "string", 8: "interface"}

    def kind(tid):
        if tid in builtin:
            return builtin[tid]
        t = types[tid]
        if t["kind"] == "struct":
            return "struct"
        if t["kind"] == "map":
            return "map[%s]%s" % (kind(t["key"]), kind(t["elem"]))
        if t["kind"] in ("slice", "array"):
            return "[]" + kind(t["elem"])
        return t["kind"]

    return {t["name"]: {f: kind(fid) for f, fid in t["fields"]} for t in types.values() if t["kind"] == "struct" and t["name"]}


def _decode_with_types(data):
    from tests import gobfmt
    src = open(gobfmt.__file__).read().replace("return (v, order) if want_types else v", "return (v, order, types) if want_types else v")
    ns = {}
    exec(compile(src, "gobfmt_types", "exec"), ns)
    return ns["decode"](data, want_types=True)


def test_encode_results_is_a_gob_node_results():
    """sybl_result_encode: a Go decoder matches an incoming stream against its own structs by struct / field
    NAME and wire kind.  Every struct the engine sends must therefore be a subset of what the reference's own
    gob-encoded NodeResults (node_results.golden.gob -> tests/golden/node_results_wiretypes.json) defines;
    the values are checked against the result rows."""
    import re
    import sybil_amd
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "node_results_wiretypes.json")))
    ctx = sybil_amd.Context(0)
    rng = np.random.default_rng(9)
    n = 20_000
    browsers, devices = ["webkit", "edge", "gecko"], ["laptop", "desktop", "phone", "tablet"]
    b = rng.integers(0, 3, size=n).astype(np.int32)
    d = rng.integers(0, 4, size=n).astype(np.int32)
    pl = rng.integers(30, 23_500, size=n).astype(np.int64)
    t = 1_700_000_000 + np.sort(rng.integers(0, 3 * 3600, size=n)).astype(np.int64)
    tb = ctx.create_table("enc")
    tb.add_column("browser", "str")
    tb.add_column("device", "str")
    tb.add_column("pageload", "int", 30, 23_500)
    tb.add_column("time", "int")
    tb.append_block(n, {"browser": {"ids": b, "strings": browsers}, "device": {"ids": d, "strings": devices}, "pageload": pl, "time": t})
    for q in (dict(groups=["browser", "device"], aggs=["pageload"], op="hist", order_by="$COUNT", limit=100),
              dict(groups=["device"], aggs=["pageload"], op="avg", time_col="time", time_bucket=3600)):
        query = tb.query(**q)
        r = query.run()
        v, _, types = _decode_with_types(r.encode())
        mine = _wire_kinds(types)
        for name, fields in mine.items():
            if name == "IntInfo":      # anonymous in the golden stream (a field of BasicHistCachedInfo)
                continue
            assert name in golden, name
            for f, k in fields.items():
                if (name, f) == ("QueryParams", "OrderAsc"):   # query_spec.go:33, newer than the golden stream
                    assert k == "bool"
                    continue
                assert f in golden[name], (name, f)
                assert re.sub(r"struct:\w*", "struct", golden[name][f]) == k, (name, f, golden[name][f], k)
        qs = v["QuerySpec"]
        assert [g["Name"] for g in qs["QueryParams"]["Groups"]] == q["groups"]
        assert qs["QueryParams"]["Aggregations"] == [{"Op": q["op"], "Name": "pageload", "HistType": "basic"}]
        res = qs["QueryResults"]
        assert res["MatchedCount"] == r.matched == n

        def check(enc, row):
            assert enc.get("Count", 0) == row["count"] and enc.get("Samples", 0) == row["samples"]
            assert enc.get("GroupByKey", "") == row["group_by_key"]
            assert enc.get("BinaryByKey", "").encode("utf-8") == row["key"]  # (small dictionary ids: plain ASCII range)
            h = row["hists"][0]
            if not h["present"]:
                assert "Hists" not in enc
                return
            iv = enc["Hists"]["pageload"]
            assert iv["@type"] == "*sybil.HistCompat"
            ci = iv["value"]["BasicHist"]["BasicHistCachedInfo"]
            assert ci.get("Count", 0) == h["count"] and ci.get("Avg", 0.0) == h["avg"]
            assert ci.get("Min", 0) == h["min"] and ci.get("Max", 0) == h["max"]
            assert ci["Info"] == {"Min": 30, "Max": 23_500}
            if q["op"] == "hist":
                assert ci["PercentileMode"] is True and ci["BucketSize"] == h["bucket_size"] and ci["NumBuckets"] == h["num_buckets"]
                assert ci["Values"] == h["values"].tolist()
                # values beyond the last bucket (30 + 1002 * 23 = 23076 and up) are remembered as Outliers (hist_basic.go:132-135)
                assert ci.get("Outliers", []) == h.get("outlier_values", np.zeros(0)).tolist() and "Underliers" not in ci
                assert len(ci.get("Outliers", [])) == (h["n_outliers"] if enc is not res["Cumulative"] else 0)
            else:
                assert "Values" not in ci

        check(res["Cumulative"], r.cumulative)
        if q.get("time_col"):
            rows = r.time_results
            assert sum(len(m) for m in res["TimeResults"].values()) == len(rows)
            for row in rows:
                check(res["TimeResults"][row["time_bucket"]][row["group_by_key"]], row)
            for row in r.results:                      # all-time Results: Count / Samples only
                check(res["Results"][row["group_by_key"]], row)
        else:
            rows = r.results
            assert set(res["Results"]) == {x["group_by_key"] for x in rows} and len(res["Sorted"]) == len(rows)
            for row, s in zip(rows, res["Sorted"]):
                check(res["Results"][row["group_by_key"]], row)
                check(s, row)
        r.free()
        query.free()
    tb.free()
    ctx.close()


def test_cli_encode_results(db):
    env = dict(os.environ, TZ="UTC")
    p = subprocess.run([CLI, "-dir", db, "-table", "pages", "-group", "browser", "-int", "load", "-op", "hist", "-encode-results"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    from tests import gobfmt
    v = gobfmt.decode(p.stdout)
    res = v["QuerySpec"]["QueryResults"]
    assert res["MatchedCount"] == 10 and res["Cumulative"]["Count"] == 10
    assert {k: r["Count"] for k, r in res["Results"].items()} == {"edge\t": 5, "gecko\t": 3, "webkit\t": 2}
    assert [r["GroupByKey"] for r in res["Sorted"]] == ["edge\t", "gecko\t", "webkit\t"]
    ci = res["Results"]["webkit\t"]["Hists"]["load"]["value"]["BasicHist"]["BasicHistCachedInfo"]
    assert ci["Count"] == 2 and ci["Avg"] == 700.0 and sum(ci["Values"]) == 2 and ci["Info"] == {"Max": 1000}


def test_c_example_runs(db, tmp_path):
    import sybil_amd
    libdir = os.path.dirname(os.path.abspath(sybil_amd.__file__))
    exe = str(tmp_path / "example_query")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "example_query.c"),
                           "-L", libdir, "-lsybilgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    p = subprocess.run([exe, db, "pages", "browser", "load", "100"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    lines = p.stdout.decode().splitlines()
    # load > 100: edge {300, 200, 500, 600}, gecko {150, 250}, webkit {1000, 400}
    assert lines[0] == "matched 8 rows, 3 groups"
    assert lines[1].split() == ["edge", "count", "4", "avg", "400.00", "p50", "500", "p99", "600"]
    assert lines[2].split()[:5] == ["gecko", "count", "2", "avg", "200.00"] and lines[3].split()[:5] == ["webkit", "count", "2", "avg", "700.00"]
    assert lines[4].startswith("-encode-results:")
    # ... and as two ranks of a multi-GPU job (the table's two blocks: one each), the collectives through the test-only RCCL
    # stand-in on this one device: rank 0 prints the same rows, rank 1 nothing
    standin = os.path.join(ROOT, "tests", "rccl_standin", "librccl_standin.so")
    env = dict(os.environ, LD_PRELOAD=standin, HSA_ENABLE_IPC_MODE_LEGACY="0", SYBL_STANDIN_TIMEOUT_S="120")
    idf = str(tmp_path / "id")
    procs = [subprocess.Popen([exe, db, "pages", "browser", "load", "100", str(r), "2", idf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
             for r in range(2)]
    outs = [pp.communicate(timeout=300) for pp in procs]
    assert [pp.returncode for pp in procs] == [0, 0], outs
    assert outs[0][0].decode().splitlines()[:4] == lines[:4] and outs[1][0] == b""
