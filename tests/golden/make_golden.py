#!/usr/bin/env python3
"""Regenerates tests/golden/node_results_hists.json from the reference's own golden
NodeResults file (src/lib/testdata/TestDecodeGoldenFiles/node_results.golden.json,
pinned by src/lib/decoding_test.go:20-74).  Run in the build container only --
/root/reference does not exist on the GPU box; the JSON it writes is committed.

What is kept: for the Cumulative result and each of the 12 group results, the
BasicHist fields the scan path produces (Values, Count, Avg, Min, Max, Outliers,
BucketSize, NumBuckets, Info.Min/Max) plus Result.Count/Samples/BinaryByKey/GroupByKey.
Bucket `Averages` are dropped (not merged by Combine, not printed).
"""
import json
import os
import sys

REFDIR = "/root/reference/src/lib/testdata/TestDecodeGoldenFiles"
REF = REFDIR + "/node_results.golden.json"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "node_results_hists.json")
# the reference's second golden (decoding_test.go:26): a gob-encoded FlagDefs.  Kept as a hex dump of
# the stream (the byte-exact pin of tests/gobfmt.py and csrc/gob.cpp) and as the expected decoded
# value (the fields Go re-marshals as non-null).
FLAGS_HEX = os.path.join(HERE, "flagdefs_stream.hex")
FLAGS_EXPECT = os.path.join(HERE, "flagdefs_expected.json")


def hist(h):
    return {"NumBuckets": h["NumBuckets"], "BucketSize": h["BucketSize"], "Values": h["Values"],
            "PercentileMode": h["PercentileMode"], "Outliers": h["Outliers"] or [],
            "Underliers": h["Underliers"] or [], "Max": h["Max"], "Min": h["Min"], "Samples": h["Samples"],
            "Count": h["Count"], "Avg": h["Avg"], "InfoMin": h["Info"]["Min"], "InfoMax": h["Info"]["Max"]}


def result(r):
    return {"GroupByKey": r["GroupByKey"], "BinaryByKeyHex": r["BinaryByKey"].encode("latin-1").hex(),
            "Count": r["Count"], "Samples": r["Samples"],
            "Hists": {k: hist(v) for k, v in r["Hists"].items()}}


def main():
    qs = json.load(open(REF))["QuerySpec"]
    out = {"source": "logv/sybil src/lib/testdata/TestDecodeGoldenFiles/node_results.golden.json",
           "Groups": [g["Name"] for g in qs["Groups"]],
           "Aggregations": qs["Aggregations"], "OrderBy": qs["OrderBy"], "Limit": qs["Limit"],
           "MatchedCount": qs["MatchedCount"],
           "Cumulative": result(qs["Cumulative"]),
           "Results": {k: result(v) for k, v in qs["Results"].items()},
           "SortedKeys": [r["GroupByKey"] for r in qs["Sorted"]]}
    json.dump(out, open(OUT, "w"), separators=(",", ":"), sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    raw = open(REFDIR + "/flag_defs.golden.gob", "rb").read()
    open(FLAGS_HEX, "w").write("\n".join(raw[i:i + 32].hex() for i in range(0, len(raw), 32)) + "\n")
    want = json.load(open(REFDIR + "/flag_defs.golden.json"))
    json.dump({k: v for k, v in want.items() if v is not None}, open(FLAGS_EXPECT, "w"), indent=1, sort_keys=True)
    print("wrote", FLAGS_HEX, "and", FLAGS_EXPECT)
    # the WIRE TYPES of the reference's gob-encoded NodeResults (node_results.golden.gob): struct / field
    # names and the kind of every field, which is what a Go decoder matches an incoming stream against --
    # the pin for sybl_result_encode (-encode-results)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import gobfmt
    src = open(gobfmt.__file__).read().replace("return (v, order) if want_types else v",
                                               "return (v, order, types) if want_types else v")
    ns = {}
    exec(compile(src, "gobfmt_types", "exec"), ns)
    _, _, types = ns["decode"](open(REFDIR + "/node_results.golden.gob", "rb").read(), want_types=True)
    builtin = {1: "bool", 2: "int", 3: "uint", 4: "float", 5: "bytes", 6: "string", 8: "interface"}

    def kind(tid):
        if tid in builtin:
            return builtin[tid]
        t = types[tid]
        if t["kind"] == "struct":
            return "struct:" + t["name"]
        if t["kind"] == "map":
            return "map[%s]%s" % (kind(t["key"]), kind(t["elem"]))
        if t["kind"] in ("slice", "array"):
            return "[]" + kind(t["elem"])
        return t["kind"]

    wire = {t["name"]: {f: kind(fid) for f, fid in t["fields"]} for t in types.values() if t["kind"] == "struct" and t["name"]}
    keep = ("NodeResults", "QuerySpec", "QueryParams", "Grouping", "Aggregation", "QueryResults", "Result", "HistCompat",
            "BasicHist", "BasicHistCachedInfo")
    json.dump({k: wire[k] for k in keep}, open(os.path.join(HERE, "node_results_wiretypes.json"), "w"), indent=1, sort_keys=True)
    print("wrote node_results_wiretypes.json")


if __name__ == "__main__":
    sys.exit(main())
