#!/usr/bin/env python3
"""Config 3 with FOUR aggregation columns (c07, c08, c09, c04) over compact storage, 1e9 rows: the run-time-count packed
body (k_scan_hash_packed<4, .., HASH = false>) against the plan-interpreting k_scan (SYBL_NO_PACKED_N=1), and avg mode
with three.  usage: bench_wide.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
cols = wl["columns"] + ["c09"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(cols))
t.compact()
for label, q, off in (("hist x4", dict(wl["query"], aggs=["c07", "c08", "c09", "c04"], order_by=None), "SYBL_NO_PACKED_N"),
                      ("avg x3", dict(wl["query"], aggs=["c07", "c08", "c09"], op="avg", order_by=None), "SYBL_NO_PACKED_N"),
                      # every bucket wanted (percentiles): two passes of the partitioned histograms against one device-scope
                      # atomic per value
                      ("hist x4 + buckets", dict(wl["query"], aggs=["c07", "c08", "c09", "c04"], want_percentiles=True, order_by=None), "SYBL_NO_PARTHIST"),
                      # config 3 with -hist-bucket 990: the top ~1 % of c07 / c08 lie beyond the last bucket (outliers): the NUL
                      # variant of k_scan_packed against the any-width GEN body
                      ("cfg3, ~1 % outliers", dict(wl["query"], hist_bucket=990, order_by=None), "SYBL_NO_PACKED")):
    for env in ({}, {off: "1"}):
        os.environ.update(env)
        qy = t.query(**q)
        qy.scan(); ctx.sync()
        ms = []
        for _ in range(3):
            qy.scan(); ctx.sync(); ms.append(qy.stats()["scan_ms"])
        st = qy.stats()
        print(json.dumps({"query": label, "env": env, "strategy": st["strategy"], "packed_kernel": st["packed_kernel"], "scan_ms": round(sorted(ms)[1], 3),
                          "GBps": round(st["algorithmic_bytes"] / (sorted(ms)[1] * 1e-3) / 1e9, 1)}))
        sys.stdout.flush()
        qy.free()
        for k in env: del os.environ[k]
t.free()
# config 4 (65 536 groups, every bucket kept) with -hist-bucket 990: ~1 % of c07 are outliers; the partitioned histograms
# (k_part_hist remembers them) against one device-scope atomic per value
wl4 = synth.WORKLOADS["cfg4_hist_highcard"]
t = ctx.synth_table("b", synth.SEED, rows, 0, rows, synth.synth_cols(wl4["columns"]))
t.compact()
for env in ({}, {"SYBL_NO_PARTHIST": "1"}):
    os.environ.update(env)
    qy = t.query(**dict(wl4["query"], hist_bucket=990, order_by=None))
    qy.scan(); ctx.sync()
    ms = []
    for _ in range(3):
        qy.scan(); ctx.sync(); ms.append(qy.stats()["scan_ms"])
    st = qy.stats()
    print(json.dumps({"query": "cfg4, ~1 % outliers", "env": env, "strategy": st["strategy"], "packed_kernel": st["packed_kernel"],
                      "scan_ms": round(sorted(ms)[1], 3), "GBps": round(st["algorithmic_bytes"] / (sorted(ms)[1] * 1e-3) / 1e9, 1)}))
    sys.stdout.flush()
    qy.free()
    for k in env: del os.environ[k]
