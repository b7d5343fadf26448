#!/usr/bin/env python3
"""Query shapes that used to leave the packed row body for the plan interpreter, over config 3's table in compact
storage: a fifth / sixth filter column, now evaluated by the filter pre-pass (k_prefilter + the packed body reading its row
bitmap), against the plan-interpreting k_scan (SYBL_NO_PREFILTER=1).  usage: bench_shapes.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"] + ["c09", "c03"]))
t.compact()
f5 = wl["query"]["filters"] + [("c09", "lt", 450), ("c03", "gt", 1000)]
f5s = wl["query"]["filters"] + [("c09", "lt", 450), ("c03", "gt", 65208)]  # the fifth filter column passes 0.5 % of the rows
cases = [("config 3 + filters on c09 and c03 (five filter columns)", dict(wl["query"], filters=f5, order_by=None)),
         ("the same with c03 > 65208 (the pre-pass column passes 0.5 %: most tiles hold no passing row)", dict(wl["query"], filters=f5s, order_by=None))]
for label, q in cases:
    for env in ({}, {"SYBL_NO_PREFILTER": "1"}):
        os.environ.update(env)
        qy = t.query(**q)
        ms = []
        for _ in range(6):
            qy.scan(); ctx.sync(); ms.append(qy.stats()["scan_ms"])
        st = qy.stats()
        k = sorted(ms[2:])[2]
        r = qy.finalize()
        print(json.dumps({"query": label, "env": env, "strategy": st["strategy"], "packed_kernel": st["packed_kernel"], "scan_ms": round(k, 3),
                          "matched": r.matched, "GBps": round(st["algorithmic_bytes"] / (k * 1e-3) / 1e9, 1)}))
        sys.stdout.flush()
        r.free()
        qy.free()
        for kk in env: del os.environ[kk]
