#!/usr/bin/env python3
"""Config 4 as a printer's query with -limit pushed into the scan (printed_only = 2, csrc/pushdown.hip) beside the full path
(printed_only = 1): scan-kernel ms of back-to-back scans.  usage: bench_pushdown.py [rows] [scans]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = synth.WORKLOADS["cfg4_hist_highcard"]
ctx = sybil_amd.Context(0)
t = ctx.synth_table("pd", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
t.compact()
for level in (1, 2):
    q = t.query(**dict(wl["query"], limit=100, order_by="$COUNT", printed_only=level))
    ms = []
    for _ in range(scans + 2):
        q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
    r = q.finalize()
    st = q.stats()
    print(json.dumps({"printed_only": level, "strategy": st["strategy"], "scan_ms_median": round(sorted(ms[2:])[len(ms[2:]) // 2], 3), "matched": r.matched}))
    r.free(); q.free()
