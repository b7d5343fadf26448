#!/usr/bin/env python3
"""One workload's scan N times in one process, the scan time of every repetition (thermal / clock drift shows here).
usage: scan_loop.py workload n [compact]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
wl = synth.WORKLOADS[[k for k in synth.WORKLOADS if sys.argv[1] in k][0]]
n = int(sys.argv[2])
ctx = sybil_amd.Context(0)
t = ctx.synth_table("t", synth.SEED, wl["rows"], 0, wl["rows"], synth.synth_cols(wl["columns"]))
if len(sys.argv) > 3: t.compact()
q = t.query(**wl["query"])
ms = []
for i in range(n):
    q.scan(); ctx.sync()
    ms.append(q.stats()["scan_ms"])
print(time.time(), " ".join("%.2f" % x for x in ms))
