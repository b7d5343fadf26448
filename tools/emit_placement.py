#!/usr/bin/env python3
"""Does config 4's scan time depend on WHERE its record buffer landed?  Several prepared queries of the same shape on one
table, each with buffers of its own (5 GB of records + a 525 MB table), kept alive together and scanned in turns: if their
times differ inside one process, the spread seen from process to process (k_emit_packed 2.6-3.0 ms) is a matter of
physical placement, not of clocks or code.   usage: emit_placement.py [queries] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = synth.WORKLOADS["cfg4_hist_highcard"]
ctx = sybil_amd.Context(0)
t = ctx.synth_table("t", synth.SEED, wl["rows"], 0, wl["rows"], synth.synth_cols(wl["columns"]))
t.compact()
qs = [t.query(**wl["query"]) for _ in range(nq)]
for q in qs:
    q.scan(); ctx.sync()
print("# config 4, %d prepared queries alive together; median scan-kernel ms of 8 back-to-back scans per visit" % nq)
for r in range(rounds):
    row = []
    for q in qs:
        ms = []
        for _ in range(8):
            q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
        row.append(sorted(ms)[4])
    print("round %d: " % r + "  ".join("%.3f" % x for x in row))
