#!/usr/bin/env python3
"""k_emit's chunk stores with and without the non-temporal hint, on the SAME prepared queries (same buffers: the placement
lottery of tools/emit_placement.py cancels): SYBL_EMIT_PLAIN_STORES is read at every scan.   usage: emit_nt.py [queries] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = synth.WORKLOADS["cfg4_hist_highcard"]
ctx = sybil_amd.Context(0)
t = ctx.synth_table("t", synth.SEED, wl["rows"], 0, wl["rows"], synth.synth_cols(wl["columns"]))
t.compact()
qs = [t.query(**wl["query"]) for _ in range(nq)]
for q in qs:
    q.scan(); ctx.sync()
print("# config 4, %d prepared queries; per query: median scan-kernel ms of 8 back-to-back scans, plain stores / nt stores" % nq)
for r in range(rounds):
    row = []
    for q in qs:
        pair = []
        for nt in (False, True):
            if nt: os.environ.pop("SYBL_EMIT_PLAIN_STORES", None)
            else: os.environ["SYBL_EMIT_PLAIN_STORES"] = "1"
            ms = []
            for _ in range(8):
                q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
            pair.append(sorted(ms)[4])
        row.append("%.3f/%.3f" % tuple(pair))
    os.environ.pop("SYBL_EMIT_PLAIN_STORES", None)
    print("round %d: " % r + "  ".join(row))
r = qs[0].finalize(); print("matched", r.matched); r.free()
