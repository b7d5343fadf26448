#!/usr/bin/env python3
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_hist_highcard"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
wl = synth.WORKLOADS[name]
ctx = sybil_amd.Context(0)
t = ctx.synth_table("b", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
kw = dict(wl["query"])
if len(sys.argv) > 3:
    kw["limit"] = int(sys.argv[3])
q = t.query(**kw)
q.run().free()
for _ in range(3):
    t0 = time.perf_counter(); q.scan(); ctx.sync(); t1 = time.perf_counter(); r = q.finalize(); t2 = time.perf_counter(); r.free(); t3 = time.perf_counter()
    print("scan+sync %.1f ms  finalize %.1f ms  free %.1f ms  kernel %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, q.stats()["scan_ms"]))
