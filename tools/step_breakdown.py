#!/usr/bin/env python3
"""Host-side timing of each phase of one bench step (scan launch, sync, finalize, stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
ctx = sybil_amd.Context(0)
t = ctx.synth_table("b", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
if len(sys.argv) > 2 and sys.argv[2] == "compact":
    t.compact()
q = t.query(**wl["query"])
for _ in range(3):
    q.run().free()
acc = {}
N = 20
for _ in range(N):
    t0 = time.perf_counter(); q.scan(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    r = q.finalize(); t3 = time.perf_counter(); st = q.stats(); t4 = time.perf_counter(); r.free(); t5 = time.perf_counter()
    for k, v in (("scan_launch", t1 - t0), ("sync_wait", t2 - t1), ("finalize", t3 - t2), ("stats", t4 - t3), ("free", t5 - t4),
                 ("kernel_ms", st["scan_ms"] / 1e3), ("fold_ms", st["reduce_ms"] / 1e3), ("total", t5 - t0)):
        acc[k] = acc.get(k, 0) + v
for k, v in acc.items():
    print("%-12s %8.1f us" % (k, v / N * 1e6))
