#!/usr/bin/env python3
"""End-to-end latency of small queries on a resident table: prepare / scan+sync / finalize / free."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
ctx = sybil_amd.Context(0)
for name in ("cfg1_count_range", "cfg2_group1_avg2"):
    wl = synth.WORKLOADS[name]
    rows = 10_000_000
    t = ctx.synth_table("x", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.query(**wl["query"]).run().free()
    acc = [0.0] * 5
    N = 20
    for _ in range(N):
        t0 = time.perf_counter(); q = t.query(**wl["query"]); t1 = time.perf_counter(); q.scan(); ctx.sync(); t2 = time.perf_counter()
        r = q.finalize(); t3 = time.perf_counter(); r.free(); q.free(); t4 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
            acc[i] += v
    print("%-18s prepare %.0f us  scan+sync %.0f us  finalize %.0f us  free %.0f us  total %.0f us" % ((name,) + tuple(a / N * 1e6 for a in acc)))
    t.free()
