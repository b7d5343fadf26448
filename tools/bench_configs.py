#!/usr/bin/env python3
"""Times every BASELINE.json workload on one GPU: rows/s, scan-kernel ms, achieved GB/s, strategy.
usage: bench_configs.py [rows_cap] [steps] [name-filter,...|all] [compact]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else None
compact = len(sys.argv) > 4 and sys.argv[4] == "compact"
ctx = sybil_amd.Context(0)
for name, wl in synth.WORKLOADS.items():
    if only and not any(o in name for o in only):
        continue
    rows = min(wl["rows"], cap) if cap else wl["rows"]
    t = ctx.synth_table(name, synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    if compact:
        t.compact()
    q = t.query(**wl["query"])
    q.run().free()
    ms, wall = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        r = q.run()
        wall.append(time.perf_counter() - t0)
        ms.append(q.stats()["scan_ms"])
        groups, matched = len(r.results) or len(r.time_results), r.matched
        r.free()
    st = q.stats()
    k = sorted(ms)[len(ms) // 2]
    w = sorted(wall)[len(wall) // 2]
    print(json.dumps({"workload": name, "rows": rows, "strategy": st["strategy"], "cells": st["n_cells"],
                      "kernel_ms": round(k, 3), "step_ms": round(w * 1e3, 3), "rows_per_s": rows / w,
                      "GBps": st["algorithmic_bytes"] / (k * 1e-3) / 1e9, "bytes_per_row": st["algorithmic_bytes"] / rows,
                      "canonical_GBps": st["canonical_bytes"] / (k * 1e-3) / 1e9, "kernel_rows_per_s": rows / (k * 1e-3),
                      "matched": matched, "groups": groups}))
    sys.stdout.flush()
    q.free()
    t.free()
