#!/usr/bin/env python3
"""Config 3 (3 ANDed int-range filters, group-by 2, moments of 2 columns) at different selectivities on compact
storage: does the row body's LDS-atomic issue still cost time when almost no row matches?
usage: bench_selectivity.py [rows] [steps] [packed|hash|wide|nul]
  nul: the NUL variant of k_scan_packed forced (SYBL_FORCE_NUL=1: validity words, per-aggregation gates)
  hash: the same queries through the hash table (SYBL_FORCE_HASH=1: k_scan_hash_packed); wide: four aggregation columns
  (k_scan_hash_packed<4, .., HASH = false>, the run-time-count direct-mapped body)
  parthist: config 4 (histograms by 65 536 groups: k_count_packed + k_emit_packed + k_part_hist) with the same filters;
  SYBL_NO_COUNT_CACHE=1 is set so that every scan pays its counting pass"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variant = sys.argv[3] if len(sys.argv) > 3 else "packed"
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
base = dict(wl["query"])
cols = list(wl["columns"])
if variant == "hash":
    os.environ["SYBL_FORCE_HASH"] = "1"
elif variant == "nul":
    os.environ["SYBL_FORCE_NUL"] = "1"
elif variant == "wide":
    cols += ["c09"]
    base["aggs"] = ["c07", "c08", "c09", "c04"]
elif variant == "parthist":
    w4 = synth.WORKLOADS["cfg4_hist_highcard"]
    base = dict(w4["query"])
    cols = list(w4["columns"]) + ["c04", "c05", "c06"]
    os.environ["SYBL_NO_COUNT_CACHE"] = "1"
ctx = sybil_amd.Context(0)
t = ctx.synth_table("sel", synth.SEED, rows, 0, rows, synth.synth_cols(cols))
t.compact()
cases = [
    ("51% (gt:99,lt:900 x3: the headline)", wl["query"]["filters"]),
    ("10% (gt:535 x3)", [("c04", "gt", 535), ("c05", "gt", 535), ("c06", "gt", 535)]),
    ("1% (gt:989 on c04 only)", [("c04", "gt", 989), ("c05", "gt", -1), ("c06", "gt", -1)]),
    ("0.1% (gt:899 x3)", [("c04", "gt", 899), ("c05", "gt", 899), ("c06", "gt", 899)]),
    ("1e-6 (gt:989 x3)", [("c04", "gt", 989), ("c05", "gt", 989), ("c06", "gt", 989)]),
    ("0 (gt:999 on c04)", [("c04", "gt", 999), ("c05", "gt", -1), ("c06", "gt", -1)]),
]
for label, filters in cases:
    q = t.query(**dict(base, filters=filters))
    ms = []
    for _ in range(steps + 3):  # (back-to-back scans, no finalize in between: the GPU stays at its working clocks)
        q.scan()
        ctx.sync()
        ms.append(q.stats()["scan_ms"])
    ms = ms[3:]
    r = q.finalize()
    matched = r.matched
    r.free()
    st = q.stats()
    k = sorted(ms)[len(ms) // 2]
    print(json.dumps({"variant": variant, "late_path": os.environ.get("SYBL_LATE_PATH", "planner's estimate"), "selectivity": label, "rows": rows, "matched": matched, "match_frac": matched / rows, "kernel_ms": round(k, 3),
                      "GBps": st["algorithmic_bytes"] / (k * 1e-3) / 1e9, "packed_kernel": st["packed_kernel"], "strategy": st["strategy"]}))
    sys.stdout.flush()
    q.free()
t.free()
