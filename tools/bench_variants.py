#!/usr/bin/env python3
"""Config 3 (compact storage, 1e9 rows) through different row bodies: usage: bench_variants.py [rows]
packed (the headline), SYBL_NO_PACKED (k_scan_fast<GEN>), hash (k_scan_hash_fast), hash generic (SYBL_NO_HASH_FAST)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
t.compact()
for label, env in (("packed", {}), ("gen", {"SYBL_NO_PACKED": "1"}), ("hash_fast", {"SYBL_FORCE_HASH": "1", "SYBL_NO_HASH_PACKED": "1"}),
                   ("hash_packed", {"SYBL_FORCE_HASH": "1"}), ("hash_generic", {"SYBL_FORCE_HASH": "1", "SYBL_NO_HASH_FAST": "1"}), ("generic", {"SYBL_NO_FAST": "1"})):
    os.environ.update(env)
    q = t.query(**dict(wl["query"], order_by=None))
    q.scan(); ctx.sync()
    ms = []
    for _ in range(3):
        q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
    print(json.dumps({"variant": label, "strategy": q.stats()["strategy"], "scan_ms": round(sorted(ms)[1], 3)}))
    sys.stdout.flush()
    q.free()
    for k in env: del os.environ[k]
