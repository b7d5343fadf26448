#!/usr/bin/env python3
"""Workgroup size A/B of the row bodies behind the packed one (round 6): config 3 over compact storage, 1e9 rows, through the
plan-interpreting k_scan / k_scan_hash, k_scan_hash_fast, k_scan_hash_packed (hashed and direct-mapped with four aggregations)
at 1024 / 768 / 512 threads per workgroup.  usage: bench_threads.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
cols = wl["columns"] + ["c09"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(cols))
t.compact()
base = dict(wl["query"], order_by=None)
CASES = [
    ("generic", base, {"SYBL_NO_FAST": "1"}, "SYBL_SCAN_THREADS"),
    ("hash_generic", base, {"SYBL_FORCE_HASH": "1", "SYBL_NO_HASH_FAST": "1"}, "SYBL_SCAN_THREADS"),
    ("hash_fast", base, {"SYBL_FORCE_HASH": "1", "SYBL_NO_HASH_PACKED": "1"}, "SYBL_HASH_FAST_THREADS"),
    ("hash_packed", base, {"SYBL_FORCE_HASH": "1"}, "SYBL_HASH_PACKED_THREADS"),
    ("hist x4 (packed_n)", dict(base, aggs=["c07", "c08", "c09", "c04"]), {}, "SYBL_HASH_PACKED_THREADS"),
    ("avg x3 (packed_n)", dict(base, aggs=["c07", "c08", "c09"], op="avg"), {}, "SYBL_HASH_PACKED_THREADS"),
    ("hist x4 generic", dict(base, aggs=["c07", "c08", "c09", "c04"]), {"SYBL_NO_PACKED_N": "1"}, "SYBL_SCAN_THREADS"),
]
only = os.environ.get("BENCH_ONLY")
for label, q, env, knob in CASES:
    if only and only not in label:
        continue
    for T, extra in ((1024, {}), (768, {}), (512, {}), (512, {"SYBL_WG_PER_CU": "2", "SYBL_REP_BUDGET_KB": "72"})):
        e = dict(env, **extra)
        e[knob] = str(T)
        os.environ.update(e)
        try:
            qy = t.query(**q)
            qy.scan(); ctx.sync()
            ms = []
            for _ in range(3):
                qy.scan(); ctx.sync(); ms.append(qy.stats()["scan_ms"])
            st = qy.stats()
            r = qy.finalize()
            print(json.dumps({"case": label, "threads": T, "extra": extra, "strategy": st["strategy"], "packed": st["packed_kernel"], "lds": st["lds_bytes"],
                              "n_wg": st["n_workgroups"], "scan_ms": round(sorted(ms)[1], 3), "matched": r.matched, "groups": r.materialize(0)}))
            r.free()
            qy.free()
        except Exception as ex:
            print(json.dumps({"case": label, "threads": T, "extra": extra, "error": str(ex)[:200]}))
        sys.stdout.flush()
        for k in e:
            del os.environ[k]
