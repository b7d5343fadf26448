#!/usr/bin/env python3
"""Headline kernel (config 3, compact, 1e9 rows) at the environment's SYBL_WG_PER_CU / SYBL_REP_BUDGET_KB."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
t.compact()
for env in ({}, {"SYBL_WG_PER_CU": "2", "SYBL_REP_BUDGET_KB": "76"}, {"SYBL_REP_BUDGET_KB": "76"}):
    os.environ.update(env)
    q = t.query(**dict(wl["query"], order_by=None))
    q.scan(); ctx.sync()
    ms = []
    for _ in range(5):
        q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
    st = q.stats()
    r = q.finalize(); m = r.matched; r.free()
    print(json.dumps({"env": env, "scan_ms": round(sorted(ms)[2], 3), "TBps": round(16.0 / sorted(ms)[2], 3), "wgs": st["n_workgroups"], "lds": st["lds_bytes"], "replicas": st["replicas"], "matched": m}))
    sys.stdout.flush()
    q.free()
    for k in env: del os.environ[k]
