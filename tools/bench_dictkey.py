#!/usr/bin/env python3
"""Config 3 (compact storage, 1e9 rows) with its second group column forced through a group dictionary
(sybl_table_set_group_dict: the digits are ranks among the distinct values, as for a sparse key): the derived rank column
(round 5: Column::rank_col, the query runs k_scan_packed) against the per-row dictionary probe of the plan-interpreting
k_scan (SYBL_NO_RANKCOL=1), and what laying the ranks out costs once.  usage: bench_dictkey.py [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = sybil_amd.Context(0)
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
t.compact()
t.set_group_dict("c02", list(range(64)))
want = None
for label, env in (("rank column", {}), ("dictionary probed per row", {"SYBL_NO_RANKCOL": "1"})):
    os.environ.update(env)
    t0 = time.perf_counter()
    q = t.query(**dict(wl["query"], order_by=None))
    prep = time.perf_counter() - t0
    q.scan(); ctx.sync()
    ms = []
    for _ in range(3):
        q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
    r = q.finalize()
    got = sorted((g["key"], g["count"], g["hists"][0]["sum"]) for g in r.results)
    assert want is None or got == want, "the two paths disagree"
    want = got
    r.free()
    st = q.stats()
    print(json.dumps({"variant": label, "strategy": st["strategy"], "packed_kernel": st["packed_kernel"], "scan_ms": round(sorted(ms)[1], 3),
                      "prepare_ms": round(prep * 1e3, 2), "groups": len(got)}))
    sys.stdout.flush()
    q.free()
    for k in env: del os.environ[k]
