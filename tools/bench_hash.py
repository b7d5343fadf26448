#!/usr/bin/env python3
"""Hash group-by (strategy 7) at scale: (A) config 3's 1024 groups forced through the hash table -- every row hits the LDS
staging table; (B) 10^8 rows grouped by (c03, c04, c01): 1.05e9 possible cells, ~9.5e7 live groups -- almost every row
goes to the table in HBM; (C) 1.2e7 rows with a key drawn from 2^40 values.  usage: bench_hash.py [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = sybil_amd.Context(0)


def run(label, t, q, rows):
    query = t.query(**q)
    query.scan()
    ctx.sync()
    ms, wall = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        query.scan()
        ctx.sync()
        wall.append(time.perf_counter() - t0)
        ms.append(query.stats()["scan_ms"])
    t0 = time.perf_counter()
    r = query.finalize()
    fin = time.perf_counter() - t0
    st = query.stats()
    k = sorted(ms)[len(ms) // 2]
    print(json.dumps({"case": label, "rows": rows, "strategy": st["strategy"], "slots": st["n_cells"], "lds_bytes": st["lds_bytes"],
                      "groups": len(query.hash_keys()), "matched": r.matched, "scan_ms": round(k, 3), "rows_per_s": rows / (k * 1e-3),
                      "GBps": st["algorithmic_bytes"] / (k * 1e-3) / 1e9, "compact_and_finalize_s": round(fin, 3)}))
    sys.stdout.flush()
    r.free()
    query.free()


wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
rows = 1_000_000_000
t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
t.compact()
os.environ["SYBL_FORCE_HASH"] = "1"
run("A: config 3 forced through the hash table (LDS staging)", t, dict(wl["query"], order_by=None), rows)
os.environ["SYBL_NO_HASH_LDS"] = "1"
run("A': the same without the LDS staging table (device-scope atomics per row)", t, dict(wl["query"], order_by=None), rows)
del os.environ["SYBL_NO_HASH_LDS"]
del os.environ["SYBL_FORCE_HASH"]
t.free()
rows = 100_000_000
t = ctx.synth_table("b", synth.SEED, rows, 0, rows, synth.synth_cols(["c03", "c04", "c01", "c07"]))
t.compact()
run("B: 1e8 rows by (c03, c04, c01): ~9.5e7 live groups", t, dict(groups=["c03", "c04", "c01"], aggs=["c07"], op="avg", order_by=None), rows)
t.free()
rows = 12_000_000
t = ctx.synth_table("c", synth.SEED, rows, 0, rows, [
    {"name": "k", "kind": synth.UNIFORM, "col_index": 40, "a": -(1 << 39), "b": 1 << 40},
    {"name": "v", "kind": synth.UNIFORM, "col_index": 41, "a": 0, "b": 1_000_000, "info_min": 0, "info_max": 999_999}])
run("C: 1.2e7 rows, key from 2^40 values", t, dict(groups=["k"], aggs=["v"], op="avg", order_by=None), rows)
t.free()
