#!/usr/bin/env python3
"""Same-box A/B of the load ring depth of k_scan_packed (SYBL_PACKED_RING) on configs 2, 3, 5 (compact storage)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
ctx = sybil_amd.Context(0)
for name in ("cfg2_group1_avg2", "cfg3_filter3_group2_stddev", "cfg5_time_rollup"):
    wl = synth.WORKLOADS[name]
    rows = wl["rows"]
    t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    res = {}
    for rnd in range(2):
        for ring in ("default", "1", "2", "3"):
            if ring != "default": os.environ["SYBL_PACKED_RING"] = ring
            q = t.query(**dict(wl["query"], order_by=None))
            q.scan(); ctx.sync()
            ms = []
            for _ in range(7):
                q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
            res.setdefault(ring, []).append(round(sorted(ms)[3], 4))
            q.free()
            os.environ.pop("SYBL_PACKED_RING", None)
    print(name, json.dumps(res)); sys.stdout.flush()
    t.free()
