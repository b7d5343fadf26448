#!/usr/bin/env python3
"""Same-box A/B of k_scan_packed with the plan words reloaded per stage (SYBL_PACKED_RING=4) on configs 2, 3, 5."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from sybil_amd import synth
ctx = sybil_amd.Context(0)
for name in sys.argv[1:] or ("cfg3_filter3_group2_stddev", "cfg2_group1_avg2", "cfg5_time_rollup"):
    wl = synth.WORKLOADS[name]
    rows = wl["rows"]
    t = ctx.synth_table("a", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.compact()
    res = {}
    for rnd in range(3):
        for ring in ("default", "4"):
            if ring != "default": os.environ["SYBL_PACKED_RING"] = ring
            q = t.query(**dict(wl["query"], order_by=None))
            q.scan(); ctx.sync()
            ms = []
            for _ in range(9):
                q.scan(); ctx.sync(); ms.append(q.stats()["scan_ms"])
            res.setdefault(ring, []).append(round(sorted(ms)[4], 4))
            q.free()
            os.environ.pop("SYBL_PACKED_RING", None)
    print(name, json.dumps(res)); sys.stdout.flush()
    t.free()
