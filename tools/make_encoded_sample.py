#!/usr/bin/env python3
"""Writes gpurun_out/encoded_results_sample.hex: the `-encode-results` gob stream of a tiny, fully known
query, produced on the GPU.  Committed as tests/golden/encoded_results_sample.hex and decoded by the CPU
suite (tests/test_gob.py) -- a regression pin for sybl_result_encode and for the decoder's interface path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sybil_amd

ctx = sybil_amd.Context(0)
tb = ctx.create_table("pages")
tb.add_column("browser", "str")
tb.add_column("load", "int", 0, 1000)
tb.add_column("time", "int")
names = ["edge", "gecko", "webkit"]
tb.append_block(6, {"browser": {"ids": np.array([0, 0, 1, 2, 1, 0], dtype=np.int32), "strings": names},
                    "load": np.array([100, 300, 50, 1000, 150, 200], dtype=np.int64),
                    "time": np.array([1700000000, 1700000100, 1700003700, 1700003800, 1700007300, 1700007400], dtype=np.int64)})
tb.append_block(4, {"browser": {"ids": np.array([2, 1, 0, 0], dtype=np.int32), "strings": names},
                    "load": np.array([400, 250, 500, 600], dtype=np.int64),
                    "time": np.array([1700007500, 1700010900, 1700011000, 1700011100], dtype=np.int64)})
out = []
for q in (dict(groups=["browser"], aggs=["load"], op="hist", order_by="$COUNT", limit=100),
          dict(groups=["browser"], aggs=["load"], op="avg", time_col="time", time_bucket=3600)):
    query = tb.query(**q)
    r = query.run()
    out.append(r.encode().hex())
    r.free()
    query.free()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/encoded_results_sample.hex", "w").write("\n".join(out) + "\n")
print("wrote gpurun_out/encoded_results_sample.hex", [len(x) // 2 for x in out])
