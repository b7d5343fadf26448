#!/usr/bin/env python3
"""cfg3-shaped query over columns WITH missing rows and a str group column (host-appended table):
GEN role-specialised kernel vs the generic kernel (SYBL_NO_FASTGEN=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sybil_amd

n_blocks, br = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 65536
rng = np.random.default_rng(0)
ctx = sybil_amd.Context(0)
tb = ctx.create_table("nul")
for c in ("f1", "f2", "g1", "v1", "v2"):
    tb.add_column(c, "int", 0, 999_999)
tb.add_column("gs", "str")
strings = ["s%02d" % i for i in range(64)]
f1 = rng.integers(0, 1000, size=br); f2 = rng.integers(0, 1000, size=br); g1 = rng.integers(0, 16, size=br)
v1 = rng.integers(0, 1_000_000, size=br); v2 = rng.integers(0, 1_000_000, size=br); gs = rng.integers(0, 64, size=br).astype(np.int32)
pop = (rng.random(br) > 0.1).astype(np.uint8)
for b in range(n_blocks):
    tb.append_block(br, {"f1": (f1, pop), "f2": f2, "g1": g1, "v1": (v1, pop), "v2": v2, "gs": {"ids": gs, "strings": strings}})
rows = tb.rows
for label, env in (("fast-gen", None), ("generic", "1"), ("compact fast-gen", None), ("compact generic", "1")):
    if label == "compact fast-gen":
        tb.compact()
    os.environ.pop("SYBL_NO_FASTGEN", None)
    if env:
        os.environ["SYBL_NO_FASTGEN"] = env
    q = tb.query(filters=[("f1", "gt", 99), ("f1", "lt", 900), ("f2", "gt", 99), ("f2", "lt", 900)], groups=["g1", "gs"],
                 aggs=["v1", "v2"], op="hist", want_percentiles=False)
    q.run().free()
    ms = []
    for _ in range(5):
        q.run().free(); ms.append(q.stats()["scan_ms"])
    st = q.stats()
    k = sorted(ms)[2]
    print("%-17s strategy %d  kernel %.3f ms  %.0f GB/s  (%d rows, %d B/row)" % (label, st["strategy"], k, st["algorithmic_bytes"] / k / 1e6,
                                                                          rows, st["algorithmic_bytes"] // rows))
    q.free()
