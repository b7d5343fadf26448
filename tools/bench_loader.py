#!/usr/bin/env python3
"""Times sybl_table_open on a reference-format table: rows/s and bytes/s of gob column files.
usage: bench_loader.py [n_blocks=128]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sybil_amd
from tests import sybil_fixture as F

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
root = tempfile.mkdtemp(prefix="sybl_loader_")
base = []
t0 = time.perf_counter()
for b in range(4):
    n = 65536
    base.append({"status": ("int", rng.integers(0, 16, size=n)),                 # bucket encoded
                 "latency": ("int", rng.integers(0, 1_000_000, size=n)),          # value encoded (> 5000 distinct)
                 "time": ("int", np.sort(1_700_000_000 + b * 3600 + rng.integers(0, 3600, size=n))),
                 "host": ("str", ["host%d" % x for x in rng.integers(0, 200, size=n)])})
F.write_table(root, "t", base, extra_dirs=False)
tdir = os.path.join(root, "t")
for b in range(4, n_blocks):
    shutil.copytree(os.path.join(tdir, "block%09d" % (b % 4 + 1)), os.path.join(tdir, "block%09d" % (b + 1)))
size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tdir) for f in fs)
print("fixture: %d blocks, %.1f MB of gob files, built in %.1f s" % (n_blocks, size / 1e6, time.perf_counter() - t0))
ctx = sybil_amd.Context(0)
for rep in range(4):
    compact = rep >= 2
    t0 = time.perf_counter()
    tb = ctx.open_table(root, "t", compact=compact)
    dt = time.perf_counter() - t0
    print("open_table(%s): %d rows in %.3f s = %.1f M rows/s, %.1f MB/s of column files, %d columns, %.1f MB in HBM" % (
        "compact" if compact else "canonical", tb.rows, dt, tb.rows / dt / 1e6, size / dt / 1e6, 4, tb.hbm_bytes / 1e6))
    print("   ", tb.load_stats())
    q = tb.query(groups=["status"], aggs=["latency"])
    r = q.run()
    assert r.matched == tb.rows
    r.free(); q.free(); tb.free()
shutil.rmtree(root)
