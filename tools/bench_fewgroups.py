import sys, time
sys.path.insert(0, '.')
import sybil_amd
from sybil_amd import synth
ctx = sybil_amd.Context(0)
rows = 1_000_000_000
t = ctx.synth_table("x", synth.SEED, rows, 0, rows, synth.synth_cols(["c01", "c02", "c07"]))
for groups in (["c01"], ["c02"], ["c01", "c02"], []):
    q = t.query(groups=groups, aggs=["c07"], op="hist")
    q.run().free()
    r = q.run(); st = q.stats()
    print(groups, "strategy", st["strategy"], "kernel_ms %.2f" % st["scan_ms"], "GB/s %.0f" % (st["algorithmic_bytes"] / st["scan_ms"] / 1e6))
    r.free(); q.free()
