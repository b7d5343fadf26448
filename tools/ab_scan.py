#!/usr/bin/env python3
"""Same-box A/B of a BASELINE workload's scan kernel: every (library, environment) variant in a process of its own, in
turns; scans queued back to back (no finalize in between), median hipEvent time.
usage: ab_scan.py <workload substring> <rounds> name=lib.so[,ENV=val...] ...      (lib '-' = the in-tree build)"""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sybil_amd
    from sybil_amd import synth
    name = [n for n in synth.WORKLOADS if sys.argv[2] in n][0]
    wl = synth.WORKLOADS[name]
    ctx = sybil_amd.Context(0)
    t = ctx.synth_table(name, synth.SEED, wl["rows"], 0, wl["rows"], synth.synth_cols(wl["columns"]))
    t.compact()
    q = t.query(**wl["query"])
    ms = []
    for _ in range(24):
        q.scan()
        ctx.sync()
        ms.append(q.stats()["scan_ms"])
    ms = sorted(ms[4:])
    print(json.dumps({"kernel_ms": ms[len(ms) // 2], "min": ms[0], "packed": q.stats()["packed_kernel"]}))
    sys.exit(0)
wl, rounds = sys.argv[1], int(sys.argv[2])
variants = []
for a in sys.argv[3:]:
    name, rest = a.split("=", 1)
    parts = rest.split(",")
    env = dict(p.split("=", 1) for p in parts[1:])
    if parts[0] != "-":
        env["SYBL_LIBRARY"] = os.path.abspath(parts[0])
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl], env=dict(os.environ, **env), capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")]
        res[name].append(json.loads(line[-1])["kernel_ms"] if line else None)
print("# %s: median scan-kernel ms of 20 back-to-back scans per process, processes in turns" % wl)
for n, v in res.items():
    print("%-14s %s" % (n, " ".join("%.4f" % x if x else "fail" for x in v)))
