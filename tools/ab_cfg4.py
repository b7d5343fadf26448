#!/usr/bin/env python3
"""Same-box A/B of config 4's scan: every (library, environment) variant in a process of its own, in turns.
usage: ab_cfg4.py rounds name=lib.so[,ENV=val...] ...      (lib '-' = the in-tree build)"""
import json, os, subprocess, sys
rounds = int(sys.argv[1])
variants = []
for a in sys.argv[2:]:
    name, rest = a.split("=", 1)
    parts = rest.split(",")
    env = dict(p.split("=", 1) for p in parts[1:])
    if parts[0] != "-":
        env["SYBL_LIBRARY"] = os.path.abspath(parts[0])
    variants.append((name, env))
here = os.path.dirname(os.path.abspath(__file__))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.join(here, "bench_configs.py"), "0", "5", "cfg4", "compact"], env=dict(os.environ, **env),
                             capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")]
        res[name].append(json.loads(line[-1])["kernel_ms"] if line else None)
for n, v in res.items():
    print("%-12s %s" % (n, " ".join("%.3f" % x if x else "fail" for x in v)))
