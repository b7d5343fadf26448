#!/usr/bin/env python3
"""Where a cold `sybil-gpu-query` spends its wall time: saves the bench's 104.9 M-row, 7-column table, then runs the CLI on it
(config 3's flags, -stats) a few times and prints its phase line.  usage: cold_cli_phases.py [rows]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sybil_amd
from sybil_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100 * 1024 * 1024 // 65536 * 65536
wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
root = tempfile.mkdtemp(prefix="sybl_cold_")
try:
    ctx = sybil_amd.Context(0)
    t = ctx.synth_table("loadbench", synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
    t.save(root)
    t.free()
    ctx.close()
    cli = os.path.join(ROOT, "sybil_amd", "sybil-gpu-query")
    for extra in ([], [], ["-limit", "5"], []):
        time.sleep(0.3)
        t0 = time.perf_counter()
        p = subprocess.run([cli, "-dir", root, "-table", "loadbench", "-stats"] + wl["flags"].split() + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        print("wall %.3f s rc %d | %s" % (dt, p.returncode, p.stderr.decode(errors="replace").strip().replace("\n", " | ")))
    # the same process without a table: start + library load + HIP context only
    t0 = time.perf_counter()
    p = subprocess.run([cli, "-dir", root, "-table", "no-such-table"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print("no table: wall %.3f s rc %d" % (time.perf_counter() - t0, p.returncode))
finally:
    shutil.rmtree(root, ignore_errors=True)
