#!/usr/bin/env python3
"""A/B of the loader's GPU varint walk (SYBL_LOADER_GPU_VARINT, csrc/gobgpu.hip) on the bench's own saved table: the same
104.9 M-row, 7-column table bench.py's `load` record opens, opened K times with the host parser, then K times with the walk on the
GPU (each side's first open also builds the pinned staging arena for its slab size); every open starts 0.3 s after the previous CPU burst.  Prints one line per open and the best of each side.

    python tools/bench_load_varint.py [--rows N] [--opens K]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100 * 1024 * 1024 // 65536 * 65536)
    ap.add_argument("--opens", type=int, default=4)
    ap.add_argument("--workload", default="cfg3_filter3_group2_stddev")
    ap.add_argument("--groups", default="", help="e.g. 1,2,4: also time the default path with SYBL_LOADER_GROUP set to each, interleaved")
    args = ap.parse_args()
    import sybil_amd
    from sybil_amd import synth
    names = synth.WORKLOADS[args.workload]["columns"]
    ctx = sybil_amd.Context(0)
    root = tempfile.mkdtemp(prefix="sybl_varint_ab_")
    try:
        t = ctx.synth_table("loadbench", synth.SEED, args.rows, 0, args.rows, synth.synth_cols(names))
        t.save(root)
        t.free()
        best = {}
        ref = None
        for i in range(2 * args.opens):
            gpu = i >= args.opens   # (not alternating: the staging arena is rebuilt when the slab size changes)
            os.environ["SYBL_LOADER_GPU_VARINT"] = "1" if gpu else "0"
            time.sleep(0.3)
            t0 = time.perf_counter()
            tb = ctx.open_table(root, "loadbench", compact=True)
            dt = time.perf_counter() - t0
            st = tb.load_stats()
            assert tb.rows == args.rows
            # the same resident table either way: a digest of two columns' values
            probe = [int(tb.read_int(n, 0, min(args.rows, 1 << 20)).sum()) for n in names]
            if ref is None:
                ref = probe
            assert probe == ref, (probe, ref)
            print("%-5s open %.4f s = %.3e rows/s  parse_cpu %.3f s  wait %.3f s  apply %.3f s  h2d %.1f MB  walked %d files, %d blocks redone"
                  % ("gpu" if gpu else "host", dt, args.rows / dt, st["parse_cpu_s"], st["wait_s"], st["apply_s"], st["h2d_bytes"] / 1e6,
                     st["gpu_varint_cols"], st["gpu_varint_redone"]), flush=True)
            if gpu not in best or dt < best[gpu][0]:
                best[gpu] = (dt, st)
            tb.free()
        os.environ.pop("SYBL_LOADER_GPU_VARINT", None)
        if args.groups:
            # blocks per launch set (SYBL_LOADER_GROUP), interleaved in one process: the default path only
            res = {}
            for rnd in range(args.opens):
                for g in args.groups.split(","):
                    os.environ["SYBL_LOADER_GROUP"] = g
                    time.sleep(0.3)
                    t0 = time.perf_counter()
                    tb = ctx.open_table(root, "loadbench", compact=True)
                    dt = time.perf_counter() - t0
                    st = tb.load_stats()
                    tb.free()
                    res.setdefault(g, []).append((dt, st["parse_cpu_s"], st["apply_s"], st["wait_s"]))
            os.environ.pop("SYBL_LOADER_GROUP", None)
            for g, r in res.items():
                print("group %-2s opens %s  best %.4f s = %.3e rows/s  (parse_cpu %.3f, apply %.3f, wait %.3f of the best)"
                      % (g, " ".join("%.4f" % x[0] for x in r), min(r)[0], args.rows / min(r)[0], min(r)[1], min(r)[2], min(r)[3]))
        for gpu in (False, True):
            dt, st = best[gpu]
            print("best %-5s %.4f s = %.3e rows/s, parse_cpu %.3f s" % ("gpu" if gpu else "host", dt, args.rows / dt, st["parse_cpu_s"]))
    finally:
        shutil.rmtree(root, ignore_errors=True)
        ctx.close()


if __name__ == "__main__":
    main()
