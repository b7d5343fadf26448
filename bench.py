#!/usr/bin/env python3
"""bench.py -- rows scanned/sec + achieved HBM GB/s of the sybil scan hot path on MI355X.

Workload (BASELINE.json metric / configs[2], "config 3" of BASELINE.md): the synthetic
1 B-row x 32-int-column table (only the 7 referenced columns are resident, as the reference
only opens referenced column files), 3 ANDed int-range filters, group-by 2 columns,
count / sum / avg / stddev of 2 columns.  A step = one full pass of the hot path over the
table: scan kernel + per-workgroup table fold (+ the SUM/MAX all-reduce of the partial group
tables over RCCL when N > 1) + finalize on rank 0.  Inputs are resident in HBM before the
timed region.  The 1 B rows are block-sharded across the N ranks (strong scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable by a copy


def cpu_baseline(workload, total_rows, budget_s=20.0):
    """The CPU oracle (a C restatement of the reference algorithm, kind "port") timed on this
    host's cores on a bounded sample of the same workload.  Reported, never the target."""
    from oracle import oracle as orc
    from sybil_amd import synth
    from tests import parity
    wl = synth.WORKLOADS[workload]
    names, q = wl["columns"], wl["query"]
    info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in names}
    # the reference keeps at most 16 blocks in flight between merges (table_query.go:230-231,
    # CHUNKS_BEFORE_GC=16) and merges them on one goroutine, so more threads do not help it
    cores = min(os.cpu_count() or 1, 16)
    kw = parity.oracle_query_kwargs(names, info, q)

    def run(nrows, columnar=False):
        cols = parity.oracle_synth_cols(orc, names, total_rows, 0, nrows)
        t0 = time.perf_counter()
        r = orc.run_query(cols, n_threads=cores, want_values=False, **kw)
        dt = time.perf_counter() - t0
        col = None
        if columnar and workload == "cfg3_filter3_group2_stddev":
            # variant (ii) of BASELINE.md section 2: a plain columnar multi-threaded scan of the same sample on
            # every host thread (direct-mapped cell table per thread, no Record rows, no maps)
            d = {n: c["data"] for n, c in zip(names, cols)}
            threads = os.cpu_count() or 1
            args = ([d["c04"], d["c05"], d["c06"]], [(100, 899)] * 3, [d["c01"], d["c02"]], [(0, 16), (0, 64)],
                    [d["c07"], d["c08"]], [(0, 999), (0, 999)])
            orc.columnar_scan(*args, n_threads=threads)  # warm the thread pool / page in
            t1 = time.perf_counter()
            m, tab = orc.columnar_scan(*args, n_threads=threads)
            dt2 = time.perf_counter() - t1
            assert m == r["matched"] and int(tab[0].sum()) == r["matched"]
            col = {"value": nrows / dt2, "unit": "rows/s", "cores": threads, "kind": "port-columnar",
                   "sample": "%d rows, oracle/sybil_oracle.c:orc_columnar_scan with %d threads, %.2f s" % (nrows, threads, dt2)}
        return dt, r["matched"], col

    probe = 2_000_000
    dt, _, _ = run(probe)
    rate = probe / dt
    sample = int(min(max(rate * budget_s, probe), 96_000_000, total_rows))
    sample = max(65536, sample // 65536 * 65536)
    dt, matched, col = run(sample, columnar=True)
    out = {"value": sample / dt, "unit": "rows/s", "cores": cores, "kind": "port",
           "sample": "%d rows (first blocks) of %s, oracle/sybil_oracle.c with %d threads, %.1f s" % (
               sample, workload, cores, dt)}
    return out, col


def measured_traffic(stats, names):
    """(HBM bytes per launch, where the figure comes from): the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, in their own runs) give bytes per scanned row, times this
    launch's rows.  (None, None) when no profile of this kernel shape has been recorded."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    key = "%d_cols_strategy_%d%s" % (len(names), stats["strategy"], "_packed" if stats.get("packed_kernel") else "")
    if key not in rec:
        return None, None
    # (a constant from the committed counter passes x this launch's rows -- not something this run measured)
    return rec[key]["hbm_bytes_per_row"] * stats["rows_scanned"], "profiles/traffic.json[%s]: %s" % (key, rec[key]["source"])


def oracle_check(workload, q, total_rows, head):
    """The headline result against the CPU oracle at the FULL size, bit for bit: orc_synth_scan regenerates the table
    on every host thread and runs the reference's row loop in its direct-mapped form (oracle/sybil_oracle.h).  Checker
    only -- outside every timed region."""
    from oracle import oracle as orc
    from sybil_amd import synth
    t0 = time.perf_counter()
    o = orc.synth_scan(synth.COLUMNS, synth.SEED, total_rows, 0, total_rows, filters=q.get("filters", ()), groups=q.get("groups", ()),
                       aggs=q.get("aggs", ()), time_col=q.get("time_col"), time_bucket=q.get("time_bucket", 0), want_buckets=False,
                       n_threads=os.cpu_count() or 1)
    dt = time.perf_counter() - t0
    assert head["matched"] == o["matched"], ("matched rows differ from the oracle", head["matched"], o["matched"])
    cards = o["cells"][1:]
    n_aggs = len(q.get("aggs", ()))
    for row in head["digest"]:
        tb, key, count, sums = row[0], row[1], row[2], row[3:]
        cell = tb // q["time_bucket"] - o["tb_min"] if q.get("time_col") else 0
        for g, card in enumerate(cards):
            cell = cell * card + (int.from_bytes(key[8 * g:8 * g + 8], "little", signed=True) - o["gmin"][g])
        assert count == o["count"][cell] and all(sums[a] == o["sum"][a][cell] for a in range(n_aggs)), ("cell differs from the oracle", row)
    assert len(head["digest"]) == int((o["count"] != 0).sum())
    return {"rows": total_rows, "matched": int(o["matched"]), "groups": len(head["digest"]),
            "checked": "matched rows, every group's Count and exact sum(v) per aggregation == orc_synth_scan (bit-exact)",
            "oracle_seconds": round(dt, 2), "threads": os.cpu_count() or 1}


def load_path(ctx, names, rows=100 * 1024 * 1024 // 65536 * 65536, scan_workload=None):
    """Disk -> HBM: the TableBlock load half of the hot path (table_block_io.go:225-310, column_store_io.go:493-780).  A
    synthetic table is written in the reference's on-disk format (sybl_table_save) and read back with the native loader
    into compact storage (sybl_table_open_flags); outside every timed scan region.  scan_workload: that workload's query
    is then run on the table exactly as the loader left it -- info.db bounds, validity as the files say -- which is the
    scan a drop-in host gets (the headline scans a generator-built table with declared bounds)."""
    import shutil
    import tempfile
    from sybil_amd import synth
    root = tempfile.mkdtemp(prefix="sybl_bench_load_")
    try:
        t = ctx.synth_table("loadbench", synth.SEED, rows, 0, rows, synth.synth_cols(names))
        t0 = time.perf_counter()
        t.save(root)
        save_s = time.perf_counter() - t0
        t.free()
        tdir = os.path.join(root, "loadbench")
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tdir) for f in fs)
        best, scanned = None, None
        opens = []
        # Three opens from idle, then three back to back.  An open is a burst of ~1.8 s of CPU in ~0.07 s (32 parser
        # threads): the GPU boxes' containers have a CFS quota of 16 CPUs per 100 ms period, so an open that starts
        # behind another burst (the save above, the previous open) is throttled in its middle and takes 0.10-0.14 s,
        # one that starts from idle is not.  Since round 6 the int columns' varints are walked on the GPU and an open burns
        # 0.4-0.5 s of CPU: the quota no longer bites, but an open that starts on an IDLE GPU pays ~0.02-0.04 s for the first
        # commands of its streams (SYBL_LOADER_TRACE: "copy 0.04 s" against 0.004).  "rows_per_s" is the best of opens 2-6
        # (the first also builds the staging arena), "open_seconds" lists all six, "back_to_back" is the mean of the last three.
        for i in range(6):
            if i < 3:
                time.sleep(0.3)
            t0 = time.perf_counter()
            tb = ctx.open_table(root, "loadbench", compact=True)
            dt = time.perf_counter() - t0
            assert tb.rows == rows
            st = tb.load_stats()
            hbm = tb.hbm_bytes
            opens.append(round(dt, 4))
            if scan_workload and scanned is None:
                q = dict(synth.WORKLOADS[scan_workload]["query"])
                qy = tb.query(**q)
                ms = []
                for _ in range(12):
                    qy.scan()
                    ctx.sync()
                    ms.append(qy.stats()["scan_ms"])
                res = qy.finalize()
                stq = qy.stats()
                k = sorted(ms[2:])[len(ms[2:]) // 2]
                kernel = ("k_scan_packed" if stq["packed_kernel"] else "k_scan_fast") if stq["strategy"] in (2, 4, 6) else "k_scan"
                scanned = {"workload": scan_workload, "rows": rows, "kernel": kernel, "strategy": stq["strategy"], "kernel_ms": round(k, 4),
                           "rows_per_s": rows / (k * 1e-3), "matched": res.matched, "groups": len(res.rows(0, want_values=False)),
                           "stored_bytes_per_row": stq["algorithmic_bytes"] / max(stq["rows_scanned"], 1),
                           "roofline": {"bound": "hbm", "achieved": stq["algorithmic_bytes"] / (k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": stq["algorithmic_bytes"] / (k * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "what": "the workload's query on the table as sybl_table_open left it (info.db bounds), median scan of 10 "
                                   "back-to-back scans; 104.9 M rows are 27 us of scan per workgroup: launch and ramp weigh in"}
                res.free()
                qy.free()
            tb.free()
            if i > 0 and (best is None or dt < best[0]):
                best = (dt, st, hbm)
        dt, st, hbm = best
        b2b = sum(opens[3:]) / len(opens[3:])
        # The same open with every column file through the host parser (SYBL_LOADER_GPU_VARINT=0: gob.cpp's AVX-512 varint
        # windows on the worker threads, the only path before round 6; the default now walks the int columns' Values / Bins
        # slices on the GPU, csrc/gobgpu.hip).  Its first open also rebuilds the staging arena for the other slab size and is
        # not counted; best of three from idle.
        hp = None
        os.environ["SYBL_LOADER_GPU_VARINT"] = "0"
        try:
            runs = []
            for i in range(4):
                time.sleep(0.3)
                t0 = time.perf_counter()
                tb = ctx.open_table(root, "loadbench", compact=True)
                dtv = time.perf_counter() - t0
                assert tb.rows == rows
                stv = tb.load_stats()
                tb.free()
                if i > 0:
                    runs.append((dtv, stv))
            dtv, stv = min(runs, key=lambda r: r[0])
            hp = {"rows_per_s": rows / dtv, "seconds": round(dtv, 4), "open_seconds": [round(r[0], 4) for r in runs], "stage_breakdown": stv,
                  "what": "SYBL_LOADER_GPU_VARINT=0: every column file parsed by the worker threads (gob.cpp); best of 3 opens from idle"}
        finally:
            os.environ.pop("SYBL_LOADER_GPU_VARINT", None)
        # (back to the default slab size before anything else is timed)
        tb = ctx.open_table(root, "loadbench", compact=True)
        tb.free()
        cold = None
        if scan_workload:
            # SURVEY 8d's end-to-end figure: `sybil query` itself, cold -- process start -> library load -> HIP context -> table
            # open (disk -> HBM, page cache warm) -> agree / compact -> prepare -> scan -> finalize -> printed result -> exit,
            # through the C ABI's own CLI (tools/sybil_gpu_query.cpp) with the reference's flags for the workload.  Wall time of
            # the child process as this process sees it; the first run and the best of three.
            import subprocess
            cli = os.path.join(ROOT, "sybil_amd", "sybil-gpu-query")
            argv = [cli, "-dir", root, "-table", "loadbench"] + synth.WORKLOADS[scan_workload]["flags"].split()
            runs, text = [], b""
            for _ in range(3):
                time.sleep(0.3)
                t0 = time.perf_counter()
                p = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                runs.append(round(time.perf_counter() - t0, 4))
                if p.returncode != 0:
                    raise RuntimeError("sybil-gpu-query failed: " + p.stderr.decode(errors="replace")[-400:])
                text = p.stdout
            total = text.decode(errors="replace").splitlines()[0].split()
            assert total[0] == "TOTAL" and (scanned is None or int(total[1]) == scanned["matched"]), total
            cold = {"seconds_first": runs[0], "seconds_best": min(runs), "runs": runs, "rows": rows, "rows_per_s": rows / min(runs),
                    "printed_bytes": len(text), "matched": int(total[1]), "argv": " ".join(["sybil-gpu-query"] + argv[1:2] + ["<dir>"] + argv[3:]),
                    "resident_step_ms": None if scanned is None else scanned["kernel_ms"],
                    "what": "wall time of the CLI process, start to exit, on the saved table (files in the page cache): what a cold "
                            "`sybil query` replacement costs end to end; `seconds` above is the open alone, `loaded_table_scan` the resident scan"}
        return {"cold_cli": cold, "loaded_table_scan": scanned, "host_parser": hp, "rows_per_s": rows / dt, "rows": rows, "columns": names, "seconds": round(dt, 3), "bytes_on_disk": size,
                "disk_bytes_per_row": size / rows, "hbm_bytes": hbm, "stage_breakdown": st, "save_seconds": round(save_s, 2),
                "open_seconds": opens, "back_to_back": {"seconds": round(b2b, 4), "rows_per_s": rows / b2b},
                "what": "sybl_table_save -> sybl_table_open_flags(SYBL_OPEN_COMPACT), page cache warm, the int columns' varints walked on the GPU "
                        "(the default; host_parser: the same open with SYBL_LOADER_GPU_VARINT=0): best of opens 2-6 -- three that each start "
                        "0.3 s after the previous one (an idle GPU), three back to back; back_to_back: mean of the last three"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="total rows (default: the workload's BASELINE size)")
    ap.add_argument("--workload", default="cfg3_filter3_group2_stddev")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish every step (finalize on the host) before the next scan is launched")
    ap.add_argument("--no-canonical", action="store_true", help="skip the secondary canonical-storage measurement")
    ap.add_argument("--storage", choices=["canonical", "compact"], default="compact",
                    help="canonical: int64 per value (the reference's in-memory IntField); compact: sybl_table_compact "
                         "(1/2/4-byte offsets from the column minimum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (process group, bound partial tables, all-reduce) even with one rank")
    ap.add_argument("--collective", choices=["torch", "rccl"], default="rccl",
                    help="rccl: the library's own communicator (sybl_comm_init / sybl_query_allreduce -- what a Go host "
                         "calls, the product path); torch: torch.distributed all-reduces of the bound partial tables")
    ap.add_argument("--no-load", action="store_true", help="skip the disk -> HBM load measurement (N=1 only)")
    ap.add_argument("--summarise-every-row", action="store_true",
                    help="config 4: derive percentiles / stddev for all 65 536 rows every step (printed_only = 0), as rounds 1-4 measured it")
    ap.add_argument("--no-configs", action="store_true",
                    help="only the headline workload: skip the cfg2 / cfg4 / cfg5 records of the `configs` key")
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="skip the full-size bit-exact check of the result against the CPU oracle (N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sybil_amd
    from sybil_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SYBL_BENCH_ONE_DEVICE=1 (tests/test_gpu_bench_multirank.py): a FUNCTIONAL check of this file's N > 1 flow on a box with
    # one GPU -- every rank on device 0, the library's collectives through the test-only shared-memory RCCL stand-in the
    # launcher preloads (tests/rccl_standin/), torch's process group on gloo.  Its timings mean nothing and are marked so.
    one_device = os.environ.get("SYBL_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
        if args.collective != "rccl":
            raise SystemExit("SYBL_BENCH_ONE_DEVICE needs --collective rccl (the in-library path)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    ctx = sybil_amd.Context(local_rank)
    dev = ctx.device_info()
    if multi and args.collective == "rccl":
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], world, rank)
    side = None
    if multi and args.collective == "torch":
        # One explicit (non-default) stream carries scan -> all-reduce -> finalize: the engine is
        # pointed at it and torch.distributed orders its collective against torch's CURRENT stream,
        # which is this one inside the `with` below.  (The default stream's handle is NULL, which
        # sybl_ctx_set_stream would read as "use the engine's own stream" -- and then nothing would
        # order the all-reduce after the scan.)
        side = torch.cuda.Stream(device=device)
        assert side.cuda_stream != 0
        ctx.set_stream(side.cuda_stream)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    STRATEGY = {0: "lds-generic", 1: "global-atomics", 2: "lds-fast", 3: "lds-window-generic", 4: "lds-window-fast",
                5: "partitioned-hist", 6: "lds-hist", 7: "hash"}
    WHAT = {"cfg3_filter3_group2_stddev": "3 ANDed int-range filters, group-by 2 cols, count/sum/avg/stddev of 2 cols",
            "cfg4_hist_highcard": "histogram (p25/p50/p99, every bucket) of 1 col grouped by a 65536-value col, -limit 100",
            "cfg5_time_rollup": "hourly time buckets x group-by 1 str-like col, sum+count",
            "cfg2_group1_avg2": "group-by 1 low-card col, sum+avg of 2 cols",
            "cfg1_count_range": "count(*) with one int-range predicate"}

    def measure(workload, total_rows, steps, warmup, storage, canonical_too):
        """One BASELINE workload: builds this rank's shard of the synthetic table, times `steps` steps (after `warmup`)
        in `storage`, optionally the same table in canonical int64 storage first.  Returns the record rank 0 prints."""
        wl = synth.WORKLOADS[workload]
        names, q = wl["columns"], dict(wl["query"])
        if workload == "cfg4_hist_highcard":
            q["limit"] = 100  # FLAGS.LIMIT defaults to 100 (cmd_query.go): only the printed rows carry their bucket arrays
            # `sybil query` is a printer: GetPercentiles / GetStdDev run at print time for the rows it prints (printer.go:60-76,
            # 291-308).  The step below is that query (sybl_query_desc.printed_only); the same step with every one of the
            # 65 536 rows summarised (what -encode-results needs) is measured beside it: `every_row_summarised`.
            q["printed_only"] = not args.summarise_every_row
        total_rows = total_rows or wl["rows"]
        bytes_per_row = 8 * len(names)
        # fit check (single GPU must hold its shard)
        row0, nrows = synth.shard(total_rows, rank, world)
        if nrows * bytes_per_row > dev["hbm_bytes"] * 0.9:
            raise SystemExit("shard of %d rows x %d B does not fit in %d B of HBM" % (nrows, bytes_per_row, dev["hbm_bytes"]))
        table = ctx.synth_table("bench", synth.SEED, total_rows, row0, nrows, synth.synth_cols(names))
        # identical direct-mapped layout on every rank: declare the generator's value bounds
        for n in names:
            kind, _, a, b, _, _ = synth.COLUMNS[n]
            hi = a + 4 * (b - 1) if kind == synth.BELL else a + b - 1
            table.set_bounds(n, a, hi)

        def run_phase(steps, warmup):
            """Prepares the query against the table as currently laid out, does `warmup` untimed steps and
            times exactly `steps` steps (barrier + synchronize on both sides, max over ranks).

            Steps are software-pipelined one deep (two prepared queries, alternating): the host-side
            finalize of step i -- waiting for its snapshot, deriving avg / stddev, building and sorting the
            result rows -- runs while the GPU already scans step i+1.  Every step does all of its work and
            every result is complete inside the timed region; --no-pipeline serialises them."""
            nq = 1 if args.no_pipeline else 2
            queries = [table.query(**q) for _ in range(nq)]
            if multi and args.collective == "torch":
                with torch.cuda.stream(side):
                    for qy in queries:
                        qy.bind_torch(device)
            scan_ms = []
            host_ms = {"launch": [], "finish": []}  # host time inside the two halves of a step

            everyone = [False]  # snapshot / finalize are collective calls (bucket arrays merged by reduce-scatter)

            def launch(i):
                qy = queries[i % nq]
                qy.scan()
                if multi:
                    if args.collective == "torch":
                        with torch.cuda.stream(side):
                            qy.allreduce_torch()
                    else:
                        qy.allreduce()
                        everyone[0] = qy.collective_finalize()
                if rank == 0 or everyone[0]:
                    qy.snapshot()  # D2H copy of the reduced table, queued behind the all-reduce

            def finish(i):
                qy = queries[i % nq]
                res = None
                if rank == 0 or everyone[0]:
                    res = qy.finalize()
                    if rank == 0:
                        scan_ms.append(qy.stats()["scan_ms"])
                    else:
                        res.free()
                        res = None
                elif nq == 1:
                    ctx.sync()
                return res

            def run_steps(n, keep):
                res = None
                for i in range(n):
                    h0 = time.perf_counter()
                    launch(i)
                    h1 = time.perf_counter()
                    host_ms["launch"].append((h1 - h0) * 1e3)
                    j = i - (nq - 1)  # the step whose result is due
                    if j >= 0:
                        r = finish(j)
                        host_ms["finish"].append((time.perf_counter() - h1) * 1e3)
                        if r is not None:
                            seen_matched.add(r.matched)
                            if res is not None:
                                res.free()
                            res = r
                for j in range(max(n - (nq - 1), 0), n):
                    r = finish(j)
                    if r is not None:
                        seen_matched.add(r.matched)
                        if res is not None:
                            res.free()
                        res = r
                if not keep and res is not None:
                    res.free()
                    res = None
                return res

            seen_matched = set()
            run_steps(warmup, False)
            fence()
            del scan_ms[:]
            del host_ms["launch"][:], host_ms["finish"][:]
            t0 = time.perf_counter()
            res = run_steps(steps, True)
            fence()
            dt = time.perf_counter() - t0
            if multi:
                tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_device else device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            stats = queries[0].stats()
            out = {"dt": dt, "stats": stats, "kernel_ms": (sum(scan_ms) / len(scan_ms)) if scan_ms else stats["scan_ms"],
                   "host_ms": dict({k: round(sum(v) / len(v), 3) if v else None for k, v in host_ms.items()},
                                   finish_median=round(sorted(host_ms["finish"])[len(host_ms["finish"]) // 2], 3) if host_ms["finish"] else None,
                                   finish_max=round(max(host_ms["finish"]), 3) if host_ms["finish"] else None,
                                   **({"finish_all": [round(x, 2) for x in host_ms["finish"]]} if os.environ.get("SYBL_BENCH_TRACE") else {}))}
            if rank == 0:
                # every step scans the same table: the merged result must not change from step to step (it
                # would if the all-reduce ever ran ahead of a rank's scan) and group counts must add up
                assert len(seen_matched) == 1, "matched count varies across steps: %r" % sorted(seen_matched)
                assert len(scan_ms) == steps
                # (a big result's rows are built when first asked for -- sybl_query_finalize finds the live cells, sorts and keeps
                # the snapshot --: what that costs is reported beside the step, from the last step's result)
                t_rows = time.perf_counter()
                res.materialize()
                out["rows_first_access_ms"] = round((time.perf_counter() - t_rows) * 1e3, 3)
                rows_out = res.time_results if q.get("time_col") else res.rows(0, want_values=False)
                assert sum(g["count"] for g in rows_out) == res.matched
                out["matched"] = res.matched
                out["groups"] = len(rows_out)
                if workload == "cfg4_hist_highcard":
                    out["printed"] = res.render("text")  # (what the CLI's text printer shows of this result)
                out["digest"] = sorted((g["time_bucket"], g["key"], g["count"]) + tuple(h["sum"] for h in g["hists"]) for g in rows_out)
                res.free()
            for qy in queries:
                qy.free()
            return out

        def roofline(ph):
            stats = ph["stats"]
            alg = stats["algorithmic_bytes"]  # this rank's rows x stored bytes per row: per launch, per GPU
            achieved = alg / (ph["kernel_ms"] * 1e-3) / 1e9
            shape = "<3,2,2,moments>" if workload == "cfg3_filter3_group2_stddev" else ""
            if stats["strategy"] in (2, 4, 6):
                kernel = ("k_scan_packed" if stats["packed_kernel"] else "k_scan_fast") + shape
            elif stats["strategy"] == 5:
                pk = "_packed" if stats["packed_kernel"] else ""
                # (one key column, no filter, one bin per partition -- config 4 -- counts with k_count_key: scan_packed.h)
                kernel = "%s + k_emit%s + k_part_hist + k_part_fix" % ("k_count_key" if pk and len(names) == 2 else "k_count" + pk, pk)
            else:
                kernel = "k_scan<%d>" % len(names)
            traffic, traffic_source = measured_traffic(stats, names)
            return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel, "kernel_ms": ph["kernel_ms"],
                    "algorithmic_bytes_per_launch": alg, "stored_bytes_per_row": alg / max(stats["rows_scanned"], 1),
                    "int64_canonical_bytes_per_launch": stats["canonical_bytes"],
                    "strategy": STRATEGY[stats["strategy"]], "lds_bytes": stats["lds_bytes"], "workgroups": stats["n_workgroups"]}

        # Secondary, untimed-for-the-headline measurement: the same table in canonical int64 storage
        # (the reference's in-memory IntField) before it is compacted.
        canon = None
        if storage == "compact" and canonical_too:
            canon = run_phase(min(steps, 10), min(warmup, 2))
        if storage == "compact":
            table.compact()
        head = run_phase(steps, warmup)
        extra = {}
        if workload == "cfg4_hist_highcard" and q.get("printed_only"):
            q["printed_only"] = False
            alt = run_phase(min(steps, 10), warmup)  # (the same warm-up: its first steps allocate the pinned 52 MB percentile buffers)
            q["printed_only"] = True
            if rank == 0:
                assert alt["digest"] == head["digest"], "a printer's query and the fully summarised one disagree"
                # (rounds 1-4 timed the every-row step as THE config 4 figure; since round 5 the record's value / ms_per_step
                # are the printer's query -- both are first-class here, each with its own rows/s, and the record says which is which)
                extra["metric_definition"] = ("value / ms_per_step: the PRINTER's query (sybl_query_desc.printed_only = 1, what `sybil query` runs; "
                                              "since round 5); every_row_summarised.value: every group's percentiles / stddev derived per step "
                                              "(printed_only = 0, what -encode-results needs; the definition rounds 1-4 reported)")
                extra["every_row_summarised"] = {"value": total_rows * min(steps, 10) / alt["dt"], "unit": "rows/s",
                                                 "ms_per_step": alt["dt"] / min(steps, 10) * 1e3, "kernel_ms": alt["kernel_ms"], "steps": min(steps, 10),
                                                 "host_ms_per_step": alt["host_ms"], "rows_first_access_ms": alt.get("rows_first_access_ms"),
                                                 "what": "the same step with percentiles / stddev derived for all 65 536 groups (k_hist_summary over the "
                                                         "525 MB table + 52 MB of percentiles to the host per step): what -encode-results needs"}
        if workload == "cfg4_hist_highcard" and q.get("printed_only") and world == 1:
            # -limit pushed INTO the scan (sybl_query_desc.printed_only = 2, csrc/pushdown.hip): the printer's query when the
            # rows beyond the limit need nothing but their Count -- group counts from the key column, the printed groups chosen
            # on the device, one pass over key + value for Cumulative and those groups.  A SEPARATE, labelled measurement: never
            # config 4's roofline figure (that stays the full path above, every group's buckets built).
            q["printed_only"] = 2
            pd = run_phase(min(steps, 10), warmup)
            q["printed_only"] = True
            if rank == 0:
                assert pd["stats"]["strategy"] == 8, pd["stats"]["strategy"]
                assert pd["matched"] == head["matched"] and pd["printed"] == head["printed"], "the pushed-down printer prints something else"
                assert [r[:3] for r in pd["digest"]] == [r[:3] for r in head["digest"]], "group counts differ"
                alg_pd = pd["stats"]["rows_scanned"] * (2 * table.column_storage(names[0])[0] + table.column_storage(names[1])[0])
                extra["printer_pushdown"] = {"value": total_rows * min(steps, 10) / pd["dt"], "unit": "rows/s", "ms_per_step": pd["dt"] / min(steps, 10) * 1e3,
                                             "kernel_ms": pd["kernel_ms"], "steps": min(steps, 10), "host_ms_per_step": pd["host_ms"],
                                             "bytes_read_per_step": alg_pd,
                                             "achieved_GBps": alg_pd / (pd["kernel_ms"] * 1e-3) / 1e9, "frac_of_hbm_peak": alg_pd / (pd["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "checked": "text printer's output byte-equal to the full path's; matched rows and every group's Count equal",
                                             "what": "printed_only = 2: -limit %d pushed into the scan (k_pd_count over the key column, k_pd_select, "
                                                     "k_pd_scan over key + value): no records, no bucket table; the rows beyond the limit carry their "
                                                     "Count only.  Not the roofline figure of config 4" % q["limit"]}
        if head["stats"].get("strategy") == 5:
            extra["count_pass"] = {"reused": bool(head["stats"].get("count_pass_reused")),
                                   "what": "the counting pass over the key column (k_count*: per-(workgroup, bin) record counts) runs in a prepared "
                                           "query's FIRST scan; rescans of the unchanged table reuse its regions, as they reuse block statistics "
                                           "(SYBL_NO_COUNT_CACHE=1: every scan) -- the timed steps are rescans"}
        if workload == "cfg4_hist_highcard":
            # scans back to back (no finalize between them: the GPU never idles, clocks stay up): the kernels' own figure
            qy = table.query(**q)
            ms = []
            for _ in range(12):
                qy.scan()
                ctx.sync()
                ms.append(qy.stats()["scan_ms"])
            qy.free()
            extra["back_to_back_scan_ms"] = {"median": round(sorted(ms[2:])[len(ms[2:]) // 2], 4), "min": round(min(ms[2:]), 4), "scans": len(ms) - 2,
                                             "what": "hipEvent time of the scan kernels over 10 scans queued back to back, no finalize in between; "
                                                     "`roofline.kernel_ms` is the mean over the pipelined steps of the timed region"}
        out = None
        if rank == 0:
            if canon is not None:
                assert canon["digest"] == head["digest"], "compact and canonical storage disagree"
            dt = head["dt"]
            out = {
                "metric": "rows scanned/sec (%s-row x %d-int-col synthetic table, %s)" % (
                    "1B" if total_rows == 1_000_000_000 else str(total_rows), wl["table_cols"], WHAT[workload]),
                "value": total_rows * steps / dt, "unit": "rows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "int64", "data": "synthetic",
                "config": {"workload": workload, "reference_flags": wl["flags"], "rows": total_rows,
                           "table_columns": wl["table_cols"], "resident_columns": names,
                           "storage": storage,
                           "pipelined": None if args.no_pipeline else "finalize(i) on the host overlaps scan(i+1) on the GPU "
                                                                      "(two prepared queries); all work of the K steps is "
                                                                      "inside the timed region",
                           "stored_widths": {n: table.column_storage(n)[0] for n in names},
                           "sharding": "contiguous 65536-row blocks per rank", "collective": args.collective if multi else None,
                           "device": dev["name"], "matched_rows": head["matched"], "groups": head["groups"],
                           "host_ms_per_step": head["host_ms"],
                           "rows_first_access_ms": head.get("rows_first_access_ms")},
                "roofline": roofline(head),
            }
            out.update(extra)
            if one_device:
                out["one_device_standin"] = "every rank on GPU 0 through the test-only RCCL stand-in: a functional check, not a measurement"
            if canon is not None:
                out["canonical_storage"] = {"value": total_rows * min(steps, 10) / canon["dt"], "unit": "rows/s",
                                            "steps": min(steps, 10), "roofline": roofline(canon)}
            if (world == 1 or one_device) and not args.no_oracle_check:
                out["oracle_check"] = oracle_check(workload, q, total_rows, head)
                # (the check burns every host thread for seconds: the container's CPU quota throttles whatever follows inside the
                # same accounting periods -- the next config's host half once took 25 ms for one finalize.  Let the periods pass.)
                time.sleep(0.5)
        table.free()
        return out

    out = measure(args.workload, args.rows, args.steps, args.warmup, args.storage, not args.no_canonical)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], columnar = cpu_baseline(args.workload, args.rows or synth.WORKLOADS[args.workload]["rows"])
            if columnar is not None:
                out["cpu_baseline_columnar"] = columnar
        if "oracle_check" in out:  # (key order of the round-2 line: cpu_baseline before oracle_check)
            out["oracle_check"] = out.pop("oracle_check")
    # The other BASELINE.json configurations that fit one GPU, driver-measured in the same run: same table generator,
    # same step (scan + fold (+ all-reduce) + finalize, pipelined one deep), compact storage, at most 20 steps.
    # (single-GPU runs only: the N > 1 scaling runs measure the headline, whose collective path is the one exercised on
    # one GPU by --force-dist; a config that fails is reported, it does not take the headline line with it)
    if not args.no_configs and args.workload == "cfg3_filter3_group2_stddev" and not args.rows and not multi:
        recs = []
        for name in ("cfg1_count_range", "cfg2_group1_avg2", "cfg4_hist_highcard", "cfg5_time_rollup"):
            try:
                # (eight warm-up steps whatever --warmup says: the host half of a step -- 360 500 result rows for config 5 --
                # runs on worker threads that take a few finalizes to reach their steady state: 3.0 ms per step with two
                # warm-up steps, 2.4 with four or more; the record names the steps and warm-up it used)
                rec = measure(name, 0, min(args.steps, 20), 8, "compact", False)
                recs.append({k: rec[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")
                             + tuple(k for k in ("oracle_check", "metric_definition", "every_row_summarised", "printer_pushdown", "back_to_back_scan_ms", "count_pass") if k in rec)})
                recs[-1]["kernel_ms"] = rec["roofline"]["kernel_ms"]
                if name == "cfg1_count_range" and not args.no_cpu_baseline:
                    # BASELINE.json configs[0] is the reference's own CPU-runnable case: the oracle on the whole table beside it
                    recs[-1]["cpu_baseline"], _ = cpu_baseline(name, synth.WORKLOADS[name]["rows"], budget_s=4.0)
            except Exception as e:  # noqa: BLE001 -- reported in the line
                recs.append({"config": {"workload": name}, "error": "%s: %s" % (type(e).__name__, e)})
        out["configs"] = recs
    if rank == 0:
        if world == 1 and not args.no_load:
            # the bench's own table (the 7 referenced columns of config 3: five bucket-encoded, two value-encoded), and
            # the mix of round 2's record: time (delta-friendly), 16 values, 1e6 values (value encoded), 500 ids
            out["load"] = load_path(ctx, synth.WORKLOADS["cfg3_filter3_group2_stddev"]["columns"], scan_workload="cfg3_filter3_group2_stddev")
            out["loaded_table_scan"] = out["load"].pop("loaded_table_scan")
            out["cold_cli"] = out["load"].pop("cold_cli")
            out["load_mixed_4col"] = load_path(ctx, ["c00", "c01", "c07", "c09"])
            out["load_mixed_4col"].pop("loaded_table_scan")
            out["load_mixed_4col"].pop("cold_cli")
        print(json.dumps(out))
        sys.stdout.flush()
    if multi and args.collective == "rccl":
        ctx.comm_free()
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
