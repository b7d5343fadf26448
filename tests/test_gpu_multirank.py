"""GPU, N > 1: the in-library collective path (sybl_comm_* / sybl_query_allreduce, csrc/rccl.cpp) with one process per
GPU -- what a Go host would run.  Needs at least two GPUs on the box: skipped on the 1-GPU boxes this project is
developed and judged on (the N = 1 forms of the same calls run in test_gpu_parity.py / test_gpu_hash.py; the host
protocol with world_size 2 runs on CPU in test_dist_gloo.py), so this file has NEVER RUN -- it is here for the day the
suite meets a node.  Every rank scans its contiguous block shard; rank 0's merged result must equal the oracle's on the
whole table: direct-mapped cells (one SUM all-reduce), tracked extrema (+ the MAX all-reduce), the reduce-scatter of a
big bucket table with collective finalize (int32 slices), outlier values gathered from every rank, and the hash
group-by's key-union protocol."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on this box")]

TOTAL = 6_000_000


def _queries():
    from sybil_amd import synth
    wl3 = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    return [
        ("direct", wl3["columns"], dict(wl3["query"]), {}),
        ("extrema", ["c01", "c07"], dict(groups=["c01"], aggs=["c07"], op="avg"), {}),
        ("scatter", ["c03", "c07"], dict(groups=["c03"], aggs=["c07"], op="hist", limit=50), {}),
        ("outliers", ["c01", "c07"], dict(groups=["c01"], aggs=["c07"], op="hist", hist_bucket=100), {}),
        ("hash", wl3["columns"], dict(wl3["query"]), {"SYBL_FORCE_HASH": "1"}),
    ]


def _worker(rank, world, uid_path, out_path):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import pickle
    import time
    import sybil_amd
    from sybil_amd import synth
    ctx = sybil_amd.Context(rank)
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(ctx.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.05)
    uid = open(uid_path, "rb").read()
    ctx.comm_init(uid, world, rank)
    row0, nrows = synth.shard(TOTAL, rank, world)
    results = {}
    for name, cols, q, env in _queries():
        for k, v in env.items():
            os.environ[k] = v
        t = ctx.synth_table("mr", synth.SEED, TOTAL, row0, nrows, synth.synth_cols(cols))
        for n in cols:  # identical direct-mapped layout on every rank: the generator's bounds
            kind, _, a, b, _, _ = synth.COLUMNS[n]
            t.set_bounds(n, a, a + 4 * (b - 1) if kind == synth.BELL else a + b - 1)
        if name == "outliers":
            t.set_bounds("c07", 0, 999_999)
        qy = t.query(**q)
        qy.scan()
        qy.allreduce()
        everyone = qy.collective_finalize()
        if rank == 0 or everyone:
            res = qy.finalize()
            if rank == 0:
                rows = res.rows(0)
                results[name] = dict(matched=res.matched, strategy=qy.stats()["strategy"],
                                     rows=[(r["key"], r["count"], [(h["count"], h["sum"], h.get("values"), h.get("outlier_values")) for h in r["hists"]])
                                           for r in rows])
            res.free()
        qy.free()
        t.free()
        for k in env:
            del os.environ[k]
    if rank == 0:
        with open(out_path, "wb") as f:
            pickle.dump(results, f)
    ctx.comm_free()
    ctx.close()


def test_inlibrary_collectives_across_ranks(tmp_path):
    import pickle
    from oracle import oracle as orc
    from sybil_amd import synth
    from tests import parity
    world = min(torch.cuda.device_count(), 8)
    uid_path, out_path = str(tmp_path / "uid"), str(tmp_path / "out.pkl")
    mp.start_processes(_worker, args=(world, uid_path, out_path), nprocs=world, join=True, start_method="spawn")
    got = pickle.load(open(out_path, "rb"))
    for name, cols, q, _ in _queries():
        info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in cols}
        ocols = parity.oracle_synth_cols(orc, cols, TOTAL, 0, TOTAL)
        o = orc.run_query(ocols, n_threads=8, **parity.oracle_query_kwargs(cols, info, q))
        g = got[name]
        assert g["matched"] == o["matched"], name
        omap = {r["key"]: r for r in o["results"]}
        assert len(g["rows"]) == len(omap), name
        for i, (key, count, hists) in enumerate(g["rows"]):
            orow = omap[key]
            assert count == orow["count"], (name, key)
            for a, (hc, hs, values, outliers) in enumerate(hists):
                oh = orow["hists"][a]
                assert (hc, hs) == (oh["count"], oh["sum_exact"]), (name, key, a)
                if values is not None:
                    assert np.array_equal(values, oh["values"]), (name, key, a)
                if name == "outliers":
                    assert outliers is not None and np.array_equal(outliers, oh["outlier_values"]), (name, key, a)
        if name == "scatter":
            assert g["strategy"] == 5 and sum(1 for r in g["rows"] if r[2][0][2] is not None) == 50   # the printed rows' buckets
        if name == "hash":
            assert g["strategy"] == 7
