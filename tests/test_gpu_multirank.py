"""GPU, N > 1: the in-library collective path (sybl_comm_* / sybl_query_allreduce, csrc/rccl.cpp) with one process per rank
-- what a Go host would run.  Every rank scans its contiguous block shard; the merged result must equal the oracle's over
the whole table (CombineResults / Result.Combine, aggregate.go:414-467, query_spec.go:138-193).

Two transports:
  * `standin` -- world 2, 4 and 8 on ONE device.  The workers run with tests/rccl_standin/librccl_standin.so in LD_PRELOAD:
    a TEST-ONLY implementation of the nine RCCL entry points the engine imports, over POSIX shared memory with host-staged,
    stream-ordered copies (its own protocol is checked on the CPU by tests/test_rccl_standin.py).  The product never links
    it.  This is how the N > 1 branches of rccl.cpp execute on the 1-GPU boxes this project is built and judged on.
  * `rccl` -- the real library, one process per GPU, min(device_count, 8) ranks; skipped below two GPUs.

Branches reached: the SUM all-reduce of direct-mapped cells (`direct`), + the MAX all-reduce of tracked
extrema (`extrema`), whole bucket tables all-reduced (`allreduce_hist`), the reduce-scatter of a big bucket table with
k_pack32 / k_unpack32, slice summaries and the collective finalize on every rank (`scatter32`), the same with int64 slices in
place (`scatter64`), outlier logs all-gathered and closed up in rank order (`outliers`), the hash group-by's key-union
protocol with SUM (`hash`) and SUM + MAX (`hash_extrema`), a time series (`timeseries`), count-distinct sketches merged with a
uint8 MAX (`distinct`), a printer's limit-aware merge -- cell fields all-reduced, Cumulative's buckets and the printed rows'
arrays summed in the collective snapshot / finalize, the bucket table never sent (`printer`), ranks without a single row -- the
finalizing rank among them -- direct-mapped and hashed (`tiny`, `tiny_hash`)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import multirank_worker as W
from tests import parity

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STANDIN = os.path.join(HERE, "rccl_standin", "librccl_standin.so")


class Dumped:
    """What a worker pickled of a sybil_amd.Result, with the accessors tests/parity.py uses."""

    def __init__(self, d):
        self.d = d
        self.matched = d["matched"]

    def rows(self, which=0):
        return self.d["rows"][which]

    @property
    def cumulative(self):
        return self.d["rows"][2][0]

    def distinct(self, which, i, registers=False):
        card, regs = self.d["registers"][which][i]
        return (card, regs) if registers else card


def run_world(tmp_path, world, standin, cases=None, extra_env=None, timeout=900):
    ndev = torch.cuda.device_count()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if standin:
        if not os.path.exists(STANDIN):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(STANDIN)])
        env["LD_PRELOAD"] = STANDIN
        env.setdefault("SYBL_STANDIN_TIMEOUT_S", "240")
    env.update(extra_env or {})
    work = str(tmp_path)
    args = [",".join(cases)] if cases else []
    logs = [open(os.path.join(work, "err%d.txt" % r), "wb") for r in range(world)]  # (files, not pipes: a full pipe would stall a rank)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.multirank_worker", work, str(world), str(r), str(0 if standin else r % ndev), *args],
                              cwd=ROOT, env=env, stderr=logs[r]) for r in range(world)]
    try:
        for p in procs:
            p.wait(timeout=timeout)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()
    errs = [open(os.path.join(work, "err%d.txt" % r), "rb").read().decode(errors="replace")[-2000:] for r in range(world)]
    assert [p.returncode for p in procs] == [0] * world, "\n".join(errs)
    return [pickle.load(open(os.path.join(work, "out%d.pkl" % r), "rb")) for r in range(world)]


_ORACLE = {}


def oracle_for(orc, name):
    """The oracle over the WHOLE table, once per case and session."""
    if name not in _ORACLE:
        from sybil_amd import synth
        cols, q, _, opt = W.CASES[name]
        info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in cols}
        for n, b in opt.get("bounds", {}).items():
            info[n] = b
        kw = parity.oracle_query_kwargs(cols, info, q)
        if q.get("distincts"):
            kw.update(distincts=[cols.index(c) for c in q["distincts"]], want_registers=True)
        total = opt.get("total", W.TOTAL)
        _ORACLE[name] = orc.run_query(parity.oracle_synth_cols(orc, cols, total, 0, total), n_threads=8, **kw)
    return _ORACLE[name]


def check_case(name, got, o):
    cols, q, _, _ = W.CASES[name]
    g = Dumped(got)
    n_aggs = len(q.get("aggs", ()))
    if q.get("distincts"):
        assert g.matched == o["matched"]
        omap = {r["key"]: r for r in o["results"]}
        assert len(g.rows(0)) == len(omap)
        for i, r in enumerate(g.rows(0)):
            card, regs = g.distinct(0, i, registers=True)
            assert np.array_equal(regs, omap[r["key"]]["registers"]) and card == omap[r["key"]]["distinct"] == r["distinct"], (name, r["key_vals"])
            assert r["count"] == omap[r["key"]]["count"]
        card, regs = g.distinct(2, 0, registers=True)
        assert np.array_equal(regs, o["cumulative"]["registers"]) and card == o["cumulative"]["distinct"]
    elif q.get("limit"):
        # only the printed rows carry bucket arrays; every row carries what the slice summaries delivered
        assert g.matched == o["matched"]
        omap = {r["key"]: r for r in o["results"]}
        rows = g.rows(0)
        assert len(rows) == len(omap)
        with_values = 0
        for r in rows:
            orow = omap[r["key"]]
            assert r["count"] == orow["count"], (name, r["key_vals"])
            for a in range(n_aggs):
                h, oh = r["hists"][a], orow["hists"][a]
                assert (h["count"], h["sum"], h["samples"]) == (oh["count"], oh["sum_exact"], oh["samples"]), (name, r["key_vals"])
                assert (h["min"], h["max"]) == (oh["min"], oh["max"])
                if "values" in h:
                    with_values += 1
                    assert np.array_equal(h["values"], oh["values"]), (name, r["key_vals"])
                if "values" in h or not q.get("printed_only"):
                    assert np.array_equal(h.get("percentiles", np.zeros(0, dtype=np.int64)), oh["percentiles"]), (name, r["key_vals"])
                    assert parity._close(h["stddev"], oh["stddev_exact"], 1e-9, max(abs(oh["avg"]), abs(oh["bucket_size"]), 1.0))
                else:  # a printer's result: rows beyond the printed ones carry neither percentiles nor stddev
                    assert "percentiles" not in h and h["stddev"] != h["stddev"], (name, r["key_vals"])
        assert with_values == q["limit"] * n_aggs
        for a in range(n_aggs):
            parity.compare_hist(g.cumulative["hists"][a], o["cumulative"]["hists"][a], "hist", True, ctx=(name, "cumulative"), cumulative=True)
    else:
        parity.compare(g, o, op=q.get("op", "avg"), full=q.get("want_percentiles", True), n_aggs=n_aggs, time_mode=bool(q.get("time_col")))
    if name.startswith("scatter") or name == "printer":
        assert got["strategy"] == 5 and got["everyone"], (name, got["strategy"], got["everyone"])
    if name.startswith("hash") or name == "tiny_hash":
        assert got["strategy"] == 7
    if name == "outliers":
        assert any(h.get("outlier_values") is not None and len(h["outlier_values"]) for r in g.rows(0) for h in r["hists"])


def check_world(outs, orc):
    assert set(outs[0]) == set(W.CASES)
    for name in W.CASES:
        o = oracle_for(orc, name)
        for r, out in enumerate(outs):
            if name in out:  # rank 0 always; every rank after a collective finalize
                check_case(name, out[name], o)
        if outs[0][name]["everyone"]:
            assert all(name in out for out in outs), name


@pytest.mark.parametrize("world", [2, 4, 8])
def test_inlibrary_collectives_on_one_device_through_the_standin(tmp_path, oracle, world):
    check_world(run_world(tmp_path, world, standin=True), oracle)


def test_standin_synchronous_steps(tmp_path, oracle):
    """The stand-in's fallback mode (no host functions) agrees: a difference would point at stream ordering."""
    outs = run_world(tmp_path, 2, standin=True, cases=["extrema", "scatter32"], extra_env={"SYBL_STANDIN_SYNC": "1"})
    for name in ("extrema", "scatter32"):
        check_case(name, outs[0][name], oracle_for(oracle, name))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the real RCCL needs >= 2 GPUs on this box")
def test_inlibrary_collectives_across_gpus(tmp_path, oracle):
    check_world(run_world(tmp_path, min(torch.cuda.device_count(), 8), standin=False), oracle)


@pytest.mark.parametrize("world", [2, 4])
def test_loaded_table_across_ranks(tmp_path, world, oracle):
    """The whole multi-GPU host protocol on a table that came from disk, through the C ABI alone
    (tests/multirank_loaded_worker.py): ranks open their block ranges, agree on bounds / str and set dictionaries / the sparse
    keys' group dictionaries inside the library (sybl_table_agree over the stand-in), merge, and rank 0's results must equal
    the ORACLE's over the table's logical content -- and those of one rank that opened the whole table: str keys, a sparse int
    key through the union dictionary's rank column, a set-member filter, a regex id mask with negative values, a time series,
    a hashed two-key group-by whose keys both go through union dictionaries."""
    import sybil_amd
    from tests import loaded_oracle as LO
    from tests import multirank_loaded_worker as LW
    from tests import sybil_fixture as F
    from tests.test_gpu_loader import _make_blocks
    blocks, logical = _make_blocks(7, 3000, seed=77, ragged=True)
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=8, int_info={"big": LO.INFO_BIG})
    lo = LO.LoadedOracle(oracle, logical, 8)
    want = [LO.summarise_oracle(lo, q, lo.run(q)) for q in LW.QUERIES]
    ctx = sybil_amd.Context(0)
    tb = ctx.open_table(root, "events", compact=True)
    one = []
    for q in LW.QUERIES:
        qy = tb.query(**q)
        r = qy.run()
        one.append(LO.summarise_engine(r, q.get("op", "avg")))
        r.free()
        qy.free()
    total_rows = tb.rows
    tb.free()
    ctx.close()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", LD_PRELOAD=STANDIN, SYBL_STANDIN_TIMEOUT_S="240")
    if not os.path.exists(STANDIN):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STANDIN)])
    work = str(tmp_path)
    logs = [open(os.path.join(work, "lerr%d.txt" % r), "wb") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, "-m", "tests.multirank_loaded_worker", work, str(world), str(r), "0", root], cwd=ROOT, env=env,
                              stdout=subprocess.DEVNULL, stderr=logs[r]) for r in range(world)]
    try:
        for p in procs:
            p.wait(timeout=600)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()
    errs = [open(os.path.join(work, "lerr%d.txt" % r), "rb").read().decode(errors="replace")[-2000:] for r in range(world)]
    assert [p.returncode for p in procs] == [0] * world, "\n".join(errs)
    got = pickle.load(open(os.path.join(work, "loaded.pkl"), "rb"))
    assert len(got["results"]) == len(want)
    for i, (g, w, o) in enumerate(zip(got["results"], want, one)):
        assert g["matched"] == w["matched"] == o["matched"], (i, LW.QUERIES[i])
        assert g["rows"] == w["rows"], (i, LW.QUERIES[i], "vs the oracle")
        assert g["rows"] == o["rows"], (i, LW.QUERIES[i], "vs one rank")
    assert got["results"][-1]["strategy"] == 7
    assert sum(1 for _ in blocks) == 7 and total_rows == sum(len(b["age"][1]) for b in blocks) == lo.n
