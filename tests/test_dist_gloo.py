"""world_size-2 gloo tests of the N>1 host protocol (sybil_amd/dist.py): block sharding, bounds
agreement and the SUM/MAX merge of partial group tables.  The tables here are produced by the CPU
oracle packed into the engine's cell layout -- the GPU scan that produces them in production is
covered by tests/test_gpu_parity.py::test_partials_add_up_like_ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack_cells(orc_result, g_lo, g_card, n_fields):
    """Oracle results -> [field][cell] int64 table (field 0 Count, 1 sum(v), 2 max(v), 3 max(-v))."""
    sums = np.zeros((2, g_card), dtype=np.int64)
    maxs = np.full((2, g_card), -(1 << 63), dtype=np.int64)
    for r in orc_result["results"]:
        cell = r["key_vals"][0] - g_lo
        sums[0, cell] = r["count"]
        sums[1, cell] = r["hists"][0]["sum_exact"]
        maxs[0, cell] = r["hists"][0]["true_max"]
        maxs[1, cell] = -r["hists"][0]["true_min"]
    return sums, maxs


def _worker(rank, world, port, total_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from sybil_amd import dist as sdist
        from sybil_amd import synth
        row0, nrows = sdist.shard(total_rows, rank, world)
        kind, idx, a, b, _, _ = synth.COLUMNS["c02"]
        g = orc.synth_fill(kind, a, b, synth.SEED, idx, row0, nrows, total_rows)
        # make the shards' value ranges differ so the bounds agreement matters
        g = np.where(g < 32, g if rank == 0 else g + 20, g) if nrows else g
        kind, idx, a, b, _, _ = synth.COLUMNS["c07"]
        v = orc.synth_fill(kind, a, b, synth.SEED, idx, row0, nrows, total_rows) - 500_000
        infos = {"g": {"exact_min": int(g.min()), "exact_max": int(g.max()), "has_missing": False},
                 "v": {"exact_min": int(v.min()), "exact_max": int(v.max()), "has_missing": rank == 1}}
        bounds = sdist.agree_bounds(infos)
        lo, hi = bounds["g"]["lo"], bounds["g"]["hi"]
        card = hi - lo + 1
        res = orc.run_query([{"type": "int", "data": g}, {"type": "int", "data": v}], groups=[0],
                            aggs=[(1, -500_000, 499_999)], op="avg")
        sums, maxs = _pack_cells(res, lo, card, 2)
        ts, tm = torch.from_numpy(sums.reshape(-1).copy()), torch.from_numpy(maxs.reshape(-1).copy())
        sdist.merge_partials(ts, tm)
        q.put((rank, row0, nrows, bounds, ts.numpy().copy(), tm.numpy().copy(), g, v))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_equals_whole_table():
    world, total = 2, 5 * 65536 + 1234
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, row0a, na, b0, s0, m0, g0, v0), (r1, row0b, nb, b1, s1, m1, g1, v1) = outs
    # shards are contiguous, block aligned, disjoint and complete
    assert row0a == 0 and row0b == na and na + nb == total and na % 65536 == 0
    # both ranks agreed on the same bounds and hold the same merged table
    assert b0 == b1 and np.array_equal(s0, s1) and np.array_equal(m0, m1)
    assert b0["v"]["has_missing"] is True
    g, v = np.concatenate([g0, g1]), np.concatenate([v0, v1])
    lo, hi = b0["g"]["lo"], b0["g"]["hi"]
    assert (lo, hi) == (int(g.min()), int(g.max()))
    card = hi - lo + 1
    sums, maxs = s0.reshape(2, card), m0.reshape(2, card)
    for k in range(card):
        sel = g == lo + k
        assert sums[0, k] == int(sel.sum())
        assert sums[1, k] == (int(v[sel].sum()) if sel.any() else 0)
        if sel.any():
            assert maxs[0, k] == int(v[sel].max()) and -maxs[1, k] == int(v[sel].min())
        else:
            assert maxs[0, k] == -(1 << 63)


def test_shard_partitions_blocks():
    from sybil_amd import synth
    for total in (1, 65536, 65537, 1_000_000_000, 10 * 65536 + 5):
        for world in (1, 2, 3, 8):
            spans = [synth.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b and (a + n) % 65536 == 0
            sizes = [n for _, n in spans]
            assert max(sizes) - min(sizes) <= 65536 or total < world * 65536


def test_single_process_is_a_noop():
    from sybil_amd import dist as sdist
    s, m = torch.arange(8), torch.arange(4)
    sdist.merge_partials(s, m)
    assert s.tolist() == list(range(8))
    b = sdist.agree_bounds({"a": {"exact_min": 3, "exact_max": 9, "has_missing": False},
                            "e": {"exact_min": 1, "exact_max": 0, "has_missing": True}})
    assert b["a"] == {"lo": 3, "hi": 9, "has_missing": False} and b["e"]["hi"] < b["e"]["lo"]
    # the agreed has_missing flag of a column nobody holds a value of is still declared (an empty range), and the
    # extreme int64 values survive the encoding of the minimum
    assert b["e"]["has_missing"] is True
    lo, hi = -(1 << 63), (1 << 63) - 1
    x = sdist.agree_bounds({"m": {"exact_min": lo, "exact_max": hi, "has_missing": False}})
    assert x["m"] == {"lo": lo, "hi": hi, "has_missing": False}

    class T:
        calls = []

        def set_bounds(self, name, lo, hi, has_missing=False):
            self.calls.append((name, lo, hi, has_missing))

    t = T()
    sdist.apply_bounds(t, b)
    assert ("a", 3, 9, False) in t.calls and any(c[0] == "e" and c[1] > c[2] and c[3] is True for c in t.calls)


class _FakeTable:
    """Stands in for sybil_amd.Table on a box without a GPU: only the dictionary accessors."""

    def __init__(self, distinct, strings, str_missing=False):
        self.distinct, self.strings = np.asarray(distinct, dtype=np.int64), list(strings)
        self.installed = {}
        self.str_missing = str_missing

    def column_info(self, name):
        return {"has_missing": self.str_missing}

    def set_bounds(self, name, lo, hi, has_missing=False):
        self.installed["b:" + name] = (lo, hi, has_missing)

    def column_distinct(self, name):
        return self.distinct

    def set_group_dict(self, name, values):
        self.installed["g:" + name] = np.asarray(values)

    def column_dict(self, name):
        return self.strings

    def set_dict(self, name, strings):
        self.installed["s:" + name] = list(strings)


def _dict_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sybil_amd import dist as sdist
        # (only rank 1 has rows without the str column: the MISSING key digit must still be declared on both)
        t = _FakeTable([5, 1 << 40, -3] if rank == 0 else [7, 5, 9], ["b", "a"] if rank == 0 else ["c", "a"], str_missing=rank == 1)
        u = sdist.agree_group_dict(t, "uid")
        s = sdist.agree_str_dict(t, "host")
        assert t.installed["b:host"] == (0, -1, True)
        q.put((rank, u.tolist(), s, t.installed["g:uid"].tolist(), t.installed["s:host"]))
    finally:
        dist.destroy_process_group()


def test_dictionary_agreement_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=120) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, u, s, gu, gs in outs:
        assert u == gu == [-3, 5, 7, 9, 1 << 40]       # sorted union: rank order == key order on every rank
        assert s == gs == ["a", "b", "c"]


def _sketch_worker(rank, world, port, total_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from sybil_amd import dist as sdist
        row0, nrows = sdist.shard(total_rows, rank, world, block_rows=4096)
        rng = np.random.default_rng(77)                      # the same table on every rank, sliced by shard
        g = rng.integers(0, 6, total_rows)[row0:row0 + nrows]
        user = rng.integers(0, 40_000, total_rows)[row0:row0 + nrows]
        res = orc.run_query([{"type": "int", "data": g}, {"type": "int", "data": user}], groups=[0], distincts=[1],
                            block_rows=4096, want_registers=True)
        regs = np.zeros((6, orc.LLB_M), dtype=np.uint8)      # direct-mapped cells, as the engine lays them out
        for r in res["results"]:
            regs[r["key_vals"][0]] = r["registers"]
        t = torch.from_numpy(regs.reshape(-1).copy())
        sdist.merge_sketches(t)
        q.put((rank, t.numpy().reshape(6, orc.LLB_M).copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sketch_merge_equals_whole_table():
    """Count distinct across ranks: the MAX all-reduce of the shards' sketches is the whole table's sketch."""
    from oracle import oracle as orc
    total = 60_000
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sketch_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(77)
    g = rng.integers(0, 6, total)
    user = rng.integers(0, 40_000, total)
    whole = orc.run_query([{"type": "int", "data": g}, {"type": "int", "data": user}], groups=[0], distincts=[1], want_registers=True)
    assert np.array_equal(got[0], got[1])
    for r in whole["results"]:
        k = r["key_vals"][0]
        assert np.array_equal(got[0][k], r["registers"])
        assert orc.LogLogBeta(got[0][k]).cardinality() == r["distinct"]
