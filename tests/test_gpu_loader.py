"""GPU: the native loader (sybl_table_open: gob column files -> decode kernels -> HBM) and set
columns, against the CPU oracle on the logical data the reference would hold after
LoadBlockFromDir (src/lib/table_block_io.go:225-310, column_store_io.go:493-780).
Tables are fabricated in the reference's on-disk format by tests/sybil_fixture.py."""
import numpy as np
import pytest

from tests import parity
from tests import sybil_fixture as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sybil_amd
    c = sybil_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["gpu_walk", "host_parser"], autouse=True)
def loader_path(request, monkeypatch):
    """Every test of this module twice: with the int column files' varints walked on the GPU (the default since round 6,
    csrc/gobgpu.hip) and with every file through the host parser (SYBL_LOADER_GPU_VARINT=0, the only path before)."""
    if request.param == "host_parser":
        monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "0")
    else:
        monkeypatch.delenv("SYBL_LOADER_GPU_VARINT", raising=False)
    return request.param


def _make_blocks(n_blocks, rows, seed=3, ragged=True):
    rng = np.random.default_rng(seed)
    blocks, logical = [], []
    for b in range(n_blocks):
        n = rows - (17 * b if ragged else 0)
        age = rng.integers(10, 30, size=n).astype(np.int64)
        t = (1_700_000_000 + b * 7200 + np.sort(rng.integers(0, 7200, size=n))).astype(np.int64)
        big = rng.integers(-(1 << 40), 1 << 40, size=n).astype(np.int64)
        big_pop = rng.random(n) > 0.2
        name = [None if rng.random() < 0.1 else "user%d" % rng.integers(0, 50) for _ in range(n)]
        tags = [None if rng.random() < 0.15 else sorted({"tag%d" % x for x in rng.integers(0, 12, size=rng.integers(1, 5))})
                for _ in range(n)]
        blk = {"age": ("int", age), "time": ("int", t), "big": ("int", big, big_pop), "name": ("str", name),
               "tags": ("set", tags)}
        if b == 1:
            del blk["big"]          # a block without the column file: unpopulated for the whole block
            big_pop = np.zeros(n, dtype=bool)
        blocks.append(blk)
        logical.append({"age": age, "time": t, "big": big, "big_pop": big_pop, "name": name, "tags": tags})
    return blocks, logical


def _logical_after_load(logical, threshold):
    """What the reference holds after unpack*: with value-encoded columns every row below
    len(Values) is populated, holes hold 0 / string id 0 (column_store_io.go:590-604,758-777)."""
    age = np.concatenate([l["age"] for l in logical])
    t = np.concatenate([l["time"] for l in logical])
    big, big_pop, names, name_pop, tags, tag_pop = [], [], [], [], [], []
    for l in logical:
        n = len(l["age"])
        b, bp = l["big"].copy(), l["big_pop"].copy()
        if bp.any() and len({int(x) for x in b[bp]}) > threshold:
            max_r = int(np.nonzero(bp)[0][-1]) + 1
            b[~bp] = 0
            bp = np.arange(n) < max_r
        big.append(b)
        big_pop.append(bp)
        nm = list(l["name"])
        present = [s for s in nm if s is not None]
        uniq = list(dict.fromkeys(present))
        np_ = np.array([s is not None for s in nm])
        if len(uniq) > threshold:
            max_r = int(np.nonzero(np_)[0][-1]) + 1
            nm = [(s if s is not None else uniq[0]) if r < max_r else None for r, s in enumerate(nm)]
            np_ = np.arange(n) < max_r
        names += nm
        name_pop.append(np_)
        tg = l["tags"]
        tp = np.array([bool(s) for s in tg])
        if len({x for s in tg if s for x in s}) > threshold and tp.any():
            tp = np.arange(n) < int(np.nonzero(tp)[0][-1]) + 1   # Values [][]int32: rows below len(Values) are SET_VAL
        tags += tg
        tag_pop.append(tp)
    return (age, t, np.concatenate(big), np.concatenate(big_pop), names, np.concatenate(name_pop), tags,
            np.concatenate(tag_pop))


@pytest.mark.parametrize("gz,threshold,compact", [(False, 5000, False), (True, 5000, False), (False, 8, False),
                                                  (False, 5000, True), (False, 8, True)])
def test_open_table_and_query(ctx, oracle, tmp_path, gz, threshold, compact):
    blocks, logical = _make_blocks(4, 3000)
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, gz=gz, threshold=threshold, int_info={"big": (-(1 << 40), 1 << 40)})
    tb = ctx.open_table(root, "events", compact=compact)   # compact: every decoded block packed as it arrives
    if compact:
        assert tb.column_storage("age")[0] == 1 and tb.column_storage("big")[0] == 8
    age, t, big, big_pop, names, name_pop, tags, tag_pop = _logical_after_load(logical, threshold)
    n = age.size
    assert tb.rows == n and tb.blocks == 4 and tb.broken_blocks == 0
    # decode check: dense columns come back exactly
    assert np.array_equal(tb.read_int("age", 0, n), age)
    assert np.array_equal(tb.read_int("time", 0, n), t)
    got_big = tb.read_int("big", 0, n)
    assert np.array_equal(got_big[big_pop], big[big_pop])
    ci = tb.column_info("big")
    assert ci["has_missing"] and ci["info_min"] == -(1 << 40) and ci["exact_min"] == int(big[big_pop].min())

    # oracle inputs: table-global dictionaries in first-seen order are private to the engine, so
    # compare through strings: the oracle gets its own ids
    uniq = sorted({s for s in names if s is not None})
    sid = np.array([uniq.index(s) if s is not None else 0 for s in names], dtype=np.int32)
    tag_names = sorted({x for s in tags if s for x in s})
    off = np.zeros(n + 1, dtype=np.int64)
    flat = []
    for r, s in enumerate(tags):
        if s:
            flat += [tag_names.index(x) for x in s]
        off[r + 1] = len(flat)
    tag_pop = tag_pop.astype(np.uint8)
    ocols = [{"type": "int", "data": age}, {"type": "int", "data": t},
             {"type": "int", "data": big, "populated": big_pop.astype(np.uint8)},
             {"type": "str", "data": sid, "populated": name_pop.astype(np.uint8)},
             {"type": "set", "data": np.array(flat, dtype=np.int32), "offsets": off, "populated": tag_pop}]
    cols = ["age", "time", "big", "name", "tags"]
    info = {"age": (10, 29), "big": (-(1 << 40), 1 << 40), "time": (int(t.min()), int(t.max()))}
    cases = [
        dict(groups=["age"], aggs=["big"], op="hist"),
        dict(groups=["name"], aggs=["age"], op="avg"),
        dict(filters=[("tags", "in", "tag3")], groups=["age"], aggs=["age"]),
        dict(filters=[("tags", "nin", "tag3")], groups=["age"]),
        dict(filters=[("tags", "in", "tag3"), ("tags", "nin", "tag5")], groups=["name"]),
        dict(filters=[("tags", "in", "no-such-tag")], groups=["age"]),
        dict(filters=[("tags", "nin", "no-such-tag")], groups=["age"]),
        dict(filters=[("name", "eq", "user7"), ("big", "gt", 0)], groups=["age"], aggs=["big"]),
        dict(groups=["age"], aggs=["big"], time_col="time", time_bucket=3600),
        dict(filters=[("time", "gt", int(t[n // 2]))], groups=["age"], block_skip=True),
    ]
    for q in cases:
        query = tb.query(**q)
        gres = query.run()
        okw = parity.oracle_query_kwargs(cols, info, q)
        fixed = []
        for f in okw["filters"]:
            if isinstance(f[2], str):
                table = uniq if cols[f[0]] == "name" else tag_names
                fixed.append((f[0], f[1], table.index(f[2]) if f[2] in table else -1))
            else:
                fixed.append(f)
        okw["filters"] = fixed
        ores = oracle.run_query(ocols, block_rows=10 ** 9, **okw)
        # group keys are dictionary ids private to each side: compare str groups through strings
        if q.get("groups") == ["name"]:
            gmap = {r["group_by_key"]: r for r in gres.results}
            omap = {("" if r["key_vals"][0] == 0xFFFFFFFFFFFFFFFF else uniq[r["key_vals"][0]]) + "\t": r for r in ores["results"]}
            assert set(gmap) == set(omap)
            for k, o in omap.items():
                assert gmap[k]["count"] == o["count"]
                for a in range(len(q.get("aggs", []))):
                    parity.compare_hist(gmap[k]["hists"][a], o["hists"][a], q.get("op", "avg"), True, ctx=(k, a))
            assert gres.matched == ores["matched"]
        else:
            parity.compare(gres, ores, op=q.get("op", "avg"), full=True, n_aggs=len(q.get("aggs", [])),
                           time_mode=bool(q.get("time_col")))
        gres.free()
        query.free()
    tb.free()


def test_blocks_larger_than_a_staging_slab(ctx, tmp_path, monkeypatch):
    """The loader lays every block out in a pinned slab sized for a reference block; a block that needs more brings a
    pinned / device pair of its own (forced here by a tiny slab size) and decodes to the same columns."""
    blocks, logical = _make_blocks(3, 5000, seed=9)
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=8, int_info={"big": (-(1 << 40), 1 << 40)})
    ref = ctx.open_table(root, "events")
    monkeypatch.setenv("SYBL_LOADER_SLAB_BYTES", "4096")
    tb = ctx.open_table(root, "events")
    n = ref.rows
    assert tb.rows == n and tb.broken_blocks == 0
    for c in ("age", "time", "big"):
        assert np.array_equal(tb.read_int(c, 0, n), ref.read_int(c, 0, n)), c
    for q in (dict(groups=["name"], aggs=["age"]), dict(filters=[("tags", "in", "tag3")], groups=["age"], aggs=["big"], op="hist")):
        qa, qb = ref.query(**q), tb.query(**q)
        ra, rb = qa.run(), qb.run()
        assert ra.matched == rb.matched
        assert {(r["group_by_key"], r["count"], tuple(h["sum"] for h in r["hists"])) for r in ra.results} == \
               {(r["group_by_key"], r["count"], tuple(h["sum"] for h in r["hists"])) for r in rb.results}
        ra.free(); rb.free(); qa.free(); qb.free()
    tb.free()
    ref.free()


def _summary(tb, q):
    query = tb.query(**q)
    r = query.run()
    out = (r.matched, sorted((x["group_by_key"], x["count"], tuple((h["count"], h["sum"]) for h in x["hists"])) for x in r.results))
    r.free()
    query.free()
    return out


def test_refresh_follows_the_directory(ctx, tmp_path):
    """sybl_table_refresh: a block that appeared is loaded, one that vanished leaves the scan, one that was rewritten
    with another NumRecords (the reference's resized-block case, table_query_test.go:11-158) is loaded again -- and the
    table then answers like a freshly opened one."""
    import shutil
    import sybil_amd
    blocks, _ = _make_blocks(4, 3000, seed=21, ragged=False)
    info = {"big": (-(1 << 40), 1 << 40)}
    root, spare = str(tmp_path / "db"), str(tmp_path / "spare")
    F.write_table(root, "events", blocks[:3], threshold=8, int_info=info)
    tb = ctx.open_table(root, "events", compact=True)
    queries = [dict(groups=["age"], aggs=["big", "time"]), dict(filters=[("tags", "in", "tag3")], groups=["name"], aggs=["age"], op="hist")]
    stale = tb.query(**queries[0])
    stale.run().free()
    assert tb.refresh() == (0, 0, 0) and tb.rows == 9000
    stale.run().free()  # nothing changed: the prepared query is still good... after being prepared again
    # the directory moves on: block 4 appears, block 2 vanishes, block 1 is rewritten with fewer rows
    resized = {c: (spec[0],) + tuple(x[:1234] for x in spec[1:]) for c, spec in blocks[0].items()}
    F.write_table(spare, "events", [resized, blocks[1], blocks[2], blocks[3]], threshold=8, int_info=info)
    tdir, sdir = str(tmp_path / "db" / "events"), str(tmp_path / "spare" / "events")
    shutil.rmtree(tdir + "/block000000002")
    shutil.rmtree(tdir + "/block000000001")
    shutil.copytree(sdir + "/block000000001", tdir + "/block000000001")
    shutil.copytree(sdir + "/block000000004", tdir + "/block000000004")
    shutil.copy(sdir + "/info.db", tdir + "/info.db")
    assert tb.refresh() == (1, 1, 1)
    assert tb.rows == 1234 + 3000 + 3000 and tb.broken_blocks == 0
    with pytest.raises(sybil_amd.SyblError):
        stale.scan()  # prepared before the table changed
    stale.free()
    fresh = ctx.open_table(root, "events", compact=True)
    assert fresh.rows == tb.rows
    for q in queries:
        assert _summary(tb, q) == _summary(fresh, q), q
    assert tb.refresh() == (0, 0, 0)
    fresh.free()
    tb.free()


def test_refresh_reuses_the_rows_of_a_rewritten_last_block(ctx, tmp_path):
    """Sybil's ingest rewrites the last, partly filled block on every digest: a resident table that follows its directory
    must not grow by a block per refresh.  The rewritten block's rows are the table's tail, so the reloaded block takes
    their place (block count and HBM footprint stay put) -- with missing values, a set column and str keys in play --
    and the table answers like a freshly opened one every time."""
    import shutil
    blocks, _ = _make_blocks(4, 3000, seed=33, ragged=False)
    info = {"big": (-(1 << 40), 1 << 40)}
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks[:3], threshold=8, int_info=info)
    tb = ctx.open_table(root, "events", compact=True)
    queries = [dict(groups=["age"], aggs=["big", "time"]), dict(filters=[("tags", "in", "tag3")], groups=["name"], aggs=["age"], op="hist")]
    assert tb.blocks == 3
    hbm = tb.hbm_bytes  # three full blocks resident
    tdir = str(tmp_path / "db" / "events")
    for turn, n in enumerate((2100, 2999, 700, 3000)):
        spare = str(tmp_path / ("spare%d" % turn))
        last = {c: (spec[0],) + tuple(x[:n] for x in spec[1:]) for c, spec in blocks[3 if turn % 2 else 2].items()}
        F.write_table(spare, "events", [blocks[0], blocks[1], last], threshold=8, int_info=info)
        sdir = spare + "/events"
        shutil.rmtree(tdir + "/block000000003")  # (the fixture numbers block directories from 1)
        shutil.copytree(sdir + "/block000000003", tdir + "/block000000003")
        shutil.copy(sdir + "/info.db", tdir + "/info.db")
        assert tb.refresh() == (0, 0, 1)
        assert tb.blocks == 3 and tb.rows == 6000 + n and tb.broken_blocks == 0
        fresh = ctx.open_table(root, "events", compact=True)
        for q in queries:
            assert _summary(tb, q) == _summary(fresh, q), (turn, q)
        fresh.free()
        # (set members, dictionaries and staging blocks move with the content; a leaked block per refresh would add a third
        # of the table every turn)
        assert tb.hbm_bytes < hbm * 1.15, (turn, tb.hbm_bytes, hbm)
    tb.free()


def test_column_subset_and_rank_sharding(ctx, tmp_path):
    blocks, logical = _make_blocks(6, 2000, ragged=False)
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks)
    whole = ctx.open_table(root, "events", columns=["age", "time"])
    q = whole.query(groups=["age"], aggs=["time"])
    rw = q.run()
    total = {r["key_vals"][0]: (r["count"], r["hists"][0]["sum"]) for r in rw.results}
    acc = {}
    rows = 0
    for rank in range(4):
        part = ctx.open_table(root, "events", columns=["age", "time"], rank=rank, nranks=4)
        assert part.blocks in (1, 2)
        rows += part.rows
        qp = part.query(groups=["age"], aggs=["time"])
        rp = qp.run()
        for r in rp.results:
            c, s = acc.get(r["key_vals"][0], (0, 0))
            acc[r["key_vals"][0]] = (c + r["count"], s + r["hists"][0]["sum"])
        rp.free()
        qp.free()
        part.free()
    assert rows == whole.rows == 12000 and acc == total
    import sybil_amd
    with pytest.raises(sybil_amd.SyblError):
        whole.query(groups=["name"])          # not loaded (LoadSpec semantics)
    with pytest.raises(sybil_amd.SyblError):
        ctx.open_table(root, "events", columns=["nope"])
    with pytest.raises(sybil_amd.SyblError):
        ctx.open_table(root, "no_such_table")
    rw.free()
    q.free()
    whole.free()


def test_resized_block_is_skipped_not_fatal(ctx, tmp_path):
    """table_query_test.go:11-158: a block whose info.db disagrees with its column files is
    skipped ("BLOCK SIZE CHANGED DURING QUERY"), the rest of the table still answers."""
    import os
    from tests import gobfmt as G
    blocks, logical = _make_blocks(3, 1000, ragged=False)
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks)
    bdir = os.path.join(root, "events", "block000000002")
    open(os.path.join(bdir, "info.db"), "wb").write(G.encode(G.saved_column_info(), {"NumRecords": 500}))
    tb = ctx.open_table(root, "events", columns=["age"])
    assert tb.blocks == 2 and tb.broken_blocks == 1 and tb.rows == 2000
    r = tb.query(groups=["age"]).run()
    assert r.matched == 2000
    r.free()
    tb.free()
    # unreadable block info => also skipped
    open(os.path.join(bdir, "info.db"), "wb").write(b"garbage")
    tb = ctx.open_table(root, "events", columns=["age"])
    assert tb.blocks == 2 and tb.broken_blocks == 1
    tb.free()


def test_append_block_with_set_column(ctx, oracle):
    n = 5000
    rng = np.random.default_rng(4)
    age = rng.integers(10, 30, size=n).astype(np.int64)
    strings = ["t%d" % i for i in range(10)]
    off = np.zeros(n + 1, dtype=np.int64)
    ids = []
    pop = (rng.random(n) > 0.1).astype(np.uint8)
    for r in range(n):
        k = int(rng.integers(0, 4))
        ids += rng.integers(0, 10, size=k).tolist()
        off[r + 1] = len(ids)
    ids = np.array(ids, dtype=np.int32)
    tb = ctx.create_table("s")
    tb.add_column("age", "int")
    tb.add_column("tags", "set")
    for r0 in range(0, n, 1024):
        r1 = min(r0 + 1024, n)
        tb.append_block(r1 - r0, {"age": age[r0:r1],
                                  "tags": {"ids": ids[off[r0]:off[r1]], "offsets": off[r0:r1 + 1] - off[r0],
                                           "strings": strings, "populated": pop[r0:r1]}})
    ocols = [{"type": "int", "data": age}, {"type": "set", "data": ids, "offsets": off, "populated": pop}]
    for op, tag in (("in", "t3"), ("nin", "t3"), ("in", "t9")):
        q = tb.query(filters=[("tags", op, tag)], groups=["age"])
        g = q.run()
        o = oracle.run_query(ocols, filters=[(1, op, strings.index(tag))], groups=[0], block_rows=1024)
        parity.compare(g, o)
        g.free()
        q.free()
    import sybil_amd
    with pytest.raises(sybil_amd.SyblError):
        tb.query(groups=["tags"])     # cmd_query.go:254: cannot group by a set column
    tb.free()


def test_refresh_gives_back_the_rows_of_blocks_that_vanished_mid_table(ctx, tmp_path, monkeypatch):
    """sybil trim / expire removes whole block directories from the middle of a table.  A resident table that follows its
    directory drops them from the scan at once; once their rows are worth it (here: always, SYBL_RECLAIM_ALWAYS) the live
    blocks close up -- rows, validity bits, set members -- so a host that runs for weeks does not keep every expired block
    in HBM.  The table answers like a freshly opened one afterwards, and blocks that appear later land behind it."""
    import shutil
    monkeypatch.setenv("SYBL_RECLAIM_ALWAYS", "1")
    blocks, _ = _make_blocks(6, 3000, seed=57, ragged=False)
    info = {"big": (-(1 << 40), 1 << 40)}
    root, spare = str(tmp_path / "db"), str(tmp_path / "spare")
    F.write_table(root, "events", blocks[:5], threshold=8, int_info=info)
    F.write_table(spare, "events", blocks, threshold=8, int_info=info)
    tb = ctx.open_table(root, "events", compact=True)
    hbm5 = tb.hbm_bytes
    queries = [dict(groups=["age"], aggs=["big", "time"]), dict(filters=[("tags", "in", "tag3")], groups=["name"], aggs=["age"], op="hist")]
    tdir, sdir = str(tmp_path / "db" / "events"), str(tmp_path / "spare" / "events")
    shutil.rmtree(tdir + "/block000000002")
    shutil.rmtree(tdir + "/block000000003")
    assert tb.refresh() == (0, 2, 0)
    assert tb.rows == 9000 and tb.broken_blocks == 0
    fresh = ctx.open_table(root, "events", compact=True)
    for q in queries:
        assert _summary(tb, q) == _summary(fresh, q), q
    fresh.free()
    # a block appears: it lands behind the closed-up rows, and the table is no bigger than it was with five blocks
    shutil.copytree(sdir + "/block000000006", tdir + "/block000000006")
    shutil.copy(sdir + "/info.db", tdir + "/info.db")
    assert tb.refresh() == (1, 0, 0)
    assert tb.rows == 12000
    fresh = ctx.open_table(root, "events", compact=True)
    for q in queries:
        assert _summary(tb, q) == _summary(fresh, q), q
    fresh.free()
    assert tb.hbm_bytes <= hbm5 * 1.05, (tb.hbm_bytes, hbm5)
    tb.free()


def test_group_dictionary_after_a_trim_only_refresh_reads_the_moved_blocks(ctx, tmp_path, monkeypatch):
    """table_reclaim_dead_rows moves the live blocks' rows, but the device copy of the block segments was only
    uploaded when some column's statistics were pending -- not after a refresh that only DROPS blocks.  The next sparse-key
    group dictionary (k_distinct walks those segments) then read the old row ranges: values of a moved block were missing
    from the dictionary and its rows fell out of the result.  Sequence: a sparse int key grouped once (dictionary built,
    segments uploaded), a block dropped without reclaim (its n = 0 uploaded with the next statistics), a second block
    dropped WITH reclaim and nothing added, the key grouped again."""
    import shutil
    blocks, _ = _make_blocks(6, 3000, seed=91, ragged=False)
    info = {"big": (-(1 << 40), 1 << 40)}
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=8, int_info=info)
    tb = ctx.open_table(root, "events", compact=True)
    q = dict(groups=["big"], aggs=["age"])
    tdir = str(tmp_path / "db" / "events")
    first = _summary(tb, q)
    assert len(first[1]) > 10_000  # (a sparse key: nearly every populated row a group of its own)
    shutil.rmtree(tdir + "/block000000002")
    assert tb.refresh() == (0, 1, 0)  # below the reclaim threshold: the block just leaves the scan
    assert _summary(tb, dict(groups=["age"], aggs=["time"]))[0] == 15000
    monkeypatch.setenv("SYBL_RECLAIM_ALWAYS", "1")
    shutil.rmtree(tdir + "/block000000001")
    assert tb.refresh() == (0, 1, 0)  # trim only: the live blocks close up, no column's statistics become pending
    fresh = ctx.open_table(root, "events", compact=True)
    got, want = _summary(tb, q), _summary(fresh, q)
    assert got == want
    assert got[0] == 12000
    fresh.free()
    tb.free()


def test_ctx_trim_gives_the_loaders_arena_back_and_the_next_load_takes_it_again(ctx, tmp_path):
    """sybl_ctx_trim: the staging arena a load leaves behind for the next one is freed on request; resident tables
    keep answering, and a refresh afterwards allocates the arena again."""
    import shutil
    blocks, _ = _make_blocks(4, 3000, seed=5, ragged=False)
    root, spare = str(tmp_path / "db"), str(tmp_path / "spare")
    F.write_table(root, "events", blocks[:3], threshold=8)
    F.write_table(spare, "events", blocks, threshold=8)
    tb = ctx.open_table(root, "events", compact=True)
    q = dict(groups=["age"], aggs=["time"])
    before = _summary(tb, q)
    ctx.trim()
    ctx.trim()  # (idempotent)
    assert _summary(tb, q) == before
    shutil.copytree(spare + "/events/block000000004", root + "/events/block000000004")
    shutil.copy(spare + "/events/info.db", root + "/events/info.db")
    assert tb.refresh() == (1, 0, 0)
    fresh = ctx.open_table(root, "events", compact=True)
    assert _summary(tb, q) == _summary(fresh, q) and tb.rows == 12000
    fresh.free()
    tb.free()
