"""GPU: the loader's default since round 6 (SYBL_LOADER_GPU_VARINT=0 turns it off) -- the varint walk of an int column's `Values` slice on the GPU (csrc/gobgpu.hip) against the
host parser (gob.cpp) on the same files: unpackIntCol, src/lib/column_store_io.go:690-780, reads them through encoding/gob
(decodeUint / decodeInt).  The two loads must produce the same resident columns, the same verdict on damaged blocks and the
same query results; a walk that meets anything it was not told to expect (values outside the block's info.db bounds, a short
or damaged slice) hands its block back to the host parser, which this file provokes on purpose."""
import os

import numpy as np
import pytest

from tests import gobfmt as G
from tests import sybil_fixture as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sybil_amd
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _open_both(ctx, monkeypatch, root, table, **kw):
    monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "0")
    host = ctx.open_table(root, table, **kw)
    monkeypatch.delenv("SYBL_LOADER_GPU_VARINT", raising=False)   # (the default: on)
    gpu = ctx.open_table(root, table, **kw)
    return host, gpu


def _same_columns(host, gpu, cols):
    assert gpu.rows == host.rows and gpu.blocks == host.blocks and gpu.broken_blocks == host.broken_blocks
    n = host.rows
    for c in cols:
        assert np.array_equal(gpu.read_int(c, 0, n), host.read_int(c, 0, n)), c
        a, b = host.column_info(c), gpu.column_info(c)
        assert a == b, (c, a, b)


def _summary(tb, **q):
    query = tb.query(**q)
    r = query.run()
    out = (r.matched, sorted((x["group_by_key"], x["count"], tuple((h["count"], h["sum"], h["min"], h["max"]) for h in x["hists"]))
                             for x in r.results))
    r.free()
    query.free()
    return out


def _values_of_every_length(rng, n):
    """Deltas whose zig-zag forms take 1..9 bytes on the wire, both signs, the int64 extremes among them."""
    bits = rng.integers(0, 63, size=n)
    mag = (rng.integers(0, 1 << 62, size=n, dtype=np.int64) >> (62 - bits)).astype(np.int64)
    v = np.where(rng.random(n) < 0.5, mag, -mag - 1)
    v = v.astype(np.int64)
    edge = np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).min, 0, -1], dtype=np.int64)
    v[:min(n, 4)] = edge[:min(n, 4)]
    return v


@pytest.mark.parametrize("compact", [False, True])
def test_value_encoded_columns_load_the_same_either_way(ctx, tmp_path, monkeypatch, compact):
    rng = np.random.default_rng(41)
    blocks = []
    for b in range(5):
        n = 65536 if b < 2 else int(rng.integers(1, 30000))
        t = (1_700_000_000 + b * 7200 + np.sort(rng.integers(0, 7200, size=n))).astype(np.int64)   # one- and two-byte deltas
        wide = rng.integers(-(1 << 40), 1 << 40, size=n).astype(np.int64)                          # six-byte deltas
        small = rng.integers(0, 1000, size=n).astype(np.int64)
        key = rng.integers(0, 7, size=n).astype(np.int64)                                          # stays bucket encoded
        blocks.append({"time": ("int", t), "wide": ("int", wide), "small": ("int", small), "key": ("int", key)})
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=100)
    host, gpu = _open_both(ctx, monkeypatch, root, "events", compact=compact)
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] == 20 and st["gpu_varint_redone"] == 0, st   # time, wide, small (values) and key (buckets) of five blocks
    assert host.load_stats()["gpu_varint_cols"] == 0
    _same_columns(host, gpu, ("time", "wide", "small", "key"))
    for q in (dict(groups=["key"], aggs=["wide", "time"], op="hist"), dict(filters=[("small", "gt", 500)], groups=["key"], aggs=["time"])):
        assert _summary(gpu, **q) == _summary(host, **q)
    gpu.free()
    host.free()


def test_unset_rows_of_a_value_encoded_column_are_zeros_the_bounds_must_cover(ctx, tmp_path, monkeypatch):
    """A value-encoded column's unset rows hold 0 (SaveIntsToColumns, column_store_io.go:97-114) and become populated on load
    (:758-766); the block's IntInfo was kept over the set values only, so 0 may lie outside its Min / Max.  Its Count says
    whether there are such rows (fewer set values than the slice is long): the block is then placed by bounds that include 0,
    exactly the extrema the host parser finds, and nothing has to be loaded twice."""
    rng = np.random.default_rng(5)
    blocks = []
    for b in range(4):
        n = 20000
        age = rng.integers(10, 300, size=n).astype(np.int64)
        pop = rng.random(n) > (0.2 if b % 2 else 0.0)     # blocks 1 and 3 have unset rows: 0 < Min = 10
        key = rng.integers(0, 5, size=n).astype(np.int64)
        blocks.append({"age": ("int", age, pop), "key": ("int", key)})
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=50)
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] == 8 and st["gpu_varint_redone"] == 0, st
    _same_columns(host, gpu, ("age", "key"))
    for q in (dict(groups=["key"], aggs=["age"], op="hist"), dict(filters=[("age", "lt", 5)], groups=["key"], aggs=["age"])):
        assert _summary(gpu, **q) == _summary(host, **q)
    gpu.free()
    host.free()


def test_a_block_whose_bounds_do_not_cover_its_values_is_loaded_again_by_the_host_parser(ctx, tmp_path, monkeypatch):
    """info.db says Min = 10 and every row is set, but the file holds a 3: k_decode_delta notices, the block's rows leave the
    scan and the host parser loads the block again behind the others -- same rows, same answers."""
    rng = np.random.default_rng(6)
    blocks = []
    for b in range(3):
        n = 9000
        blocks.append({"age": ("int", rng.integers(10, 300, size=n).astype(np.int64)), "key": ("int", rng.integers(0, 5, size=n).astype(np.int64))})
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks, threshold=50)
    # block 2's file rewritten with a value below the recorded minimum
    vals = np.asarray(blocks[1]["age"][1]).copy()
    vals[4000] = 3
    with open(os.path.join(root, "events", "block000000002", "int_age.db"), "wb") as f:
        f.write(F.int_column("age", vals, None, 50))
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] == 6 and st["gpu_varint_redone"] == 1, st
    assert gpu.rows == host.rows and gpu.broken_blocks == host.broken_blocks == 0
    for q in (dict(groups=["key"], aggs=["age"], op="hist"), dict(filters=[("age", "lt", 5)], groups=["key"], aggs=["age"])):
        assert _summary(gpu, **q) == _summary(host, **q)
    assert sorted(gpu.read_int("age", 0, gpu.rows).tolist()) == sorted(host.read_int("age", 0, host.rows).tolist())
    gpu.free()
    host.free()


def _write_block(tdir, index, cols, nrows, infos):
    bdir = os.path.join(tdir, "block%09d" % index)
    os.makedirs(bdir, exist_ok=True)
    for name, data in cols.items():
        with open(os.path.join(bdir, "int_%s.db" % name), "wb") as f:
            f.write(data)
    with open(os.path.join(bdir, "info.db"), "wb") as f:
        f.write(G.encode(G.saved_column_info(), {"NumRecords": nrows, "IntInfoMap": infos}))
    return bdir


def _info(vals):
    return {"Min": int(vals.min()), "Max": int(vals.max()), "Avg": 0.0, "M2": 0.0, "Count": int(vals.size)}


def _table_info(tdir, table, names, lo, hi):
    tbl = {"Name": table, "KeyTable": {n: i for i, n in enumerate(names)}, "KeyTypes": {i: F.INT_VAL for i in range(len(names))},
           "IntInfo": {i: {"Min": lo, "Max": hi, "Avg": 0.0, "M2": 0.0, "Count": 1} for i in range(len(names))}}
    with open(os.path.join(tdir, "info.db"), "wb") as f:
        f.write(G.encode(G.table_info(), tbl))


def test_every_varint_length_plain_and_value_encoded(ctx, tmp_path, monkeypatch):
    """`Values` as plain numbers (ValueEncoded false) and as deltas, with zig-zag forms of one to nine bytes and the int64
    extremes; files of 1 value, of a few, and of a size that leaves most of the kernel's threads without a chunk."""
    rng = np.random.default_rng(77)
    root = str(tmp_path / "db")
    tdir = os.path.join(root, "events")
    os.makedirs(tdir)
    sizes = [1, 2, 7, 63, 64, 65, 1000, 65536]
    for bi, n in enumerate(sizes):
        plain = _values_of_every_length(rng, n)
        deltas = _values_of_every_length(rng, n) >> 8        # (running sums stay inside int64)
        running = np.cumsum(deltas)
        cols = {"plain": G.encode(G.saved_int_column(), {"Name": "plain", "DeltaEncodedIDs": True, "VERSION": 1, "Values": [int(x) for x in plain]}),
                "run": G.encode(G.saved_int_column(), {"Name": "run", "DeltaEncodedIDs": True, "ValueEncoded": True, "VERSION": 1,
                                                       "Values": [int(x) for x in deltas]})}
        _write_block(tdir, bi + 1, cols, n, {"plain": _info(plain), "run": _info(running)})
    _table_info(tdir, "events", ["plain", "run"], -(1 << 62), 1 << 62)
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] == 2 * len(sizes) and st["gpu_varint_redone"] == 0, st
    _same_columns(host, gpu, ("plain", "run"))
    gpu.free()
    host.free()


def test_damaged_value_slices_get_the_host_parsers_verdict(ctx, tmp_path, monkeypatch):
    """A slice that announces more values than its bytes hold, one cut in the middle of a value, a byte that cannot start a
    value, more values than NumRecords, and info.db bounds that do not cover the values: whatever the host parser makes of
    each block (skipped, column empty, loaded), the GPU path ends up with the same."""
    rng = np.random.default_rng(8)
    root = str(tmp_path / "db")
    tdir = os.path.join(root, "events")
    os.makedirs(tdir)
    n = 5000
    vals = rng.integers(0, 1 << 20, size=n).astype(np.int64)
    good = G.encode(G.saved_int_column(), {"Name": "v", "DeltaEncodedIDs": True, "VERSION": 1, "Values": [int(x) for x in vals]})
    other = G.encode(G.saved_int_column(), {"Name": "w", "DeltaEncodedIDs": True, "VERSION": 1, "Values": [int(x) for x in vals[::-1]]})
    info = {"v": _info(vals), "w": _info(vals)}
    _write_block(tdir, 1, {"v": good, "w": other}, n, info)
    # 2: cut in the middle (the message length no longer matches: a decode error, the column stays empty)
    _write_block(tdir, 2, {"v": good[: len(good) // 2], "w": other}, n, info)
    # 3: a byte 0x90 where a value starts (the last value's first byte: 3-byte values are FE hi lo ... find one from the end)
    bad = bytearray(good)
    at = len(bad) - 12
    bad[at] = 0x90
    _write_block(tdir, 3, {"v": bytes(bad), "w": other}, n, info)
    # 4: more values than NumRecords ("BLOCK SIZE CHANGED DURING QUERY": the block is skipped)
    _write_block(tdir, 4, {"v": good, "w": other}, n - 1, info)
    # 5: bounds that do not cover the values (the walk notices; the host parser loads the block as it is)
    _write_block(tdir, 5, {"v": good, "w": other}, n, {"v": {"Min": 5, "Max": 6, "Avg": 0.0, "M2": 0.0, "Count": 1}, "w": _info(vals)})
    # 6: no IntInfoMap entry at all (the host parser from the start)
    _write_block(tdir, 6, {"v": good, "w": other}, n, {"w": _info(vals)})
    _table_info(tdir, "events", ["v", "w"], 0, 1 << 20)
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] > 0 and st["gpu_varint_redone"] >= 1, st
    # (a block loaded again leaves its first copy behind as an empty block: rows and verdicts must agree, not the block count)
    assert gpu.rows == host.rows and gpu.broken_blocks == host.broken_blocks
    assert host.broken_blocks >= 1
    for c in ("v", "w"):
        assert sorted(gpu.read_int(c, 0, gpu.rows).tolist()) == sorted(host.read_int(c, 0, host.rows).tolist()), c
    q = dict(filters=[("v", "gt", 1000)], aggs=["v", "w"], op="hist")
    assert _summary(gpu, **q) == _summary(host, **q)
    gpu.free()
    host.free()


def test_refresh_loads_new_blocks_through_the_same_path(ctx, tmp_path, monkeypatch):
    rng = np.random.default_rng(3)

    def block(b):
        n = 10000
        return {"t": ("int", (1000 * b + np.sort(rng.integers(0, 1000, size=n))).astype(np.int64)), "k": ("int", rng.integers(0, 4, size=n).astype(np.int64))}

    blocks = [block(b) for b in range(3)]
    root = str(tmp_path / "db")
    F.write_table(root, "events", blocks[:2], threshold=100)
    monkeypatch.delenv("SYBL_LOADER_GPU_VARINT", raising=False)
    tb = ctx.open_table(root, "events")
    assert tb.load_stats()["gpu_varint_cols"] == 4
    F.write_table(root, "events", blocks, threshold=100)
    tb.refresh()
    assert tb.load_stats()["gpu_varint_cols"] >= 1
    monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "0")
    ref = ctx.open_table(root, "events")
    q = dict(groups=["k"], aggs=["t"], op="hist")
    assert _summary(tb, **q) == _summary(ref, **q)
    assert sorted(tb.read_int("t", 0, tb.rows).tolist()) == sorted(ref.read_int("t", 0, ref.rows).tolist())
    tb.free()
    ref.free()


def _bucket_file(name, values, pop=None, delta=True):
    """A bucket-encoded int column file (SaveIntsToColumns, column_store_io.go:64-131): buckets in order of first appearance,
    record ids ascending, as differences when DeltaEncodedIDs."""
    values = np.asarray(values, dtype=np.int64)
    rows = np.arange(len(values)) if pop is None else np.nonzero(pop)[0]
    order, ids = [], {}
    for r in rows:
        v = int(values[r])
        if v not in ids:
            ids[v] = []
            order.append(v)
        ids[v].append(int(r))
    bins = []
    for v in order:
        r = np.asarray(ids[v], dtype=np.int64)
        bins.append({"Value": v, "Records": [int(x) for x in (np.diff(r, prepend=0) if delta else r)]})
    col = {"Name": name, "BucketEncoded": True, "Bins": bins, "VERSION": 1}
    if delta:
        col["DeltaEncodedIDs"] = True
    return G.encode(G.saved_int_column(), col)


@pytest.mark.parametrize("compact", [False, True])
def test_bucket_encoded_columns_load_the_same_either_way(ctx, tmp_path, monkeypatch, compact):
    """The buckets of bucket-encoded columns (k_gob_bins) in the shapes the wire format has: the bucket of row 0 (its first
    record is a 0 among the terminators) with one record, with many, and first or last among the buckets; Value 0 (the field is
    left out); negative and 64-bit values; ids as differences and absolute; rows without a value; one bucket holding every
    row; thousands of buckets; a one-row block."""
    rng = np.random.default_rng(19)
    root = str(tmp_path / "db")
    tdir = os.path.join(root, "events")
    os.makedirs(tdir)
    names = ["few", "zero", "neg", "abs", "holes", "one", "many", "solo0"]
    sizes = [65536, 1, 2, 300, 5000, 20000]
    for bi, n in enumerate(sizes):
        few = rng.integers(0, 7, size=n) * 1000 + 5
        zero = rng.integers(0, 3, size=n)                                  # a bucket with Value 0
        neg = rng.choice(np.array([-(1 << 62), -5, 0, 7, (1 << 62) + 12345], dtype=np.int64), size=n)
        ab = rng.integers(0, 40, size=n)
        holes = rng.integers(10, 20, size=n)
        hp = rng.random(n) > 0.3
        hp[0] = bi % 2 == 0                                                # row 0 with and without a value
        if not hp.any():
            hp[-1] = True
        one = np.full(n, 42)
        many = rng.integers(0, min(n, 6000), size=n)
        solo0 = np.arange(n) % min(max(n - 1, 1), 3000) + 100                       # row 0's value appears once when n > 2: a bucket [2][1][0][0]
        solo0[0] = 7
        vals = {"few": few, "zero": zero, "neg": neg, "abs": ab, "holes": holes, "one": one, "many": many, "solo0": solo0}
        cols = {c: _bucket_file(c, v, pop=hp if c == "holes" else None, delta=c != "abs") for c, v in vals.items()}
        infos = {c: _info(np.asarray(v, dtype=np.int64)[hp] if c == "holes" else np.asarray(v, dtype=np.int64)) for c, v in vals.items()}
        _write_block(tdir, bi + 1, cols, n, infos)
    _table_info(tdir, "events", names, -(1 << 62), 1 << 62)
    host, gpu = _open_both(ctx, monkeypatch, root, "events", compact=compact)
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] == len(names) * len(sizes) and st["gpu_varint_redone"] == 0, st
    _same_columns(host, gpu, names)
    for q in (dict(groups=["few"], aggs=["neg", "holes"], op="hist"), dict(filters=[("holes", "gt", 14)], groups=["zero"], aggs=["many"])):
        assert _summary(gpu, **q) == _summary(host, **q)
    gpu.free()
    host.free()


def test_damaged_buckets_get_the_host_parsers_verdict(ctx, tmp_path, monkeypatch):
    """Buckets that do not hold what they announce: a record id beyond NumRecords ("BLOCK SIZE CHANGED DURING QUERY": the
    block is skipped), a count larger than the records that follow, a file cut short, info.db announcing another number of set
    rows or bounds that do not cover the values (the host parser loads those as they are)."""
    rng = np.random.default_rng(23)
    root = str(tmp_path / "db")
    tdir = os.path.join(root, "events")
    os.makedirs(tdir)
    n = 4000
    vals = rng.integers(0, 9, size=n).astype(np.int64) + 3
    other = rng.integers(0, 5, size=n).astype(np.int64)
    good, good2 = _bucket_file("v", vals), _bucket_file("w", other)
    info = {"v": _info(vals), "w": _info(other)}
    _write_block(tdir, 1, {"v": good, "w": good2}, n, info)
    _write_block(tdir, 2, {"v": good, "w": good2}, n - 100, {"v": dict(_info(vals), Count=n - 100), "w": dict(_info(other), Count=n - 100)})   # ids beyond NumRecords
    _write_block(tdir, 3, {"v": good[: len(good) - 40], "w": good2}, n, info)                                 # cut short
    wrong = dict(_info(vals), Count=n - 1)
    _write_block(tdir, 4, {"v": good, "w": good2}, n, {"v": wrong, "w": _info(other)})                         # another Count
    _write_block(tdir, 5, {"v": good, "w": good2}, n, {"v": dict(_info(vals), Min=4), "w": _info(other)})      # bounds too narrow
    c0 = int((vals == vals[0]).sum())                                      # the first bucket's count: one more than there are
    head = b"\x01" + G.enc_int(int(vals[0])) + b"\x01"
    assert len(G.enc_uint(c0)) == len(G.enc_uint(c0 + 1)) and good.count(head + G.enc_uint(c0)) == 1
    _write_block(tdir, 6, {"v": good.replace(head + G.enc_uint(c0), head + G.enc_uint(c0 + 1)), "w": good2}, n, info)
    _table_info(tdir, "events", ["v", "w"], 0, 100)
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] > 0 and st["gpu_varint_redone"] >= 3, st
    assert gpu.rows == host.rows and gpu.broken_blocks == host.broken_blocks
    for c in ("v", "w"):
        assert sorted(gpu.read_int(c, 0, gpu.rows).tolist()) == sorted(host.read_int(c, 0, host.rows).tolist()), c
    q = dict(groups=["w"], aggs=["v"], op="hist")
    assert _summary(gpu, **q) == _summary(host, **q)
    gpu.free()
    host.free()


def _mutate(rng, data, lo):
    """One random piece of damage at or behind byte `lo`: a flipped byte, a run overwritten, a cut, bytes inserted or removed."""
    data = bytearray(data)
    kind = rng.integers(0, 6)
    at = int(rng.integers(lo, len(data)))
    if kind == 0:
        data[at] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        data[at] = int(rng.integers(0, 256))
    elif kind == 2:
        n = int(rng.integers(1, 9))
        data[at:at + n] = bytes(rng.integers(0, 256, size=min(n, len(data) - at), dtype=np.uint8))
    elif kind == 3:
        del data[at:]
    elif kind == 4:
        data[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 5)), dtype=np.uint8))
    else:
        del data[at:at + int(rng.integers(1, 5))]
    return bytes(data), "%s at %d" % (["bit flip", "byte", "run", "cut", "insert", "remove"][kind], at)


# (SYBL_FUZZ_SEEDS=n: n more seeds -- profiles/r06_gpu_varint_fuzz.txt is such a run)
@pytest.mark.parametrize("seed", [1, 2, 3] + list(range(100, 100 + int(os.environ.get("SYBL_FUZZ_SEEDS", "0")))))
def test_damaged_files_load_like_the_host_parser_loads_them(ctx, tmp_path, monkeypatch, seed):
    """Fuzz: 120 blocks, each with one value-encoded and one bucket-encoded column file damaged somewhere behind its type
    definitions (the same damage for both loads), beside an intact column that names the block.  Whatever the host parser
    makes of a block -- skipped, the column empty, the column loaded with other values -- the default load must end up with
    exactly the same rows per block: the walk's checks send everything it is not sure of back through the host parser."""
    rng = np.random.default_rng(seed)
    root = str(tmp_path / "db")
    tdir = os.path.join(root, "events")
    os.makedirs(tdir)
    n_blocks, what = 120, {}
    for b in range(n_blocks):
        n = int(rng.integers(1, 3000))
        vals = np.cumsum(rng.integers(-50, 60, size=n)).astype(np.int64) if rng.random() < 0.7 else _values_of_every_length(rng, n) >> 16
        few = rng.integers(0, int(rng.integers(1, 40)), size=n).astype(np.int64) * 7 - 20
        fv = F.int_column("v", vals, None, 0)            # value encoded
        fb = _bucket_file("b", few, delta=bool(rng.integers(0, 2)))
        blk = _bucket_file("blk", np.full(n, b))
        # (the first 60 bytes or so are the type definitions and the struct's first fields: damage there never reaches the walk)
        if b % 10 != 0:
            fv, w1 = _mutate(rng, fv, int(rng.integers(0, len(fv) - 1)) if rng.random() < 0.15 else max(len(fv) - int(rng.integers(1, 3 * n + 40)), 0))
            fb, w2 = _mutate(rng, fb, int(rng.integers(0, len(fb) - 1)) if rng.random() < 0.15 else max(len(fb) - int(rng.integers(1, 2 * n + 60)), 0))
            what[b] = (w1, w2)
        _write_block(tdir, b + 1, {"v": fv, "b": fb, "blk": blk}, n, {"v": _info(vals), "b": _info(few), "blk": _info(np.full(n, b))})
    _table_info(tdir, "events", ["v", "b", "blk"], -(1 << 62), 1 << 62)
    host, gpu = _open_both(ctx, monkeypatch, root, "events")
    st = gpu.load_stats()
    assert st["gpu_varint_cols"] > n_blocks and st["gpu_varint_redone"] > 0, st
    assert gpu.rows == host.rows and gpu.broken_blocks == host.broken_blocks, (gpu.rows, host.rows, gpu.broken_blocks, host.broken_blocks)

    def per_block(tb):
        out = {}
        for col in ("v", "b"):
            query = tb.query(groups=["blk"], aggs=[col], op="hist")
            r = query.run()
            for x in r.results:
                h = x["hists"][0]
                # (an aggregation no row of the group has a value for is absent from the output: its other fields mean nothing)
                out[(int(x["group_by_key"]), col)] = (x["count"], h["present"]) + ((h["count"], h["sum"], h["min"], h["max"]) if h["present"] else ())
            r.free()
            query.free()
        return out

    a, b = per_block(host), per_block(gpu)
    diff = {k: (a.get(k), b.get(k), what.get(k[0])) for k in set(a) | set(b) if a.get(k) != b.get(k)}
    assert not diff, dict(list(diff.items())[:5])
    # (no comparison of the stored columns here: what a row WITHOUT a value holds is nobody's business, and damaged buckets leave such rows)
    gpu.free()
    host.free()
