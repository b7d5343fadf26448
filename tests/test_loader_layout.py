"""The loader's worker half on the CPU (csrc/loader.cpp:prepare_block through the sybl_debug_block_layout hook): what one
block directory in the reference's on-disk format (column_store_io.go:64-358) becomes before it crosses PCIe -- element
widths, counts, extrema, and a digest of every region of the staging slab.  The decode variants must agree byte for byte:
the AVX-512 VBMI varint windows against the scalar loops, slices decoded straight to uint16 / int32 against int64 slices
narrowed afterwards, the AVX-512 copies / scans against the plain loops.  (What the GPU then makes of the slab is
tests/test_gpu_loader.py's business.)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import sybil_fixture as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"int": 1, "str": 2, "set": 3}


def layout(block_dir, cols):
    """cols: [(name, "int" | "str" | "set")]"""
    from sybil_amd import _native as N
    names = (C.c_char_p * len(cols))(*[c.encode() for c, _ in cols])
    types = (C.c_int32 * len(cols))(*[TYPES[t] for _, t in cols])
    s = N.lib().sybl_debug_block_layout(block_dir.encode(), names, types, len(cols))
    assert s is not None, N.lib().sybl_last_error()
    return s.decode()


def _fields(line):
    name, rest = line.split(" ", 1)
    return name, dict(kv.split("=", 1) for kv in rest.split())


def _blocks(root):
    """A table whose blocks walk through the encodings: (block rows, {column: spec})."""
    rng = np.random.default_rng(21)
    out = []
    for n in (1, 100, 5003, 65536):
        pop = rng.random(n) > 0.25
        vocab = ["host%05d" % i for i in range(7000)]
        few = [None if rng.random() < 0.1 else vocab[int(i)] for i in rng.integers(0, 40, n)]
        many = [vocab[int(i)] for i in rng.integers(0, 7000, n)]
        tags = [None if rng.random() < 0.2 else ["t%d" % x for x in rng.integers(0, 9, size=int(rng.integers(1, 4)))] for _ in range(n)]
        out.append({
            "low": ("int", rng.integers(-5, 40, n)),                               # bucket encoded, one-byte id deltas
            "mid": ("int", rng.integers(0, 1000, n)),                              # bucket encoded, one/two/three-byte deltas
            "holes": ("int", rng.integers(0, 50, n), pop),                         # bucket encoded with missing rows
            "vals": ("int", rng.integers(0, 1_000_000, n)),                        # value encoded above 5000 distinct: int32 deltas
            "wide": ("int", rng.integers(-(1 << 45), 1 << 45, n)),                 # value encoded, deltas beyond int32
            "vholes": ("int", rng.integers(0, 1_000_000, n), pop),                 # value encoded with missing rows
            "const": ("int", np.full(n, 7)),
            "few": ("str", few), "many": ("str", many), "tags": ("set", tags)})
    F.write_table(root, "t", out, extra_dirs=False)
    return out


COLS = [("low", "int"), ("mid", "int"), ("holes", "int"), ("vals", "int"), ("wide", "int"), ("vholes", "int"), ("const", "int"),
        ("few", "str"), ("many", "str"), ("tags", "set"), ("absent", "int")]


@pytest.fixture(scope="module")
def table(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("layout"))
    return root, _blocks(root)


def _all_layouts(root, n_blocks):
    return [layout(os.path.join(root, "t", "block%09d" % (b + 1)), COLS) for b in range(n_blocks)]


def test_layout_of_known_blocks(table):
    root, blocks = table
    for b, text in enumerate(_all_layouts(root, len(blocks))):
        lines = text.strip().split("\n")
        head = dict(kv.split("=") for kv in lines[0].split()[:3])
        n = len(blocks[b]["low"][1])
        assert head == {"rows": str(n), "unreadable": "0", "broken": "0"}, lines[0]
        cols = dict(_fields(l) for l in lines[1:])
        assert set(cols) == {c for c, _ in COLS}
        # fully populated int columns: the worker's extrema are the column's
        for name in ("low", "mid", "vals", "wide", "const"):
            v = np.asarray(blocks[b][name][1], dtype=np.int64)
            f = cols[name]
            assert f["stats"] == "1" and int(f["min"]) == v.min() and int(f["max"]) == v.max() and int(f["pop"]) == n, (b, name, f)
        # encodings as the writer chose them (CARDINALITY_THRESHOLD = 5000): bins below, per-row values above
        assert cols["low"]["kind"] == "1" and int(cols["low"]["bins"]) == len(set(blocks[b]["low"][1].tolist()))
        assert int(cols["low"]["recs"]) == n and cols["low"]["rec_w"] == "2"
        assert cols["few"]["kind"] == "3" and cols["tags"]["kind"] == "5" and cols["absent"]["kind"] == "0"
        if n == 65536:
            assert cols["vals"]["kind"] == "2" and cols["vals"]["venc"] == "1" and cols["vals"]["val_w"] == "4" and int(cols["vals"]["vals"]) == n
            assert cols["wide"]["kind"] == "2" and cols["wide"]["val_w"] == "8"
            assert cols["many"]["kind"] == "4" and cols["many"]["local_w"] == "2" and int(cols["many"]["strings"].split(":")[0]) > 5000
        # rows without a value: the ids cover the populated rows only
        holes = blocks[b]["holes"]
        assert int(cols["holes"]["recs"]) == int(np.count_nonzero(holes[2]))


SWITCHES = [{"SYBL_LOADER_WIDE_DECODE": "1"}, {"SYBL_GOB_NO_VBMI": "1", "SYBL_LOADER_NO_AVX512": "1"},
            {"SYBL_LOADER_WIDE_DECODE": "1", "SYBL_GOB_NO_VBMI": "1", "SYBL_LOADER_NO_AVX512": "1"}]


@pytest.mark.parametrize("switches", SWITCHES, ids=["wide", "scalar", "wide-scalar"])
def test_decode_variants_lay_out_the_same_slab(table, switches):
    """(the switches are read when the library loads: a process each)"""
    root, blocks = table
    want = _all_layouts(root, len(blocks))
    code = ("import sys, json; sys.path.insert(0, %r); from tests.test_loader_layout import _all_layouts; "
            "print(json.dumps(_all_layouts(%r, %d)))" % (ROOT, root, len(blocks)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **switches), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().split("\n")[-1])
    for b in range(len(blocks)):
        assert got[b] == want[b], "block %d:\n%s\n--- default:\n%s" % (b + 1, got[b], want[b])


def test_damaged_blocks_are_reported_not_laid_out(tmp_path):
    """A record id at or beyond NumRecords, more values than rows: "BLOCK SIZE CHANGED DURING QUERY" (column_store_io.go:
    524-526,572-574,733-735) -- the block is marked broken whichever decode path saw it; a missing info.db: unreadable."""
    from tests import gobfmt as G
    root = str(tmp_path)
    rng = np.random.default_rng(2)
    n = 3000
    F.write_table(root, "t", [{"a": ("int", rng.integers(0, 20, n)), "b": ("int", rng.integers(0, 1_000_000, n))}], threshold=10, extra_dirs=False)
    bdir = os.path.join(root, "t", "block000000001")
    cols = [("a", "int"), ("b", "int")]
    assert "broken=0" in layout(bdir, cols)
    good = open(os.path.join(bdir, "int_a.db"), "rb").read()
    # an id beyond the block (delta-encoded: the bin's ids sum past NumRecords)
    col = {"Name": "a", "DeltaEncodedIDs": True, "BucketEncoded": True, "VERSION": 1,
           "Bins": [{"Value": 1, "Records": [5, 10, 2990]}, {"Value": 2, "Records": [0, 1]}]}
    open(os.path.join(bdir, "int_a.db"), "wb").write(G.encode(G.saved_int_column(), col))
    assert "broken=1" in layout(bdir, cols)
    col["Bins"][0]["Records"] = [5, 10, 70000]  # (does not fit uint16: the narrow reader goes back for int64)
    open(os.path.join(bdir, "int_a.db"), "wb").write(G.encode(G.saved_int_column(), col))
    assert "broken=1" in layout(bdir, cols)
    col["Bins"][0]["Records"] = [5, 10, 2984]   # the last row exactly: fine
    open(os.path.join(bdir, "int_a.db"), "wb").write(G.encode(G.saved_int_column(), col))
    assert "broken=0" in layout(bdir, cols)
    open(os.path.join(bdir, "int_a.db"), "wb").write(good)
    # more values than NumRecords
    vcol = {"Name": "b", "ValueEncoded": True, "Values": [1] * (n + 1), "VERSION": 1}
    open(os.path.join(bdir, "int_b.db"), "wb").write(G.encode(G.saved_int_column(), vcol))
    assert "broken=1" in layout(bdir, cols)
    os.remove(os.path.join(bdir, "info.db"))
    assert "unreadable=1" in layout(bdir, cols)


def test_damaged_blocks_under_the_sanitizers(table, tmp_path):
    """tools/micro/loader_block_fuzz.cpp: prepare_block over copies of two of the table's blocks with one file damaged per
    trial (bytes overwritten / inserted / removed, the tail cut off, a huge length planted, the file removed) into a heap slab
    of exactly a worker's size -- now and then one that is too small --, built with AddressSanitizer and
    UndefinedBehaviorSanitizer: a write past the slab, a read past a decoded array or undefined arithmetic on a hostile
    length ends the run.  Both forms of the pass (column by column, and SYBL_LOADER_TWO_PASS=1), and the worker half of the default
    load (SYBL_LOADER_GPU_VARINT=1: slices located and copied as file bytes instead of decoded)."""
    root, _ = table
    exe = str(tmp_path / "loader_block_fuzz")
    b = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-x", "hip",
                        os.path.join(ROOT, "tools", "micro", "loader_block_fuzz.cpp"), "-o", exe, "-I", os.path.join(ROOT, "sybil_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), "-L", os.path.join(ROOT, "sybil_amd"), "-lsybilgpu",
                        "-Wl,-rpath," + os.path.join(ROOT, "sybil_amd"), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-3000:]
    specs = ["%s:%d" % (n, TYPES[t]) for n, t in COLS]
    env = dict(os.environ, ASAN_OPTIONS="allocator_may_return_null=1:detect_leaks=0")
    for block, trials, extra in ((3, 1200, {}), (4, 250, {}), (3, 500, {"SYBL_LOADER_TWO_PASS": "1"}), (3, 1200, {"SYBL_LOADER_GPU_VARINT": "1"}),
                                 (4, 250, {"SYBL_LOADER_GPU_VARINT": "1"})):
        r = subprocess.run([exe, os.path.join(root, "t", "block%09d" % block), str(tmp_path / ("scratch%d" % block)), str(trials)] + specs,
                           env=dict(env, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "laid out" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
        assert " 0 exceptions" in r.stdout, r.stdout  # (std::bad_alloc / length_error would mean a hostile length sized something)


def _fnv(data, h=1469598103934665603):
    for byte in data:
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _check_regions(bdir, n, cols_spec, seen):
    from tests import gobfmt as G
    prefix = {"int": "int_", "str": "str_", "set": "set_"}
    cols = dict(_fields(l) for l in layout(bdir, cols_spec).strip().split("\n")[1:])
    for name, typ in cols_spec:
        f = cols[name]
        path = os.path.join(bdir, prefix[typ] + name + ".db")
        if not os.path.exists(path):
            assert f["kind"] == "0", (name, f)
            continue
        col = G.decode(open(path, "rb").read())
        if typ != "int":
            strings = col.get("StringTable", [])
            assert f["strings"] == "%d:%016x" % (len(strings), _fnv(b"".join(x.encode() + b"\0" for x in strings))), (bdir, name)
        if typ == "set":
            continue  # (sets stay on the host as CSR: the GPU tests' business)
        if col.get("BucketEncoded"):
            bins = col.get("Bins", [])
            vals = np.array([x.get("Value", 0) for x in bins], dtype=np.int64)
            recs = [x.get("Records", []) for x in bins]
            off = np.cumsum([0] + [len(r) for r in recs]).astype(np.int64)
            flat = np.array([d for r in recs for d in r], dtype=np.uint16)
            assert (int(f["bins"]), int(f["recs"]), f["rec_w"], f["delta"]) == (len(bins), flat.size, "2", "1"), (bdir, name, f)
            assert f["binval"] == "%016x" % _fnv(vals.tobytes()), (bdir, name)
            assert f["binoff"] == "%016x" % _fnv(off.tobytes()), (bdir, name)
            assert f["rec"] == "%016x" % _fnv(flat.tobytes()), (bdir, name)
            seen.add(typ + "-bins")
            continue
        values = col.get("Values", [])
        if typ == "int":
            wide = any(v < -(1 << 31) or v >= (1 << 31) for v in values)
            arr = np.array(values, dtype=np.int64 if wide else np.int32)
            assert (int(f["vals"]), f["val_w"], f["venc"]) == (len(values), "8" if wide else "4", "1" if col.get("ValueEncoded") else "0"), (bdir, name, f)
            assert f["val"] == "%016x" % _fnv(arr.tobytes()), (bdir, name)
            seen.add("int-values-%d" % (8 if wide else 4))
        else:
            arr = np.array(values, dtype=np.uint16)
            assert (int(f["local"]), f["local_w"]) == (len(values), "2"), (bdir, name, f)
            assert f["loc"] == "%016x" % _fnv(arr.tobytes()), (bdir, name)
            seen.add("str-values")
        if 0 < len(values) < n:
            words = (n + 31) // 32
            bits = np.zeros(words, dtype=np.uint32)
            for w in range(words):
                lo = w * 32
                bits[w] = 0xFFFFFFFF if len(values) >= lo + 32 else ((1 << (len(values) - lo)) - 1 if len(values) > lo else 0)
            assert int(f["bits"]) == words and f["valid"] == "%016x" % _fnv(bits.tobytes()), (bdir, name)
            seen.add("validity-prefix")
        else:
            assert f["bits"] == "0", (bdir, name, f)


def test_slab_regions_hold_what_the_files_hold(table, tmp_path):
    """Independent of every C++ decode path: the column files are read back with the Python gob reader (tests/gobfmt.py, pinned
    on the reference's golden gobs) and each region of the slab is computed here -- bin values and offsets as int64, the record
    ids as they are in the file (deltas) at uint16, value-encoded columns' deltas at int32 (int64 when one does not fit), str
    columns' block-local ids at uint16, the validity prefix of a column with fewer values than rows, the string table -- and
    its FNV-1a digest held against the one the worker's slab gives (sybl_debug_block_layout)."""
    root, blocks = table
    seen = set()
    for b in (0, 1, 2):  # 1, 100 and 5003 rows: pure-Python digests of a 65 536-row block would take a minute
        _check_regions(os.path.join(root, "t", "block%09d" % (b + 1)), len(blocks[b]["low"][1]), COLS, seen)
    # the per-row encodings at a size Python digests in a second: a table written with a cardinality threshold of 8
    rng = np.random.default_rng(5)
    n = 3000
    tail = np.arange(n) < n - 77  # (the last 77 rows have no value: Values is shorter than the block)
    vocab = ["name%04d" % i for i in range(900)]
    small = {"vals": ("int", rng.integers(0, 1_000_000, n)), "wide": ("int", rng.integers(-(1 << 45), 1 << 45, n)),
             "short": ("int", rng.integers(0, 1_000_000, n), tail), "few": ("int", rng.integers(0, 5, n)),
             "many": ("str", [vocab[int(i)] if keep else None for i, keep in zip(rng.integers(0, 900, n), tail)])}
    F.write_table(str(tmp_path), "v", [small], threshold=8, extra_dirs=False)
    _check_regions(os.path.join(str(tmp_path), "v", "block000000001"), n,
                   [("vals", "int"), ("wide", "int"), ("short", "int"), ("few", "int"), ("many", "str")], seen)
    assert seen >= {"int-bins", "str-bins", "int-values-4", "int-values-8", "str-values", "validity-prefix"}, seen


def test_gpu_varint_worker_half_hands_over_the_files_value_bytes(tmp_path, monkeypatch):
    """SYBL_LOADER_GPU_VARINT (csrc/gobgpu.hip walks the varints): what travels for an int column stored as `Values` is the
    file's own bytes from the slice's first value to the end of the message -- here held against the encoder's bytes
    (tests/gobfmt.py) -- with the announced count and the block's info.db bounds; bucket-encoded columns and columns the
    block's info.db says nothing about go the host parser's way."""
    from tests import gobfmt as G
    rng = np.random.default_rng(12)
    n = 4000
    t = np.sort(rng.integers(0, 10 ** 6, n)).astype(np.int64)
    wide = rng.integers(-(1 << 50), 1 << 50, n).astype(np.int64)
    few = rng.integers(0, 5, n).astype(np.int64)
    root = str(tmp_path / "db")
    F.write_table(root, "t", [{"t": ("int", t), "wide": ("int", wide), "few": ("int", few)}], threshold=8)
    bdir = os.path.join(root, "t", "block000000001")
    cols = [("t", "int"), ("wide", "int"), ("few", "int")]
    plain = dict(_fields(l) for l in layout(bdir, cols).splitlines()[1:])
    monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "1")
    lines = dict(_fields(l) for l in layout(bdir, cols).splitlines()[1:])
    monkeypatch.delenv("SYBL_LOADER_GPU_VARINT")
    for name, vals in (("t", t), ("wide", wide)):
        f = lines[name]
        deltas = np.diff(vals, prepend=0)
        body = b"".join(G.enc_int(int(d)) for d in deltas)
        data = open(os.path.join(bdir, "int_%s.db" % name), "rb").read()
        at = data.index(body)
        region = data[at:]
        assert data[:at].endswith(G.enc_uint(n))                       # the slice's count stands right before it
        assert f["raw"] == "%d:%016x" % (len(region), _fnv(region)), name
        assert f["vals"] == str(n) and f["val_w"] == "8" and f["venc"] == "1" and f["kind"] == plain[name]["kind"]
        assert (int(f["min"]), int(f["max"]), int(f["pop"])) == (int(vals.min()), int(vals.max()), n)
        assert (plain[name]["min"], plain[name]["max"]) == (f["min"], f["max"]) and "raw" not in plain[name]
    # the bucket-encoded column: its `Bins` region travels, from the first bucket to the end of the message; the count of the
    # buckets' records is info.db's Count
    f = lines["few"]
    data = open(os.path.join(bdir, "int_few.db"), "rb").read()
    ids = {int(v): np.nonzero(few == v)[0] for v in np.unique(few)}
    starts = [data.index(b"".join([b"\x01", G.enc_int(v), b"\x01"] if v else [b"\x02"]) + G.enc_uint(len(r)) +
                         b"".join(G.enc_uint(int(d)) for d in np.diff(r, prepend=0))) for v, r in ids.items()]
    region = data[min(starts):]
    assert data[:min(starts)].endswith(G.enc_uint(len(ids)))
    assert f["raw"] == "%d:%016x" % (len(region), _fnv(region))
    assert (f["bins"], f["recs"], f["rec_w"], f["delta"]) == (str(len(ids)), str(n), "8", "1")
    assert (int(f["min"]), int(f["max"]), int(f["pop"])) == (int(few.min()), int(few.max()), n) and "raw" not in plain["few"]
    # no IntInfoMap entry for the column: nothing to place the block by, the host parser takes it
    info = G.decode(open(os.path.join(bdir, "info.db"), "rb").read())
    del info["IntInfoMap"]["wide"]
    open(os.path.join(bdir, "info.db"), "wb").write(G.encode(G.saved_column_info(), info))
    monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "1")
    lines = dict(_fields(l) for l in layout(bdir, cols).splitlines()[1:])
    assert "raw" in lines["t"] and lines["wide"] == plain["wide"]


def test_gpu_varint_reader_only_stops_at_a_slice_the_struct_ends_plainly_behind(tmp_path, monkeypatch):
    """The reader hands a slice over unparsed only when every field behind it is an integer -- then "[0] or [d][value][0]" is
    all it would accept there itself, which is what the calling thread holds the walk's tail values against.  A type definition
    damaged in a LATER field (found by tests/test_gpu_loader_varint.py's fuzzer: VERSION's type id) must send the file the
    whole reader's way, whose verdict -- the column stays empty -- is then the same with and without the switch."""
    rng = np.random.default_rng(3)
    n = 500
    vals = np.sort(rng.integers(0, 10 ** 6, n)).astype(np.int64)
    few = rng.integers(0, 5, n).astype(np.int64)
    root = str(tmp_path / "db")
    F.write_table(root, "t", [{"v": ("int", vals), "b": ("int", few)}], threshold=8)
    bdir = os.path.join(root, "t", "block000000001")
    cols = [("v", "int"), ("b", "int")]
    for name in ("v", "b"):
        path = os.path.join(bdir, "int_%s.db" % name)
        data = open(path, "rb").read()
        assert data.count(b"\x07VERSION\x01\x04") == 1          # field name, then its type id (int = 2, zig-zag 4)
        open(path, "wb").write(data.replace(b"\x07VERSION\x01\x04", b"\x07VERSION\x01\x17"))
    plain = dict(_fields(l) for l in layout(bdir, cols).splitlines()[1:])
    monkeypatch.setenv("SYBL_LOADER_GPU_VARINT", "1")
    walked = dict(_fields(l) for l in layout(bdir, cols).splitlines()[1:])
    assert walked == plain and all("raw" not in f for f in walked.values())
    assert all(f["kind"] == "0" for f in plain.values())       # (kAbsent: "DECODE COL ERR", the reference carries on with an empty column)
