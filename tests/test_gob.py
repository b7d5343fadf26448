"""Pins the Python gob encoder/decoder (tests/gobfmt.py) and, through it, the fixtures the
native loader is tested with, against the reference's golden gob stream
(src/lib/testdata/TestDecodeGoldenFiles/flag_defs.golden.gob, decoding_test.go:20-74): a hex dump
of the 941-byte stream and the expected decoded value, written by tests/golden/make_golden.py."""
import json

import numpy as np
import os

from tests import gobfmt as G

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _golden_stream():
    return bytes.fromhex("".join(open(os.path.join(GOLD, "flagdefs_stream.hex")).read().split()))


def _golden_expected():
    return json.load(open(os.path.join(GOLD, "flagdefs_expected.json")))


def test_decode_golden_flag_defs_matches_golden_json():
    data = _golden_stream()
    v, order = G.decode(data, want_types=True)
    assert order == [(65, "struct", "FlagDefs")]
    # Go re-marshals nil pointers as null (dropped from the expectation); gob omitted them
    assert v == _golden_expected()


def test_reencode_golden_flag_defs_byte_exact():
    """Decode the golden stream, rebuild the struct type from its own type definition, encode it
    with our Encoder: the bytes must equal Go's (PrintBytes appends '\\n', printer.go:272-282)."""
    data = _golden_stream()
    v = G.decode(data)
    # recover the field list (name, type id) from the definition message
    h = G.Reader(data)
    n = h.uint()
    r = G.Reader(data, h.p, h.p + n)
    assert r.int() == -65
    td = G._read_wiretype(r)
    basic = {G.BOOL: G.Bool, G.INT: G.Int, G.UINT: G.Uint, G.FLOAT: G.Float, G.STRING: G.String}
    t = G.Struct(td["name"], [(fname, basic[fid]) for fname, fid in td["fields"]])
    assert G.encode(t, v) + b"\n" == data


def test_primitives():
    assert G.enc_uint(0) == b"\x00" and G.enc_uint(127) == b"\x7f" and G.enc_uint(850) == b"\xfe\x03\x52"
    assert G.enc_int(-1) == b"\x01" and G.enc_int(1) == b"\x02" and G.enc_int(-129) == b"\xfe\x01\x01"
    assert G.enc_float(17.0) == b"\xfe\x31\x40"       # gob doc: 17.0 -> fe 31 40
    for x in (0, 1, -1, 2 ** 40, -2 ** 40, 2 ** 62, -2 ** 63):
        assert G.Reader(G.enc_int(x)).int() == x
    for x in (0.0, 1.5, -2.25e300, 3.141592653589793):
        assert G.Reader(G.enc_float(x)).float() == x


def test_round_trip_column_structs():
    col = {"Name": "age", "DeltaEncodedIDs": True, "BucketEncoded": True, "VERSION": 1,
           "Bins": [{"Value": -5, "Records": [0, 1, 1]}, {"Value": 1 << 50, "Records": [1, 7]}]}
    data = G.encode(G.saved_int_column(), col)
    v, order = G.decode(data, want_types=True)
    assert v == col
    # id allocation follows encoding/gob: struct ids at creation, slice ids after their element
    assert [o[0] for o in order] == [65, 68, 66, 67, 69]
    assert dict((o[0], o[2]) for o in order) == {65: "SavedIntColumn", 66: "SavedIntBucket", 67: "[]uint32",
                                                  68: "[]sybil.SavedIntBucket", 69: "[]int64"}
    sc = {"Name": "tags", "Values": [[1, 2], [], [3]], "StringTable": ["a", "b", "c", "d"], "VERSION": 1}
    assert G.decode(G.encode(G.saved_set_column(), sc)) == {k: v for k, v in sc.items()}
    info = {"NumRecords": 100, "IntInfoMap": {"age": {"Min": 10, "Max": 29, "Avg": 19.5, "M2": 3.25, "Count": 100}}}
    assert G.decode(G.encode(G.saved_column_info(), info)) == info


# ---- the library's C++ gob reader (csrc/gob.cpp) through the sybl_debug_gob_to_json hook

def _cxx_json(path):
    from sybil_amd import _native as N
    s = N.lib().sybl_debug_gob_to_json(path.encode())
    assert s is not None, N.lib().sybl_last_error()
    return json.loads(s)


def test_cxx_reader_on_golden_flag_defs(tmp_path):
    p = tmp_path / "flagdefs.gob"
    p.write_bytes(_golden_stream())
    assert _cxx_json(str(p)) == _golden_expected()


def test_cxx_reader_on_fixture_column_files(tmp_path):
    import numpy as np
    from tests import sybil_fixture as F
    rng = np.random.default_rng(0)
    n = 3000
    vals = rng.integers(-1000, 1 << 45, size=n)
    pop = rng.random(n) > 0.1
    strs = [None if rng.random() < 0.1 else "s%d" % rng.integers(0, 40) for _ in range(n)]
    sets = [None if rng.random() < 0.2 else ["t%d" % x for x in rng.integers(0, 9, size=rng.integers(1, 4))] for _ in range(n)]
    for gz in (False, True):
        for thr in (5000, 10):   # bucket-encoded and value-encoded variants
            root = str(tmp_path / ("t_%s_%d" % (gz, thr)))
            F.write_table(root, "tab", [{"v": ("int", vals, pop), "s": ("str", strs), "z": ("set", sets)}], gz=gz, threshold=thr)
            bdir = os.path.join(root, "tab", "block000000001")
            for fname, schema in (("int_v.db", G.saved_int_column), ("str_s.db", G.saved_str_column),
                                  ("set_z.db", G.saved_set_column), ("info.db", G.saved_column_info)):
                path = os.path.join(bdir, fname)
                raw = open(path + ".gz", "rb").read() if gz else open(path, "rb").read()
                if gz:
                    import gzip
                    raw = gzip.decompress(raw)
                want = G.decode(raw)
                got = _cxx_json(path)
                # the C++ reader renders maps as [[k, v], ...]
                def norm(x):
                    if isinstance(x, dict):
                        return {k: norm(v) for k, v in x.items()}
                    if isinstance(x, list):
                        if x and all(isinstance(p, list) and len(p) == 2 and not isinstance(p[0], (list, dict)) for p in x) \
                                and fname == "info.db":
                            return {str(p[0]): norm(p[1]) for p in x}
                        return [norm(v) for v in x]
                    return x
                assert norm(got) == norm(want), (fname, gz, thr)
            tinfo = _cxx_json(os.path.join(root, "tab", "info.db"))
            assert tinfo["Name"] == "tab" and sorted(k for k, _ in tinfo["KeyTable"]) == ["s", "v", "z"]


def test_cxx_reader_rejects_garbage(tmp_path):
    from sybil_amd import _native as N
    p = tmp_path / "bad.db"
    p.write_bytes(b"\x05\xff\x81\x03\x01")
    assert N.lib().sybl_debug_gob_to_json(str(p).encode()) is None
    assert b"gob" in N.lib().sybl_last_error()
    assert N.lib().sybl_debug_gob_to_json(str(tmp_path / "missing.db").encode()) is None


def test_encoded_results_sample_decodes():
    """tests/golden/encoded_results_sample.hex: two `-encode-results` streams written by sybl_result_encode on the
    GPU (tools/make_encoded_sample.py) for a ten-row table whose answers are known by hand -- decoded here with
    the same decoder that reads the reference's own golden gob files (interface values included)."""
    import os
    from tests import gobfmt
    lines = open(os.path.join(os.path.dirname(__file__), "golden", "encoded_results_sample.hex")).read().split()
    hist, times = (gobfmt.decode(bytes.fromhex(x)) for x in lines)
    qs = hist["QuerySpec"]
    assert qs["QueryParams"] == {"Groups": [{"Name": "browser"}], "Aggregations": [{"Op": "hist", "Name": "load", "HistType": "basic"}],
                                 "OrderBy": "$COUNT", "Limit": 100}
    res = qs["QueryResults"]
    assert res["MatchedCount"] == 10
    assert [(r["GroupByKey"], r["Count"]) for r in res["Sorted"]] == [("edge\t", 5), ("gecko\t", 3), ("webkit\t", 2)]
    edge = res["Results"]["edge\t"]
    assert edge["BinaryByKey"] == "\x00" * 8 and edge["Samples"] == 5
    iv = edge["Hists"]["load"]
    assert iv["@type"] == "*sybil.HistCompat"
    ci = iv["value"]["BasicHist"]["BasicHistCachedInfo"]
    # Info [0, 1000]: BucketSize 1, NumBuckets 1001 -> 1002 Values (hist_basic.go:34-70)
    assert (ci["NumBuckets"], ci["BucketSize"], len(ci["Values"]), ci["PercentileMode"]) == (1001, 1, 1002, True)
    assert [i for i, x in enumerate(ci["Values"]) if x] == [100, 200, 300, 500, 600]
    assert (ci["Count"], ci["Avg"], ci["Max"], ci["Info"]) == (5, 340.0, 1000, {"Max": 1000})
    total = res["Cumulative"]["Hists"]["load"]["value"]["BasicHist"]["BasicHistCachedInfo"]
    assert total["Count"] == 10 and total["Avg"] == 355.0 and sum(total["Values"]) == 10
    # time series: TimeResults[bucket][key] carries the hists, Results only Count / Samples (aggregate.go:156-183)
    tres = times["QuerySpec"]["QueryResults"]
    assert times["QuerySpec"]["QueryParams"]["TimeBucket"] == 3600
    assert sorted(tres["TimeResults"]) == [1699999200, 1700002800, 1700006400, 1700010000]
    assert {k: v["Count"] for k, v in tres["TimeResults"][1700006400].items()} == {"edge\t": 1, "gecko\t": 1, "webkit\t": 1}
    assert tres["TimeResults"][1699999200]["edge\t"]["Hists"]["load"]["value"]["BasicHist"]["BasicHistCachedInfo"]["Avg"] == 200.0
    assert "Hists" not in tres["Results"]["edge\t"] and tres["Results"]["edge\t"]["Count"] == 5


def _encode_column(kind, name, vals, pop=None, dict_strings=None):
    import ctypes as C
    from sybil_amd import _native as N
    vals = np.ascontiguousarray(vals, dtype=np.int64)
    popa = None if pop is None else np.ascontiguousarray(pop, dtype=np.uint8)
    n = C.c_int64(0)
    ds = [s.encode() for s in (dict_strings or [])]
    arr = (C.c_char_p * max(len(ds), 1))(*ds)
    p = N.lib().sybl_debug_encode_column(kind, name.encode(), vals.ctypes.data, None if popa is None else popa.ctypes.data,
                                         len(vals), arr, len(ds), C.byref(n))
    assert p, N.lib().sybl_last_error()
    return C.string_at(p, n.value)


def test_native_column_writer_matches_the_python_writer():
    """The C++ gob encoder behind sybl_table_save (csrc/writer.cpp, csrc/gobenc.h) against the Go-faithful Python
    writer (tests/sybil_fixture.py) -- byte for byte, for bucket- and value-encoded int columns with and without
    missing rows and for str columns (no GPU involved)."""
    from tests import sybil_fixture as F
    rng = np.random.default_rng(3)
    n = 20_000
    low = rng.integers(-5, 40, size=n)                      # <= 5000 distinct: bucket encoded
    high = rng.integers(-(1 << 45), 1 << 45, size=n)        # > 5000 distinct: value encoded, wide deltas
    pop = (rng.random(n) > 0.3).astype(np.uint8)
    for name, vals, p in (("low", low, None), ("low", low, pop), ("high", high, None), ("high", high, pop),
                          ("one", np.array([7]), None), ("none", np.array([1, 2, 3]), np.zeros(3, dtype=np.uint8)),
                          ("empty", np.zeros(0, dtype=np.int64), None)):
        want = F.int_column(name, np.asarray(vals, dtype=np.int64), p)
        got = _encode_column(1, name, vals, p)
        assert got == want, (name, len(got), len(want))
    vocab = ["host%04d" % i for i in range(6000)]
    few = rng.integers(0, 50, size=n)
    many = rng.integers(0, 6000, size=n)                    # > 5000 distinct strings: per-row Values
    spop = rng.random(n) > 0.2
    for name, ids, p in (("few", few, None), ("few", few, spop), ("many", many, None), ("many", many, spop)):
        strings = [vocab[i] if (p is None or p[k]) else None for k, i in enumerate(ids)]
        want = F.str_column(name, strings)
        got = _encode_column(2, name, ids, None if p is None else p.astype(np.uint8), vocab)
        assert got == want, (name, len(got), len(want))


def test_cxx_reader_survives_truncated_and_corrupted_column_files(tmp_path):
    """The loader reads files it did not write (a block the reference's digest left half-written is skipped, not fatal:
    table_query_test.go:11-158): every prefix of a column file's first bytes and a few thousand random byte flips must come
    back as a value or as an error -- never as a crash, a hang or an absurd allocation."""
    from sybil_amd import _native as N
    from tests import sybil_fixture as F
    rng = np.random.default_rng(0)
    n = 2000
    vals = rng.integers(-1000, 1 << 45, size=n)
    pop = rng.random(n) > 0.1
    strs = [None if rng.random() < 0.1 else "s%d" % rng.integers(0, 40) for _ in range(n)]
    sets = [None if rng.random() < 0.2 else ["t%d" % x for x in rng.integers(0, 9, size=rng.integers(1, 4))] for _ in range(n)]
    p = str(tmp_path / "x.db").encode()
    tried = 0
    for thr in (5000, 10):  # bucket-encoded and value-encoded
        root = str(tmp_path / ("t%d" % thr))
        F.write_table(root, "tab", [{"v": ("int", vals, pop), "s": ("str", strs), "z": ("set", sets)}], gz=False, threshold=thr)
        bdir = os.path.join(root, "tab", "block000000001")
        for fname in ("int_v.db", "str_s.db", "set_z.db", "info.db"):
            raw = open(os.path.join(bdir, fname), "rb").read()
            assert N.lib().sybl_debug_gob_to_json(os.path.join(bdir, fname).encode()) is not None
            cuts = sorted(set(list(range(0, min(len(raw), 200))) + rng.integers(0, len(raw), 60).tolist()))
            for c in cuts:
                open(p, "wb").write(raw[:c])
                N.lib().sybl_debug_gob_to_json(p)
                tried += 1
            for k in range(150):
                b = bytearray(raw)
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, min(len(b), 400) if k % 2 else len(b)))] = int(rng.integers(0, 256))
                open(p, "wb").write(bytes(b))
                N.lib().sybl_debug_gob_to_json(p)
                tried += 1
    assert tried > 2500


def _check_int_slices(tmp):
    """Int / uint slices of every encoded length through the C++ reader (Reader::ints: the 64-byte windows of ints_vbmi on a
    host that has AVX-512 VBMI, the scalar loops elsewhere and under SYBL_GOB_NO_VBMI): value-encoded `Values []int64` and
    the bins' `Records []uint32`, in the mixes that steer the loops -- runs of one-byte values, one/two/three-byte deltas at
    random, markers of five to eight data bytes, slices shorter than a window, values that straddle a window's end."""
    rng = np.random.default_rng(11)

    def ints(n, kind):
        if kind == "bytes":  # one-byte values only (zig-zag of -64..63)
            return [int(x) for x in rng.integers(-64, 64, size=n)]
        if kind == "mixed":  # every length, equally likely
            bits = rng.integers(1, 64, size=n)
            return [int(rng.integers(0, 1 << 62) >> (62 - b)) * (1 if rng.random() < 0.5 else -1) for b in bits]
        if kind == "deltas":  # what a value-encoded column holds: three data bytes, now and then two
            return [int(x) for x in rng.integers(-1_000_000, 1_000_000, size=n)]
        if kind == "extremes":
            pool = [0, -1, 1, 63, 64, -64, -65, 127, 128, 255, 256, 65535, 65536, (1 << 31) - 1, -(1 << 31), (1 << 55), (1 << 56) - 1, 1 << 56,
                    (1 << 62), (1 << 63) - 1, -(1 << 63), -(1 << 56), -(1 << 55) - 1]
            return [pool[int(i)] for i in rng.integers(0, len(pool), size=n)]
        if kind == "runs3":  # stretches of 1..20 values of three data bytes (the reader's shuffle path takes six or more) between others
            out = []
            while len(out) < n:
                out += [int(x) * (1 if rng.random() < 0.5 else -1) for x in rng.integers(1 << 15, 1 << 23, size=int(rng.integers(1, 21)))]
                out.append([5, -300, 1 << 40, -(1 << 62), 70000][int(rng.integers(0, 5))])
            return out[:n]
        runs, out = [], []  # "runs": long one-byte stretches broken by wide values
        while len(out) < n:
            out += [int(x) for x in rng.integers(-60, 60, size=int(rng.integers(1, 200)))]
            out += [int(rng.integers(-(1 << 40), 1 << 40)) for _ in range(int(rng.integers(1, 4)))]
        return out[:n]

    k = 0
    for kind in ("bytes", "mixed", "deltas", "extremes", "runs", "runs3"):
        for n in (0, 1, 7, 15, 16, 17, 63, 64, 65, 200, 1000, 5003):
            vals = ints(n, kind)
            col = {"Name": "v", "ValueEncoded": True, "Values": vals, "VERSION": 1}
            p = os.path.join(tmp, "v%d.db" % k)
            k += 1
            open(p, "wb").write(G.encode(G.saved_int_column(), col))
            got = _cxx_json(p)
            assert got.get("Values", []) == vals, (kind, n)
    # Records []uint32 (unsigned; ids below 65536 are at most two data bytes, the type allows four) inside bins of many sizes
    for trial in range(12):
        bins = []
        for b in range(int(rng.integers(1, 40))):
            m = int(rng.choice([0, 1, 2, 15, 16, 17, 64, 65, 130, 700]))
            top = int(rng.choice([100, 128, 300, 70000, (1 << 32) - 1]))
            bins.append({"Value": int(rng.integers(-(1 << 40), 1 << 40)), "Records": [int(x) for x in rng.integers(0, top, size=m, endpoint=True)]})
        col = {"Name": "b", "DeltaEncodedIDs": True, "BucketEncoded": True, "Bins": bins, "VERSION": 1}
        p = os.path.join(tmp, "b%d.db" % trial)
        open(p, "wb").write(G.encode(G.saved_int_column(), col))
        got = _cxx_json(p)
        want = [{kk: vv for kk, vv in b.items() if vv not in (0, [])} for b in bins]  # (gob omits zero fields)
        assert [{kk: vv for kk, vv in b.items() if vv not in (0, [])} for b in got["Bins"]] == want, trial


def test_cxx_reader_int_slices_of_every_length(tmp_path):
    _check_int_slices(str(tmp_path))


import pytest


@pytest.mark.parametrize("switches", [{"SYBL_GOB_NO_VBMI": "1"}, {"SYBL_DEBUG_GOB_NARROW": "1"},
                                      {"SYBL_DEBUG_GOB_NARROW": "1", "SYBL_GOB_NO_VBMI": "1"}],
                         ids=["scalar", "narrow", "narrow-scalar"])
def test_cxx_reader_int_slices_other_paths(tmp_path, switches):
    """The same check with the AVX-512 windows switched off (a host without VBMI runs that one twice, harmlessly) and with
    the narrow slices the loader asks for (uint16 records / int32 values where they fit, a second pass as int64 where one
    does not: the JSON is the same)."""
    import subprocess
    import sys
    env = dict(os.environ, **switches)
    code = "import sys; sys.path.insert(0, %r); from tests.test_gob import _check_int_slices; _check_int_slices(%r)" % (
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_varint_windows_against_the_checked_reader_on_random_streams(tmp_path):
    """tools/micro/gobints_fuzz.cpp: Reader::ints -- the AVX-512 VBMI windows where the host has them -- against the checked
    value-by-value reader on random byte strings (valid streams of the shapes column files have, marker soup, bytes that
    are no marker, damage), as int64 and as the narrow element types the loader asks for: the same values, the same
    position behind them, the same verdict, `misfit` set exactly when a value does not fit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gobints_fuzz")
    b = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "sybil_amd", "csrc"), "-I", os.path.join(root, "include"),
                        os.path.join(root, "tools", "micro", "gobints_fuzz.cpp"), "-lz", "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    for env in ({}, {"SYBL_GOB_NO_VBMI": "1"}):
        r = subprocess.run([exe, "60000"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout[-500:], r.stderr[-500:])
