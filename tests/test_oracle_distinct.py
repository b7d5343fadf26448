"""Count-distinct in the oracle (aggregate.go:205-243, query_spec.go:87,100,180-188): the restated MetroHash64 is
pinned on the hash's published test vectors; the LogLog-Beta sketch (github.com/logv/loglogbeta -- not in the
reference tree, no pinned version: PARITY UNPINNED) is checked for its defining properties."""
import struct

import numpy as np
import pytest

from oracle import oracle as orc


def test_metrohash64_published_vectors():
    # metrohash64.cpp: MetroHash64::test_string, test_seed_0, test_seed_1 (63 bytes: every tail branch is taken)
    key = b"012345678901234567890123456789012345678901234567890123456789012"
    assert len(key) == 63
    assert struct.pack("<Q", orc.metro64(key, 0)) == bytes([0x6B, 0x75, 0x3D, 0xAE, 0x06, 0x70, 0x4B, 0xAD])
    assert struct.pack("<Q", orc.metro64(key, 1)) == bytes([0x3B, 0x0D, 0x48, 0x1C, 0xF4, 0xB9, 0xB8, 0xDF])


def test_metrohash64_matches_a_plain_python_restatement():
    """An independent restatement (Python big ints) over every length 0..80: guards the C tail handling."""
    M = (1 << 64) - 1
    k0, k1, k2, k3 = 0xD6D018F5, 0xA2AA033B, 0x62992FC1, 0x30BC5B29

    def rotr(v, k):
        return ((v >> k) | (v << (64 - k))) & M

    def rd(b, i, n):
        return int.from_bytes(b[i:i + n], "little")

    def metro(b, seed):
        n, i = len(b), 0
        h = ((seed + k2) * k0) & M
        if n >= 32:
            v = [h, h, h, h]
            while True:
                v[0] = (v[0] + rd(b, i, 8) * k0) & M; i += 8; v[0] = (rotr(v[0], 29) + v[2]) & M
                v[1] = (v[1] + rd(b, i, 8) * k1) & M; i += 8; v[1] = (rotr(v[1], 29) + v[3]) & M
                v[2] = (v[2] + rd(b, i, 8) * k2) & M; i += 8; v[2] = (rotr(v[2], 29) + v[0]) & M
                v[3] = (v[3] + rd(b, i, 8) * k3) & M; i += 8; v[3] = (rotr(v[3], 29) + v[1]) & M
                if i > n - 32:
                    break
            v[2] ^= (rotr((((v[0] + v[3]) * k0) + v[1]) & M, 37) * k1) & M
            v[3] ^= (rotr((((v[1] + v[2]) * k1) + v[0]) & M, 37) * k0) & M
            v[0] ^= (rotr((((v[0] + v[2]) * k0) + v[3]) & M, 37) * k1) & M
            v[1] ^= (rotr((((v[1] + v[3]) * k1) + v[2]) & M, 37) * k0) & M
            h = (h + (v[0] ^ v[1])) & M
        if n - i >= 16:
            v0 = (h + rd(b, i, 8) * k2) & M; i += 8; v0 = (rotr(v0, 29) * k3) & M
            v1 = (h + rd(b, i, 8) * k2) & M; i += 8; v1 = (rotr(v1, 29) * k3) & M
            v0 ^= (rotr((v0 * k0) & M, 21) + v1) & M
            v1 ^= (rotr((v1 * k3) & M, 21) + v0) & M
            h = (h + v1) & M
        for width, rot in ((8, 55), (4, 26), (2, 48), (1, 37)):
            if n - i >= width:
                h = (h + rd(b, i, width) * k3) & M; i += width
                h ^= (rotr(h, rot) * k1) & M
        h ^= rotr(h, 28)
        h = (h * k0) & M
        h ^= rotr(h, 29)
        return h

    rng = np.random.default_rng(7)
    for n in range(0, 81):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1337):
            assert orc.metro64(b, seed) == metro(b, seed), (n, seed)


def test_add_hash_register_and_rank():
    s = orc.LogLogBeta()
    s.add_hash(0)                         # register 0; the 50 bits below are zero: rank 51 (the guard bits stop the count)
    assert s.registers[0] == 51 and s.registers[1:].max() == 0
    s.add_hash((5 << 50) | (1 << 49))     # register 5, first bit below the index set: rank 1
    assert s.registers[5] == 1
    s.add_hash((5 << 50) | (1 << 40))     # nine leading zeros: rank 10, replaces the 1
    assert s.registers[5] == 10
    s.add_hash((5 << 50) | (1 << 45))     # rank 5 does not lower it
    assert s.registers[5] == 10
    s.add_hash(((1 << 14) - 1) << 50)     # last register
    assert s.registers[(1 << 14) - 1] == 51


def test_empty_and_small_cardinalities():
    assert orc.LogLogBeta().cardinality() == 0
    s = orc.LogLogBeta()
    for i in range(10):
        s.add(struct.pack("<q", i))
    assert s.cardinality() == 10          # the beta correction is what makes small counts come out right
    for i in range(10):                   # duplicates change nothing
        s.add(struct.pack("<q", i))
    assert s.cardinality() == 10


@pytest.mark.parametrize("n", [1000, 50_000, 1_000_000])
def test_estimate_within_the_sketch_error(n):
    # standard error 1.04 / sqrt(16384) = 0.8 %: allow 3 %
    s = orc.LogLogBeta()
    hashes = np.random.default_rng(n).integers(0, 1 << 63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    for x in np.unique(hashes).tolist():
        s.add_hash(int(x))
    true = len(np.unique(hashes))
    assert abs(s.cardinality() - true) <= 0.03 * true, (s.cardinality(), true)


def test_merge_is_the_sketch_of_the_union():
    a, b, u = orc.LogLogBeta(), orc.LogLogBeta(), orc.LogLogBeta()
    for i in range(0, 30_000):
        a.add(struct.pack("<q", i))
        u.add(struct.pack("<q", i))
    for i in range(20_000, 60_000):
        b.add(struct.pack("<q", i))
        u.add(struct.pack("<q", i))
    a.merge(b)
    assert np.array_equal(a.registers, u.registers)
    assert a.cardinality() == u.cardinality()


def _cols(n, seed):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, n)
    user = rng.integers(0, 5000, n)
    t = np.sort(rng.integers(1_700_000_000, 1_700_000_000 + 3 * 3600, n))
    pop = (rng.random(n) > 0.1).astype(np.uint8)
    return g, user, t, pop


def test_query_int_fast_path_matches_a_direct_sketch():
    n = 200_000
    g, user, t, pop = _cols(n, 3)
    cols = [{"type": "int", "data": g}, {"type": "int", "data": user, "populated": pop}]
    res = orc.run_query(cols, groups=[0], distincts=[1], block_rows=65536, n_threads=4, want_registers=True)
    assert res["matched"] == n
    for r in res["results"]:
        k = r["key_vals"][0]
        s = orc.LogLogBeta()
        sel = g == k
        vals = np.where(pop[sel] != 0, user[sel], -1)  # unpopulated: MISSING_VALUE = all ones
        for v in np.unique(vals).tolist():
            s.add(struct.pack("<q", v))
        assert np.array_equal(r["registers"], s.registers)
        assert r["distinct"] == s.cardinality()
        true = len(np.unique(vals))
        assert abs(r["distinct"] - true) <= 0.03 * true
    # Cumulative: the union over every group (query_spec.go:180-188, aggregate.go:431-434)
    s = orc.LogLogBeta()
    for v in np.unique(np.where(pop != 0, user, -1)).tolist():
        s.add(struct.pack("<q", v))
    assert np.array_equal(res["cumulative"]["registers"], s.registers)


def test_query_blocks_and_threads_do_not_change_the_sketch():
    n = 150_000
    g, user, t, pop = _cols(n, 5)
    cols = [{"type": "int", "data": g}, {"type": "int", "data": user}]
    a = orc.run_query(cols, groups=[0], distincts=[1], block_rows=1 << 20, n_threads=1, want_registers=True)
    b = orc.run_query(cols, groups=[0], distincts=[1], block_rows=4096, n_threads=8, want_registers=True)
    for ra, rb in zip(a["results"], b["results"]):
        assert ra["key"] == rb["key"] and np.array_equal(ra["registers"], rb["registers"]) and ra["distinct"] == rb["distinct"]


def test_query_two_int_columns_hash_sixteen_bytes():
    n = 50_000
    g, user, t, pop = _cols(n, 9)
    cols = [{"type": "int", "data": g}, {"type": "int", "data": user}]
    res = orc.run_query(cols, distincts=[0, 1], want_registers=True)
    s = orc.LogLogBeta()
    for a, b in set(zip(g.tolist(), user.tolist())):
        s.add(struct.pack("<qq", a, b))
    assert np.array_equal(res["results"][0]["registers"], s.registers)


def test_query_str_slow_path_hashes_strings_with_delimiters():
    n = 20_000
    rng = np.random.default_rng(11)
    ids = rng.integers(0, 300, n).astype(np.int32)
    num = rng.integers(-50, 50, n)
    pop = (rng.random(n) > 0.2).astype(np.uint8)
    strs = ["agent-%03d" % i for i in range(300)]
    cols = [{"type": "str", "data": ids}, {"type": "int", "data": num, "populated": pop}]
    res = orc.run_query(cols, distincts=[0, 1], distinct_dicts={0: strs}, want_registers=True)
    s = orc.LogLogBeta()
    for i, v, p in set(zip(ids.tolist(), num.tolist(), pop.tolist())):
        s.add((strs[i] + "\t" + (str(v) if p else "") + "\t").encode())  # aggregate.go:226-236
    assert np.array_equal(res["results"][0]["registers"], s.registers)


def test_time_series_sketches_live_in_the_time_results():
    n = 100_000
    g, user, t, pop = _cols(n, 13)
    cols = [{"type": "int", "data": g}, {"type": "int", "data": user}, {"type": "int", "data": t}]
    res = orc.run_query(cols, groups=[0], distincts=[1], time_col=2, time_bucket=3600, want_registers=True)
    # aggregate.go:146-183: the all-time Results only count; the (bucket, group) results carry the sketches
    assert all(r["distinct"] == 0 for r in res["results"])
    seen = 0
    for r in res["time_results"]:
        sel = (g == r["key_vals"][0]) & ((t // 3600) * 3600 == r["time_bucket"])
        true = len(np.unique(user[sel]))
        assert abs(r["distinct"] - true) <= max(0.03 * true, 2)
        seen += 1
    assert seen >= 8
