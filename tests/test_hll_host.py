"""The count-distinct building blocks of the engine (csrc/hll.h: the functions k_scan_distinct calls, built for the host)
against the oracle's restatement -- no GPU needed.  The kernel around them is covered by tests/test_gpu_distinct.py."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import oracle as orc
from sybil_amd import _native as N


def _regs():
    return np.zeros(16384, dtype=np.uint8)


@pytest.mark.parametrize("ncols", [1, 2, 3, 4, 5, 6, 7, 8])
def test_int_path_registers_equal_the_oracles(ncols):
    rng = np.random.default_rng(100 + ncols)
    n = 20_000
    vals = rng.integers(-1 << 62, 1 << 62, size=(n, ncols), dtype=np.int64)
    vals[: n // 2] = rng.integers(-5, 5, size=(n // 2, ncols))      # small values and duplicates too
    pop = (rng.random((n, ncols)) > 0.15).astype(np.uint8)
    regs = _regs()
    N.check(N.lib().sybl_debug_hll_ints(vals.ctypes.data, pop.ctypes.data, n, ncols, regs.ctypes.data))
    want = orc.LogLogBeta()
    masked = np.where(pop != 0, vals, -1)                            # MISSING_VALUE = all ones (aggregate.go:215)
    for row in masked:
        want.add(struct.pack("<%dq" % ncols, *row.tolist()))
    assert np.array_equal(regs, want.registers)
    assert N.lib().sybl_debug_hll_cardinality(regs.ctypes.data) == want.cardinality()


def test_byte_hash_equals_the_oracles_over_every_length():
    rng = np.random.default_rng(5)
    for n in range(0, 100):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert N.lib().sybl_debug_hll_bytes(b, n, None) == orc.metro64(b, 1337), n


def test_str_path_registers_equal_the_oracles():
    strs = [("agent-%d" % i).encode() + b"\t" for i in range(5000)] + [b"\t"]
    regs = _regs()
    want = orc.LogLogBeta()
    for s in strs:
        N.lib().sybl_debug_hll_bytes(s, len(s), regs.ctypes.data)
        want.add(s)
    assert np.array_equal(regs, want.registers)
    assert N.lib().sybl_debug_hll_cardinality(regs.ctypes.data) == want.cardinality()


def test_cardinality_of_arbitrary_registers():
    rng = np.random.default_rng(9)
    for fill in (0.0, 0.01, 0.5, 1.0):
        regs = np.where(rng.random(16384) < fill, rng.integers(1, 52, 16384), 0).astype(np.uint8)
        assert N.lib().sybl_debug_hll_cardinality(regs.ctypes.data) == orc.LogLogBeta(regs).cardinality()
