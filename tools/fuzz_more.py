#!/usr/bin/env python3
"""Runs the seeded random-query parity tests of tests/test_gpu_fuzz.py for seeds beyond the ones the suite
pins: fuzz_more.py <first> <last>.  Prints the failing seeds (none expected)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from oracle import oracle as orc
from tests import test_gpu_fuzz as T

a, b = int(sys.argv[1]), int(sys.argv[2])
ctx = sybil_amd.Context(0)
bad = []
for seed in range(a, b):
    for fn in (T.test_random_queries, T.test_random_queries_with_strings_and_sets):
        try:
            fn.__wrapped__(ctx, orc, seed) if hasattr(fn, "__wrapped__") else fn(ctx, orc, seed)
        except Exception as e:  # noqa
            bad.append((seed, fn.__name__, str(e)[:300]))
            traceback.print_exc(limit=2)
print("seeds %d..%d: %d failures" % (a, b - 1, len(bad)))
for x in bad:
    print(x)
