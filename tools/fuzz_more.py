#!/usr/bin/env python3
"""Runs the seeded random-query parity tests of tests/test_gpu_fuzz.py for seeds beyond the ones the suite
pins: fuzz_more.py <first> <last>, or fuzz_more.py <seed>,<seed>,...  Prints the failing seeds (none expected)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sybil_amd
from oracle import oracle as orc
from tests import test_gpu_fuzz as T



class _Env:
    """the two methods of pytest's monkeypatch the tests use"""
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=True):
        os.environ.pop(k, None)


if "," in sys.argv[1] or len(sys.argv) == 2:
    seeds = [int(x) for x in sys.argv[1].split(",") if x]
else:
    seeds = list(range(int(sys.argv[1]), int(sys.argv[2])))
ctx = sybil_amd.Context(0)
bad = []
for seed in seeds:
    # every fourth seed sends the grouped queries through the hash table (strategy 7), half of those without LDS staging
    os.environ.pop("SYBL_FORCE_HASH", None)
    os.environ.pop("SYBL_NO_HASH_LDS", None)
    if seed % 4 == 0:
        os.environ["SYBL_FORCE_HASH"] = "1"
        if seed % 8 == 0:
            os.environ["SYBL_NO_HASH_LDS"] = "1"
    for fn in (T.test_random_queries, T.test_random_queries_many_tiles_per_workgroup, T.test_random_queries_with_strings_and_sets,
               T.test_random_queries_round5_shapes):
        if fn is T.test_random_queries_many_tiles_per_workgroup and seed % 5:
            continue  # (bigger tables: every fifth seed)
        try:
            if fn is T.test_random_queries_round5_shapes:
                fn(ctx, orc, seed)
            elif fn is not T.test_random_queries_with_strings_and_sets:
                fn(ctx, orc, seed, _Env())
            else:
                fn(ctx, orc, seed)
        except Exception as e:  # noqa
            bad.append((seed, fn.__name__, str(e)[:300]))
            traceback.print_exc(limit=2)
print("seeds %d..%d (%d): %d failures" % (seeds[0], seeds[-1], len(seeds), len(bad)))
for x in bad:
    print(x)
