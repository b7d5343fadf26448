#!/usr/bin/env python3
"""Two fixture blocks with the column mix of the bench's 7-column table (five bucket-encoded columns of 1000 / 1000 / 1000 /
16 / 64 values, two value-encoded ones), in the reference's on-disk format, for tools/micro/loader_parse.cpp:
    python tools/micro/loader_parse_blocks.py <dir>      ->  <dir>/t/block00000000{1,2}/"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import sybil_fixture as F

rng = np.random.default_rng(7)
n = 65536
blocks = []
for b in range(2):
    bell = np.clip((rng.normal(0.5, 0.12, n) * 250000).astype(np.int64) * 4, 0, 999996)
    blocks.append({"c04": ("int", rng.integers(0, 1000, n)), "c05": ("int", rng.integers(0, 1000, n)), "c06": ("int", rng.integers(0, 1000, n)),
                   "c01": ("int", rng.integers(0, 16, n)), "c02": ("int", rng.integers(0, 64, n)),
                   "c07": ("int", rng.integers(0, 1000000, n)), "c08": ("int", bell)})
F.write_table(sys.argv[1], "t", blocks)
