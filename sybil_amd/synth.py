"""The deterministic synthetic table and the five BASELINE.json workloads.

Column recipe: SURVEY.md 8(d) / BASELINE.md section 3 (seed 20241022, counter-based
splitmix64, every row populated, dense int64 storage).  The generator itself lives in
csrc/kernels.hip:k_synth and is restated in oracle/sybil_oracle.c:orc_synth_fill.
"""
SEED = 20241022
UNIFORM, TIME, BELL = 0, 1, 2

# name -> (kind, col_index, a, b, info_min, info_max)
COLUMNS = {
    "c00": (TIME, 0, 1_700_000_000, 2_592_000, 1_700_000_000, 1_702_591_999),  # time: 30 days
    "c01": (UNIFORM, 1, 0, 16, 0, 15),            # low-card group
    "c02": (UNIFORM, 2, 0, 64, 0, 63),            # 2nd group
    "c03": (UNIFORM, 3, 0, 65536, 0, 65535),      # high-card group
    "c04": (UNIFORM, 4, 0, 1000, 0, 999),         # filter columns
    "c05": (UNIFORM, 5, 0, 1000, 0, 999),
    "c06": (UNIFORM, 6, 0, 1000, 0, 999),
    "c07": (UNIFORM, 7, 0, 1_000_000, 0, 999_999),  # agg A: BucketSize 999, 1002 buckets
    "c08": (BELL, 8, 0, 250_000, 0, 999_996),       # agg B: bell-shaped
    "c09": (UNIFORM, 9, 0, 500, 0, 499),            # str-like group (global dictionary ids)
}
for _i in range(10, 32):                            # filler: never referenced, never resident
    COLUMNS["c%02d" % _i] = (UNIFORM, _i, 0, 1 << 31, 0, (1 << 31) - 1)


def synth_cols(names):
    out = []
    for n in names:
        kind, idx, a, b, imin, imax = COLUMNS[n]
        out.append({"name": n, "kind": kind, "col_index": idx, "a": a, "b": b, "info_min": imin, "info_max": imax})
    return out


_RANGE3 = [("c04", "gt", 99), ("c04", "lt", 900), ("c05", "gt", 99), ("c05", "lt", 900),
           ("c06", "gt", 99), ("c06", "lt", 900)]

# BASELINE.json configs -> reference CLI flags -> query kwargs
WORKLOADS = {
    "cfg1_count_range": {
        "rows": 10_000_000, "table_cols": 8, "columns": ["c04"],
        "flags": "-int-filter c04:gt:99,c04:lt:900",
        "query": dict(filters=[("c04", "gt", 99), ("c04", "lt", 900)]),
    },
    "cfg2_group1_avg2": {
        "rows": 100_000_000, "table_cols": 16, "columns": ["c01", "c07", "c08"],
        "flags": "-group c01 -int c07,c08 -op avg",
        "query": dict(groups=["c01"], aggs=["c07", "c08"], op="avg"),
    },
    "cfg3_filter3_group2_stddev": {
        "rows": 1_000_000_000, "table_cols": 32,
        "columns": ["c04", "c05", "c06", "c01", "c02", "c07", "c08"],
        "flags": "-int-filter c04:gt:99,c04:lt:900,c05:gt:99,c05:lt:900,c06:gt:99,c06:lt:900 "
                 "-group c01,c02 -int c07,c08 -op hist",
        # graded outputs are count/sum/avg/stddev: moments mode (no per-bucket arrays)
        "query": dict(filters=_RANGE3, groups=["c01", "c02"], aggs=["c07", "c08"], op="hist", want_percentiles=False),
    },
    "cfg4_hist_highcard": {
        "rows": 1_000_000_000, "table_cols": 32, "columns": ["c03", "c07"],
        "flags": "-group c03 -int c07 -op hist",
        "query": dict(groups=["c03"], aggs=["c07"], op="hist", want_percentiles=True),
    },
    "cfg5_time_rollup": {
        "rows": 1_000_000_000, "table_cols": 32, "columns": ["c00", "c09", "c07"],
        "flags": "-time -time-col c00 -time-bucket 3600 -group c09 -int c07 -op avg",
        "query": dict(groups=["c09"], aggs=["c07"], op="avg", time_col="c00", time_bucket=3600),
    },
}


def shard(total_rows, rank, nranks, block_rows=65536):
    """Contiguous block ranges per rank: returns (row0, nrows)."""
    nblocks = (total_rows + block_rows - 1) // block_rows
    b0 = nblocks * rank // nranks
    b1 = nblocks * (rank + 1) // nranks
    row0 = b0 * block_rows
    row1 = min(b1 * block_rows, total_rows)
    return row0, row1 - row0
