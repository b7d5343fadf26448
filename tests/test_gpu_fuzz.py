"""GPU: seeded random query shapes over random small tables, every planner path (role-specialised /
generic / LDS window / global atomics / partitioned histograms) against the CPU oracle."""
import numpy as np
import pytest

import sybil_amd
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sybil_amd
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _random_table(rng, n):
    cols = {}
    cols["g_small"] = rng.integers(0, int(rng.integers(1, 40)), size=n)
    cols["g_neg"] = rng.integers(-7, 9, size=n)
    cols["g_wide"] = rng.integers(0, int(rng.integers(100, 5000)), size=n)
    cols["f1"] = rng.integers(0, 1000, size=n)
    cols["f2"] = rng.integers(-1000, 1000, size=n)
    cols["v_pos"] = rng.integers(0, int(rng.integers(10, 2_000_000)), size=n)
    cols["v_any"] = rng.integers(-50_000, 50_000, size=n)
    cols["t"] = np.sort(1_700_000_000 + rng.integers(0, int(rng.integers(1, 40)) * 3600, size=n))
    cols["w"] = rng.integers(1, 5, size=n)
    pops = {}
    for name in ("g_neg", "f2", "v_any"):
        if rng.random() < 0.5:
            pops[name] = (rng.random(n) > 0.2).astype(np.uint8)
    return {k: v.astype(np.int64) for k, v in cols.items()}, pops


def _random_query(rng, info):
    q = {}
    ops = ["gt", "lt", "eq", "neq"]
    nf = int(rng.integers(0, 4))
    filters = []
    for _ in range(nf):
        c = str(rng.choice(["f1", "f2", "g_small", "v_any"]))
        op = str(rng.choice(ops))
        lo, hi = info[c]
        filters.append((c, op, int(rng.integers(lo - 5, hi + 5))))
    q["filters"] = filters
    q["groups"] = [str(x) for x in rng.choice(["g_small", "g_neg", "g_wide"], size=int(rng.integers(0, 3)), replace=False)]
    q["aggs"] = [str(x) for x in rng.choice(["v_pos", "v_any", "f1"], size=int(rng.integers(0, 3)), replace=False)]
    q["op"] = str(rng.choice(["avg", "hist"]))
    if q["op"] == "hist":
        q["want_percentiles"] = bool(rng.random() < 0.6)
        if rng.random() < 0.3:
            q["hist_bucket"] = int(rng.integers(1, 5000))
    if rng.random() < 0.3:
        q["time_col"] = "t"
        q["time_bucket"] = int(rng.choice([600, 3600, 86400]))
    if rng.random() < 0.2:
        q["weight_col"] = "w"
    if rng.random() < 0.3:
        q["block_skip"] = True
    return q


@pytest.mark.parametrize("seed", range(24))
def test_random_queries(ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 60_000))
    block_rows = int(rng.choice([997, 4096, 65536]))
    cols, pops = _random_table(rng, n)
    names = list(cols)
    # IntInfo: sometimes exact, sometimes narrower than the data (rejects, outliers)
    info = {}
    for c in names:
        lo, hi = int(cols[c].min()), int(cols[c].max())
        if rng.random() < 0.3 and hi - lo > 10:
            lo, hi = lo + (hi - lo) // 10, hi - (hi - lo) // 3
        info[c] = (lo, hi)
    tb = ctx.create_table("fuzz")
    for c in names:
        tb.add_column(c, "int", info[c][0], info[c][1])
    for r0 in range(0, n, block_rows):
        r1 = min(r0 + block_rows, n)
        tb.append_block(r1 - r0, {c: ((cols[c][r0:r1], pops[c][r0:r1]) if c in pops else cols[c][r0:r1]) for c in names})
    ocols = [{"type": "int", "data": cols[c], **({"populated": pops[c]} if c in pops else {})} for c in names]
    seen = set()
    for k in range(6):
        q = _random_query(rng, info)
        try:
            query = tb.query(**q)
        except sybil_amd.SyblError as e:
            # documented limits of the direct-mapped layout
            assert "direct-mapped cells" in str(e) or "histogram budget" in str(e) or "exceeds 2^27" in str(e), str(e)
            continue
        gres = query.run()
        seen.add(query.stats()["strategy"])
        ores = oracle.run_query(ocols, block_rows=block_rows, **parity.oracle_query_kwargs(names, info, q))
        try:
            parity.compare(gres, ores, op=q["op"], full=q.get("want_percentiles", True) and q["op"] == "hist",
                           n_aggs=len(q["aggs"]), time_mode=bool(q.get("time_col")))
        except AssertionError as e:
            raise AssertionError("seed %d query %d %r strategy %d: %s" % (seed, k, q, query.stats()["strategy"], e))
        gres.free()
        query.free()
    tb.free()
