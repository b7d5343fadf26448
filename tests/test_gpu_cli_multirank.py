"""GPU, N > 1 from the COMMAND LINE: `sybil-gpu-query ... -gpu-rank i -gpu-ranks n -gpu-id-file f`, one process per rank, no
Python and no collective runtime on the host side -- the reference's only distributed mechanism is driven from its CLI too
(src/lib/node_aggregator.go:147-177, src/cmd/cmd_aggregate.go:10, scripts/basic_aggregation_test.sh:12-21).  The ranks load
their block ranges from disk, agree inside the library (sybl_table_agree), merge (sybl_query_allreduce) and rank 0 prints.

World 2 / 4 / 8 share ONE device through the test-only RCCL stand-in in LD_PRELOAD (tests/rccl_standin/: it pins the
protocol, not the transport).  Rank 0's stdout must be BYTE-EQUAL to the one-process CLI's in all three output formats (text,
-json, -encode-results), the other ranks print nothing, and the -json content is held against the CPU oracle over the
table's logical rows."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import loaded_oracle as LO
from tests import sybil_fixture as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "sybil_amd", "sybil-gpu-query")
STANDIN = os.path.join(HERE, "rccl_standin", "librccl_standin.so")

# (CLI arguments, the same query for the oracle or None)
QUERIES = {
    "str_key_text": (["-group", "name", "-int", "age,time"], None),
    "str_key_json": (["-group", "name", "-int", "age,time", "-json", "-limit", "1000"], dict(groups=["name"], aggs=["age", "time"], op="avg")),
    "sparse_key_json": (["-group", "big", "-int", "age", "-json", "-limit", "100000"], dict(groups=["big"], aggs=["age"], op="avg")),
    "set_filter_hist_json": (["-group", "name", "-int", "age", "-op", "hist", "-set-filter", "tags:in:tag3", "-json", "-limit", "1000"],
                             dict(filters=[("tags", "in", "tag3")], groups=["name"], aggs=["age"], op="hist")),
    "regex_text": (["-group", "age", "-int", "big", "-str-filter", "name:re:user[1-3].*", "-sort", "big"], None),
    "timeseries_text": (["-group", "age", "-int", "time", "-op", "hist", "-time", "-time-col", "time", "-time-bucket", "7200"], None),
    "timeseries_json": (["-group", "age", "-int", "age", "-time", "-time-col", "time", "-time-bucket", "7200", "-json"], None),
    "hashed_json": (["-group", "big,time", "-int", "age", "-json", "-limit", "50"], None),
    "encode": (["-group", "name", "-int", "age", "-op", "hist", "-encode-results"], None),
    "encode_timeseries": (["-group", "age", "-int", "big", "-time", "-time-col", "time", "-time-bucket", "7200", "-encode-results"], None),
    "no_groups": (["-int", "age,big", "-op", "hist"], None),
    "distinct": (["-group", "age", "-distinct", "time", "-json"], None),
}


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    from tests.test_gpu_loader import _make_blocks
    root = str(tmp_path_factory.mktemp("climr"))
    blocks, logical = _make_blocks(9, 3000, seed=55, ragged=True)
    F.write_table(root, "events", blocks, threshold=8, int_info={"big": LO.INFO_BIG})
    return root, logical


def _one(db_root, args):
    p = subprocess.run([CLI, "-dir", db_root, "-table", "events"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, TZ="UTC"), timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def _world(db_root, args, world, work, tag):
    if not os.path.exists(STANDIN):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STANDIN)])
    env = dict(os.environ, TZ="UTC", HSA_ENABLE_IPC_MODE_LEGACY="0", LD_PRELOAD=STANDIN, SYBL_CLI_BACKTRACE="1",
               SYBL_STANDIN_TIMEOUT_S=os.environ.get("SYBL_STANDIN_TIMEOUT_S", "120"))
    idf = os.path.join(work, "id_%s_%d" % (tag, world))
    procs = [subprocess.Popen([CLI, "-dir", db_root, "-table", "events"] + args + ["-gpu-rank", str(r), "-gpu-ranks", str(world), "-gpu-id-file", idf],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [0] * world, (tag, b"\n".join(e for _, e in outs).decode(errors="replace")[-6000:])
    return [o for o, _ in outs]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cli_ranks_print_what_one_process_prints(db, tmp_path, world, oracle):
    root, logical = db
    lo = LO.LoadedOracle(oracle, logical, 8)
    for name, (args, oq) in QUERIES.items():
        if world == 8 and name not in ("str_key_json", "sparse_key_json", "hashed_json", "encode", "timeseries_text"):
            continue  # (world 8 has ranks with a single block -- and, 9 blocks over 8 ranks, none without: the shapes that matter there)
        want = _one(root, args)
        outs = _world(root, args, world, str(tmp_path), name)
        assert outs[0] == want, (name, world, outs[0][:400], want[:400])
        assert all(o == b"" for o in outs[1:]), (name, "only rank 0 prints")
        assert len(want) > 0, name
        if oq is not None and world == 2:
            _check_json(name, json.loads(outs[0]), lo, oq)


def _check_json(name, rows, lo, q):
    """-json rows (printer.go:109-181) against the oracle: Count / Samples per group, avg or the percentiles per aggregation."""
    ores = lo.run(q)
    op = q.get("op", "avg")
    omap = {lo.key_string(q, r["key_vals"]): r for r in ores["results"]}
    assert len(rows) == len(omap), name
    for r in rows:
        key = "".join(r[g] + "\t" for g in q["groups"])  # (group values are the parts of GroupByKey, as strings: printer.go:127-131)
        o = omap[key]
        assert r["Count"] == o["count"] and r["Samples"] == o["samples"], (name, key)
        for a, col in enumerate(q["aggs"]):
            oh = o["hists"][a]
            if not oh["present"]:
                continue
            if op == "hist":
                assert r[col]["percentiles"] == [int(x) for x in oh["percentiles"]], (name, key, col)
                assert r[col]["samples"] == oh["count"]  # ("samples" = TotalCount(), printer.go:123)
            else:
                assert r[col] == pytest.approx(oh["sum_exact"] / oh["count"], rel=1e-6), (name, key, col)


def test_ranks_without_a_block(tmp_path, oracle):
    """Three blocks over four ranks: one rank opens nothing -- no rows, no extrema, empty dictionaries -- and still takes part in
    every collective of the agreement and the merge; the output is the one-process CLI's."""
    from tests.test_gpu_loader import _make_blocks
    root = str(tmp_path / "db3")
    blocks, _ = _make_blocks(3, 2000, seed=5, ragged=True)
    F.write_table(root, "events", blocks, threshold=8, int_info={"big": LO.INFO_BIG})
    for name in ("str_key_json", "sparse_key_json", "hashed_json", "timeseries_text", "set_filter_hist_json"):
        args = QUERIES[name][0]
        want = _one(root, args)
        outs = _world(root, args, 4, str(tmp_path), "e" + name)
        assert outs[0] == want and len(want) > 0, (name, outs[0][:300], want[:300])


def test_cli_rank_arguments_are_checked(db):
    root, _ = db
    for extra in (["-gpu-ranks", "2"], ["-gpu-rank", "2", "-gpu-ranks", "2", "-gpu-id-file", "/tmp/x"], ["-gpu-rank", "-1", "-gpu-id-file", "/tmp/x"]):
        p = subprocess.run([CLI, "-dir", root, "-table", "events", "-group", "age"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == 2 and b"-gpu-rank" in p.stderr, (extra, p.stderr)


def test_layout_mismatch_is_an_error_on_every_rank(db, tmp_path):
    """Ranks that skipped the agreement hold different layouts (their own extrema): the first sybl_query_allreduce says so on
    every rank instead of summing unrelated words (SYBL_CLI_SKIP_AGREE: a test switch of the CLI)."""
    root, _ = db
    env = dict(os.environ, TZ="UTC", HSA_ENABLE_IPC_MODE_LEGACY="0", LD_PRELOAD=STANDIN, SYBL_STANDIN_TIMEOUT_S="120", SYBL_CLI_SKIP_AGREE="1")
    idf = str(tmp_path / "id")
    procs = [subprocess.Popen([CLI, "-dir", root, "-table", "events", "-group", "time", "-int", "age", "-gpu-rank", str(r), "-gpu-ranks", "2", "-gpu-id-file", idf],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [1, 1], outs
    assert all(b"partial tables differ across ranks" in e for _, e in outs), outs


@pytest.mark.parametrize("world,n_blocks", [(2, 9), (4, 9), (4, 3)])
def test_limit_pushed_into_the_scan_across_ranks(tmp_path, world, n_blocks):
    """A printer's $COUNT-sorted histogram query with -limit (strategy 8, csrc/pushdown.hip) on N ranks: the groups' counts are
    all-reduced between the two passes, so every rank prints -- and fills -- the same cells; the merge is the limit-aware one.
    Output byte-equal to the one-process CLI's in text and -json; (4, 3): three blocks over four ranks -- the rank without rows
    cannot plan the pushed-down scan, the ranks find out at their first scan (one MIN all-reduce) and all take the full path."""
    rng = np.random.default_rng(7)
    root = str(tmp_path / "dbp")
    # (the shape the pushdown takes, tests/test_gpu_pushdown.py: a few thousand groups, values that cannot be outliers)
    blocks = []
    for b in range(n_blocks):
        n = 40_000 - 1_000 * b
        blocks.append({"k": ("int", rng.integers(0, 3001, n).astype(np.int64)), "v": ("int", rng.integers(0, 1000, n).astype(np.int64)),
                       "w": ("int", rng.integers(0, 1000, n).astype(np.int64))})
    blocks[0]["k"][1][:3001] = np.arange(3001)   # (every key and both ends of the values somewhere: the bounds are the full ranges)
    blocks[0]["v"][1][:2] = [0, 999]
    blocks[0]["w"][1][:2] = [0, 999]
    F.write_table(root, "events", blocks, threshold=8)
    for tag, args in (("pd_text", ["-group", "k", "-int", "v", "-op", "hist", "-limit", "12"]),
                      ("pd_json", ["-group", "k", "-int", "v,w", "-op", "hist", "-limit", "7", "-json"])):
        want = _one(root, args)
        assert len(want) > 0
        # (-stats: rank 0 says on stderr which strategy its scan took)
        env = dict(os.environ, TZ="UTC", HSA_ENABLE_IPC_MODE_LEGACY="0", LD_PRELOAD=STANDIN, SYBL_CLI_BACKTRACE="1",
                   SYBL_STANDIN_TIMEOUT_S=os.environ.get("SYBL_STANDIN_TIMEOUT_S", "120"))
        idf = os.path.join(str(tmp_path), "id_%s_%d_%d" % (tag, world, n_blocks))
        procs = [subprocess.Popen([CLI, "-dir", root, "-table", "events"] + args + ["-stats", "-gpu-rank", str(r), "-gpu-ranks", str(world), "-gpu-id-file", idf],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(world)]
        outs = [p.communicate(timeout=600) for p in procs]
        assert [p.returncode for p in procs] == [0] * world, (tag, b"\n".join(e for _, e in outs).decode(errors="replace")[-4000:])
        # (a rank without rows plans no pushed-down scan; the ranks ask each other at the first scan and then none takes it)
        assert (b"strategy=8" if n_blocks >= world else b"strategy=5") in outs[0][1], (tag, outs[0][1][-600:])
        assert outs[0][0] == want, (tag, world, outs[0][0][:400], want[:400])
