"""GPU: lifetimes and threads (round 5).

(a) Direct-mapped results whose rows are built on first access (>= 2048 rows: result.cpp result_ensure_rows) must stay valid
    -- equal to the oracle of the scan they came from, and byte-equal in both printers to a result built eagerly -- whatever
    happens to their query and table between sybl_query_finalize and the first look at the rows: the query scanned again
    (the bench's own pattern: finalize step i, scan step i + 1, read rows later), scanned and finalized again, freed; the
    table freed; the table appended to and queried anew.  The reference's Results are plain Go values that outlive everything
    (aggregate.go:186-203); config 5's shape (a time series of 60 000 rows) and config 4's (65 536 groups with bucket arrays,
    -limit) are the two the bench runs lazily.
(b) Threads (table_query.go:110,230-231: the reference queries from 16 goroutines): the library serialises calls per ctx
    (include/sybilgpu.h, SYBL_API_GUARD) -- one thread reads a result's lazily built rows while another rescans and frees the
    query behind it, on ONE ctx, without a lock on this side; and threads with a ctx each run whole queries concurrently."""
import threading

import numpy as np
import pytest

import sybil_amd
from sybil_amd import _native as N
from sybil_amd import synth
from tests import parity

pytestmark = pytest.mark.gpu

SHAPES = {
    # 120 six-hour buckets x 500 groups = 60 000 TimeResults rows + 500 all-time rows
    "cfg5": dict(cols=["c00", "c09", "c07"], rows=1_200_000, compact=True,
                 q=dict(groups=["c09"], aggs=["c07"], op="avg", time_col="c00", time_bucket=21600)),
    # 65 536 groups x 1002 buckets, the GPU summary path, bucket arrays for the printed rows only
    "cfg4": dict(cols=["c03", "c07"], rows=1_000_000, compact=True,
                 q=dict(groups=["c03"], aggs=["c07"], op="hist", limit=100)),
}
ACTIONS = ["nothing", "rescan", "rescan_finalize", "free_query", "free_table", "append_requery"]


@pytest.fixture(scope="module")
def ctx():
    c = sybil_amd.Context(0)
    yield c
    c.close()


def _table(ctx, orc, shape, extra_blocks=0):
    s = SHAPES[shape]
    t = ctx.synth_table("life", synth.SEED, s["rows"], 0, s["rows"], synth.synth_cols(s["cols"]))
    if s["compact"]:
        t.compact()
    return t


_ORACLE = {}


def _oracle(orc, shape):
    if shape not in _ORACLE:
        s = SHAPES[shape]
        info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in s["cols"]}
        ocols = parity.oracle_synth_cols(orc, s["cols"], s["rows"], 0, s["rows"])
        _ORACLE[shape] = orc.run_query(ocols, n_threads=8, **parity.oracle_query_kwargs(s["cols"], info, s["q"]))
    return _ORACLE[shape]


def _check(shape, r, o):
    q = SHAPES[shape]["q"]
    if q.get("limit"):
        assert r.matched == o["matched"]
        omap = {x["key"]: x for x in o["results"]}
        rows = r.rows(0)
        assert len(rows) == len(omap)
        with_values = 0
        for g in rows:
            h, oh = g["hists"][0], omap[g["key"]]["hists"][0]
            assert (g["count"], h["count"], h["sum"]) == (omap[g["key"]]["count"], oh["count"], oh["sum_exact"])
            assert np.array_equal(h.get("percentiles", np.zeros(0, dtype=np.int64)), oh["percentiles"])
            if "values" in h:
                with_values += 1
                assert np.array_equal(h["values"], oh["values"])
        assert with_values == q["limit"]
        parity.compare_hist(r.cumulative["hists"][0], o["cumulative"]["hists"][0], "hist", True, cumulative=True)
    else:
        parity.compare(r, o, op=q["op"], full=True, n_aggs=len(q["aggs"]), time_mode=bool(q.get("time_col")))


def _printed(r):
    out = {"text": r.render("text"), "json": r.render("json")}
    try:
        out["gob"] = r.encode()
    except N.SyblError as e:  # (a shape the encoder refuses is refused the same way built eagerly or lazily)
        out["gob"] = "refused: %s" % str(e)[:60]
    return out


@pytest.mark.parametrize("action", ACTIONS)
@pytest.mark.parametrize("shape", list(SHAPES))
def test_lazy_direct_mapped_rows_outlive_what_happens_to_query_and_table(ctx, oracle, monkeypatch, shape, action):
    s = SHAPES[shape]
    o = _oracle(oracle, shape)
    t = _table(ctx, oracle, shape)
    q = t.query(**s["q"])
    monkeypatch.setenv("SYBL_EAGER_ROWS", "1")
    ref = q.scan().finalize()
    want = _printed(ref)
    _check(shape, ref, o)
    ref.free()
    monkeypatch.delenv("SYBL_EAGER_ROWS")
    r = q.scan().finalize()  # >= 2048 rows: nothing built yet
    freed_q = freed_t = False
    if action == "rescan":
        q.scan()  # (left in flight: the rows are read underneath it)
    elif action == "rescan_finalize":
        for _ in range(3):
            q.scan().snapshot()
            r2 = q.finalize()
            r2.free()
        q.scan().snapshot()
    elif action == "free_query":
        q.free()
        freed_q = True
    elif action == "free_table":
        t.free()
        q.free()
        freed_q = freed_t = True
    elif action == "append_requery":
        n = 70_000
        blk = {c: oracle.synth_fill(synth.COLUMNS[c][0], synth.COLUMNS[c][2], synth.COLUMNS[c][3], synth.SEED + 1, synth.COLUMNS[c][1], 0, n, n)
               for c in s["cols"]}
        t.append_block(n, blk)
        with pytest.raises(N.SyblError):
            q.scan()  # the table changed under a prepared query: refused, the first result is untouched
        q2 = t.query(**s["q"])
        r2 = q2.scan().finalize()
        assert r2.matched == o["matched"] + n
        r2.materialize(0)
        r2.free()
        q2.free()
    got = _printed(r)
    for k in want:
        assert got[k] == want[k], (shape, action, k)
    _check(shape, r, o)
    r.free()
    if not freed_q:
        q.free()
    if not freed_t:
        t.free()


def test_pipelined_steps_read_rows_late(ctx, oracle):
    """bench.py's step: two prepared queries alternate, finalize(i) after scan(i + 1) was queued, and the rows of step i
    are looked at only after step i + 2 has finalized into the same query's buffers."""
    shape = "cfg5"
    s = SHAPES[shape]
    o = _oracle(oracle, shape)
    t = _table(ctx, oracle, shape)
    qs = [t.query(**s["q"]), t.query(**s["q"])]
    held = []
    pending = None
    for i in range(7):
        q = qs[i % 2]
        q.scan().snapshot()
        if pending is not None:
            held.append(pending.finalize())
        pending = q
    held.append(pending.finalize())
    for q in qs:
        q.free()
    t.free()
    for r in held:  # seven results alive at once, none of them built before its query and table went away
        _check(shape, r, o)
        r.free()


def test_threads_on_one_ctx_reader_against_rescans_and_free(ctx, oracle):
    """No lock on this side: the library serialises the calls.  Whatever the interleaving, the reader's rows are those of
    the scan its result came from."""
    for shape, hashed in (("cfg5", False), ("cfg4", False), ("cfg5", True)):
        s = SHAPES[shape]
        o = _oracle(oracle, shape)
        t = _table(ctx, oracle, shape)
        import os
        if hashed:
            os.environ["SYBL_FORCE_HASH"] = "1"  # rows with keys of their own: registered with the query, built when it goes away
        try:
            q = t.query(**s["q"])
        finally:
            os.environ.pop("SYBL_FORCE_HASH", None)
        r = q.scan().finalize()
        errors = []
        start = threading.Barrier(2)

        def reader():
            try:
                start.wait()
                _check(shape, r, o)
                r.render("text")
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        def owner():
            try:
                start.wait()
                for _ in range(4):
                    q.scan().snapshot()
                    r2 = q.finalize()
                    r2.free()
                q.free()
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=reader), threading.Thread(target=owner)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, (shape, hashed, errors)
        _check(shape, r, o)
        r.free()
        t.free()


def test_threads_with_a_ctx_each(oracle):
    wl = synth.WORKLOADS["cfg3_filter3_group2_stddev"]
    rows = 600_000
    info = {n: (synth.COLUMNS[n][4], synth.COLUMNS[n][5]) for n in wl["columns"]}
    queries = [dict(wl["query"], want_percentiles=True), dict(groups=["c01"], aggs=["c07", "c08"], op="avg"),
               dict(wl["query"]), dict(filters=[("c04", "gt", 499)], groups=["c02"], aggs=["c08"], op="hist")]
    ocols = parity.oracle_synth_cols(oracle, wl["columns"], rows, 0, rows)
    want = [oracle.run_query(ocols, n_threads=4, **parity.oracle_query_kwargs(wl["columns"], info, q)) for q in queries]
    errors = []
    start = threading.Barrier(len(queries))

    def work(i):
        try:
            c = sybil_amd.Context(0)
            t = c.synth_table("thr%d" % i, synth.SEED, rows, 0, rows, synth.synth_cols(wl["columns"]))
            if i % 2 == 0:
                t.compact()
            start.wait()
            for _ in range(5):
                qy = t.query(**queries[i])
                r = qy.run()
                parity.compare(r, want[i], op=queries[i]["op"], full=queries[i].get("want_percentiles", True), n_aggs=len(queries[i]["aggs"]))
                r.free()
                qy.free()
            t.free()
            c.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((i, e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(queries))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
