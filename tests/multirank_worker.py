"""One rank of tests/test_gpu_multirank.py: `python -m tests.multirank_worker <dir> <world> <rank> <device>`.

Scans this rank's contiguous block shard of the synthetic table for every case in CASES, merges through the library's own
collective path (sybl_comm_* / sybl_query_allreduce, csrc/rccl.cpp) and dumps what sybl_query_finalize returned -- on rank
0, and on every rank when the finalize is collective -- for the parent to compare with the oracle over the whole table.
The parent decides which librccl answers: the real one (one process per GPU) or the test-only shared-memory stand-in in
LD_PRELOAD (tests/rccl_standin/, processes sharing one GPU)."""
import os
import pickle
import sys
import time

TOTAL = 6_000_000

_RANGE3 = [("c04", "gt", 99), ("c04", "lt", 900), ("c05", "gt", 99), ("c05", "lt", 900), ("c06", "gt", 99), ("c06", "lt", 900)]
_CFG3 = dict(filters=_RANGE3, groups=["c01", "c02"], aggs=["c07", "c08"], op="hist", want_percentiles=False)

# name -> (columns, query kwargs, environment for prepare .. finalize, options)
CASES = {
    "direct": (["c04", "c05", "c06", "c01", "c02", "c07", "c08"], _CFG3, {}, dict(compact=True)),
    "extrema": (["c01", "c07"], dict(groups=["c01"], aggs=["c07"], op="avg"), {}, {}),
    "allreduce_hist": (["c01", "c07", "c08"], dict(groups=["c01"], aggs=["c07", "c08"], op="hist"), {}, {}),
    "scatter32": (["c03", "c07"], dict(groups=["c03"], aggs=["c07"], op="hist", limit=50), {}, dict(compact=True)),
    "scatter64": (["c03", "c07"], dict(groups=["c03"], aggs=["c07"], op="hist", limit=50), {"SYBL_NO_SCATTER32": "1"}, {}),
    # (-hist-bucket 990 puts the top ~1 % of c07 beyond the last bucket: ~60 000 outliers, spread over every rank's log)
    # a printer's merge (sybl_query_desc.printed_only): the bucket table stays rank-local, the cell fields are all-reduced, every
    # rank derives the same top-`limit` order, and only Cumulative's buckets and those rows' arrays are summed
    "printer": (["c03", "c07"], dict(groups=["c03"], aggs=["c07"], op="hist", limit=50, printed_only=True), {}, dict(compact=True)),
    "outliers": (["c01", "c07"], dict(groups=["c01"], aggs=["c07"], op="hist", hist_bucket=990), {}, {}),
    "hash": (["c04", "c05", "c06", "c01", "c02", "c07", "c08"], _CFG3, {"SYBL_FORCE_HASH": "1"}, {}),
    "hash_extrema": (["c02", "c09", "c08"], dict(groups=["c02", "c09"], aggs=["c08"], op="avg"), {"SYBL_FORCE_HASH": "1"}, dict(compact=True)),
    "timeseries": (["c00", "c09", "c07"], dict(groups=["c09"], aggs=["c07"], op="avg", time_col="c00", time_bucket=3600), {}, dict(compact=True)),
    # three blocks over 2 / 4 / 8 ranks: ranks WITHOUT A ROW (at world 4 and 8 rank 0, the one that finalizes, is one of them)
    "tiny": (["c04", "c01", "c02", "c07"], dict(filters=[("c04", "gt", 99)], groups=["c01", "c02"], aggs=["c07"], op="avg"), {}, dict(compact=True, total=150_000)),
    "tiny_hash": (["c01", "c02", "c07"], dict(groups=["c01", "c02"], aggs=["c07"], op="hist", want_percentiles=False), {"SYBL_FORCE_HASH": "1"},
                  dict(total=150_000)),
    "distinct": (["c01", "c05", "c06"], dict(groups=["c01"], distincts=["c05", "c06"], filters=[("c06", "lt", 500)]), {}, {}),
    # ... over a hashed group-by (round 6): the sketch pass runs on every rank over the UNION of the ranks' keys, then the MAX
    "distinct_hash": (["c01", "c02", "c05"], dict(groups=["c01", "c02"], distincts=["c05"]), {"SYBL_FORCE_HASH": "1"}, {}),
}


def dump_result(res, n_distincts=0):
    out = {"matched": res.matched, "rows": [res.rows(0), res.rows(1), res.rows(2)]}
    if n_distincts:
        out["registers"] = [[res.distinct(w, i, registers=True) for i in range(len(out["rows"][w]))] for w in range(3)]
    return out


def main():
    work, world, rank, device = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    names = sys.argv[5].split(",") if len(sys.argv) > 5 else list(CASES)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import sybil_amd
    from sybil_amd import synth
    ctx = sybil_amd.Context(device)
    uid_path = os.path.join(work, "uid")
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(ctx.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.02)
    ctx.comm_init(open(uid_path, "rb").read(), world, rank)
    results = {}
    for name in names:
        cols, q, env, opt = CASES[name]
        os.environ.update(env)
        total = opt.get("total", TOTAL)
        row0, nrows = synth.shard(total, rank, world)
        t = ctx.synth_table("mr", synth.SEED, total, row0, nrows, synth.synth_cols(cols))
        for n in cols:  # identical direct-mapped layout on every rank: the generator's bounds
            kind, _, a, b, _, _ = synth.COLUMNS[n]
            t.set_bounds(n, a, a + 4 * (b - 1) if kind == synth.BELL else a + b - 1)
        for n, (lo, hi) in opt.get("bounds", {}).items():
            t.set_bounds(n, lo, hi)
        if opt.get("compact"):
            t.compact()
        qy = t.query(**q)
        for rep in range(opt.get("reps", 1)):  # (a query object is scanned and merged more than once in a serving loop)
            qy.scan()
            qy.allreduce()
            everyone = qy.collective_finalize()
            res = None
            if rank == 0 or everyone:
                res = qy.finalize()
                st = qy.stats()
                results[name] = dict(dump_result(res, len(q.get("distincts", ()))), strategy=st["strategy"], packed=st["packed_kernel"],
                                     everyone=everyone)
                res.free()
        qy.free()
        t.free()
        for k in env:
            del os.environ[k]
    with open(os.path.join(work, "out%d.pkl" % rank), "wb") as f:
        pickle.dump(results, f)
    ctx.comm_free()
    ctx.close()


if __name__ == "__main__":
    main()
