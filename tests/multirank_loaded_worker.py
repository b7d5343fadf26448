"""One rank of tests/test_gpu_multirank.py::test_loaded_table_across_ranks: `python -m tests.multirank_loaded_worker <dir>
<world> <rank> <device> <table root>`.  The whole host protocol a multi-GPU sybil would run (INTEGRATION.md "Multi-GPU from
Go"), on a table that came from DISK: every rank opens its contiguous range of block directories (sybl_table_open with rank /
nranks), the ranks agree on column bounds, on the str / set dictionaries and on the sparse key's group dictionary
(sybil_amd.dist over a gloo process group -- the part a Go host does over its own channel), then scan, merge in the library
(sybl_query_allreduce: the real RCCL, or the test-only stand-in the parent preloads) and rank 0 finalizes."""
import os
import pickle
import sys
import time

QUERIES = [
    dict(groups=["name"], aggs=["age", "time"], op="avg"),                                    # str key: dictionary ids agreed
    dict(groups=["big"], aggs=["age"], op="avg"),                                             # sparse int key: union dictionary -> rank column
    dict(filters=[("tags", "in", "tag3")], groups=["name"], aggs=["age"], op="hist"),         # set member filter (pre-pass / generic)
    dict(filters=[("name", "re", "user[1-3].*")], groups=["age"], aggs=["big"], op="avg"),     # str regex as an id mask, negative values
    dict(groups=["age"], aggs=["time"], op="hist", time_col="time", time_bucket=7200),        # a time series
]


def summarise(res):
    def hist(h):
        return (h["count"], h["sum"], h["min"], h["max"], None if "percentiles" not in h else tuple(h["percentiles"].tolist()))
    return {"matched": res.matched,
            "rows": [sorted((r["time_bucket"], r["group_by_key"], r["count"], tuple(hist(h) for h in r["hists"])) for r in res.rows(w)) for w in (0, 1)]}


def main():
    work, world, rank, device, root = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch.distributed as dist
    import sybil_amd
    from sybil_amd import dist as sdist
    dist.init_process_group("gloo", init_method="file://" + os.path.join(work, "rendezvous"), rank=rank, world_size=world)
    ctx = sybil_amd.Context(device)
    uid = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], world, rank)
    tb = ctx.open_table(root, "events", rank=rank, nranks=world, compact=True)
    ints = ["age", "time", "big"]
    sdist.apply_bounds(tb, sdist.agree_bounds({c: tb.column_info(c) for c in ints}))
    sdist.agree_str_dict(tb, "name")
    sdist.agree_str_dict(tb, "tags")
    sdist.agree_group_dict(tb, "big")
    out = []
    for q in QUERIES:
        qy = tb.query(**q)
        sdist.check_layout(qy)
        qy.scan()
        qy.allreduce()
        if rank == 0 or qy.collective_finalize():
            res = qy.finalize()
            if rank == 0:
                out.append(dict(summarise(res), strategy=qy.stats()["strategy"]))
            res.free()
        qy.free()
    if rank == 0:
        with open(os.path.join(work, "loaded.pkl"), "wb") as f:
            pickle.dump({"rows": tb.rows, "results": out}, f)
    rows = tb.rows
    tb.free()
    ctx.comm_free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d: %d rows" % (rank, rows))


if __name__ == "__main__":
    main()
