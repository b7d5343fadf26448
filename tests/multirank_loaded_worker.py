"""One rank of tests/test_gpu_multirank.py::test_loaded_table_across_ranks: `python -m tests.multirank_loaded_worker <dir>
<world> <rank> <device> <table root>`.  The whole host protocol a multi-GPU sybil runs (INTEGRATION.md "Multi-GPU from Go"),
on a table that came from DISK and through the C ABI alone -- no torch.distributed, no gloo: every rank opens its contiguous
range of block directories (sybl_table_open with rank / nranks), rank 0 hands the communicator id over in a file, the ranks
agree on column bounds, the str / set dictionaries and the sparse keys' group dictionaries inside the library
(sybl_table_agree), then scan, merge (sybl_query_allreduce: the real RCCL, or the test-only stand-in the parent preloads) and
rank 0 finalizes."""
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
    dict(groups=["big", "time"], aggs=["age"], op="avg"),                                     # hashed: 2^41 x 50 000 key space, both keys through union dictionaries
]


def main():
    work, world, rank, device, root = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import sybil_amd
    from tests import loaded_oracle as LO
    ctx = sybil_amd.Context(device)
    uid_path = os.path.join(work, "uid")
    if world > 1:
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(ctx.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 120, "no communicator id from rank 0"
            time.sleep(0.01)
        ctx.comm_init(open(uid_path, "rb").read(), world, rank)
        assert ctx.comm_info() == (rank, world)
    tb = ctx.open_table(root, "events", rank=rank, nranks=world, compact=True)
    out = []
    for q in QUERIES:
        tb.agree(q.get("groups", []))          # collective; the same call with one rank sorts the dictionaries
        qy = tb.query(**q)
        qy.scan()
        if world > 1:
            qy.allreduce()                    # (its first call on a query also compares the ranks' layouts)
        if rank == 0 or qy.collective_finalize():
            res = qy.finalize()
            if rank == 0:
                out.append(dict(LO.summarise_engine(res, q.get("op", "avg")), strategy=qy.stats()["strategy"]))
            res.free()
        qy.free()
    if rank == 0:
        with open(os.path.join(work, "loaded.pkl"), "wb") as f:
            pickle.dump({"rows": tb.rows, "results": out}, f)
    rows = tb.rows
    tb.free()
    if world > 1:
        ctx.comm_free()
    ctx.close()
    print("rank %d: %d rows" % (rank, rows))


if __name__ == "__main__":
    main()
