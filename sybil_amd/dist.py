"""Multi-rank host protocol (one process per GPU): block sharding, bounds agreement and the
merge of partial group tables.

Reference: the only distributed mechanism in logv/sybil is `sybil aggregate` merging per-host gob
results with CombineResults (node_aggregator.go:147-177, aggregate.go:414-467).  Here every rank
holds an identically laid out integer table, so the merge is one SUM all-reduce (counts, sums,
moments, buckets) and one MAX all-reduce (extrema, minima negated) over RCCL.  The functions take
any torch.distributed process group, so the protocol is exercised on CPU with gloo in
tests/test_dist_gloo.py and on GPUs with the nccl(=RCCL) backend in bench.py.
"""
import torch
import torch.distributed as dist

from .synth import shard  # noqa: F401  (re-exported: contiguous 65536-row block ranges per rank)

INT64_MIN = -(1 << 63)


def merge_partials(sum_words, max_words, group=None, has_max=True):
    """In-place merge of the partial tables of all ranks (the contract of sybl_query_partials)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(sum_words, op=dist.ReduceOp.SUM, group=group)
    if has_max:
        dist.all_reduce(max_words, op=dist.ReduceOp.MAX, group=group)


def merge_sketches(registers, group=None):
    """Count-distinct queries: the ranks' sketches ([cell][16384] uint8 registers) combine by the register-wise maximum
    (Result.Combine -> Distinct.Merge, query_spec.go:180-188) -- one MAX all-reduce, in place.  (sybl_query_allreduce
    does the same over RCCL inside the library.)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(registers, op=dist.ReduceOp.MAX, group=group)


def merge_hash_partials(query, device, group=None):
    """Hash group-by (sybl_query_hash_keys / sybl_query_hash_install_union): the ranks found different key sets, so
    they first install the sorted union of their keys -- the dense partial arrays then line up -- and merge with the
    usual SUM / MAX all-reduce.  Call after query.scan(); query.finalize() on any rank then gives the whole result."""
    import numpy as np
    mine = query.hash_keys()
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if multi:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, mine, group=group)
        union = np.unique(np.concatenate(parts))
    else:
        union = mine
    query.hash_install_union(union)
    if multi:
        s, m = query.partials_torch(device)
        query.table.ctx.sync()  # the install ran on the ctx stream, the collective runs on torch's
        merge_partials(s, m, group=group, has_max=query.stats()["n_max_fields"] > 0)
        import torch
        torch.cuda.synchronize(device)
    return union


def agree_bounds(infos, device="cpu", group=None):
    """infos: {column: {"exact_min", "exact_max", "has_missing"}} of this rank's shard (a rank with no
    populated row reports exact_min > exact_max).  Returns the bounds every rank must declare with
    sybl_table_set_bounds so that the direct-mapped layout is identical everywhere.

    One MAX all-reduce: minima travel as their bitwise complement (~x = -x - 1 is order-reversing and, unlike
    -x, cannot overflow at INT64_MIN), and a separate flag says whether any rank holds a value at all."""
    names = sorted(infos)
    has = [infos[n]["exact_min"] <= infos[n]["exact_max"] for n in names]
    lo = torch.tensor([~infos[n]["exact_min"] if h else INT64_MIN for n, h in zip(names, has)], dtype=torch.int64, device=device)
    hi = torch.tensor([infos[n]["exact_max"] if h else INT64_MIN for n, h in zip(names, has)], dtype=torch.int64, device=device)
    miss = torch.tensor([1 if infos[n].get("has_missing") else 0 for n in names], dtype=torch.int64, device=device)
    rows = torch.tensor([1 if h else 0 for h in has], dtype=torch.int64, device=device)
    packed = torch.cat([lo, hi, miss, rows])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    k = len(names)
    out = {}
    for i, n in enumerate(names):
        nlo, h, m, any_rows = int(packed[i]), int(packed[k + i]), int(packed[2 * k + i]), int(packed[3 * k + i])
        if not any_rows:  # no rank holds a value
            out[n] = {"lo": 0, "hi": -1, "has_missing": bool(m)}
        else:
            out[n] = {"lo": ~nlo, "hi": h, "has_missing": bool(m)}
    return out


def apply_bounds(table, bounds):
    """Declares the agreed bounds on this rank.  A column no rank holds a value of still gets its agreed
    has_missing flag (an empty range): the MISSING key digit and the populated-count field must exist on
    every rank or on none, or the partial tables would differ in size."""
    for name, b in bounds.items():
        if b["hi"] >= b["lo"]:
            table.set_bounds(name, b["lo"], b["hi"], b["has_missing"])
        else:
            table.set_bounds(name, 0, -1, b["has_missing"])


def check_layout(query, group=None):
    """Every rank must hold the same partial-table layout before the merge: a mismatch would make the all-reduce
    hang or add unrelated words.  Raises on every rank when the sizes differ."""
    ns, nm = query.partial_sizes()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ns, nm
    t = torch.tensor([ns, -ns, nm, -nm], dtype=torch.int64)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t[0]) != -int(t[1]) or int(t[2]) != -int(t[3]):
        raise RuntimeError("partial tables differ across ranks: SUM words %d..%d, MAX words %d..%d -- declare the same bounds / "
                           "dictionaries on every rank (agree_bounds, agree_group_dict, agree_str_dict)"
                           % (-int(t[1]), int(t[0]), -int(t[3]), int(t[2])))
    return ns, nm


def agree_group_dict(table, column, group=None):
    """Sparse group keys (sybl_table_column_distinct / sybl_table_set_group_dict): every rank installs
    the sorted union of the ranks' distinct values, so digits (= ranks in that union) and therefore
    the partial tables line up across ranks."""
    import numpy as np
    mine = table.column_distinct(column)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, mine, group=group)
        union = np.unique(np.concatenate(parts)) if parts else mine
    else:
        union = mine
    table.set_group_dict(column, union)
    return union


def agree_str_dict(table, column, group=None):
    """Str / set columns: every rank installs the sorted union of the ranks' dictionaries, so a
    dictionary id -- and with it a str group cell or a per-id filter mask -- means the same string on
    every rank (the reference merges blocks by translated string key, aggregate.go:284-324)."""
    mine = table.column_dict(column)
    missing = bool(table.column_info(column)["has_missing"])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, (mine, missing), group=group)
        union = sorted(set(s for p, _ in parts for s in p))
        missing = any(m for _, m in parts)
    else:
        union = sorted(set(mine))
    table.set_dict(column, union)
    # ... and whether ANY rank has a row without the column: the MISSING key digit of a str group-by (aggregate.go:138) must
    # exist on every rank or on none, or the partial tables differ by a cell (found by check_layout on a loaded table whose
    # missing names all sat in one rank's blocks: tests/test_gpu_multirank.py::test_loaded_table_across_ranks)
    if missing:
        table.set_bounds(column, 0, -1, True)
    return union
