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


def agree_bounds(infos, device="cpu", group=None):
    """infos: {column: {"exact_min", "exact_max", "has_missing"}} of this rank's shard (a rank with no
    populated row reports exact_min > exact_max).  Returns the bounds every rank must declare with
    sybl_table_set_bounds so that the direct-mapped layout is identical everywhere."""
    names = sorted(infos)
    lo = torch.tensor([-infos[n]["exact_min"] if infos[n]["exact_min"] <= infos[n]["exact_max"] else INT64_MIN
                       for n in names], dtype=torch.int64, device=device)
    hi = torch.tensor([infos[n]["exact_max"] if infos[n]["exact_min"] <= infos[n]["exact_max"] else INT64_MIN
                       for n in names], dtype=torch.int64, device=device)
    miss = torch.tensor([1 if infos[n].get("has_missing") else 0 for n in names], dtype=torch.int64, device=device)
    packed = torch.cat([lo, hi, miss])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    k = len(names)
    out = {}
    for i, n in enumerate(names):
        nlo, h, m = int(packed[i]), int(packed[k + i]), int(packed[2 * k + i])
        if nlo == INT64_MIN:  # no rank holds a value
            out[n] = {"lo": 0, "hi": -1, "has_missing": bool(m)}
        else:
            out[n] = {"lo": -nlo, "hi": h, "has_missing": bool(m)}
    return out


def apply_bounds(table, bounds):
    for name, b in bounds.items():
        if b["hi"] >= b["lo"]:
            table.set_bounds(name, b["lo"], b["hi"], b["has_missing"])


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
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, mine, group=group)
        union = sorted(set(s for p in parts for s in p))
    else:
        union = sorted(set(mine))
    table.set_dict(column, union)
    return union
