"""Test infrastructure: the CPU oracle on the LOGICAL content of the fabricated on-disk table of tests/test_gpu_loader.py
(_make_blocks), for tests that run the engine on that table from disk -- several ranks (tests/test_gpu_multirank.py), the CLI
with -gpu-ranks (tests/test_gpu_cli_multirank.py).  Str group keys are compared through their strings: dictionary ids are
private to each side."""
import numpy as np

from tests import parity

MISSING = 0xFFFFFFFFFFFFFFFF
INFO_BIG = (-(1 << 40), 1 << 40)


class LoadedOracle:
    def __init__(self, oracle, logical, threshold):
        from tests.test_gpu_loader import _logical_after_load
        self.oracle = oracle
        age, t, big, big_pop, names, name_pop, tags, tag_pop = _logical_after_load(logical, threshold)
        n = age.size
        self.n = n
        self.uniq = sorted({s for s in names if s is not None})
        ix = {s: i for i, s in enumerate(self.uniq)}
        sid = np.array([ix[s] if s is not None else 0 for s in names], dtype=np.int32)
        self.tag_names = sorted({x for s in tags if s for x in s})
        tix = {s: i for i, s in enumerate(self.tag_names)}
        off = np.zeros(n + 1, dtype=np.int64)
        flat = []
        for r, s in enumerate(tags):
            if s:
                flat += [tix[x] for x in s]
            off[r + 1] = len(flat)
        self.names = names
        self.ocols = [{"type": "int", "data": age}, {"type": "int", "data": t},
                      {"type": "int", "data": big, "populated": big_pop.astype(np.uint8)},
                      {"type": "str", "data": sid, "populated": name_pop.astype(np.uint8)},
                      {"type": "set", "data": np.array(flat, dtype=np.int32), "offsets": off, "populated": tag_pop.astype(np.uint8)}]
        self.cols = ["age", "time", "big", "name", "tags"]
        self.info = {"age": (10, 29), "big": INFO_BIG, "time": (int(t.min()), int(t.max()))}

    def run(self, q):
        import re
        okw = parity.oracle_query_kwargs(self.cols, self.info, {k: v for k, v in q.items() if k != "filters"})
        fixed = []
        for f in q.get("filters", []):
            c = self.cols.index(f[0])
            if f[1] in ("re", "nre"):
                # the oracle takes the host's verdict per dictionary id (Go's regexp on the reference side; Python's here)
                rx = re.compile(f[2])
                fixed.append((c, f[1], 0, np.array([1 if rx.search(s) else 0 for s in self.uniq], dtype=np.uint8)))
            elif isinstance(f[2], str):
                table = self.uniq if f[0] == "name" else self.tag_names
                fixed.append((c, f[1], table.index(f[2]) if f[2] in table else -1))
            else:
                fixed.append((c,) + tuple(f[1:]))
        okw["filters"] = fixed
        return self.oracle.run_query(self.ocols, block_rows=10 ** 9, **okw)

    def key_string(self, q, key_vals):
        out = ""
        for g, v in zip(q.get("groups", []), key_vals):
            if v == MISSING:
                out += "\t"
            elif g == "name":
                out += self.uniq[v] + "\t"
            else:
                out += "%d\t" % (v - (1 << 64) if v >= 1 << 63 else v)
        return out


def hist_tuple(h, op):
    if not h["present"]:
        return None
    pct = h.get("percentiles")
    pct = None if op != "hist" or pct is None or len(pct) == 0 else tuple(int(x) for x in pct)
    return (h["count"], h["sum_exact"] if "sum_exact" in h else h["sum"], h["min"], h["max"], pct)


def summarise_engine(res, op):
    """matched + the rows of Results and TimeResults as comparable tuples (sybil_amd.Result or a dumped one)."""
    return {"matched": res.matched,
            "rows": [sorted((r["time_bucket"], r["group_by_key"], r["count"], tuple(hist_tuple(h, op) for h in r["hists"])) for r in res.rows(w))
                     for w in (0, 1)]}


def summarise_oracle(LO, q, ores):
    op = q.get("op", "avg")
    return {"matched": ores["matched"],
            "rows": [sorted((r["time_bucket"], LO.key_string(q, r["key_vals"]), r["count"], tuple(hist_tuple(h, op) for h in r["hists"])) for r in ores[name])
                     for name in ("results", "time_results")]}
