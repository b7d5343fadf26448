"""Writes sybil tables in the reference's on-disk format (test infrastructure).

Mirrors SaveToColumns / SaveIntsToColumns / SaveStrsToColumns / SaveSetsToColumns /
SaveInfoToColumns and SaveTableInfo (src/lib/column_store_io.go:64-358,419-491,
table_io.go:40-78): one directory per block holding info.db + int_/str_/set_<col>.db gob
files; <= CARDINALITY_THRESHOLD distinct values => bins of delta-encoded ascending record
ids, otherwise per-row Values (delta-encoded for ints); VERSION = 1.  Optionally gzips the
files (".db.gz", file_decoder.go:10).
"""
import gzip
import os

import numpy as np

from tests import gobfmt as G

INT_VAL, STR_VAL, SET_VAL = 1, 2, 3
CARDINALITY_THRESHOLD = 5000  # column_store_io.go:18


def _write(path, data, gz):
    if gz:
        with gzip.open(path + ".gz", "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def _bins(value_of_row, rows):
    """{value: ascending record ids} -> [(value, delta-encoded ids)] (delta_encode_col, :21-30)."""
    by = {}
    for r in rows:
        by.setdefault(value_of_row(r), []).append(int(r))
    out = []
    for v, recs in by.items():
        prev, d = 0, []
        for r in recs:
            d.append(r - prev)
            prev = r
        out.append((v, d))
    return out


def int_column(name, values, populated=None, threshold=CARDINALITY_THRESHOLD):
    n = len(values)
    rows = [r for r in range(n) if populated is None or populated[r]]
    col = {"Name": name, "DeltaEncodedIDs": True, "VERSION": 1}
    distinct = len({int(values[r]) for r in rows})
    if distinct <= threshold:
        col["BucketEncoded"] = True
        col["Bins"] = [{"Value": int(v), "Records": d} for v, d in _bins(lambda r: int(values[r]), rows)]
    else:
        # SaveIntsToColumns :97-114: Values[max_r], rows without a value hold 0, then delta-encode
        max_r = rows[-1] + 1 if rows else 0
        dense = [0] * max_r
        for r in rows:
            dense[r] = int(values[r])
        prev, deltas = 0, []
        for v in dense:
            deltas.append(v - prev)
            prev = v
        col["ValueEncoded"] = True
        col["Values"] = deltas
    return G.encode(G.saved_int_column(), col)


def str_column(name, strings, threshold=CARDINALITY_THRESHOLD):
    """strings: list of str or None (missing)."""
    rows = [r for r, s in enumerate(strings) if s is not None]
    table, ids = [], {}
    for r in rows:
        if strings[r] not in ids:
            ids[strings[r]] = len(table)
            table.append(strings[r])
    col = {"Name": name, "DeltaEncodedIDs": True, "StringTable": table, "VERSION": 1}
    if len(table) <= threshold:
        col["BucketEncoded"] = True
        col["Bins"] = [{"Value": v, "Records": d} for v, d in _bins(lambda r: ids[strings[r]], rows)]
    else:
        max_r = rows[-1] + 1 if rows else 0
        vals = [0] * max_r
        for r in rows:
            vals[r] = ids[strings[r]]
        col["Values"] = vals
    return G.encode(G.saved_str_column(), col)


def set_column(name, sets, threshold=CARDINALITY_THRESHOLD):
    """sets: list of (list of str) or None (missing)."""
    rows = [r for r, s in enumerate(sets) if s]
    table, ids = [], {}
    members = {}
    for r in rows:
        for s in sets[r]:
            if s not in ids:
                ids[s] = len(table)
                table.append(s)
            members.setdefault(ids[s], []).append(r)
    col = {"Name": name, "DeltaEncodedIDs": True, "StringTable": table, "VERSION": 1}
    if len(table) <= threshold:
        col["BucketEncoded"] = True
        bins = []
        for v, recs in members.items():
            prev, d = 0, []
            for r in recs:
                d.append(r - prev)
                prev = r
            bins.append({"Value": v, "Records": d})
        col["Bins"] = bins
    else:
        max_r = rows[-1] + 1 if rows else 0
        col["Values"] = [[ids[s] for s in (sets[r] or [])] for r in range(max_r)]
    return G.encode(G.saved_set_column(), col)


def write_table(root, table, blocks, gz=False, threshold=CARDINALITY_THRESHOLD, int_info=None, extra_dirs=True):
    """blocks: list of {column: spec}; spec = ("int", values[, populated]) | ("str", [str|None]) |
    ("set", [[str]|None]).  Returns {column: type}.  int_info: {col: (min, max)} override for the
    table-level IntInfo (default: exact extrema)."""
    tdir = os.path.join(root, table)
    os.makedirs(tdir, exist_ok=True)
    key_table, key_types = {}, {}
    tmin, tmax, tcount, tsum = {}, {}, {}, {}
    for bi, blk in enumerate(blocks):
        bdir = os.path.join(tdir, "block%09d" % (bi + 1))
        os.makedirs(bdir, exist_ok=True)
        nrows = None
        int_infos = {}
        for cname, spec in blk.items():
            kind = spec[0]
            if cname not in key_table:
                key_table[cname] = len(key_table)
                key_types[key_table[cname]] = {"int": INT_VAL, "str": STR_VAL, "set": SET_VAL}[kind]
            n = len(spec[1])
            nrows = n if nrows is None else nrows
            assert n == nrows
            if kind == "int":
                vals = np.asarray(spec[1], dtype=np.int64)
                pop = spec[2] if len(spec) > 2 else None
                _write(os.path.join(bdir, "int_%s.db" % cname), int_column(cname, vals, pop, threshold), gz)
                sel = vals if pop is None else vals[np.asarray(pop, dtype=bool)]
                if sel.size:
                    int_infos[cname] = {"Min": int(sel.min()), "Max": int(sel.max()), "Avg": float(sel.mean()),
                                        "M2": float(((sel - sel.mean()) ** 2).sum()), "Count": int(sel.size)}
                    tmin[cname] = min(tmin.get(cname, int(sel.min())), int(sel.min()))
                    tmax[cname] = max(tmax.get(cname, int(sel.max())), int(sel.max()))
                    tcount[cname] = tcount.get(cname, 0) + int(sel.size)
                    tsum[cname] = tsum.get(cname, 0) + int(sel.sum())
            elif kind == "str":
                _write(os.path.join(bdir, "str_%s.db" % cname), str_column(cname, spec[1], threshold), gz)
            else:
                _write(os.path.join(bdir, "set_%s.db" % cname), set_column(cname, spec[1], threshold), gz)
        info = {"NumRecords": nrows or 0, "IntInfoMap": int_infos}
        _write(os.path.join(bdir, "info.db"), G.encode(G.saved_column_info(), info), gz)
    tinfo = {}
    for cname in tmin:
        lo, hi = (int_info or {}).get(cname, (tmin[cname], tmax[cname]))
        tinfo[key_table[cname]] = {"Min": lo, "Max": hi, "Avg": tsum[cname] / max(tcount[cname], 1), "M2": 0.0,
                                   "Count": tcount[cname]}
    tbl = {"Name": table, "KeyTable": key_table, "KeyTypes": key_types, "IntInfo": tinfo}
    _write(os.path.join(tdir, "info.db"), G.encode(G.table_info(), tbl), False)
    if extra_dirs:
        # directories the reference skips (file_looks_like_block, table_io.go:214-239)
        for d in ("cache", "ingest", "stomache_123", "block_old", "x.lock"):
            os.makedirs(os.path.join(tdir, d), exist_ok=True)
    return {c: {INT_VAL: "int", STR_VAL: "str", SET_VAL: "set"}[key_types[i]] for c, i in key_table.items()}
