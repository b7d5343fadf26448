"""ctypes binding of the CPU oracle (oracle/sybil_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the sybil_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsybil_oracle.so")

NO_VAL, INT_VAL, STR_VAL, SET_VAL = 0, 1, 2, 3
OP_GT, OP_LT, OP_EQ, OP_NEQ, OP_RE, OP_NRE, OP_IN, OP_NIN = range(8)
OPS = {"gt": OP_GT, "lt": OP_LT, "eq": OP_EQ, "neq": OP_NEQ, "re": OP_RE, "nre": OP_NRE, "in": OP_IN, "nin": OP_NIN}
AGG_AVG, AGG_HIST = 0, 1
SYN_UNIFORM, SYN_TIME, SYN_BELL = 0, 1, 2
MAX_GROUPS = 8
MAX_AGGS = 8
LLB_M = 1 << 14  # ORC_LLB_M: registers of a count-distinct sketch


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "sybil_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(src[:-2] + ".h"))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


class _Col(C.Structure):
    _fields_ = [("type", C.c_int32), ("ints", C.c_void_p), ("strs", C.c_void_p),
                ("set_off", C.c_void_p), ("set_vals", C.c_void_p), ("populated", C.c_void_p)]


class _Filter(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32), ("value", C.c_int64),
                ("idtable", C.c_void_p), ("idtable_len", C.c_int64)]


class _Agg(C.Structure):
    _fields_ = [("col", C.c_int32), ("info_min", C.c_int64), ("info_max", C.c_int64)]


class _Query(C.Structure):
    _fields_ = [("n_filters", C.c_int32), ("filters", C.POINTER(_Filter)),
                ("n_groups", C.c_int32), ("group_cols", C.c_int32 * MAX_GROUPS),
                ("n_aggs", C.c_int32), ("aggs", _Agg * MAX_AGGS),
                ("op", C.c_int32), ("hist_bucket", C.c_int64),
                ("time_col", C.c_int32), ("time_bucket", C.c_int64),
                ("weight_col", C.c_int32), ("block_skip", C.c_int32),
                ("block_rows", C.c_int64), ("n_threads", C.c_int32), ("loghist", C.c_int32),
                ("n_distincts", C.c_int32), ("distinct_cols", C.c_int32 * MAX_GROUPS),
                ("distinct_dicts", C.POINTER(C.c_char_p) * MAX_GROUPS), ("distinct_dict_len", C.c_int64 * MAX_GROUPS)]


class HistInfo(C.Structure):
    _fields_ = [("present", C.c_int32), ("percentile_mode", C.c_int32),
                ("num_buckets", C.c_int64), ("bucket_size", C.c_int64), ("n_values", C.c_int64),
                ("count", C.c_int64), ("samples", C.c_int64), ("min", C.c_int64), ("max", C.c_int64),
                ("avg", C.c_double), ("sum_exact", C.c_int64),
                ("true_min", C.c_int64), ("true_max", C.c_int64),
                ("n_outliers", C.c_int64), ("n_underliers", C.c_int64),
                ("stddev_ref", C.c_double), ("stddev_exact", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_query_run.restype = C.c_void_p
        L.orc_query_run.argtypes = [C.POINTER(_Query), C.POINTER(_Col), C.c_int32, C.c_int64]
        L.orc_results_free.argtypes = [C.c_void_p]
        for f in ("orc_matched_count", "orc_blocks_scanned", "orc_blocks_skipped"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_num_results.restype = C.c_int64
        L.orc_num_results.argtypes = [C.c_void_p, C.c_int]
        L.orc_result_get.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_result_hist.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(HistInfo)]
        L.orc_result_distinct.restype = C.c_int64
        L.orc_result_distinct.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
        L.orc_metro64.restype = C.c_uint64
        L.orc_metro64.argtypes = [C.c_char_p, C.c_int64, C.c_uint64]
        L.orc_llb_add_hash.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_llb_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.orc_llb_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_llb_cardinality.restype = C.c_uint64
        L.orc_llb_cardinality.argtypes = [C.c_void_p]
        L.orc_result_hist_values.restype = C.c_int64
        L.orc_result_hist_values.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
        L.orc_result_percentiles.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]
        L.orc_result_n_sub.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int]
        L.orc_result_sub.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.orc_result_sparse.restype = C.c_int64
        L.orc_result_sparse.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_hist_new_multi.restype = C.c_void_p
        L.orc_hist_new_multi.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int]
        L.orc_hist_n_sub.argtypes = [C.c_void_p]
        L.orc_hist_sub.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_hist_sparse.restype = C.c_int64
        L.orc_hist_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_result_outliers.restype = C.c_int64
        L.orc_result_outliers.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
        L.orc_setup_buckets.argtypes = [C.c_int64, C.c_int64, C.c_int64] + [C.POINTER(C.c_int64)] * 3
        L.orc_percentiles_from_values.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_stddev_from_values.restype = C.c_double
        L.orc_stddev_from_values.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                                             C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_combine_avg.restype = C.c_double
        L.orc_combine_avg.argtypes = [C.c_double, C.c_int64, C.c_double, C.c_int64]
        L.orc_time_bucket.restype = C.c_int64
        L.orc_time_bucket.argtypes = [C.c_int64, C.c_int64]
        L.orc_hist_new.restype = C.c_void_p
        L.orc_hist_new.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int]
        L.orc_hist_add.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.orc_hist_combine.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hist_info_get.argtypes = [C.c_void_p, C.POINTER(HistInfo)]
        L.orc_hist_values.restype = C.c_int64
        L.orc_hist_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_hist_percentiles.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hist_outliers.restype = C.c_int64
        L.orc_hist_outliers.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_hist_free.argtypes = [C.c_void_p]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_synth_fill.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_int32,
                                     C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_synth_scan.restype = C.c_int64
        L.orc_synth_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_columnar_scan.restype = C.c_int64
        L.orc_columnar_scan.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def metro64(data, seed=1337):
    """MetroHash64 of bytes (go-metro Hash64; loglogbeta hashes with seed 1337)."""
    return lib().orc_metro64(bytes(data), len(data), seed)


class LogLogBeta:
    """The reference's count-distinct sketch (github.com/logv/loglogbeta, restated: see sybil_oracle.h)."""

    def __init__(self, registers=None):
        self.registers = np.zeros(LLB_M, dtype=np.uint8) if registers is None else np.array(registers, dtype=np.uint8)

    def add(self, value):
        lib().orc_llb_add(_ptr(self.registers), bytes(value), len(value))

    def add_hash(self, x):
        lib().orc_llb_add_hash(_ptr(self.registers), x)

    def merge(self, other):
        lib().orc_llb_merge(_ptr(self.registers), _ptr(other.registers))

    def cardinality(self):
        return lib().orc_llb_cardinality(_ptr(self.registers))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- single hist (KATs)
class Hist:
    def __init__(self, info_min, info_max, op="avg", hist_bucket=0, weight_col=False, loghist=False):
        new = lib().orc_hist_new_multi if loghist else lib().orc_hist_new
        self.h = new(info_min, info_max, AGG_HIST if op == "hist" else AGG_AVG, hist_bucket, 1 if weight_col else 0)

    def subhists(self):
        """-loghist: [(Info.Min, Info.Max, BucketSize, NumBuckets, len(Values), offset into values())]"""
        out = []
        for k in range(lib().orc_hist_n_sub(self.h)):
            six = np.zeros(6, dtype=np.int64)
            lib().orc_hist_sub(self.h, k, _ptr(six))
            out.append(tuple(int(x) for x in six))
        return out

    def sparse(self):
        """-loghist: the union of the sub-histograms' sparse buckets, {key: count}"""
        n = lib().orc_hist_sparse(self.h, None, None, 0)
        keys, counts = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
        lib().orc_hist_sparse(self.h, _ptr(keys), _ptr(counts), n)
        return dict(zip(keys[:n].tolist(), counts[:n].tolist()))

    def add(self, v, w=1):
        lib().orc_hist_add(self.h, int(v), int(w))

    def combine(self, other):
        lib().orc_hist_combine(self.h, other.h)

    def info(self):
        hi = HistInfo()
        lib().orc_hist_info_get(self.h, C.byref(hi))
        return hi.as_dict()

    def values(self):
        n = self.info()["n_values"]
        out = np.zeros(max(n, 1), dtype=np.int64)
        lib().orc_hist_values(self.h, _ptr(out), n)
        return out[:n]

    def percentiles(self):
        out = np.zeros(100, dtype=np.int64)
        n = lib().orc_hist_percentiles(self.h, _ptr(out))
        return out[:n]

    def outliers(self):
        out = np.zeros(1 << 16, dtype=np.int64)
        n = lib().orc_hist_outliers(self.h, _ptr(out), out.size)
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hist_free(self.h)
            self.h = None


def setup_buckets(info_min, info_max, hist_bucket=0):
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    lib().orc_setup_buckets(info_min, info_max, hist_bucket, C.byref(a), C.byref(b), C.byref(c))
    return {"bucket_size": a.value, "num_buckets": b.value, "n_values": c.value}


def percentiles_from_values(values, bucket_size, hmin, count):
    values = np.ascontiguousarray(values, dtype=np.int64)
    out = np.zeros(100, dtype=np.int64)
    n = lib().orc_percentiles_from_values(_ptr(values), values.size, bucket_size, hmin, count, _ptr(out))
    return out[:n]


def stddev_from_values(values, bucket_size, hmin, count, avg, outliers=(), underliers=()):
    values = np.ascontiguousarray(values, dtype=np.int64)
    o = np.ascontiguousarray(outliers, dtype=np.int64)
    u = np.ascontiguousarray(underliers, dtype=np.int64)
    return lib().orc_stddev_from_values(_ptr(values), values.size, bucket_size, hmin, count, avg,
                                        _ptr(o), o.size, _ptr(u), u.size)


def combine_avg(a, ca, b, cb):
    return lib().orc_combine_avg(a, ca, b, cb)


def time_bucket(t, bucket):
    return lib().orc_time_bucket(t, bucket)


def splitmix64(x):
    return lib().orc_splitmix64(x & 0xFFFFFFFFFFFFFFFF)


def synth_fill(kind, a, b, seed, col_index, row0, n, total_rows):
    out = np.empty(n, dtype=np.int64)
    lib().orc_synth_fill(kind, a, b, seed, col_index, row0, n, total_rows, _ptr(out))
    return out


def columnar_scan(fcols, ranges, gcols, gbounds, acols, ageom, n_threads=0):
    """The columnar CPU baseline (orc_columnar_scan): fcols / gcols / acols are lists of int64 arrays,
    ranges = [(lo, hi)] inclusive, gbounds = [(gmin, gcard)], ageom = [(hmin, bucket_size)].
    Returns (matched, table) with table[field][cell]: Count, then sum(v), sum(b), sum(b^2) per aggregation."""
    def ptrs(arrs):
        a = (C.c_void_p * max(len(arrs), 1))(*[x.ctypes.data for x in arrs])
        return a
    def vec(vals):
        return np.asarray(vals, dtype=np.int64)
    nrows = len((fcols + gcols + acols)[0])
    lo, hi = vec([r[0] for r in ranges]), vec([r[1] for r in ranges])
    gmin, gcard = vec([g[0] for g in gbounds]), vec([g[1] for g in gbounds])
    hmin, bs = vec([a[0] for a in ageom]), vec([a[1] for a in ageom])
    cells = int(np.prod(gcard)) if len(gbounds) else 1
    out = np.zeros((1 + 3 * len(acols), cells), dtype=np.int64)
    fp, gp, ap = ptrs(fcols), ptrs(gcols), ptrs(acols)
    m = lib().orc_columnar_scan(nrows, len(fcols), fp, _ptr(lo), _ptr(hi), len(gcols), gp, _ptr(gmin), _ptr(gcard), len(acols), ap,
                                _ptr(hmin), _ptr(bs), n_threads, _ptr(out))
    return m, out


def group_count_sum(key_cols, value_col, mask=None):
    """Exact group-by for high-cardinality checks (numpy; the map-based orc_query_run takes minutes per million groups):
    the reference's  result.Count++ / hist sum  per distinct key tuple (aggregate.go:125-143,186-203), with
    mean = sum / count.  key_cols: list of int64 arrays; returns (keys[n_groups][n_cols] in ascending tuple order,
    count[n_groups], sum[n_groups]) -- int64 throughout, so the sums are exact."""
    keys = np.stack([np.asarray(k, dtype=np.int64) for k in key_cols], axis=1)
    vals = np.asarray(value_col, dtype=np.int64)
    if mask is not None:
        keys, vals = keys[mask], vals[mask]
    order = np.lexsort(tuple(keys[:, c] for c in range(keys.shape[1] - 1, -1, -1)))
    keys, vals = keys[order], vals[order]
    if len(keys) == 0:
        return keys, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    new = np.ones(len(keys), dtype=bool)
    new[1:] = np.any(keys[1:] != keys[:-1], axis=1)
    starts = np.flatnonzero(new)
    counts = np.diff(np.append(starts, len(keys))).astype(np.int64)
    sums = np.add.reduceat(vals, starts).astype(np.int64)
    return keys[starts], counts, sums


# ---------------------------------------------------------------- full-size checker
class _SynCol(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col_index", C.c_int32), ("a", C.c_int64), ("b", C.c_int64)]


class _SynthQuery(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("total_rows", C.c_int64), ("row0", C.c_int64), ("nrows", C.c_int64),
                ("nf", C.c_int32), ("ng", C.c_int32), ("na", C.c_int32), ("has_time", C.c_int32),
                ("fcol", _SynCol * 4), ("gcol", _SynCol * 4), ("acol", _SynCol * 4), ("tcol", _SynCol),
                ("lo", C.c_int64 * 4), ("hi", C.c_int64 * 4), ("gmin", C.c_int64 * 4), ("gcard", C.c_int64 * 4),
                ("hmin", C.c_int64 * 4), ("bucket_size", C.c_int64 * 4), ("n_values", C.c_int64 * 4),
                ("time_bucket", C.c_int64), ("tb_min", C.c_int64), ("n_tb", C.c_int64), ("nv_max", C.c_int64),
                ("n_threads", C.c_int32), ("pad_", C.c_int32)]


def synth_scan(columns, seed, total_rows, row0, nrows, filters=(), groups=(), aggs=(), time_col=None, time_bucket=0,
               want_buckets=False, n_threads=0):
    """orc_synth_scan over rows [row0, row0 + nrows) of the synthetic table: `columns` maps a column name to
    (kind, col_index, a, b, info_min, info_max) as sybil_amd.synth.COLUMNS does (passed in: the oracle never
    imports the product package); filters = [(col, 'gt'|'lt'|'eq', value)] folded to inclusive ranges per
    column like IntFilter (filter.go:171-195); groups / aggs = column names; bucket geometry from
    orc_setup_buckets on the columns' IntInfo.  Returns a dict: matched, cells = (n_tb, gcard...),
    count[cell], sum[a][cell], sb[a][cell], sb2[a][cell], buckets[cell][a][nv_max] (want_buckets), tb_min."""
    q = _SynthQuery()
    q.seed, q.total_rows, q.row0, q.nrows = seed, total_rows, row0, nrows

    def syn(name):
        kind, idx, a, b = columns[name][:4]
        return _SynCol(kind, idx, a, b)

    def bounds(name):
        kind, idx, a, b = columns[name][:4]
        if kind == SYN_UNIFORM:
            return a, a + b - 1
        if kind == SYN_BELL:
            return a, a + 4 * (b - 1)
        return a, a + b - 1  # TIME: a + floor(i * b / N) < a + b

    ranges = {}
    for col, op, val in filters:
        lo, hi = ranges.get(col, (-(1 << 62), 1 << 62))
        if op == "gt":
            lo = max(lo, val + 1)
        elif op == "lt":
            hi = min(hi, val - 1)
        elif op == "eq":
            lo, hi = max(lo, val), min(hi, val)
        else:
            raise ValueError(op)
        ranges[col] = (lo, hi)
    q.nf = len(ranges)
    for i, (col, (lo, hi)) in enumerate(ranges.items()):
        q.fcol[i], q.lo[i], q.hi[i] = syn(col), lo, hi
    q.ng = len(groups)
    gcard = []
    for i, col in enumerate(groups):
        lo, hi = bounds(col)
        q.gcol[i], q.gmin[i], q.gcard[i] = syn(col), lo, hi - lo + 1
        gcard.append(hi - lo + 1)
    q.na = len(aggs)
    nv_max = 1
    for i, col in enumerate(aggs):
        info_min, info_max = columns[col][4], columns[col][5]
        geo = setup_buckets(info_min, info_max, 0)
        bs, nv = geo["bucket_size"], geo["n_values"]
        q.acol[i], q.hmin[i], q.bucket_size[i], q.n_values[i] = syn(col), info_min, bs, nv
        nv_max = max(nv_max, nv)
    q.nv_max = nv_max
    n_tb, tb_min = 1, 0
    if time_col:
        lo, hi = bounds(time_col)
        q.has_time, q.tcol, q.time_bucket = 1, syn(time_col), time_bucket
        tb_min = lib().orc_time_bucket(lo, time_bucket) // time_bucket
        n_tb = lib().orc_time_bucket(hi, time_bucket) // time_bucket - tb_min + 1
        q.tb_min, q.n_tb = tb_min, n_tb
    q.n_threads = n_threads
    cells = n_tb * int(np.prod(gcard)) if gcard else n_tb
    fields = np.zeros((1 + 3 * len(aggs), cells), dtype=np.int64)
    hist = np.zeros((cells, max(len(aggs), 1), nv_max), dtype=np.int64) if want_buckets else None
    m = lib().orc_synth_scan(C.byref(q), _ptr(fields), _ptr(hist) if want_buckets else None)
    out = {"matched": m, "cells": (n_tb,) + tuple(gcard), "tb_min": tb_min, "count": fields[0],
           "sum": [fields[1 + 3 * a] for a in range(len(aggs))], "sb": [fields[2 + 3 * a] for a in range(len(aggs))],
           "sb2": [fields[3 + 3 * a] for a in range(len(aggs))], "buckets": hist,
           "gmin": [q.gmin[i] for i in range(len(groups))], "bucket_size": [q.bucket_size[i] for i in range(len(aggs))],
           "n_values": [q.n_values[i] for i in range(len(aggs))], "hmin": [q.hmin[i] for i in range(len(aggs))]}
    return out


# ---------------------------------------------------------------- full query
def run_query(cols, filters=(), groups=(), aggs=(), op="avg", hist_bucket=0, time_col=-1, time_bucket=0,
              weight_col=-1, block_skip=False, block_rows=65536, n_threads=1, want_values=True, loghist=False,
              distincts=(), distinct_dicts=None, want_registers=False):
    """cols: list of dicts {type: 'int'|'str'|'set', data, populated(optional), offsets(set)}
    distincts: col indices of a count-distinct query (-distinct); distinct_dicts: {col index: [str, ...]} for str columns
    (the reference's slow path hashes the strings).  Every result then carries "distinct" (+ "registers").
    filters: list of (col_index, op_name, value[, idtable]); groups: col indices;
    aggs: list of (col_index, info_min, info_max).
    Returns a dict with matched, results, time_results, cumulative (canonical order)."""
    L = lib()
    keep = []
    ccols = (_Col * max(len(cols), 1))()
    nrows = None
    for i, c in enumerate(cols):
        t = c["type"]
        pop = c.get("populated")
        if pop is not None:
            pop = np.ascontiguousarray(pop, dtype=np.uint8)
            keep.append(pop)
        if t == "int":
            d = np.ascontiguousarray(c["data"], dtype=np.int64)
            keep.append(d)
            ccols[i] = _Col(INT_VAL, _ptr(d), None, None, None, _ptr(pop))
            n = d.size
        elif t == "str":
            d = np.ascontiguousarray(c["data"], dtype=np.int32)
            keep.append(d)
            ccols[i] = _Col(STR_VAL, None, _ptr(d), None, None, _ptr(pop))
            n = d.size
        elif t == "set":
            off = np.ascontiguousarray(c["offsets"], dtype=np.int64)
            d = np.ascontiguousarray(c["data"], dtype=np.int32)
            keep += [off, d]
            ccols[i] = _Col(SET_VAL, None, None, _ptr(off), _ptr(d), _ptr(pop))
            n = off.size - 1
        else:
            raise ValueError(t)
        nrows = n if nrows is None else nrows
        assert n == nrows, "ragged columns"
    nrows = nrows or 0

    cf = (_Filter * max(len(filters), 1))()
    for i, f in enumerate(filters):
        col, opn, val = f[0], f[1], f[2]
        idt = None
        if len(f) > 3 and f[3] is not None:
            idt = np.ascontiguousarray(f[3], dtype=np.uint8)
            keep.append(idt)
        cf[i] = _Filter(col, OPS[opn] if isinstance(opn, str) else opn, int(val), _ptr(idt),
                        idt.size if idt is not None else 0)
    q = _Query()
    q.n_filters = len(filters)
    q.filters = C.cast(cf, C.POINTER(_Filter))
    q.n_groups = len(groups)
    for i, g in enumerate(groups):
        q.group_cols[i] = g
    q.n_aggs = len(aggs)
    for i, a in enumerate(aggs):
        q.aggs[i] = _Agg(a[0], int(a[1]), int(a[2]))
    q.op = AGG_HIST if op == "hist" else AGG_AVG
    q.hist_bucket = hist_bucket
    q.time_col = time_col
    q.time_bucket = time_bucket
    q.weight_col = weight_col
    q.block_skip = 1 if block_skip else 0
    q.block_rows = block_rows
    q.n_threads = n_threads
    q.loghist = 1 if loghist else 0
    q.n_distincts = len(distincts)
    for i, d in enumerate(distincts):
        q.distinct_cols[i] = d
        strs = (distinct_dicts or {}).get(d)
        if strs is not None:
            arr = (C.c_char_p * max(len(strs), 1))(*[x.encode() if isinstance(x, str) else x for x in strs])
            keep.append(arr)
            q.distinct_dicts[i] = C.cast(arr, C.POINTER(C.c_char_p))
            q.distinct_dict_len[i] = len(strs)

    R = L.orc_query_run(C.byref(q), ccols, len(cols), nrows)
    try:
        out = {"matched": L.orc_matched_count(R), "blocks_scanned": L.orc_blocks_scanned(R),
               "blocks_skipped": L.orc_blocks_skipped(R)}
        ng, na = len(groups), len(aggs)
        for which, name in ((0, "results"), (1, "time_results"), (2, "cumulative")):
            lst = []
            for idx in range(L.orc_num_results(R, which)):
                key = (C.c_uint8 * max(8 * ng, 1))()
                tb, cnt, smp = C.c_int64(), C.c_int64(), C.c_int64()
                L.orc_result_get(R, which, idx, key, C.byref(tb), C.byref(cnt), C.byref(smp))
                kb = bytes(key)[:8 * ng]
                r = {"key": kb, "key_vals": tuple(int.from_bytes(kb[8 * g:8 * g + 8], "little") for g in range(ng)),
                     "time_bucket": tb.value, "count": cnt.value, "samples": smp.value, "hists": []}
                if distincts:
                    regs = np.zeros(LLB_M, dtype=np.uint8) if want_registers else None
                    r["distinct"] = L.orc_result_distinct(R, which, idx, _ptr(regs))
                    if want_registers:
                        r["registers"] = regs
                for a in range(na):
                    hi = HistInfo()
                    L.orc_result_hist(R, which, idx, a, C.byref(hi))
                    h = hi.as_dict()
                    if h["present"] and h["percentile_mode"]:
                        if want_values:
                            v = np.zeros(max(h["n_values"], 1), dtype=np.int64)
                            L.orc_result_hist_values(R, which, idx, a, _ptr(v), v.size)
                            h["values"] = v[:h["n_values"]]
                        p = np.zeros(100, dtype=np.int64)
                        n = L.orc_result_percentiles(R, which, idx, a, _ptr(p))
                        h["percentiles"] = p[:max(n, 0)]
                        no = h["n_outliers"] + h["n_underliers"]
                        ov = np.zeros(max(no, 1), dtype=np.int64)
                        got = L.orc_result_outliers(R, which, idx, a, _ptr(ov), ov.size)
                        h["outlier_values"] = ov[:max(min(got, no), 0)]
                        ns = L.orc_result_n_sub(R, which, idx, a)
                        if ns > 0:  # -loghist
                            h["subhists"] = []
                            for k in range(ns):
                                six = np.zeros(6, dtype=np.int64)
                                L.orc_result_sub(R, which, idx, a, k, _ptr(six))
                                h["subhists"].append(tuple(int(x) for x in six))
                            nk = L.orc_result_sparse(R, which, idx, a, None, None, 0)
                            keys, counts = np.zeros(max(nk, 1), dtype=np.int64), np.zeros(max(nk, 1), dtype=np.int64)
                            L.orc_result_sparse(R, which, idx, a, _ptr(keys), _ptr(counts), nk)
                            h["sparse"] = dict(zip(keys[:nk].tolist(), counts[:nk].tolist()))
                    r["hists"].append(h)
                lst.append(r)
            out[name] = lst[0] if which == 2 else lst
        return out
    finally:
        L.orc_results_free(R)
