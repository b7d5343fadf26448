"""Host-side mirror of the reference's query surface over the C ABI.

Names follow logv/sybil (src/lib): a Table holds columns, a query is described by
filters (`IntFilter`/`StrFilter`/`SetFilter`, filter.go:143-168), `Grouping`s and
`Aggregation`s (query_spec.go:73-83) plus the FLAGS the hot loop reads
(op avg|hist, -int-bucket, -time/-time-col/-time-bucket, -weight-col, -sort, -limit),
and `LoadAndQueryRecords` (table_query.go:18) becomes prepare -> scan -> [all-reduce] ->
finalize.  Everything computes on the GPU through libsybilgpu.so; nothing here falls
back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else s


class Context:
    """One GPU (one process per GPU under RCCL)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        N.check(N.lib().sybl_init(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            N.lib().sybl_shutdown(self._h)
            self._h = C.c_void_p()

    def set_stream(self, hip_stream_ptr):
        N.check(N.lib().sybl_ctx_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def sync(self):
        N.check(N.lib().sybl_ctx_sync(self._h))

    def trim(self):
        """Frees the loader's staging arena (sybl_ctx_trim); the next open / refresh allocates it again."""
        N.check(N.lib().sybl_ctx_trim(self._h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_int64()
        N.check(N.lib().sybl_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "n_cus": cus.value, "hbm_bytes": hbm.value}

    # ---- RCCL inside the library (hosts without a collective runtime of their own)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        N.check(N.lib().sybl_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, nranks, rank):
        N.check(N.lib().sybl_comm_init(self._h, unique_id, nranks, rank))

    def comm_free(self):
        N.check(N.lib().sybl_comm_free(self._h))

    def comm_info(self):
        r, n = C.c_int32(), C.c_int32()
        N.check(N.lib().sybl_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def create_table(self, name):
        h = C.c_void_p()
        N.check(N.lib().sybl_table_create(self._h, _b(name), C.byref(h)))
        return Table(self, h, name)

    def synth_table(self, name, seed, total_rows, row0, nrows, cols):
        """cols: list of dicts {name, kind, col_index, a, b, info_min, info_max}."""
        arr = (N.SynthCol * len(cols))()
        keep = []
        for i, c in enumerate(cols):
            nm = _b(c["name"])
            keep.append(nm)
            arr[i] = N.SynthCol(nm, c["kind"], c["col_index"], c["a"], c["b"], c.get("info_min", 1), c.get("info_max", 0))
        h = C.c_void_p()
        N.check(N.lib().sybl_table_create_synth(self._h, _b(name), seed, total_rows, row0, nrows, len(cols), arr,
                                                C.byref(h)))
        return Table(self, h, name)

    def open_table(self, directory, table, columns=None, rank=0, nranks=1, compact=False):
        h = C.c_void_p()
        if columns:
            names = [_b(c) for c in columns]
            arr = (C.c_char_p * len(names))(*names)
            n = len(names)
        else:
            arr, n = None, 0
        N.check(N.lib().sybl_table_open_flags(self._h, _b(directory), _b(table), arr, n, rank, nranks, 1 if compact else 0,
                                              C.byref(h)))
        return Table(self, h, table)


class Table:
    def __init__(self, ctx, handle, name):
        self.ctx, self._h, self.name = ctx, handle, name

    def free(self):
        if self._h:
            N.lib().sybl_table_free(self._h)
            self._h = C.c_void_p()

    def add_column(self, name, type="int", info_min=1, info_max=0):
        t = {"int": N.INT_VAL, "str": N.STR_VAL, "set": N.SET_VAL}[type]
        N.check(N.lib().sybl_table_add_column(self._h, _b(name), t, info_min, info_max))

    def append_block(self, nrows, columns):
        """columns: {name: spec}.  int: ndarray or (ndarray, populated);  str: dict(ids=, strings=,
        populated=);  set: dict(offsets=, ids=, strings=, populated=)."""
        views = (N.ColView * max(len(columns), 1))()
        keep = []
        for i, (name, spec) in enumerate(columns.items()):
            nm = _b(name)
            keep.append(nm)
            v = N.ColView()
            v.name = nm
            pop = None
            if isinstance(spec, dict):
                pop = spec.get("populated")
                strings = [_b(s) for s in spec["strings"]]
                sarr = (C.c_char_p * max(len(strings), 1))(*strings)
                keep += [strings, sarr]
                v.strings = C.cast(sarr, C.POINTER(C.c_char_p))
                v.n_strings = len(strings)
                ids = np.ascontiguousarray(spec["ids"], dtype=np.int32)
                keep.append(ids)
                if "offsets" in spec:
                    off = np.ascontiguousarray(spec["offsets"], dtype=np.int64)
                    keep.append(off)
                    v.type = N.SET_VAL
                    v.set_off = off.ctypes.data
                    v.set_ids = ids.ctypes.data
                else:
                    v.type = N.STR_VAL
                    v.str_ids = ids.ctypes.data
            else:
                if isinstance(spec, tuple):
                    spec, pop = spec
                d = np.ascontiguousarray(spec, dtype=np.int64)
                keep.append(d)
                v.type = N.INT_VAL
                v.ints = d.ctypes.data
            if pop is not None:
                p = np.ascontiguousarray(pop, dtype=np.uint8)
                keep.append(p)
                v.populated = p.ctypes.data
            views[i] = v
        N.check(N.lib().sybl_table_append_block(self._h, nrows, len(columns), views))

    @property
    def rows(self):
        return N.lib().sybl_table_rows(self._h)

    @property
    def blocks(self):
        return N.lib().sybl_table_blocks(self._h)

    @property
    def broken_blocks(self):
        return N.lib().sybl_table_broken_blocks(self._h)

    def load_stats(self):
        """Where the time of the sybl_table_open that built this table went (sybl_table_load_stats)."""
        st = N.LoadStats()
        N.check(N.lib().sybl_table_load_stats(self._h, C.byref(st)))
        return st.as_dict()

    @property
    def hbm_bytes(self):
        return N.lib().sybl_table_hbm_bytes(self._h)

    def column_info(self, name):
        t, hm = C.c_int(), C.c_int()
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        N.check(N.lib().sybl_table_column_info(self._h, _b(name), C.byref(t), C.byref(a), C.byref(b), C.byref(c),
                                               C.byref(d), C.byref(hm)))
        return {"type": t.value, "exact_min": a.value, "exact_max": b.value, "info_min": c.value,
                "info_max": d.value, "has_missing": bool(hm.value)}

    def set_bounds(self, name, lo, hi, has_missing=False):
        N.check(N.lib().sybl_table_set_bounds(self._h, _b(name), lo, hi, 1 if has_missing else 0))

    def agree(self, group_cols=()):
        """COLLECTIVE over the ctx's communicator (sybl_table_agree): bounds, has_missing, str / set dictionaries and the
        group dictionaries of the sparse int keys among group_cols, identical on every rank afterwards."""
        arr = (C.c_char_p * max(1, len(group_cols)))(*[_b(g) for g in group_cols])
        N.check(N.lib().sybl_table_agree(self._h, arr if group_cols else None, len(group_cols)))

    def column_distinct(self, name):
        vals, n = C.POINTER(C.c_int64)(), C.c_int64()
        N.check(N.lib().sybl_table_column_distinct(self._h, _b(name), C.byref(vals), C.byref(n)))
        return np.ctypeslib.as_array(vals, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)

    def set_group_dict(self, name, values):
        v = np.ascontiguousarray(values, dtype=np.int64)
        N.check(N.lib().sybl_table_set_group_dict(self._h, _b(name), v.ctypes.data, v.size))

    def column_dict(self, name):
        arr, n = C.POINTER(C.c_char_p)(), C.c_int64()
        N.check(N.lib().sybl_table_column_dict(self._h, _b(name), C.byref(arr), C.byref(n)))
        return [arr[i].decode("utf-8", "replace") for i in range(n.value)]

    def set_dict(self, name, strings):
        bs = [_b(s) for s in strings]
        arr = (C.c_char_p * max(len(bs), 1))(*bs)
        N.check(N.lib().sybl_table_set_dict(self._h, _b(name), arr, len(bs)))

    def save(self, directory):
        """Write the table under directory/<name>/ in the reference's on-disk format (sybl_table_save)."""
        N.check(N.lib().sybl_table_save(self._h, _b(directory)))

    def refresh(self):
        """Follow the directory the table was opened from (sybl_table_refresh): returns (added, dropped, reloaded) blocks."""
        a, d, r = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(N.lib().sybl_table_refresh(self._h, C.byref(a), C.byref(d), C.byref(r)))
        return a.value, d.value, r.value

    def compact(self):
        """Re-encode every int / str column at the narrowest width that holds max - min
        (sybl_table_compact); query results are unchanged, scans stream fewer bytes."""
        N.check(N.lib().sybl_table_compact(self._h))
        return self

    def column_storage(self, name):
        """(bytes per stored value, value base) of a column as laid out in HBM."""
        w, b = C.c_int32(0), C.c_int64(0)
        N.check(N.lib().sybl_table_column_storage(self._h, _b(name), C.byref(w), C.byref(b)))
        return int(w.value), int(b.value)

    def read_int(self, name, row0, n):
        out = np.empty(n, dtype=np.int64)
        N.check(N.lib().sybl_table_read_int(self._h, _b(name), row0, n, out.ctypes.data))
        return out

    # reference-named constructors (filter.go:287-318, query_spec.go:214-219)
    @staticmethod
    def IntFilter(col, op, value):
        return (col, op, int(value))

    @staticmethod
    def StrFilter(col, op, value):
        return (col, op, str(value))

    @staticmethod
    def SetFilter(col, op, value):
        return (col, op, str(value))

    def query(self, filters=(), groups=(), aggs=(), op="avg", hist_bucket=0, want_percentiles=True, time_col=None,
              time_bucket=0, weight_col=None, order_by="$COUNT", order_asc=False, limit=0, block_skip=False, loghist=False, str_replace=(),
              distincts=(), printed_only=False):
        """str_replace: [(col, pattern, replacement)] or [(col, [replaced string per dictionary id])] (-str-replace).
        distincts: columns of a count-distinct query (-distinct): rows then carry "distinct" (Result.distinct()).
        printed_only: the caller is a printer (sybl_query_desc.printed_only): with limit > 0 and many histogram groups only the
        first `limit` rows of the sort order and Cumulative carry percentiles / stddev / bucket arrays."""
        keep = []
        farr = (N.Filter * max(len(filters), 1))()
        for i, f in enumerate(filters):
            col, opn, val = f[0], f[1], f[2]
            ff = N.Filter()
            ff.col = _b(col)
            ff.op = N.OPS[opn]
            if isinstance(val, (int, np.integer)):
                ff.int_value = int(val)
            else:
                ff.str_value = _b(val)
            if len(f) > 3 and f[3] is not None:
                m = np.ascontiguousarray(f[3], dtype=np.uint8)
                keep.append(m)
                ff.id_match = m.ctypes.data
                ff.id_match_len = m.size
            farr[i] = ff
        g = [_b(x) for x in groups]
        a = [_b(x) for x in aggs]
        garr = (C.c_char_p * max(len(g), 1))(*g)
        aarr = (C.c_char_p * max(len(a), 1))(*a)
        d = N.QueryDesc()
        d.n_filters, d.filters = len(filters), C.cast(farr, C.POINTER(N.Filter))
        d.n_groups, d.groups = len(g), C.cast(garr, C.POINTER(C.c_char_p))
        d.n_aggs, d.aggs = len(a), C.cast(aarr, C.POINTER(C.c_char_p))
        d.op = N.AGG_HIST if op == "hist" else N.AGG_AVG
        d.hist_bucket = hist_bucket
        d.want_percentiles = 1 if want_percentiles else 0
        d.time_col = _b(time_col) if time_col else None
        d.time_bucket = time_bucket
        d.weight_col = _b(weight_col) if weight_col else None
        d.order_by = _b(order_by) if order_by else None
        d.order_asc = 1 if order_asc else 0
        d.limit = limit
        d.block_skip = 1 if block_skip else 0
        d.loghist = 1 if loghist else 0
        sarr = (N.StrReplace * max(len(str_replace), 1))()
        for i, sr in enumerate(str_replace):
            sarr[i].col = _b(sr[0])
            if len(sr) == 2:
                strs = [_b(x) for x in sr[1]]
                arr = (C.c_char_p * max(len(strs), 1))(*strs)
                keep += [strs, arr]
                sarr[i].replaced = C.cast(arr, C.POINTER(C.c_char_p))
                sarr[i].n_replaced = len(strs)
            else:
                sarr[i].pattern, sarr[i].replace = _b(sr[1]), _b(sr[2])
        d.n_str_replace, d.str_replace = len(str_replace), C.cast(sarr, C.POINTER(N.StrReplace))
        dn = [_b(x) for x in distincts]
        darr = (C.c_char_p * max(len(dn), 1))(*dn)
        d.n_distincts, d.distincts = len(dn), C.cast(darr, C.POINTER(C.c_char_p))
        d.printed_only = int(printed_only)  # (False / True / 2: the rows beyond the limit need their Count only -- limit pushdown)
        h = C.c_void_p()
        N.check(N.lib().sybl_query_prepare(self._h, C.byref(d), C.byref(h)))
        qy = Query(self, h, list(groups), list(aggs))
        qy.n_distincts = len(dn)
        return qy


class Query:
    def __init__(self, table, handle, groups, aggs):
        self.table, self._h, self.groups, self.aggs = table, handle, groups, aggs
        self._bound = None
        self._has_max = None

    def free(self):
        if self._h:
            N.lib().sybl_query_free(self._h)
            self._h = C.c_void_p()

    def scan(self):
        N.check(N.lib().sybl_query_scan(self._h))
        return self

    def partial_sizes(self):
        ns, nm = C.c_int64(), C.c_int64()
        N.check(N.lib().sybl_query_partials(self._h, None, C.byref(ns), None, C.byref(nm)))
        return ns.value, nm.value

    def partials(self):
        ps, pm = C.c_void_p(), C.c_void_p()
        ns, nm = C.c_int64(), C.c_int64()
        N.check(N.lib().sybl_query_partials(self._h, C.byref(ps), C.byref(ns), C.byref(pm), C.byref(nm)))
        return ps.value, ns.value, pm.value, nm.value

    def bind_partials(self, d_sum_ptr, d_max_ptr):
        N.check(N.lib().sybl_query_bind_partials(self._h, C.c_void_p(d_sum_ptr), C.c_void_p(d_max_ptr)))

    def bind_torch(self, device):
        """Allocates the partial tables as torch tensors (so torch.distributed can all-reduce
        them over RCCL) and makes the scan write into them."""
        import torch
        ns, nm = self.partial_sizes()
        s = torch.zeros(ns, dtype=torch.int64, device=device)
        m = torch.zeros(nm, dtype=torch.int64, device=device)
        torch.cuda.synchronize(device)  # the fills ran on torch's stream, the scan runs on the ctx stream
        self.bind_partials(s.data_ptr(), m.data_ptr())
        self._bound = (s, m)
        return s, m

    def allreduce_torch(self, group=None):
        """The one collective on the path: SUM over counts/sums/buckets, MAX over extrema.
        torch.distributed orders the collective against torch's CURRENT stream: call this inside
        `with torch.cuda.stream(s)` for the stream `s` the Context was pointed at (Context.set_stream),
        or synchronise the Context first."""
        from . import dist as sdist
        s, m = self._bound
        if self._has_max is None:
            self._has_max = self.stats()["n_max_fields"] > 0
        sdist.merge_partials(s, m, group=group, has_max=self._has_max)

    def allreduce(self):
        N.check(N.lib().sybl_query_allreduce(self._h))

    def hash_keys(self):
        """Hash group-by: the ascending composite keys of the groups this rank found (sybl_query_hash_keys); an empty
        array for a direct-mapped query."""
        keys, n = C.POINTER(C.c_uint64)(), C.c_int64()
        N.check(N.lib().sybl_query_hash_keys(self._h, C.byref(keys), C.byref(n)))
        return np.ctypeslib.as_array(keys, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint64)

    def hash_install_union(self, keys):
        """Re-lays this rank's dense partial arrays out over the sorted union of every rank's keys."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        N.check(N.lib().sybl_query_hash_install_union(self._h, k.ctypes.data, k.size))

    def partials_torch(self, device):
        """The library-owned partial tables as torch tensors sharing the memory (hash group-by: the dense arrays, which
        cannot be bound to caller tensors because their size follows the keys found)."""
        import torch
        ps, ns, pm, nm = self.partials()

        class _Dev:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}
        return (torch.as_tensor(_Dev(ps, ns), device=device), torch.as_tensor(_Dev(pm, nm), device=device))

    def collective_finalize(self):
        """True when the last allreduce() scattered the bucket arrays over the ranks: snapshot() and finalize() are
        then collective calls (every rank makes them and gets the full result)."""
        return bool(N.lib().sybl_query_collective_finalize(self._h))

    def stats(self):
        st = N.RunStats()
        N.check(N.lib().sybl_query_stats(self._h, C.byref(st)))
        return st.as_dict()

    def debug_cells(self, which, agg=0):
        """Test hook (sybl_debug_query_cells): the exact per-cell integers behind the last finalized result;
        which = "count" | "sum" | "sb" | "sb2"."""
        w = {"count": 0, "sum": 1, "sb": 2, "sb2": 3}[which]
        n = C.c_int64(0)
        probe = np.zeros(1, dtype=np.int64)
        N.check(N.lib().sybl_debug_query_cells(self._h, w, agg, probe.ctypes.data, 1, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int64)
        N.check(N.lib().sybl_debug_query_cells(self._h, w, agg, out.ctypes.data, out.size, C.byref(n)))
        return out

    def snapshot(self):
        """Enqueue the device -> host copy of the (reduced) partial tables; finalize() then waits for
        that copy only, so the scan of another query can run underneath it."""
        N.check(N.lib().sybl_query_snapshot(self._h))
        return self

    def finalize(self):
        h = C.c_void_p()
        N.check(N.lib().sybl_query_finalize(self._h, C.byref(h)))
        res = Result(h, len(self.groups), len(self.aggs))
        res.has_distinct = getattr(self, "n_distincts", 0) > 0
        return res

    def run(self):
        return self.scan().finalize()


class Result:
    def __init__(self, handle, n_groups, n_aggs):
        self._h, self.n_groups, self.n_aggs = handle, n_groups, n_aggs
        self.has_distinct = False

    def free(self):
        if self._h:
            N.lib().sybl_result_free(self._h)
            self._h = C.c_void_p()

    @property
    def matched(self):
        return N.lib().sybl_result_matched(self._h)

    def materialize(self, which=0):
        """sybl_result_rows without converting anything: the number of rows (a big result builds its rows on this first call)."""
        rows = C.POINTER(N.GroupRow)()
        n = C.c_int64()
        N.check(N.lib().sybl_result_rows(self._h, which, C.byref(rows), C.byref(n)))
        return n.value

    def rows(self, which=0, want_values=True):
        rows = C.POINTER(N.GroupRow)()
        n = C.c_int64()
        N.check(N.lib().sybl_result_rows(self._h, which, C.byref(rows), C.byref(n)))
        out = []
        for i in range(n.value):
            r = rows[i]
            kb = bytes(bytearray(r.binary_key[k] for k in range(8 * self.n_groups))) if self.n_groups and which != 2 else b""
            row = {"key": kb,
                   "key_vals": tuple(int.from_bytes(kb[8 * g:8 * g + 8], "little") for g in range(len(kb) // 8)),
                   "group_by_key": r.group_by_key.decode("utf-8", "replace"),
                   "time_bucket": r.time_bucket, "count": r.count, "samples": r.samples, "hists": []}
            for a in range(self.n_aggs):
                g = r.aggs[a]
                h = {k: getattr(g, k) for k, _ in N.AggOut._fields_ if k not in ("values", "percentiles", "outlier_values")}
                if g.n_outlier_values > 0:
                    h["outlier_values"] = np.ctypeslib.as_array(g.outlier_values, shape=(g.n_outlier_values,)).copy()
                if want_values and g.values:
                    h["values"] = np.ctypeslib.as_array(g.values, shape=(g.n_values,)).copy()
                if g.percentiles:
                    h["percentiles"] = np.ctypeslib.as_array(g.percentiles, shape=(100,)).copy()
                row["hists"].append(h)
            if self.has_distinct:
                row["distinct"] = self.distinct(which, i)
            out.append(row)
        return out

    def distinct(self, which=0, row=0, registers=False):
        """Count-distinct queries: Result.Distinct.Cardinality() of a row of rows(which) (+ the sketch's registers)."""
        card, regs = C.c_int64(), C.POINTER(C.c_uint8)()
        N.check(N.lib().sybl_result_distinct(self._h, which, row, C.byref(card), C.byref(regs)))
        if registers:
            return card.value, np.ctypeslib.as_array(regs, shape=(16384,)).copy()
        return card.value

    def subhists(self, agg=0):
        """-loghist: the layout of the aggregation's `values` arrays (sybl_result_subhists) as a list of dicts."""
        subs, n = C.POINTER(N.SubHist)(), C.c_int64()
        N.check(N.lib().sybl_result_subhists(self._h, agg, C.byref(subs), C.byref(n)))
        return [{k: getattr(subs[i], k) for k, _ in N.SubHist._fields_} for i in range(n.value)]

    @property
    def results(self):
        return self.rows(0)

    @property
    def time_results(self):
        return self.rows(1)

    @property
    def cumulative(self):
        return self.rows(2)[0]

    def encode(self):
        """`-encode-results`: the gob-encoded NodeResults (bytes)."""
        n = C.c_int64(0)
        p = N.lib().sybl_result_encode(self._h, C.byref(n))
        if not p:
            raise N.SyblError(N.E_INVAL, (N.lib().sybl_last_error() or b"").decode())
        return C.string_at(p, n.value)

    def render(self, fmt="text"):
        s = N.lib().sybl_result_render(self._h, 1 if fmt == "json" else 0)
        if s is None:
            raise N.SyblError(N.E_INVAL, (N.lib().sybl_last_error() or b"").decode())
        return s.decode("utf-8", "replace")
