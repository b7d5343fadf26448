"""ctypes view of the C ABI in include/sybilgpu.h (libsybilgpu.so).

The library is built in-tree by `__graft_entry__.build()` / `make -C sybil_amd/csrc`.
There is no CPU fallback: if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The library reads its diagnostic switches (SYBL_*) from the environment once and answers from that snapshot afterwards (a
# cgo host may setenv from other threads); the test suite and the tools flip switches between queries: live reads for them.
os.environ.setdefault("SYBL_ENV_LIVE", "1")
# (SYBL_LIBRARY: another build of the same library, for same-box A/B timing runs)
LIB_PATH = os.environ.get("SYBL_LIBRARY") or os.path.join(_HERE, "libsybilgpu.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sybilgpu.h")

OK, E_INVAL, E_NODEVICE, E_NOMEM, E_IO, E_STATE, E_BLOCK = 0, -1, -2, -3, -4, -5, -6
NO_VAL, INT_VAL, STR_VAL, SET_VAL = 0, 1, 2, 3
OPS = {"gt": 0, "lt": 1, "eq": 2, "neq": 3, "re": 4, "nre": 5, "in": 6, "nin": 7}
AGG_AVG, AGG_HIST = 0, 1
SYN_UNIFORM, SYN_TIME, SYN_BELL = 0, 1, 2
MAX_GROUPS, MAX_AGGS, MAX_FILTERS = 8, 6, 16


class SyblError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sybilgpu error %d: %s" % (code, msg))
        self.code = code


class ColView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("ints", C.c_void_p), ("str_ids", C.c_void_p),
                ("set_off", C.c_void_p), ("set_ids", C.c_void_p), ("populated", C.c_void_p),
                ("strings", C.POINTER(C.c_char_p)), ("n_strings", C.c_int32)]


class SynthCol(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("col_index", C.c_int32), ("a", C.c_int64),
                ("b", C.c_int64), ("info_min", C.c_int64), ("info_max", C.c_int64)]


class Filter(C.Structure):
    _fields_ = [("col", C.c_char_p), ("op", C.c_int32), ("int_value", C.c_int64), ("str_value", C.c_char_p),
                ("id_match", C.c_void_p), ("id_match_len", C.c_int64)]


class StrReplace(C.Structure):
    _fields_ = [("col", C.c_char_p), ("pattern", C.c_char_p), ("replace", C.c_char_p), ("replaced", C.POINTER(C.c_char_p)),
                ("n_replaced", C.c_int64)]


class QueryDesc(C.Structure):
    _fields_ = [("n_filters", C.c_int32), ("filters", C.POINTER(Filter)),
                ("n_groups", C.c_int32), ("groups", C.POINTER(C.c_char_p)),
                ("n_aggs", C.c_int32), ("aggs", C.POINTER(C.c_char_p)),
                ("op", C.c_int32), ("hist_bucket", C.c_int64), ("want_percentiles", C.c_int32),
                ("time_col", C.c_char_p), ("time_bucket", C.c_int64), ("weight_col", C.c_char_p),
                ("order_by", C.c_char_p), ("order_asc", C.c_int32), ("limit", C.c_int32),
                ("block_skip", C.c_int32), ("loghist", C.c_int32), ("n_str_replace", C.c_int32), ("str_replace", C.POINTER(StrReplace)),
                ("n_distincts", C.c_int32), ("distincts", C.POINTER(C.c_char_p)), ("printed_only", C.c_int32)]


class AggOut(C.Structure):
    _fields_ = [("present", C.c_int32), ("count", C.c_int64), ("samples", C.c_int64), ("sum", C.c_int64),
                ("avg", C.c_double), ("stddev", C.c_double), ("min", C.c_int64), ("max", C.c_int64),
                ("bucket_size", C.c_int64), ("num_buckets", C.c_int64), ("n_values", C.c_int64),
                ("values", C.POINTER(C.c_int64)), ("percentiles", C.POINTER(C.c_int64)),
                ("n_outliers", C.c_int64), ("outlier_values", C.POINTER(C.c_int64)), ("n_outlier_values", C.c_int64)]


class SubHist(C.Structure):
    _fields_ = [("info_min", C.c_int64), ("info_max", C.c_int64), ("bucket_size", C.c_int64), ("num_buckets", C.c_int64),
                ("n_values", C.c_int64), ("offset", C.c_int64), ("ext_first", C.c_int64), ("n_ext", C.c_int64),
                ("ext_offset", C.c_int64)]


class GroupRow(C.Structure):
    _fields_ = [("binary_key", C.POINTER(C.c_uint8)), ("group_by_key", C.c_char_p), ("time_bucket", C.c_int64),
                ("count", C.c_int64), ("samples", C.c_int64), ("aggs", C.POINTER(AggOut))]


class LoadStats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("parse_cpu_s", C.c_double), ("wait_s", C.c_double), ("apply_s", C.c_double),
                ("file_bytes", C.c_int64), ("h2d_bytes", C.c_int64), ("workers", C.c_int32), ("blocks", C.c_int32),
                ("gpu_varint_cols", C.c_int32), ("gpu_varint_redone", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RunStats(C.Structure):
    _fields_ = [("rows_scanned", C.c_int64), ("blocks_scanned", C.c_int64), ("blocks_skipped", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("canonical_bytes", C.c_int64), ("scan_ms", C.c_double), ("reduce_ms", C.c_double),
                ("n_cells", C.c_int32), ("strategy", C.c_int32), ("lds_bytes", C.c_int32),
                ("n_workgroups", C.c_int32), ("replicas", C.c_int32), ("n_sum_fields", C.c_int32),
                ("n_max_fields", C.c_int32), ("packed_kernel", C.c_int32), ("count_pass_reused", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/sybilgpu.h declares: (restype, argtypes)
P = C.c_void_p
SIGNATURES = {
    "sybl_abi_version": (C.c_int, []),
    "sybl_last_error": (C.c_char_p, []),
    "sybl_init": (C.c_int, [C.c_int, C.POINTER(P)]),
    "sybl_shutdown": (None, [P]),
    "sybl_ctx_set_stream": (C.c_int, [P, P]),
    "sybl_ctx_sync": (C.c_int, [P]),
    "sybl_ctx_trim": (C.c_int, [P]),
    "sybl_device_info": (C.c_int, [P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "sybl_table_create": (C.c_int, [P, C.c_char_p, C.POINTER(P)]),
    "sybl_table_free": (None, [P]),
    "sybl_table_add_column": (C.c_int, [P, C.c_char_p, C.c_int, C.c_int64, C.c_int64]),
    "sybl_table_append_block": (C.c_int, [P, C.c_int64, C.c_int32, C.POINTER(ColView)]),
    "sybl_table_create_synth": (C.c_int, [P, C.c_char_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                          C.POINTER(SynthCol), C.POINTER(P)]),
    "sybl_table_open": (C.c_int, [P, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(P)]),
    "sybl_table_open_flags": (C.c_int, [P, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.POINTER(P)]),
    "sybl_table_save": (C.c_int, [P, C.c_char_p]),
    "sybl_table_refresh": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sybl_debug_encode_column": (C.c_void_p, [C.c_int, C.c_char_p, P, P, C.c_int64, C.POINTER(C.c_char_p), C.c_int64,
                                              C.POINTER(C.c_int64)]),
    "sybl_table_broken_blocks": (C.c_int64, [P]),
    "sybl_table_load_stats": (C.c_int, [P, C.POINTER(LoadStats)]),
    "sybl_debug_gob_to_json": (C.c_char_p, [C.c_char_p]),
    "sybl_debug_block_layout": (C.c_char_p, [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32]),
    "sybl_debug_regex_match": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64]),
    "sybl_debug_regex_replace": (C.c_char_p, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "sybl_table_rows": (C.c_int64, [P]),
    "sybl_table_blocks": (C.c_int64, [P]),
    "sybl_table_hbm_bytes": (C.c_int64, [P]),
    "sybl_table_column_info": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_int)] + [C.POINTER(C.c_int64)] * 4
                               + [C.POINTER(C.c_int)]),
    "sybl_table_set_bounds": (C.c_int, [P, C.c_char_p, C.c_int64, C.c_int64, C.c_int]),
    "sybl_table_read_int": (C.c_int, [P, C.c_char_p, C.c_int64, C.c_int64, P]),
    "sybl_table_compact": (C.c_int, [P]),
    "sybl_table_column_storage": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "sybl_table_column_distinct": (C.c_int, [P, C.c_char_p, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    "sybl_table_set_group_dict": (C.c_int, [P, C.c_char_p, P, C.c_int64]),
    "sybl_table_column_dict": (C.c_int, [P, C.c_char_p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_int64)]),
    "sybl_table_set_dict": (C.c_int, [P, C.c_char_p, C.POINTER(C.c_char_p), C.c_int64]),
    "sybl_query_prepare": (C.c_int, [P, C.POINTER(QueryDesc), C.POINTER(P)]),
    "sybl_query_free": (None, [P]),
    "sybl_query_scan": (C.c_int, [P]),
    "sybl_query_partials": (C.c_int, [P, C.POINTER(P), C.POINTER(C.c_int64), C.POINTER(P), C.POINTER(C.c_int64)]),
    "sybl_query_bind_partials": (C.c_int, [P, P, P]),
    "sybl_comm_unique_id": (C.c_int, [P]),
    "sybl_comm_init": (C.c_int, [P, P, C.c_int32, C.c_int32]),
    "sybl_comm_free": (C.c_int, [P]),
    "sybl_query_allreduce": (C.c_int, [P]),
    "sybl_comm_info": (C.c_int, [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sybl_table_agree": (C.c_int, [P, C.POINTER(C.c_char_p), C.c_int32]),
    "sybl_query_hash_keys": (C.c_int, [P, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int64)]),
    "sybl_query_hash_install_union": (C.c_int, [P, P, C.c_int64]),
    "sybl_query_finalize": (C.c_int, [P, C.POINTER(P)]),
    "sybl_query_snapshot": (C.c_int, [P]),
    "sybl_query_collective_finalize": (C.c_int, [P]),
    "sybl_result_rows": (C.c_int, [P, C.c_int, C.POINTER(C.POINTER(GroupRow)), C.POINTER(C.c_int64)]),
    "sybl_result_matched": (C.c_int64, [P]),
    "sybl_result_subhists": (C.c_int, [P, C.c_int, C.POINTER(C.POINTER(SubHist)), C.POINTER(C.c_int64)]),
    "sybl_result_distinct": (C.c_int, [P, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_uint8))]),
    "sybl_debug_hll_ints": (C.c_int, [P, P, C.c_int64, C.c_int32, P]),
    "sybl_debug_hll_bytes": (C.c_uint64, [C.c_char_p, C.c_int64, P]),
    "sybl_debug_hll_cardinality": (C.c_int64, [P]),
    "sybl_result_free": (None, [P]),
    "sybl_query_stats": (C.c_int, [P, C.POINTER(RunStats)]),
    "sybl_debug_query_cells": (C.c_int, [P, C.c_int, C.c_int, P, C.c_int64, C.POINTER(C.c_int64)]),
    "sybl_result_render": (C.c_char_p, [P, C.c_int]),
    "sybl_result_encode": (C.c_void_p, [P, C.POINTER(C.c_int64)]),
}

_lib = None


def lib():
    """Loads libsybilgpu.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C sybil_amd/csrc). sybil_amd has no CPU fallback." % LIB_PATH)
        # torch wheels bundle their own ROCm runtime (libamdhip64.so.7, libhsa-runtime64.so.1,
        # librccl.so.1) under torch/lib with the same sonames as /opt/rocm/lib.  Whichever copy is
        # loaded first serves the whole process, so load torch's first: the engine then shares ONE
        # HIP runtime (streams, device memory, RCCL) with torch instead of mixing two.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SyblError(rc, (lib().sybl_last_error() or b"").decode("utf-8", "replace"))
    return rc
