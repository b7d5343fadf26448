"""sybil_amd -- MI355X-native scan/aggregate engine for logv/sybil's query hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of
include/sybilgpu.h), the ctypes view of that ABI, a host-side mirror of the
reference's query surface, and the synthetic table / workloads of BASELINE.md.
"""
from .engine import Context, Query, Result, Table  # noqa: F401
from ._native import SyblError  # noqa: F401
