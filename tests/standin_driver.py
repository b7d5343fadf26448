"""Worker of tests/test_rccl_standin.py: one rank of the test-only RCCL stand-in (tests/rccl_standin/) over HOST memory
(libfakehip.so preloaded), run as `python -m tests.standin_driver <uid file> <world> <rank> <out file> [mismatch]`."""
import ctypes as C
import os
import sys
import time

import numpy as np

INT8, UINT8, INT32, UINT32, INT64, UINT64 = 0, 1, 2, 3, 4, 5
SUM, MAX = 0, 2


def main():
    uid_path, world, rank, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mismatch = len(sys.argv) > 5 and sys.argv[5] == "mismatch"
    here = os.path.dirname(os.path.abspath(__file__))
    lib = C.CDLL(os.path.join(here, "rccl_standin", "librccl_standin.so"))
    assert lib.sybl_rccl_standin_marker() == 1
    uid = C.create_string_buffer(128)
    if rank == 0:
        assert lib.ncclGetUniqueId(uid) == 0
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.01)
    uid = C.create_string_buffer(open(uid_path, "rb").read(), 128)
    comm = C.c_void_p()

    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    u = Uid()
    C.memmove(C.byref(u), uid, 128)
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert lib.ncclCommInitRank(C.byref(comm), world, u, rank) == 0
    for f in (lib.ncclAllReduce, lib.ncclReduceScatter):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    out = {}
    rng = np.random.default_rng(1234 + rank)

    def ptr(a, off=0):
        return a.ctypes.data + off * a.itemsize

    # all-reduce, in place, a count that is not a multiple of the slot (SYBL_STANDIN_SLOT_MB=1 -> several steps)
    n = 300_001
    a = rng.integers(-1 << 40, 1 << 40, n, dtype=np.int64)
    out["ar_in"] = a.copy()
    assert lib.ncclAllReduce(ptr(a), ptr(a), n, INT64, SUM, comm, None) == 0
    out["ar_sum"] = a
    b = rng.integers(-1 << 40, 1 << 40, 1000, dtype=np.int64)
    out["mx_in"] = b.copy()
    r = np.zeros_like(b)
    lib.ncclGroupStart()
    assert lib.ncclAllReduce(ptr(b), ptr(r), b.size, INT64, MAX, comm, None) == 0
    u8 = rng.integers(0, 256, 70_000, dtype=np.uint8)
    out["u8_in"] = u8.copy()
    assert lib.ncclAllReduce(ptr(u8), ptr(u8), u8.size, UINT8, MAX, comm, None) == 0
    lib.ncclGroupEnd()
    out["mx"], out["u8_max"] = r, u8
    # reduce-scatter int32 out of place, int64 in place (recv = send + rank * count, as csrc/rccl.cpp does)
    per = 200_003
    s32 = rng.integers(0, 1 << 20, per * world, dtype=np.int32)
    out["rs32_in"] = s32.copy()
    r32 = np.zeros(per, dtype=np.int32)
    assert lib.ncclReduceScatter(ptr(s32), ptr(r32), per, INT32, SUM, comm, None) == 0
    out["rs32"] = r32
    s64 = rng.integers(0, 1 << 50, per * world, dtype=np.int64)
    out["rs64_in"] = s64.copy()
    assert lib.ncclReduceScatter(ptr(s64), ptr(s64, rank * per), per, INT64, SUM, comm, None) == 0
    out["rs64"] = s64[rank * per:(rank + 1) * per].copy()
    # all-gather in place (send = recv + rank * count), uint64, and a single word
    g = np.zeros(per * world, dtype=np.uint64)
    mine = rng.integers(0, 1 << 62, per, dtype=np.uint64)
    g[rank * per:(rank + 1) * per] = mine
    out["ag_in"] = mine
    assert lib.ncclAllGather(ptr(g, rank * per), ptr(g), per, UINT64, comm, None) == 0
    out["ag"] = g
    one = np.zeros(world, dtype=np.int64)
    one[rank] = 100 + rank
    assert lib.ncclAllGather(ptr(one, rank), ptr(one), 1, INT64, comm, None) == 0
    out["ag1"] = one
    np.savez(out_path, **out)
    if mismatch:  # ranks disagree on the next collective: the stand-in must abort (exit code 86), not hang
        x = np.zeros(16, dtype=np.int64)
        lib.ncclAllReduce(ptr(x), ptr(x), 16 if rank == 0 else 8, INT64, SUM, comm, None)
        sys.exit(3)  # (not reached)
    assert lib.ncclCommDestroy(comm) == 0


if __name__ == "__main__":
    main()
