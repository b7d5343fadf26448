"""The test-only RCCL stand-in (tests/rccl_standin/rccl_standin.cpp) checked on its own, on the CPU: R processes run its
nine entry points over host memory (libfakehip.so in LD_PRELOAD supplies the few HIP calls it makes) and every rank's
output is compared with numpy.  This pins the checker's protocol -- chunked steps, barriers, in-place operands, all five
(type, op) pairs csrc/rccl.cpp uses, the mismatch abort -- before tests/test_gpu_multirank.py trusts it on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STANDIN = os.path.join(HERE, "rccl_standin")


def _built():
    if not (os.path.exists(os.path.join(STANDIN, "librccl_standin.so")) and os.path.exists(os.path.join(STANDIN, "libfakehip.so"))):
        subprocess.check_call(["make", "-s", "-C", STANDIN])


def _run(tmp_path, world, extra=(), sync=False):
    _built()
    env = dict(os.environ, LD_PRELOAD=os.path.join(STANDIN, "libfakehip.so"), SYBL_STANDIN_SLOT_MB="1", SYBL_STANDIN_TIMEOUT_S="30",
               PYTHONPATH=ROOT)
    if sync:
        env["SYBL_STANDIN_SYNC"] = "1"
    uid = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.standin_driver", uid, str(world), str(r), str(tmp_path / ("out%d.npz" % r)), *extra],
                              cwd=ROOT, env=env, stderr=subprocess.PIPE) for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], [o[1].decode() for o in outs]


@pytest.mark.parametrize("world,sync", [(2, False), (3, False), (8, False), (2, True)])
def test_standin_collectives_match_numpy(tmp_path, world, sync):
    codes, errs = _run(tmp_path, world, sync=sync)
    assert codes == [0] * world, errs
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(world)]
    ar = sum(o["ar_in"] for o in outs)
    mx = np.maximum.reduce([o["mx_in"] for o in outs])
    u8 = np.maximum.reduce([o["u8_in"] for o in outs])
    rs32 = sum(o["rs32_in"].astype(np.int64) for o in outs).astype(np.int32)
    rs64 = sum(o["rs64_in"] for o in outs)
    ag = np.concatenate([o["ag_in"] for o in outs])
    for r, o in enumerate(outs):
        per = o["rs32"].size
        assert np.array_equal(o["ar_sum"], ar) and np.array_equal(o["mx"], mx) and np.array_equal(o["u8_max"], u8)
        assert np.array_equal(o["rs32"], rs32[r * per:(r + 1) * per])
        assert np.array_equal(o["rs64"], rs64[r * per:(r + 1) * per])
        assert np.array_equal(o["ag"], ag)
        assert np.array_equal(o["ag1"], 100 + np.arange(world))


def test_standin_aborts_on_mismatched_collectives(tmp_path):
    codes, errs = _run(tmp_path, 2, extra=("mismatch",))
    assert codes == [86, 86], (codes, errs)
    assert all("collective mismatch" in e for e in errs)
