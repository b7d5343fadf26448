"""GPU: bench.py's own N > 1 flow, end to end, on ONE device -- the launch line the driver uses (`python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`), every rank on GPU 0, the library's collectives through the
test-only RCCL stand-in (tests/rccl_standin/, LD_PRELOAD), torch's process group on gloo (SYBL_BENCH_ONE_DEVICE=1).  What it
pins: block sharding, the in-library communicator set-up over torch's broadcast, scan -> all-reduce -> (collective) snapshot ->
finalize pipelined one deep, the max-over-ranks timing, and the printed line -- whose merged result is checked bit for bit
against the CPU oracle over the whole table by bench.py itself (`oracle_check`).  Config 3 is the driver's scaling workload
(one SUM all-reduce per step); config 4 takes the printer's limit-aware merge and, for its `every_row_summarised` leg, the
reduce-scatter with the collective finalize on every rank.  A FUNCTIONAL check: the line says so, its timings mean nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STANDIN = os.path.join(HERE, "rccl_standin", "librccl_standin.so")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,workload,rows", [(2, "cfg3_filter3_group2_stddev", 40_000_000), (4, "cfg3_filter3_group2_stddev", 40_000_000),
                                                 (2, "cfg4_hist_highcard", 30_000_000)])
def test_bench_line_from_n_ranks_on_one_device(world, workload, rows):
    if not os.path.exists(STANDIN):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STANDIN)])
    env = dict(os.environ, LD_PRELOAD=STANDIN, SYBL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SYBL_STANDIN_TIMEOUT_S="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--steps", "4", "--warmup", "2", "--workload", workload,
           "--rows", str(rows), "--no-cpu-baseline", "--no-load", "--no-configs", "--no-canonical"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["config"]["workload"] == workload and d["config"]["rows"] == rows
    assert "one_device_standin" in d
    oc = d["oracle_check"]  # (bench.py asserted equality before it wrote this)
    assert oc["rows"] == rows and oc["matched"] == d["config"]["matched_rows"] and oc["groups"] == d["config"]["groups"]
    if workload.startswith("cfg4"):
        assert d["every_row_summarised"]["steps"] >= 1  # the reduce-scatter leg ran too
