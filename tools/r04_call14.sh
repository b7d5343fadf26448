#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q --tb=short -x -k "outlier or wrap or rescanned or wrapping" 2>&1 | grep "passed\|failed" | tail -2
timeout -k 10 200 python tools/bench_wide.py 2>&1 | grep "outliers" | tee gpurun_out/r04_c14_wide.txt
echo "cfg2 occupancy"
for w in 1 2 3; do for kb in 152 40 20; do echo -n "wg/cu=$w rep_kb=$kb: "; SYBL_WG_PER_CU=$w SYBL_REP_BUDGET_KB=$kb python tools/scan_loop.py cfg2 40 compact | cut -d' ' -f 20-30; done; done
