#!/bin/bash
# Round 3, GPU call 1: count distinct on hardware + config 4 counters (what k_emit waits for)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_TEST_DISTINCT=1 timeout -k 10 600 python -m pytest tests/test_gpu_zz_distinct.py -q --tb=short > gpurun_out/r03_distinct.log 2>&1
echo "distinct: $(grep -n 'passed\|failed\|error' gpurun_out/r03_distinct.log | tail -2)"
tail -60 gpurun_out/r03_distinct.log | cut -c1-200
WL=cfg4 TAG=r03_cfg4 EXTRA=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg4.log 2>&1
tail -70 gpurun_out/r03_prof_cfg4.log | cut -c1-170
