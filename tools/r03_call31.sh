#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r03_c31_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c31_tests.log | tail -1)"; grep -n "^FAILED\|^ERROR" gpurun_out/r03_c31_tests.log | head -10
timeout -k 5 150 python tools/bench_wide.py 2>&1 | grep query | tee gpurun_out/r03_c31_wide.txt
