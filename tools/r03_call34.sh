#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 165 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_hash.py -q --tb=short -x > gpurun_out/r03_c34.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c34.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c34.log | head -8; tail -c 300 gpurun_out/r03_c34.log | head -5
