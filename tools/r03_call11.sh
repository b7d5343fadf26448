#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests/test_gpu_hash.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q --tb=short > gpurun_out/r03_c11.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c11.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c11.log | head -12
