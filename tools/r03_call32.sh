#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -q --tb=short -x -k "with_outliers or aggregations_with_full or cfg4 or cfg3_with_full or wrap or one_partition or outlier" > gpurun_out/r03_c32.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c32.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c32.log | head -8
