#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_compact.py tests/test_gpu_hash.py -q --tb=short -x -k "aggregation_columns or group_columns or three_row_bodies or forced or neq" > gpurun_out/r03_c29.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c29.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c29.log | head -8
