#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_compact.py -q --tb=short -x -k "outlier or aggregation_columns or nullable or rejects or cli_" > gpurun_out/r03_c33.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c33.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c33.log | head -8
timeout -k 5 200 python tools/bench_wide.py 2>&1 | grep "outliers" | tee gpurun_out/r03_wide4.txt
