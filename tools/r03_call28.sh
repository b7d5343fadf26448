#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 240 python -m pytest tests/test_gpu_compact.py tests/test_gpu_hash.py -q --tb=short -x > gpurun_out/r03_c28.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c28.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c28.log | head -8
timeout -k 5 120 python tools/bench_fresh.py 2>&1 | grep cfg | tee gpurun_out/r03_c28_fresh.txt
timeout -k 5 150 python tools/bench_variants.py 2>&1 | grep variant | tee gpurun_out/r03_c28_variants.txt
