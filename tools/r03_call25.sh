#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 240 python -m pytest tests/test_gpu_hash.py tests/test_gpu_compact.py -q --tb=short -x > gpurun_out/r03_c25.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c25.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c25.log | head -8
grep -q "passed" gpurun_out/r03_c25.log || exit 1
timeout -k 5 200 python tools/bench_variants.py 2>&1 | grep variant | tee gpurun_out/r03_c25_variants.txt
