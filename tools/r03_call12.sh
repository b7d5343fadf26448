#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests/test_gpu_hash.py tests/test_gpu_fuzz.py tests/test_gpu_compact.py -q --tb=short > gpurun_out/r03_c12.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c12.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c12.log | head -12
timeout -k 10 300 python tools/bench_variants.py 2>&1 | grep "^{" | tee gpurun_out/r03_hash_variants.txt; timeout -k 10 600 python tools/bench_hash.py 3 2>&1 | grep "^{" | cut -c1-330 | tee gpurun_out/r03_hash_bench.txt
