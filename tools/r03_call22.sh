#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py -q --tb=short -x > gpurun_out/r03_c22.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c22.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c22.log | head -8
timeout -k 10 300 python tools/bench_variants.py 2>&1 | tee gpurun_out/r03_c22_variants.txt
timeout -k 10 200 python tools/bench_nullable.py 2>&1 | tail -6 | tee gpurun_out/r03_c22_nullable.txt
