#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_shapes.py | tee gpurun_out/r04_c26_shapes.txt
timeout -k 10 600 python -m pytest tests/test_gpu_compact.py tests/test_gpu_hash.py -q --tb=short -x > gpurun_out/r04_c26_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c26_tests.log | tail -3; grep -n "^E " gpurun_out/r04_c26_tests.log | head -8
