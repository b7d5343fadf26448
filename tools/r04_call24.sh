#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_hash.py tests/test_gpu_loader.py tests/test_gpu_cli.py -q --tb=short -x > gpurun_out/r04_c24_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c24_tests.log | tail -3; grep -n "^E " gpurun_out/r04_c24_tests.log | head -8
timeout -k 10 100 python tools/bench_nullable.py 2>&1 | tail -8 | cut -c1-300
