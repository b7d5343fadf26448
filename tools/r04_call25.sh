#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_wide.py 2>&1 | grep "cfg3, ~1" | tee gpurun_out/r04_c25_wide.txt
timeout -k 10 500 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q --tb=short -x -k "outlier or fuzz or random or compact or nullable or hist_bucket" > gpurun_out/r04_c25_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c25_tests.log | tail -3; grep -n "^E " gpurun_out/r04_c25_tests.log | head -8
