#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_LAZY_ROWS=1 timeout -k 10 900 python -m pytest tests/test_gpu_hash.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -q --tb=short -x > gpurun_out/r04_c17_lazy.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c17_lazy.log | tail -3; grep -n "^E " gpurun_out/r04_c17_lazy.log | head -5
timeout -k 10 400 python -m pytest tests/test_gpu_hash.py tests/test_gpu_distinct.py tests/test_gpu_loghist.py -q --tb=short -x 2>&1 | grep "passed\|failed" | tail -2
SYBL_FINALIZE_TRACE=1 timeout -k 10 400 python tools/bench_hash.py 2>&1 | grep "case\|finalize:\|snapshot:" | cut -c1-400 | tee gpurun_out/r04_c17_hash.txt
