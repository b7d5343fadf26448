#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -q --tb=short -x -k "cfg4 or rescanned or wrapping or outlier or wrap or three_and_four or scatter or partition or full_hist or hist" > gpurun_out/r04_c3_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c3_tests.log | tail -3
timeout 200 python tools/bench_configs.py 0 5 cfg4 compact | cut -c1-260
WL=cfg4 TAG=r04_cfg4_c LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r04_c3_prof.log 2>&1
head -12 gpurun_out/prof_r04_cfg4_c/r04_cfg4_c_kernel_trace.txt | cut -c1-160
