#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q --tb=short -x -k "cfg4 or rescanned or wrapping or outlier or wrap or three_and_four or scatter or partition or full_hist" > gpurun_out/r04_c2_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c2_tests.log | tail -3
timeout 200 python tools/bench_configs.py 0 5 cfg4 compact | cut -c1-260
SYBL_PARTHIST_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r04_ph_trace2.txt timeout 200 python tools/bench_configs.py 0 1 cfg4 compact | cut -c1-100
python tools/parthist_trace.py gpurun_out/r04_ph_trace2.txt | tee gpurun_out/r04_ph_trace2_summary.txt
WL=cfg4 TAG=r04_cfg4_b LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r04_c2_prof.log 2>&1
head -12 gpurun_out/prof_r04_cfg4_b/r04_cfg4_b_kernel_trace.txt | cut -c1-160
