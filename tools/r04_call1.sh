#!/bin/bash
# round 4, first GPU call: the ADVICE fixes' tests, config 4 re-profiled at HEAD, k_part_hist's phase trace, outlier log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q --tb=short -x -k "rescanned or wrapping or outlier or wrap or three_and_four or scatter" > gpurun_out/r04_c1_tests.log 2>&1
tail -5 gpurun_out/r04_c1_tests.log
SYBL_PARTHIST_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r04_ph_trace.txt timeout 200 python tools/bench_configs.py 0 2 cfg4 compact | cut -c1-300
python tools/parthist_trace.py gpurun_out/r04_ph_trace.txt | tee gpurun_out/r04_ph_trace_summary.txt
timeout -k 10 200 python tools/bench_wide.py 2>&1 | grep query | tee gpurun_out/r04_c1_wide.txt
WL=cfg4 TAG=r04_cfg4_head bash tools/prof_cfg.sh > gpurun_out/r04_c1_prof.log 2>&1
head -12 gpurun_out/prof_r04_cfg4_head/r04_cfg4_head_kernel_trace.txt | cut -c1-160
