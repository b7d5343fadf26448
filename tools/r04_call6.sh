#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -q --tb=short -x -k "cfg4 or rescanned or wrapping or outlier or wrap or three_and_four or scatter or partition or hist" > gpurun_out/r04_c6_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c6_tests.log | tail -3
python tools/ab_cfg4.py 3 A=ab/A.so D=- | tee gpurun_out/r04_c6_ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_tl; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python tools/bench_configs.py 0 2 cfg4 compact > $OUT/kt.log 2>&1
python tools/rocpd_timeline.py $OUT/kt/*.db 14 | tee gpurun_out/r04_c6_timeline.txt
rm -rf $OUT/kt
WL=cfg4 TAG=r04_cfg4_f bash tools/prof_cfg.sh > gpurun_out/r04_c6_prof.log 2>&1
head -12 gpurun_out/prof_r04_cfg4_f/r04_cfg4_f_kernel_trace.txt | grep "k_emit\|k_part_hist\|k_count" | cut -c1-150
