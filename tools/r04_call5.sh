#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py -q --tb=short -x > gpurun_out/r04_c5_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c5_tests.log | tail -3
python tools/ab_cfg4.py 3 A=ab/A.so B=-,SYBL_PARTHIST_TAIL=0 C=- | tee gpurun_out/r04_c5_ab.txt
for v in A B C; do
  if [ $v = A ]; then export SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/A.so; else unset SYBL_LIBRARY; fi
  if [ $v = B ]; then export SYBL_PARTHIST_TAIL=0; else unset SYBL_PARTHIST_TAIL; fi
  WL=cfg4 TAG=r04_cfg4_e$v LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r04_c5_prof$v.log 2>&1; echo $v; head -12 gpurun_out/prof_r04_cfg4_e$v/r04_cfg4_e${v}_kernel_trace.txt | grep "k_emit\|k_part_hist\|k_count" | cut -c1-150
done
