#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_compact.py tests/test_gpu_fullsize.py -q -x --tb=short > gpurun_out/pytest_div32.log 2>&1; tail -4 gpurun_out/pytest_div32.log
for M in div32 nodiv32; do
  if [ $M = nodiv32 ]; then export SYBL_NO_DIV32=1; else unset SYBL_NO_DIV32; fi
  echo "== $M"; timeout 300 python tools/bench_configs.py 0 7 cfg3,cfg4,cfg2 compact 2>&1 | grep '^{' | cut -c1-230
done
