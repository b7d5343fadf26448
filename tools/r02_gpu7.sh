#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_loghist.py -q --tb=short > gpurun_out/pytest_fuzz2.log 2>&1; tail -6 gpurun_out/pytest_fuzz2.log
timeout 1200 python tools/fuzz_more.py 1000 1200 > gpurun_out/fuzz_more.log 2>&1; tail -12 gpurun_out/fuzz_more.log
