#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "str_replace or strings or outlier" > gpurun_out/pytest_sr.log 2>&1; tail -25 gpurun_out/pytest_sr.log
