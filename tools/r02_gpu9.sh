#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_loghist.py tests/test_gpu_cli.py -q -x --tb=short > gpurun_out/pytest_enc.log 2>&1; tail -25 gpurun_out/pytest_enc.log
