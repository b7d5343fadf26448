#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loghist.py -q -x --tb=short > gpurun_out/pytest_loghist.log 2>&1; tail -30 gpurun_out/pytest_loghist.log
