#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short --deselect tests/test_gpu_fullsize.py > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
