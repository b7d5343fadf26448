#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_writer.py tests/test_gpu_cli.py tests/test_gpu_compact.py -q -x --tb=short > gpurun_out/pytest_loader.log 2>&1; tail -15 gpurun_out/pytest_loader.log
