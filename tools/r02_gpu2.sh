#!/bin/bash
# GPU session 2: loader (pinned slabs) tests + load bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_writer.py tests/test_gpu_cli.py -q -x --tb=short > gpurun_out/pytest_loader.log 2>&1; tail -15 gpurun_out/pytest_loader.log
timeout 300 python tools/bench_loader.py 1600 > gpurun_out/bench_loader.log 2>&1; cat gpurun_out/bench_loader.log
SYBL_LOADER_THREADS=32 timeout 300 python tools/bench_loader.py 1600 > gpurun_out/bench_loader_32.log 2>&1; tail -4 gpurun_out/bench_loader_32.log
