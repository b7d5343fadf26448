#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 240 python -m pytest tests/test_gpu_loader.py tests/test_gpu_writer.py -q --tb=short > gpurun_out/r03_c26.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c26.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c26.log | head -12
