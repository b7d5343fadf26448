#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_loader.py tests/test_gpu_writer.py tests/test_gpu_cli.py -q --tb=short -x 2>&1 | grep "passed\|failed" | tail -2
timeout -k 10 300 python tools/bench_loader.py 2>&1 | tail -12 | cut -c1-400
