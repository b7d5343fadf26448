#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_writer.py tests/test_gpu_cli.py -q -x --tb=short > gpurun_out/pytest_loader.log 2>&1; tail -5 gpurun_out/pytest_loader.log
for T in 128 32; do echo "== threads $T"; SYBL_LOADER_THREADS=$T SYBL_LOADER_TRACE=1 timeout 300 python tools/bench_loader.py 1600 2>&1 | grep -v "^fixture\|amdgpu.ids" ; done > gpurun_out/bench_loader_sweep.log 2>&1; cat gpurun_out/bench_loader_sweep.log
