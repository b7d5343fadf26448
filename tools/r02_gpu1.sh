#!/bin/bash
# GPU session 1 of round 2 (second sitting): hash group-by tests, fuzz, selectivity sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hash.py -q -x --tb=short > gpurun_out/pytest_hash.log 2>&1; tail -15 gpurun_out/pytest_hash.log
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q --tb=short > gpurun_out/pytest_fuzz.log 2>&1; tail -8 gpurun_out/pytest_fuzz.log
timeout 300 python tools/bench_selectivity.py > gpurun_out/selectivity.log 2>&1; cat gpurun_out/selectivity.log
