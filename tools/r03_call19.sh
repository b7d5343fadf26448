#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_configs.py 0 5 all compact 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['kernel_ms'], 'ms', round(d['GBps']), 'GB/s', 'step', d['step_ms'])"
timeout -k 10 1500 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q --tb=short > gpurun_out/r03_c19.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c19.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c19.log | head -8
