#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_configs.py 0 5 all compact 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['kernel_ms'], 'ms', round(d['GBps']), 'GB/s', 'step', d['step_ms'])"
timeout -k 10 600 python -m pytest tests/test_gpu_compact.py -q --tb=short 2>&1 | tail -1
