#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_compact.py tests/test_gpu_cli.py -q --tb=short > gpurun_out/r03_c21.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c21.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c21.log | head -8
for env in "" "SYBL_NO_FUSED_SUMMARY=1"; do
env $env timeout -k 10 600 python bench.py --workload cfg4_hist_highcard --no-cpu-baseline --no-load --no-canonical --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 [$env]', round(d['ms_per_step'],3), 'ms/step kernel', round(d['roofline']['kernel_ms'],3), d['config']['host_ms_per_step'], 'oracle' in str(d.get('oracle_check')))"
done
