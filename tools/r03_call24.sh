#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -q --tb=short -x -k "cfg4 or wrap" > gpurun_out/r03_c24.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c24.log | tail -1)"; grep -n "Error\|assert \|^FAILED\|fault" gpurun_out/r03_c24.log | head -8
grep -q "passed" gpurun_out/r03_c24.log || exit 1
for env in "" "SYBL_PART_BRANCHY=1" "" "SYBL_PART_BRANCHY=1"; do
env $env timeout -k 5 120 python bench.py --workload cfg4_hist_highcard --no-cpu-baseline --no-load --no-canonical --no-configs --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 [$env]', round(d['ms_per_step'],3), 'ms/step kernel', round(d['roofline']['kernel_ms'],3), 'check', str(d.get('oracle_check'))[:60])"
done
