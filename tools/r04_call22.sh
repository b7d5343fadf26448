#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in cfg4_hist_highcard cfg5_time_rollup; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-load --no-configs --no-canonical --no-oracle-check --workload $wl --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$wl', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['host_ms_per_step'])
"
done
timeout -k 10 300 python -m pytest tests/test_gpu_hash.py tests/test_gpu_parity.py -q --tb=short -x -k "hash or cfg4 or scatter or outlier" 2>&1 | grep "passed\|failed" | tail -2
