#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in cfg2_group1_avg2 cfg5_time_rollup; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-load --no-configs --no-canonical --no-oracle-check --workload $wl --warmup 8 2>gpurun_out/r04_c13.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$wl', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['host_ms_per_step'], d['config'].get('rows_first_access_ms'))
"
done
SYBL_FINALIZE_TRACE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-load --no-configs --no-canonical --no-oracle-check --workload cfg2_group1_avg2 --warmup 8 --steps 6 2>&1 | grep -v "^{" | tail -40
