#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {  # label, steps, warmup, extra env
  env $4 SYBL_FINALIZE_TRACE=1 timeout -k 5 100 python bench.py --workload cfg5_time_rollup --no-cpu-baseline --no-load --no-canonical --no-oracle-check --steps $2 --warmup $3 2> gpurun_out/r03_c27_$1.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],3), 'ms/step kernel', round(d['roofline']['kernel_ms'],3), d['config']['host_ms_per_step'])"
  grep "^finalize:" gpurun_out/r03_c27_$1.err | tail -12 | cut -c1-260
}
run s10w2 10 2 A=1
run s20w4 20 4 A=1
