#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q --tb=short > gpurun_out/r03_c18.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c18.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c18.log | head -8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c18kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 0 3 cfg4 compact > $GRAFT_REPO_ROOT/gpurun_out/r03_c18_kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py gpurun_out/c18kt/*.db 2>&1 | grep "k_hist\|k_part\|k_emit\|k_count" | cut -c1-150; rm -rf gpurun_out/c18kt
timeout -k 10 600 python bench.py --workload cfg4_hist_highcard --no-cpu-baseline --no-load --no-canonical --no-oracle-check --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4', round(d['ms_per_step'],3), 'ms/step kernel', round(d['roofline']['kernel_ms'],3), d['config']['host_ms_per_step'])"
