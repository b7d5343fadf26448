#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_LAZY_ROWS=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_loghist.py tests/test_gpu_hash.py -q --tb=short -x > gpurun_out/r04_c12_lazy.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c12_lazy.log | tail -3
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_distinct.py -q --tb=short -x > gpurun_out/r04_c12_full.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c12_full.log | tail -3
timeout -k 10 400 python bench.py --no-cpu-baseline --no-load 2>gpurun_out/r04_c12_bench.err | tee gpurun_out/r04_c12_bench.json | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['host_ms_per_step'], d['config'].get('rows_first_access_ms'))
for c in d.get('configs',[]): print('  ', c.get('config',{}).get('workload'), c.get('ms_per_step'), c.get('kernel_ms'), c.get('config',{}).get('host_ms_per_step'), c.get('config',{}).get('rows_first_access_ms'), c.get('error'))
"
tail -3 gpurun_out/r04_c12_bench.err
