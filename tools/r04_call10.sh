#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q --tb=short -x > gpurun_out/r04_c10_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c10_tests.log | tail -3
for i in 1 2; do
echo "--- no late (D)"; SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/D.so python tools/bench_selectivity.py 1000000000 7 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-40s %.3f ms  %.0f GB/s' % (d['selectivity'], d['kernel_ms'], d['GBps']))"
echo "--- late (L)"; python tools/bench_selectivity.py 1000000000 7 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-40s %.3f ms  %.0f GB/s' % (d['selectivity'], d['kernel_ms'], d['GBps']))"
done
