#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='
import sys,json
for l in sys.stdin:
    d=json.loads(l); print("   %-40s %.3f ms  %.0f GB/s" % (d["selectivity"], d["kernel_ms"], d["GBps"]))'
for i in 1 2; do
echo "--- dropped loads (L)"; SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/L.so python tools/bench_selectivity.py 1000000000 7 | python -c "$fmt"
echo "--- branch (L2)"; python tools/bench_selectivity.py 1000000000 7 | python -c "$fmt"
done
timeout -k 10 300 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fuzz.py -q --tb=short -x 2>&1 | grep "passed\|failed" | tail -2
