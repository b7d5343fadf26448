#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -q --tb=short -x -k "cfg4 or rescanned or wrapping or outlier or wrap or three_and_four or partition or hist" 2>&1 | grep "passed\|failed" | tail -2
for i in 1 2 3; do
SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/L.so bash tools/scan_profile.sh A cfg4 | cut -c1-200
bash tools/scan_profile.sh B cfg4 | cut -c1-200
done
