#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in "" "SYBL_FORCE_NUL=1"; do echo "== $e"; env $e python tools/scan_loop.py cfg3 24 compact | cut -d" " -f 14-26; env $e python tools/scan_loop.py cfg2 24 compact | cut -d" " -f 14-26; env $e python tools/scan_loop.py cfg5 24 compact | cut -d" " -f 14-26; done
timeout -k 10 700 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_loader.py -q --tb=short -x > gpurun_out/r04_c23_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c23_tests.log | tail -3; grep -n "^E " gpurun_out/r04_c23_tests.log | head -8
SYBL_FORCE_NUL=1 timeout -k 10 300 python -m pytest tests/test_gpu_compact.py tests/test_gpu_fullsize.py -q --tb=short -x 2>&1 | grep "passed\|failed" | tail -2
