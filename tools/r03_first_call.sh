#!/bin/bash
# Round 3, first GPU call (gpurun --timeout 1500 -- 'bash tools/r03_first_call.sh'): what round 2 could not run once
# its GPU minutes were gone.  Every step is bounded (timeout -k: a rocprofv3 run that does not exit took the whole of
# round 2's remaining budget with it).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# 1. count distinct has never run on hardware: register-by-register against the oracle
SYBL_TEST_DISTINCT=1 timeout -k 10 600 python -m pytest tests/test_gpu_zz_distinct.py -q --tb=short > gpurun_out/r03_distinct.log 2>&1
echo "distinct: $(grep -n 'passed\|failed\|error' gpurun_out/r03_distinct.log | tail -2)"
# 2. the suites that were last run before the final changes of round 2
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_hash.py tests/test_gpu_loader.py tests/test_gpu_loghist.py tests/test_gpu_writer.py -q -x --tb=short > gpurun_out/r03_rest.log 2>&1
echo "rest: $(grep -n 'passed\|failed' gpurun_out/r03_rest.log | tail -1)"
# 3. what k_emit waits for: kernel trace + the usual and the extra counter passes of config 4
WL=cfg4 TAG=r03_cfg4 EXTRA=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg4.log 2>&1
tail -40 gpurun_out/r03_prof_cfg4.log | cut -c1-160
