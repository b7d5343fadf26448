#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py -q --tb=short -x > gpurun_out/r04_c4_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r04_c4_tests.log | tail -3
for i in 1 2 3; do timeout 200 python tools/bench_configs.py 0 5 cfg4 compact | cut -c1-160; done
SYBL_PARTHIST_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r04_ph_trace4.txt timeout 200 python tools/bench_configs.py 0 1 cfg4 compact | cut -c1-100
python tools/parthist_trace.py gpurun_out/r04_ph_trace4.txt | tee gpurun_out/r04_ph_trace4_summary.txt
cd /tmp && export TMPDIR=/tmp; rocprofv3 -L 2>/dev/null | grep -io "[A-Z_0-9]*UTCL[A-Z_0-9]*\|[A-Z_0-9]*TLB[A-Z_0-9]*" | sort -u | tr '\n' ' ' | cut -c1-1500; echo
cd $GRAFT_REPO_ROOT
for i in 1 2; do WL=cfg4 TAG=r04_cfg4_d$i LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r04_c4_prof$i.log 2>&1; head -12 gpurun_out/prof_r04_cfg4_d$i/r04_cfg4_d${i}_kernel_trace.txt | grep "k_emit\|k_part_hist\|k_count" | cut -c1-150; done
