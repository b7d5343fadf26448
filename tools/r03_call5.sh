#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_hash.py -q -x --tb=short > gpurun_out/r03_c5_parity.log 2>&1
echo "parity+hash: $(grep -n 'passed\|failed' gpurun_out/r03_c5_parity.log | tail -1)"; grep -n "Error\|assert " gpurun_out/r03_c5_parity.log | head -5
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -q -x --tb=short > gpurun_out/r03_c5_full.log 2>&1
echo "fullsize: $(tail -1 gpurun_out/r03_c5_full.log | cut -c1-200)"
WL=cfg4 TAG=r03_cfg4_v5 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg4_v5.log 2>&1
head -12 gpurun_out/prof_r03_cfg4_v5/r03_cfg4_v5_kernel_trace.txt | cut -c1-140
grep "k_emit_packed" gpurun_out/prof_r03_cfg4_v5/r03_cfg4_v5_pmc.txt | awk '{ for(i=1;i<=NF;i++) if ($i ~ /^(SQ|TCC|TCP|TA|FETCH|WRITE|GRBM|SQC)/) {printf "%-32s %16s\n", $i, $(NF); break} }'
