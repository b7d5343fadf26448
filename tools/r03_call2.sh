#!/bin/bash
# Round 3, GPU call 2: the reworked counting sort (k_count -> k_part_bases -> k_emit -> k_part_hist): parity, then timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -q -x --tb=short > gpurun_out/r03_c2_parity.log 2>&1
echo "parity: $(tail -3 gpurun_out/r03_c2_parity.log | cut -c1-200)"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -q -x --tb=short -k "config4 or cfg4 or percentile" > gpurun_out/r03_c2_full.log 2>&1
echo "fullsize: $(tail -3 gpurun_out/r03_c2_full.log | cut -c1-200)"
timeout -k 10 600 python tools/bench_configs.py 0 5 cfg4 compact > gpurun_out/r03_c2_cfg4.log 2>&1; tail -2 gpurun_out/r03_c2_cfg4.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c2kt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 0 3 cfg4 compact > $GRAFT_REPO_ROOT/gpurun_out/r03_c2_kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py gpurun_out/c2kt/*.db 2>&1 | head -16 | cut -c1-150; rm -rf gpurun_out/c2kt
