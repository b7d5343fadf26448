#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q --tb=short -x -k "cfg4 or hist or wrap or bucket or part" > gpurun_out/r03_c23.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c23.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c23.log | head -8
for env in "" "SYBL_PART_BRANCHY=1" "" "SYBL_PART_BRANCHY=1"; do
env $env timeout -k 10 600 python bench.py --workload cfg4_hist_highcard --no-cpu-baseline --no-load --no-canonical --no-oracle-check --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 [$env]', round(d['ms_per_step'],3), 'ms/step kernel', round(d['roofline']['kernel_ms'],3))"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_c23; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python tools/bench_configs.py 0 3 cfg4 compact > $OUT/kt.log 2>&1
python tools/rocpd_summary.py $OUT/kt/*.db | grep -v "rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack" > gpurun_out/r03_c23_trace.txt; grep '^{' $OUT/kt.log >> gpurun_out/r03_c23_trace.txt
rm -rf $OUT/kt; cut -c1-150 gpurun_out/r03_c23_trace.txt
