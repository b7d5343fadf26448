#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q --tb=short -x -k "cfg4 or config4 or wrap or scatter or highcard or hist" 2>&1 | grep "passed\|failed" | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline --no-load --no-configs --no-canonical --workload cfg4_hist_highcard --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cfg4', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['host_ms_per_step'], str(d.get('oracle_check'))[:60])
"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_hs; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python tools/bench_configs.py 0 4 cfg4 compact > $OUT/kt.log 2>&1
python tools/rocpd_summary.py $OUT/kt/*.db | grep "k_hist\|k_part_hist\|k_emit" | cut -c1-150
rm -rf $OUT/kt
