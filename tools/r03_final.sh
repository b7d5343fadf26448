#!/bin/bash
# Round-3 closing run: the whole GPU suite, the default bench line, a kernel trace of the headline, the N > 1 code path on one
# rank, the wide-aggregation bench, smoke().
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 540 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r03_final_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_final_tests.log | tail -1)"; grep -n "^FAILED\|^ERROR" gpurun_out/r03_final_tests.log | head -10
timeout -k 10 300 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_final_bench.json') if l.startswith('{')][-1])
    print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'], 'load', d.get('load',{}).get('rows_per_s'))
    for c in d.get('configs',[]): print('  ', c.get('config',{}).get('workload'), c.get('ms_per_step'), c.get('kernel_ms'), c.get('steps'), c.get('warmup'), c.get('error'))
except Exception as e: print('bench parse failed', e)
P
timeout -k 10 120 python bench.py --force-dist --no-cpu-baseline --no-load --no-canonical --no-configs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('force-dist', d['ms_per_step'], d['n_gpus'], str(d.get('oracle_check'))[:80])"
timeout -k 10 150 python tools/bench_wide.py 2>&1 | grep query | tee gpurun_out/r03_final_wide.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_final; mkdir -p $OUT; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-load --no-configs --no-oracle-check > $OUT/kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-load --no-configs --no-oracle-check   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"; python tools/rocpd_summary.py $OUT/kt/*.db | grep -v "rocclr\|k_fill"; echo; grep '^{' $OUT/kt.log | cut -c1-600; } > gpurun_out/r03_final_kernel_trace.txt
rm -rf $OUT/kt; head -8 gpurun_out/r03_final_kernel_trace.txt | cut -c1-150
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
