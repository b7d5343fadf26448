#!/bin/bash
# Round-4 closing run: the whole GPU suite, the default bench line, and every profile the bench line cites regenerated from
# the same HEAD (headline kernel trace + PMC, configs 2 / 4 / 5 kernel trace + PMC), smoke().  Afterwards, in the repo:
#   cp gpurun_out/prof/r04_* profiles/; for c in cfg2 cfg4 cfg5; do cp gpurun_out/prof_r04_$c/r04_${c}_* profiles/; done
#   python tools/make_traffic.py r04; cp gpurun_out/r04_final_bench.json profiles/r04_bench_1gpu.json
# (not prof_r04_cfg*: gpurun_out keeps the round's experiment directories -- prof_r04_cfg4_b, ... -- beside the closing run's)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 700 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r04_final_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r04_final_tests.log | tail -1)"; grep -n "^FAILED\|^ERROR" gpurun_out/r04_final_tests.log | head -10
fi
timeout -k 10 400 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
python - <<'Q'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_final_bench.json') if l.startswith('{')][-1])
    print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'], 'load', d.get('load',{}).get('rows_per_s'), 'loaded scan', d.get('loaded_table_scan',{}).get('kernel_ms'))
    for c in d.get('configs',[]): print('  ', c.get('config',{}).get('workload'), c.get('ms_per_step'), c.get('kernel_ms'), c.get('config',{}).get('host_ms_per_step'), c.get('error'))
except Exception as e: print('bench parse failed', e)
Q
PROF_TAG=r04 BENCH_ARGS="--no-load --no-configs --no-oracle-check --no-canonical" bash tools/prof_r01.sh > gpurun_out/r04_final_prof.log 2>&1
head -8 gpurun_out/prof/r04_kernel_trace_stats.txt | cut -c1-150
for wl in cfg2 cfg4 cfg5; do WL=$wl TAG=r04_$wl bash tools/prof_cfg.sh > gpurun_out/r04_final_prof_$wl.log 2>&1; head -7 gpurun_out/prof_r04_$wl/r04_${wl}_kernel_trace.txt | tail -4 | cut -c1-150; done
timeout -k 10 120 python bench.py --force-dist --no-cpu-baseline --no-load --no-canonical --no-configs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('force-dist', d['ms_per_step'], d['n_gpus'], str(d.get('oracle_check'))[:80])"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
