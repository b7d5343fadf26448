#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1800 python -m pytest tests -q -m gpu --tb=short > gpurun_out/r03_c17_all.log 2>&1
echo "all gpu tests: $(grep -n 'passed\|failed' gpurun_out/r03_c17_all.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c17_all.log | head -8
timeout -k 10 300 python tools/bench_variants.py 2>&1 | grep "^{"
timeout -k 10 600 python bench.py --no-configs --no-cpu-baseline --no-oracle-check > gpurun_out/r03_c17_bench.json 2> gpurun_out/r03_c17.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03_c17_bench.json') if l.startswith('{')][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("load","load_mixed_4col"): print(k, round(d[k]["rows_per_s"]/1e6), "M rows/s", d[k]["seconds"], d[k]["stage_breakdown"]["parse_cpu_s"])
P
