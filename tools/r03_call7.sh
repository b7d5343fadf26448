#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -x -m gpu --tb=short > gpurun_out/r03_c7_all.log 2>&1
echo "all gpu tests: $(grep -n 'passed\|failed' gpurun_out/r03_c7_all.log | tail -1)"; grep -n "Error\|assert " gpurun_out/r03_c7_all.log | head -8
for wl in "cfg4_hist_highcard 1000000000 100" "cfg5_time_rollup 1000000000"; do
  echo "== $wl"; SYBL_FINALIZE_TRACE=1 timeout -k 10 300 python tools/finalize_breakdown.py $wl 2>&1 | tail -5 | cut -c1-400
done
timeout -k 10 900 python bench.py --no-cpu-baseline --no-load > gpurun_out/r03_c7_bench.json 2> gpurun_out/r03_c7_bench.err; tail -c 300 gpurun_out/r03_c7_bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_c7_bench.json') if l.startswith('{')][-1])
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
    for c in d.get("configs", []): print(c["config"]["workload"], round(c["value"]/1e9,1), "Grows/s", round(c["ms_per_step"],3), "ms/step kernel", round(c["kernel_ms"],3), "frac", round(c["roofline"]["frac"],3), c["config"]["host_ms_per_step"])
except Exception as e: print("bench parse failed", e)
P
