#!/bin/bash
# Round 3, GPU call 4: wrap test, the bench line with `configs`, HBM traffic passes of cfg2 / cfg5, all passes of cfg4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "wrap or cfg4 or partition or full_hist" > gpurun_out/r03_c4_wrap.log 2>&1
echo "wrap: $(grep -n 'passed\|failed' gpurun_out/r03_c4_wrap.log | tail -1)"; grep -n "Error\|assert" gpurun_out/r03_c4_wrap.log | head -5
timeout -k 10 900 python bench.py > gpurun_out/r03_c4_bench.json 2> gpurun_out/r03_c4_bench.err; tail -c 600 gpurun_out/r03_c4_bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_c4_bench.json') if l.startswith('{')][-1])
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
    for c in d.get("configs", []): print(c["config"]["workload"], round(c["value"]/1e9,1), "Grows/s", round(c["ms_per_step"],3), "ms/step kernel", round(c["kernel_ms"],3), "frac", round(c["roofline"]["frac"],3), c.get("oracle_check",{}).get("oracle_seconds"))
except Exception as e: print("bench parse failed", e)
P
WL=cfg2 TAG=r03_cfg2 LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg2.log 2>&1
WL=cfg5 TAG=r03_cfg5 LEAN=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg5.log 2>&1
WL=cfg4 TAG=r03_cfg4_v4 EXTRA=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg4_v4.log 2>&1
grep -h "FETCH_SIZE\|WRITE_SIZE" gpurun_out/prof_r03_cfg*/r03_cfg*_pmc.txt | cut -c1-150
