#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1800 python -m pytest tests -q -x -m gpu --tb=short > gpurun_out/r03_c10_all.log 2>&1
echo "all gpu tests: $(grep -n 'passed\|failed' gpurun_out/r03_c10_all.log | tail -1)"; grep -n "Error\|assert " gpurun_out/r03_c10_all.log | head -8
timeout -k 10 900 python bench.py > gpurun_out/r03_c10_bench.json 2> gpurun_out/r03_c10_bench.err; tail -c 300 gpurun_out/r03_c10_bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_c10_bench.json') if l.startswith('{')][-1])
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
    for c in d.get("configs", []): print(c["config"]["workload"], round(c["value"]/1e9,1), "Grows/s", round(c["ms_per_step"],3), "ms/step kernel", round(c["kernel_ms"],3), "frac", round(c["roofline"]["frac"],3), c["config"]["host_ms_per_step"])
    for k in ("load","load_mixed_4col"): print(k, round(d[k]["rows_per_s"]/1e6), "M rows/s", d[k]["seconds"], d[k]["stage_breakdown"])
except Exception as e: print("bench parse failed", e)
P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_loader -o kt -- python $GRAFT_REPO_ROOT/tools/bench_loader.py 1600 > $GRAFT_REPO_ROOT/gpurun_out/r03_loader_bench.txt 2>&1
cd $GRAFT_REPO_ROOT; { echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_loader.py 1600   (MI355X, round 3; 4 loads of a 104.9 M-row, 4-column table)"; python tools/rocpd_summary.py gpurun_out/prof_loader/*.db; } > gpurun_out/r03_loader_kernel_trace.txt 2>&1; rm -rf gpurun_out/prof_loader
head -12 gpurun_out/r03_loader_kernel_trace.txt | cut -c1-150; tail -6 gpurun_out/r03_loader_bench.txt | cut -c1-200
