#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_distinct.py tests/test_gpu_loghist.py tests/test_gpu_hash.py tests/test_gpu_parity.py -q -x --tb=short > gpurun_out/r03_c8_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c8_tests.log | tail -1)"; grep -n "Error\|assert " gpurun_out/r03_c8_tests.log | head -8
for wl in cfg4_hist_highcard cfg5_time_rollup; do
timeout -k 10 600 python bench.py --workload $wl --no-cpu-baseline --no-load --no-canonical --steps 10 --warmup 2 > gpurun_out/r03_c8_$wl.json 2> gpurun_out/r03_c8_$wl.err; tail -c 300 gpurun_out/r03_c8_$wl.err
python - $wl <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/r03_c8_%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d["value"]/1e9,1), "Grows/s", round(d["ms_per_step"],3), "ms/step kernel", round(d["roofline"]["kernel_ms"],3), d["config"]["host_ms_per_step"], d.get("oracle_check",{}).get("checked","")[:40])
P
done
