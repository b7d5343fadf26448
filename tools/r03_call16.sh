#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests/test_gpu_compact.py -q --tb=short > gpurun_out/r03_c16.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c16.log | tail -1)"; grep -n "Error\|assert \|^FAILED" gpurun_out/r03_c16.log | head -12
for wl in; do
timeout -k 10 600 python bench.py --workload $wl --no-cpu-baseline --no-load --no-canonical --steps 10 --warmup 2 > gpurun_out/r03_c16_$wl.json 2> gpurun_out/r03_c16_$wl.err
python - $wl <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/r03_c16_%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d["value"]/1e9,1), "Grows/s", round(d["ms_per_step"],3), "ms/step kernel", round(d["roofline"]["kernel_ms"],3), d["config"]["host_ms_per_step"])
P
done
