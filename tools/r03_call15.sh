#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_FINALIZE_TRACE=1 timeout -k 10 900 python bench.py --no-load --no-cpu-baseline > gpurun_out/r03_c15.json 2> gpurun_out/r03_c15.err
grep "^finalize" gpurun_out/r03_c15.err | awk 'length($0) > 0' | tail -6 | cut -c1-220
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03_c15.json') if l.startswith('{')][-1])
for c in d.get("configs", []): print(c["config"]["workload"], round(c["ms_per_step"],3), "ms/step kernel", round(c["kernel_ms"],3), c["config"]["host_ms_per_step"])
P
