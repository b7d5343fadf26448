#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_FINALIZE_TRACE=1 timeout -k 10 600 python bench.py --workload cfg4_hist_highcard --no-cpu-baseline --no-load --no-canonical --no-oracle-check --steps 10 --warmup 4 > gpurun_out/r03_c9.json 2> gpurun_out/r03_c9.err; grep "^snapshot" gpurun_out/r03_c9.err | tail -13; grep "^finalize" gpurun_out/r03_c9.err | tail -4 | cut -c1-250
