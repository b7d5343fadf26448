#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showtemp --csv 2>&1 | head -5
bash tools/clock_watch.sh gpurun_out/r04_clock.txt -- python tools/scan_loop.py cfg4 300 compact | tee gpurun_out/r04_c7_loop.txt | cut -c1-1200
awk 'NR%5==0' gpurun_out/r04_clock.txt | cut -c1-200 | head -40
