#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WL=cfg4 TAG=r03_cfg4_v3 EXTRA=1 bash tools/prof_cfg.sh > gpurun_out/r03_prof_cfg4_v3.log 2>&1
tail -5 gpurun_out/r03_prof_cfg4_v3.log | cut -c1-170
