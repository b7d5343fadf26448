#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in "cfg4_hist_highcard 1000000000 100" "cfg5_time_rollup 1000000000"; do
  echo "== $wl"; SYBL_FINALIZE_TRACE=1 timeout -k 10 300 python tools/finalize_breakdown.py $wl 2>&1 | tail -12 | cut -c1-400
done
