#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/A.so bash tools/scan_profile.sh A cfg4
bash tools/scan_profile.sh D cfg4
done
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -i "power\|sclk" | head
