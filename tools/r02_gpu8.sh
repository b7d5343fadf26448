#!/bin/bash
# round-2 profile of the headline bench command (kernel trace + PMC passes), as tools/prof_r01.sh
export PROF_TAG=r02b BENCH_ARGS="--no-load --no-canonical"
bash $GRAFT_REPO_ROOT/tools/prof_r01.sh > $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.log 2>&1
cd $GRAFT_REPO_ROOT; head -12 gpurun_out/prof/r02b_kernel_trace_stats.txt | cut -c1-150; grep "k_scan_packed" gpurun_out/prof/r02b_pmc.txt | cut -c1-150 | head -30
