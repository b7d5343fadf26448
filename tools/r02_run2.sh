#!/bin/bash
# Round 2, second sitting: the whole GPU suite, the bench lines (config 3 = the headline, config 4), the loader under
# rocprofv3.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; tail -c 2500 gpurun_out/bench_cfg3.json
timeout 600 python bench.py --steps 10 --warmup 3 --workload cfg4_hist_highcard --no-load > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; head -c 600 gpurun_out/bench_cfg4.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_loader -o kt -- python $GRAFT_REPO_ROOT/tools/bench_loader.py 1600 > $GRAFT_REPO_ROOT/gpurun_out/prof_loader.log 2>&1
cd $GRAFT_REPO_ROOT
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_loader.py 1600   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ); 4 loads of a 104.9 M-row, 4-column table)"; python tools/rocpd_summary.py gpurun_out/prof_loader/*.db 2>/dev/null || python tools/rocpd_summary.py gpurun_out/prof_loader/*/*.db; echo; grep "open_table" gpurun_out/prof_loader.log; } > gpurun_out/r02_loader_kernel_trace.txt 2>&1
rm -rf gpurun_out/prof_loader
head -30 gpurun_out/r02_loader_kernel_trace.txt | cut -c1-160
