#!/bin/bash
# One GPU-box session of round 2: the GPU test suite, the bench line (config 3) and the config 4 line, and the
# single-rank runs of the multi-GPU code path (--force-dist: process group, in-library RCCL communicator, all-reduce /
# reduce-scatter merge, collective finalize).  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; tail -c 1800 gpurun_out/bench_cfg3.json
timeout 600 python bench.py --steps 10 --warmup 3 --workload cfg4_hist_highcard --no-load > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; head -c 1500 gpurun_out/bench_cfg4.json
timeout 300 python bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-load --no-canonical > gpurun_out/bench_cfg3_dist1.json 2> gpurun_out/bench_cfg3_dist1.err; head -c 400 gpurun_out/bench_cfg3_dist1.json
timeout 300 python bench.py --steps 20 --warmup 5 --force-dist --rows 125000000 --no-cpu-baseline --no-load --no-canonical > gpurun_out/bench_cfg3_dist1_shard.json 2> gpurun_out/bench_cfg3_dist1_shard.err; head -c 400 gpurun_out/bench_cfg3_dist1_shard.json
SYBL_FORCE_SCATTER=1 timeout 300 python bench.py --steps 5 --warmup 2 --workload cfg4_hist_highcard --force-dist --no-cpu-baseline --no-load --no-canonical > gpurun_out/bench_cfg4_dist1.json 2> gpurun_out/bench_cfg4_dist1.err; head -c 400 gpurun_out/bench_cfg4_dist1.json; tail -3 gpurun_out/bench_cfg4_dist1.err
