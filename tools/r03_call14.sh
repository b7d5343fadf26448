#!/bin/bash
# Round 3: the artefacts the documents cite -- the bench command under rocprofv3 (kernel trace + counter passes), the bench
# line itself, the counter pass of the row-body variants, the writer / loader tests after the last changes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_writer.py tests/test_gpu_loader.py tests/test_gpu_fullsize.py -q --tb=short > gpurun_out/r03_c14_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r03_c14_tests.log | tail -1)"
timeout -k 10 900 python bench.py > gpurun_out/r03_bench_cfg3_1gpu.json 2> gpurun_out/r03_c14_bench.err; tail -c 200 gpurun_out/r03_c14_bench.err
BENCH_ARGS="--no-configs --no-load --no-oracle-check" PROF_TAG=r03 bash tools/prof_r01.sh > gpurun_out/r03_prof_headline.log 2>&1
head -8 gpurun_out/prof/r03_kernel_trace_stats.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pv -o pv -- python $GRAFT_REPO_ROOT/tools/bench_variants.py > $GRAFT_REPO_ROOT/gpurun_out/pv.log 2>&1
cd $GRAFT_REPO_ROOT; { echo "# rocprofv3 --pmc ... -- python tools/bench_variants.py   (MI355X, round 3; config 3, 1e9 rows, compact storage, through each row body)"; grep "^{" gpurun_out/pv.log; python tools/rocpd_summary.py gpurun_out/pv/*.db 2>/dev/null | grep "k_scan\|^kernel"; } > gpurun_out/r03_hash_variants_pmc.txt; rm -rf gpurun_out/pv
head -12 gpurun_out/r03_hash_variants_pmc.txt | cut -c1-150
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_cfg3_1gpu.json') if l.startswith('{')][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for c in d.get("configs", []): print(c["config"]["workload"], round(c["value"]/1e9,1), "Grows/s", round(c["ms_per_step"],3), "ms/step kernel", round(c["kernel_ms"],3), "frac", round(c["roofline"]["frac"],3), c["config"]["host_ms_per_step"])
for k in ("load","load_mixed_4col"): print(k, round(d[k]["rows_per_s"]/1e6), "M rows/s")
print("cpu", d["cpu_baseline"]["value"], d.get("cpu_baseline_columnar",{}).get("value"))
P
