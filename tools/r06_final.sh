#!/bin/bash
# Round-6 closing run (adds: the selectivity table and FETCH_SIZE pass of k_scan_hash_packed's late path, the LDS atomic rate
# microbenchmark, the cold CLI's phase line; the bench line now carries cold_cli and config 4's count_pass): the whole GPU suite, the default bench line, every profile the bench line cites regenerated from the
# same HEAD (headline kernel trace + PMC, configs 2 / 4 / 5 kernel trace + PMC), the variant / wide / selectivity tables the
# round-4 profiles lacked with a FETCH_SIZE pass at 0.1 % and 0 % selectivity (item 6), a clock / power /
# temperature log beside config 4's kernel trace (item 3d), the loads-only microbenchmark of config 2's shape (item 7), smoke().
# Afterwards, in the repo:
#   cp gpurun_out/prof/r06_* profiles/; for c in cfg2 cfg4 cfg5; do cp gpurun_out/prof_r06_$c/r06_${c}_* profiles/; done
#   for f in gpurun_out/r06/r06_*.txt; do grep -v amdgpu.ids $f > profiles/$(basename $f); done; python tools/make_traffic.py r06; cp gpurun_out/r06_final_bench.json profiles/r06_bench_1gpu.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
O=gpurun_out/r06
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r06_final_tests.log 2>&1
echo "tests: $(grep -n 'passed\|failed' gpurun_out/r06_final_tests.log | tail -1)"; grep -n "^FAILED\|^ERROR" gpurun_out/r06_final_tests.log | head -10
fi
timeout -k 10 500 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
python - <<'Q'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r06_final_bench.json') if l.startswith('{')][-1])
    print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'], 'load', d.get('load',{}).get('rows_per_s'), d.get('load',{}).get('stage_breakdown',{}).get('parse_cpu_s'), 'host parser', (d.get('load',{}).get('host_parser') or {}).get('rows_per_s'), 'loaded scan', d.get('loaded_table_scan',{}).get('kernel_ms'))
    for c in d.get('configs',[]): print('  ', c.get('config',{}).get('workload'), c.get('ms_per_step'), c.get('roofline',{}).get('kernel_ms'), c.get('config',{}).get('host_ms_per_step'), c.get('every_row_summarised',{}).get('ms_per_step'), c.get('back_to_back_scan_ms'), c.get('error'))
except Exception as e: print('bench parse failed', e)
Q
if [ -z "$SKIP_PROFILES" ]; then
PROF_TAG=r06 BENCH_ARGS="--no-load --no-configs --no-oracle-check --no-canonical" bash tools/prof_r01.sh > gpurun_out/r06_final_prof.log 2>&1
head -8 gpurun_out/prof/r06_kernel_trace_stats.txt | cut -c1-150
for wl in cfg2 cfg5; do WL=$wl TAG=r06_$wl bash tools/prof_cfg.sh > gpurun_out/r06_final_prof_$wl.log 2>&1; head -7 gpurun_out/prof_r06_$wl/r06_${wl}_kernel_trace.txt | tail -4 | cut -c1-150; done
# (config 4's clock / power / temperature evidence is tools/clock_scan.py below: scans back to back on the process' own GPU)
env WL=cfg4 TAG=r06_cfg4 bash tools/prof_cfg.sh > gpurun_out/r06_final_prof_cfg4.log 2>&1
head -7 gpurun_out/prof_r06_cfg4/r06_cfg4_kernel_trace.txt | tail -4 | cut -c1-150
fi
if [ -z "$SKIP_TABLES" ]; then
{ echo "# tools/bench_variants.py at HEAD (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 3, compact storage, 1e9 rows, through each row body"; timeout -k 10 400 python tools/bench_variants.py; } > $O/r06_variants.txt 2>&1
{ echo "# tools/bench_wide.py at HEAD (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"; timeout -k 10 400 python tools/bench_wide.py; } > $O/r06_wide_aggs.txt 2>&1
{ echo "# tools/bench_selectivity.py at HEAD (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 3 at different selectivities, back-to-back scans"; timeout -k 10 300 python tools/bench_selectivity.py; } > $O/r06_selectivity.txt 2>&1
# the skipped bytes, counter-proven: FETCH_SIZE of the scan kernel at the headline's selectivity, at 0.1 % and at 0 %
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/sel_pmc -o sel -- python $GRAFT_REPO_ROOT/tools/bench_selectivity.py 1000000000 1 > $GRAFT_REPO_ROOT/$O/sel_pmc.log 2>&1 )
{ echo "# rocprofv3 --pmc FETCH_SIZE -- python tools/bench_selectivity.py 1000000000 1   (KiB per dispatch; x2 on gfx950; dispatches in the order of the"
  echo "# selectivity table: 51 %, 10 %, 1 %, 0.1 %, 1e-6, 0 -- four scans each)"; python tools/rocpd_summary.py $O/sel_pmc/*.db 2>/dev/null | grep -v "rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack"; python tools/rocpd_dispatches.py $O/sel_pmc/*.db k_scan_packed FETCH_SIZE 2>/dev/null; } > $O/r06_selectivity_fetch_size.txt
rm -rf $O/sel_pmc
for v in hash wide nul; do { echo "# tools/bench_selectivity.py 1000000000 5 $v at HEAD (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 3's shape through k_scan_hash_packed (hash: SYBL_FORCE_HASH=1; wide: four aggregations, direct-mapped) at different selectivities"; timeout -k 10 300 python tools/bench_selectivity.py 1000000000 5 $v; } > $O/r06_selectivity_$v.txt 2>&1; done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/selh_pmc -o sel -- python $GRAFT_REPO_ROOT/tools/bench_selectivity.py 1000000000 1 hash > $GRAFT_REPO_ROOT/$O/selh_pmc.log 2>&1 )
{ echo "# rocprofv3 --pmc FETCH_SIZE -- python tools/bench_selectivity.py 1000000000 1 hash   (k_scan_hash_packed with late materialisation, round 6; KiB per dispatch; x2 on gfx950;"
  echo "# dispatches in the order of the selectivity table: 51 %, 10 %, 1 %, 0.1 %, 1e-6, 0 -- four scans each)"; python tools/rocpd_dispatches.py $O/selh_pmc/*.db k_scan_hash_packed FETCH_SIZE 2>/dev/null; } > $O/r06_selectivity_hash_fetch_size.txt
rm -rf $O/selh_pmc
# config 4's kernels with filters (k_count_packed / k_emit_packed<.., LATE>): the planner's choice, then each form forced
{ echo "# tools/bench_selectivity.py 1000000000 5 parthist at HEAD (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 4 (65 536 groups, full histograms: counting pass + emit + k_part_hist, the counting pass in every scan) with config 3's three filter columns; the planner's estimate, then SYBL_LATE_PATH=0 and =1"; timeout -k 10 300 python tools/bench_selectivity.py 1000000000 5 parthist; SYBL_LATE_PATH=0 timeout -k 10 300 python tools/bench_selectivity.py 1000000000 5 parthist; SYBL_LATE_PATH=1 timeout -k 10 300 python tools/bench_selectivity.py 1000000000 5 parthist; } > $O/r06_selectivity_parthist.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/selp_pmc -o sel -- python $GRAFT_REPO_ROOT/tools/bench_selectivity.py 1000000000 1 parthist > $GRAFT_REPO_ROOT/$O/selp_pmc.log 2>&1 )
{ echo "# rocprofv3 --pmc FETCH_SIZE -- python tools/bench_selectivity.py 1000000000 1 parthist   (KiB per dispatch; x2 on gfx950; dispatches in the order of the"
  echo "# selectivity table: 51 %, 10 %, 1 %, 0.1 %, 1e-6, 0 -- four scans each; the planner takes the LATE kernels from 0.1 % down)"; python tools/rocpd_dispatches.py $O/selp_pmc/*.db k_emit_packed FETCH_SIZE 2>/dev/null; python tools/rocpd_dispatches.py $O/selp_pmc/*.db k_count_packed FETCH_SIZE 2>/dev/null; } > $O/r06_selectivity_parthist_fetch_size.txt
rm -rf $O/selp_pmc
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/pdkt -o kt -- python $GRAFT_REPO_ROOT/tools/bench_pushdown.py > $GRAFT_REPO_ROOT/$O/pd.log 2>&1 )
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_pushdown.py (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 4 as a printer's query, the full path (printed_only = 1, strategy 5) and -limit 100 pushed into the scan (printed_only = 2, strategy 8), ten back-to-back scans each"; grep "^{" $O/pd.log; python tools/rocpd_summary.py $O/pdkt/*.db 2>/dev/null | grep -v "rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack"; } > $O/r06_pushdown_kernel_trace.txt
rm -rf $O/pdkt
{ echo "# tools/micro/ldsrate (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): LDS atomic issue rates per CU with k_scan_packed's low-cardinality table layout"; timeout -k 10 120 tools/micro/ldsrate; } > $O/r06_ldsrate.txt 2>&1
{ echo "# tools/cold_cli_phases.py (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): sybil-gpu-query -stats on the saved 104.9 M-row, 7-column table, a fresh process each; the last line: the process without a table"; SYBL_LOADER_TRACE=1 timeout -k 10 300 python tools/cold_cli_phases.py; } > $O/r06_cold_cli_phases.txt 2>&1
{ echo "# tools/bench_load_varint.py (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): the bench's saved table opened four times by the host parser (SYBL_LOADER_GPU_VARINT=0), then four times with the int columns' varints walked on the GPU (the default)"; timeout -k 10 300 python tools/bench_load_varint.py; } > $O/r06_gpu_varint_ab.txt 2>&1
{ echo "# tools/micro/loadpat_cfg2 (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"; timeout -k 10 120 tools/micro/loadpat_cfg2; } > $O/r06_loadpat_cfg2.txt 2>&1
{ echo "# tools/bench_dictkey.py (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"; timeout -k 10 300 python tools/bench_dictkey.py; } > $O/r06_dictkey.txt 2>&1
{ echo "# tools/clock_scan.py cfg4 25 3 (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): config 4's scan kernels back to back for 25 s after 3 s of idle"; timeout -k 10 200 python tools/clock_scan.py cfg4 25 3; } > $O/r06_cfg4_clock_power_temp_scan.txt 2>&1
{ echo "# tools/emit_placement.py (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"; timeout -k 10 200 python tools/emit_placement.py 6 3; } > $O/r06_emit_placement.txt 2>&1
{ echo "# tools/ab_scan.py cfg4 (MI355X, $(date -u +%Y-%m-%dT%H:%MZ)): the counting pass reused across rescans against counted by every scan (SYBL_NO_COUNT_CACHE=1)"; timeout -k 10 400 python tools/ab_scan.py cfg4 3 count_cached=- count_every_scan=-,SYBL_NO_COUNT_CACHE=1; } > $O/r06_ab_count_cache.txt 2>&1
tail -3 $O/r06_selectivity_hash.txt | cut -c1-160; tail -6 $O/r06_cold_cli_phases.txt | cut -c1-400; tail -4 $O/r06_variants.txt; tail -7 $O/r06_selectivity.txt | cut -c1-160; tail -14 $O/r06_loadpat_cfg2.txt; tail -3 $O/r06_dictkey.txt; tail -4 $O/r06_emit_placement.txt; tail -3 $O/r06_ab_count_cache.txt; tail -6 $O/r06_cfg4_clock_power_temp_scan.txt
fi
timeout -k 10 120 python bench.py --force-dist --no-cpu-baseline --no-load --no-canonical --no-configs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('force-dist', d['ms_per_step'], d['n_gpus'], str(d.get('oracle_check'))[:80])"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
