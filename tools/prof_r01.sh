#!/bin/bash
# Round-1 profile of the bench command: kernel trace + the PMC passes of MI355X_MICROARCH.md
# (counters in their own runs; FETCH_SIZE and WRITE_SIZE in separate passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS"
TAG=${PROF_TAG:-r01}
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $BENCH   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"
  python tools/rocpd_summary.py $OUT/kt/*.db
  echo; echo "# bench line of the traced run"; grep '^{' $OUT/kt.log
} > $OUT/${TAG}_kernel_trace_stats.txt
{
  echo "# rocprofv3 --pmc passes -- $BENCH   (one pass per counter group; FETCH_SIZE is in KiB and on gfx950"
  echo "# reports 1/2 of a wide coalesced read stream: MI355X_MICROARCH.md section HBM)"
  for p in pmc1 pmc2 pmc3 pmc4; do python tools/rocpd_summary.py $OUT/$p/*.db | grep -v "rocclr\|k_synth\|k_block_minmax\|k_fill"; done
} > $OUT/${TAG}_pmc.txt
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
ls -la $OUT
