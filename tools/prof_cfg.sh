#!/bin/bash
# Profile of one BASELINE.json workload through tools/bench_configs.py: kernel trace + the PMC passes of
# MI355X_MICROARCH.md (counters in their own runs; FETCH_SIZE and WRITE_SIZE in separate passes).
# usage (on the GPU box): WL=cfg4 TAG=r02_cfg4 [STORAGE=compact] [EXTRA=1 | LEAN=1 (kernel trace + the two HBM passes only)] bash tools/prof_cfg.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${WL:-cfg4}
TAG=${TAG:-r02_$WL}
STORAGE=${STORAGE:-compact}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
CMD="python tools/bench_configs.py ${ROWS:-0} 3 $WL $STORAGE"
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
if [ -z "$LEAN" ]; then
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
fi
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
[ -z "$LEAN" ] && timeout -k 10 900 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
# EXTRA=1: the passes DESIGN 3.2 asks for to find what k_emit waits for (counter names as rocprofv3 -L lists them on
# gfx942/gfx950; a pass whose counter does not exist fails on its own and is reported below)
EXTRA_PASSES=""
if [ -n "$EXTRA" ]; then
  timeout -k 10 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $OUT/pmc6 -o pmc6 -- $CMD > $OUT/pmc6.log 2>&1
  timeout -k 10 900 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT -d $OUT/pmc7 -o pmc7 -- $CMD > $OUT/pmc7.log 2>&1
  timeout -k 10 900 rocprofv3 --pmc TA_BUSY_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum -d $OUT/pmc8 -o pmc8 -- $CMD > $OUT/pmc8.log 2>&1
  timeout -k 10 900 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_WRITE_sum -d $OUT/pmc9 -o pmc9 -- $CMD > $OUT/pmc9.log 2>&1
  EXTRA_PASSES="pmc6 pmc7 pmc8 pmc9"
fi
FILT="rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack"
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"
  python tools/rocpd_summary.py $OUT/kt/*.db
  echo; echo "# bench_configs line of the traced run"; grep '^{' $OUT/kt.log
} > $OUT/${TAG}_kernel_trace.txt
{
  echo "# rocprofv3 --pmc passes -- $CMD   (one pass per counter group; FETCH_SIZE / WRITE_SIZE are in KiB;"
  echo "# on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream: MI355X_MICROARCH.md section HBM)"
  for p in pmc1 pmc2 pmc3 pmc4 pmc5 $EXTRA_PASSES; do python tools/rocpd_summary.py $OUT/$p/*.db 2>/dev/null | grep -v "$FILT"; done
} > $OUT/${TAG}_pmc.txt
for p in pmc1 pmc2 pmc3 pmc4 pmc5 $EXTRA_PASSES; do grep -i "error\|invalid\|not supported" $OUT/$p.log | head -3; done
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 $OUT/pmc7 $OUT/pmc8 $OUT/pmc9
cat $OUT/${TAG}_kernel_trace.txt | cut -c1-150 | head -14
cat $OUT/${TAG}_pmc.txt | cut -c1-150
