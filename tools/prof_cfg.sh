#!/bin/bash
# Profile of one BASELINE.json workload through tools/bench_configs.py: kernel trace + the PMC passes of
# MI355X_MICROARCH.md (counters in their own runs; FETCH_SIZE and WRITE_SIZE in separate passes).
# usage (on the GPU box): WL=cfg4 TAG=r02_cfg4 [STORAGE=compact] bash tools/prof_cfg.sh
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
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
timeout -k 10 900 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
FILT="rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack"
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ))"
  python tools/rocpd_summary.py $OUT/kt/*.db
  echo; echo "# bench_configs line of the traced run"; grep '^{' $OUT/kt.log
} > $OUT/${TAG}_kernel_trace.txt
{
  echo "# rocprofv3 --pmc passes -- $CMD   (one pass per counter group; FETCH_SIZE / WRITE_SIZE are in KiB;"
  echo "# on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream: MI355X_MICROARCH.md section HBM)"
  for p in pmc1 pmc2 pmc3 pmc4 pmc5; do python tools/rocpd_summary.py $OUT/$p/*.db | grep -v "$FILT"; done
} > $OUT/${TAG}_pmc.txt
for p in pmc1 pmc2 pmc3 pmc4 pmc5; do grep -i "error\|invalid\|not supported" $OUT/$p.log | head -3; done
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
cat $OUT/${TAG}_kernel_trace.txt | cut -c1-150 | head -14
cat $OUT/${TAG}_pmc.txt | cut -c1-150
