#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_variants.py 2>&1 | grep "^{"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pv -o pv -- python $GRAFT_REPO_ROOT/tools/bench_variants.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py gpurun_out/pv/*.db 2>/dev/null | grep "k_scan" | cut -c1-160; rm -rf gpurun_out/pv
