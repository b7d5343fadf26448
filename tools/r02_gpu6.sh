#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in async sync; do
  if [ $MODE = async ]; then export SYBL_EMIT_ASYNC=1; else unset SYBL_EMIT_ASYNC; fi
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$MODE -o kt -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 0 3 cfg4,cfg3 compact > $GRAFT_REPO_ROOT/gpurun_out/prof_$MODE.log 2>&1
  echo "== $MODE"; grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_$MODE.log | cut -c1-200
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_$MODE/*.db 2>/dev/null | grep -v "rocclr\|k_synth\|k_block_minmax\|k_fill\|k_repack" | cut -c1-150 | head -14
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$MODE
done
