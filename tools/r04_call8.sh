#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/A.so python tools/scan_loop.py cfg4 60 compact | cut -c1-140
python tools/scan_loop.py cfg4 60 compact | cut -c1-140
SYBL_LIBRARY=$GRAFT_REPO_ROOT/ab/A.so python tools/scan_loop.py cfg4 60 compact | cut -c1-140
python tools/scan_loop.py cfg4 60 compact | cut -c1-140
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_loop; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python tools/scan_loop.py cfg4 60 compact > $OUT/kt.log 2>&1
python tools/rocpd_summary.py $OUT/kt/*.db | grep "k_emit\|k_part\|k_count\|kernel " | cut -c1-150
python - <<'Q'
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/prof_loop/kt/*.db')[0])
for nm in ('k_emit_packed','k_part_hist','k_count_packed'):
    rows=[r[0] for r in c.execute("select end-start from kernels where name like ? order by start", ('%'+nm+'%',)).fetchall()]
    print(nm, 'last 20 avg us', sum(rows[-20:])/20/1e3, 'first 3', [round(x/1e3) for x in rows[:3]])
Q
rm -rf $OUT/kt
