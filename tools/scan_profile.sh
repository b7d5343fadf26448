#!/bin/bash
# steady-state per-kernel times of one workload's scan: rocprofv3 kernel trace over tools/scan_loop.py (back-to-back scans,
# no finalize in between: the GPU stays at its working clocks), the average of the last 20 dispatches of every kernel.
# usage (on the GPU box): scan_profile.sh <label> <workload> [n]      (SYBL_LIBRARY / switches from the environment)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_sp_$1; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python tools/scan_loop.py $2 ${3:-60} compact > $OUT/kt.log 2>&1
python - "$1" "$OUT" <<'Q'
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(sys.argv[2] + '/kt/*.db')[0])
names = [r[0] for r in c.execute("select distinct name from kernels where name like '%sybl::k_%'").fetchall()]
out = []
for nm in names:
    rows = [r[0] for r in c.execute("select end-start from kernels where name = ? order by start", (nm,)).fetchall()]
    if len(rows) >= 20 and 'k_synth' not in nm and 'k_repack' not in nm and 'k_block_minmax' not in nm:
        out.append((sum(rows[-20:]) / 20 / 1e3, nm.split('sybl::')[1].split('(')[0]))
print(sys.argv[1], " ".join("%s=%.1f" % (n, t) for t, n in sorted(out, reverse=True)), "sum=%.1f us" % sum(t for t, _ in out))
Q
rm -rf $OUT
