cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof5
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5/kt -o kt -- python tools/bench_fewgroups.py > gpurun_out/prof5/kt.log 2>&1
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/prof5/kt/*.db')[0])
rows = c.execute("select name, (end-start)/1e6 from kernels where name like '%k_emit%' or name like '%k_part_hist%' order by start").fetchall()
for r in rows: print("%-60s %.2f ms" % (r[0][:60], r[1]))
PY
rm -rf gpurun_out/prof5/kt
