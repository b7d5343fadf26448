cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof4
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4/kt -o kt -- python tools/bench_configs.py 0 3 cfg4 > gpurun_out/prof4/kt.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof4/kt/*.db | cut -c1-170 | head -12
grep "^{" gpurun_out/prof4/kt.log
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4/kt3 -o kt3 -- python -c "
import sys; sys.path.insert(0,'.')
import sybil_amd
from sybil_amd import synth
ctx=sybil_amd.Context(0)
wl=synth.WORKLOADS['cfg3_filter3_group2_stddev']
t=ctx.synth_table('x',synth.SEED,1000000000,0,1000000000,synth.synth_cols(wl['columns']))
q=t.query(**dict(wl['query'],want_percentiles=True))
for i in range(3):
    r=q.run(); print(q.stats()['strategy'], q.stats()['scan_ms']); r.free()
" > gpurun_out/prof4/kt3.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof4/kt3/*.db | cut -c1-170 | head -8
grep -v "^W\|^E\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/prof4/kt3.log | tail -4
rm -rf gpurun_out/prof4/kt gpurun_out/prof4/kt3
