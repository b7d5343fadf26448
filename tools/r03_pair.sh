#!/bin/bash
# the default bench line and a kernel trace of the headline on ONE box (profiles/r03_bench_cfg3_1gpu.json + r03_kernel_trace_stats.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_final; mkdir -p $OUT; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-load --no-configs --no-oracle-check > $OUT/kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-load --no-configs --no-oracle-check   (MI355X, $(date -u +%Y-%m-%dT%H:%MZ); same box as profiles/r03_bench_cfg3_1gpu.json)"; python tools/rocpd_summary.py $OUT/kt/*.db | grep -v "rocclr\|k_fill"; echo; grep '^{' $OUT/kt.log | cut -c1-600; } > gpurun_out/r03_final_kernel_trace.txt
rm -rf $OUT/kt
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03_final_bench.json') if l.startswith('{')][-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
for c in d.get('configs',[]): print('  ', c.get('config',{}).get('workload'), c.get('ms_per_step'), c.get('kernel_ms'), c.get('error'))
P
sed -n 3,6p gpurun_out/r03_final_kernel_trace.txt | cut -c1-150
