mkdir -p gpurun_out
run() { env $1 timeout 200 python bench.py --workload cfg2_group1_avg2 --no-load --no-configs --no-canonical --no-cpu-baseline --no-oracle-check --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'wgs', d['roofline'].get('workgroups'), 'lds', d['roofline'].get('lds_bytes'))"; }
for rep in 1 2; do
run "SYBL_X=0"
run "SYBL_WG_PER_CU=2 SYBL_REP_BUDGET_KB=56"
run "SYBL_WG_PER_CU=2 SYBL_REP_BUDGET_KB=72"
run "SYBL_WG_PER_CU=2 SYBL_REP_BUDGET_KB=28"
run "SYBL_WG_PER_CU=3 SYBL_REP_BUDGET_KB=48"
run "SYBL_WG_PER_CU=2 SYBL_REP_BUDGET_KB=56 SYBL_PACKED_RING=2"
done
