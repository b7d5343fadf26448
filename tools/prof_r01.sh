cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 -L > gpurun_out/prof/counters.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/kt.log 2>&1
ls -R gpurun_out/prof/kt | head -30
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d gpurun_out/prof/pmc1 -o pmc1 -- python bench.py --steps 2 --warmup 1 --rows 400000000 --no-cpu-baseline > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof/pmc2 -o pmc2 -- python bench.py --steps 2 --warmup 1 --rows 400000000 --no-cpu-baseline > gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc3 -o pmc3 -- python bench.py --steps 2 --warmup 1 --rows 400000000 --no-cpu-baseline > gpurun_out/prof/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof/pmc4 -o pmc4 -- python bench.py --steps 2 --warmup 1 --rows 400000000 --no-cpu-baseline > gpurun_out/prof/pmc4.log 2>&1
find gpurun_out/prof -name "*.csv" | head -30
