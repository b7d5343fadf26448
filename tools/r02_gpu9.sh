#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_writer.py tests/test_gpu_loader.py -q -x --tb=short > gpurun_out/pytest_fz.log 2>&1; tail -8 gpurun_out/pytest_fz.log
python - <<'PY'
import time, tempfile, shutil, sybil_amd
from sybil_amd import synth
ctx = sybil_amd.Context(0)
rows = 100 * 1024 * 1024 // 65536 * 65536
t = ctx.synth_table("w", synth.SEED, rows, 0, rows, synth.synth_cols(["c00", "c01", "c07", "c09"]))
root = tempfile.mkdtemp()
t0 = time.perf_counter(); t.save(root); print("sybl_table_save of %d rows x 4 columns: %.2f s" % (rows, time.perf_counter() - t0))
shutil.rmtree(root)
PY
