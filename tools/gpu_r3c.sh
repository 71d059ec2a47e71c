#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fused.py tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r3c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3c_tests.log
tail -15 gpurun_out/r3c_tests.log
CUADMM_CU_DBG=1 timeout 300 python bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-breakdown --batch 40 2>&1 | grep "cu debug" | tail -2
for b in 0 32 100; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown --batch $b > gpurun_out/r3c_b$b.json 2> gpurun_out/r3c_b$b.err
python - gpurun_out/r3c_b$b.json b$b <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f frac %.4f steps %.2f" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"]))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
