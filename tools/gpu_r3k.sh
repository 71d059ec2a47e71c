#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fused.py tests/test_gpu_solver.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r3k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3k_tests.log
tail -6 gpurun_out/r3k_tests.log
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline $BARGS > gpurun_out/r3k_$name.json 2> gpurun_out/r3k_$name.err
  python - gpurun_out/r3k_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f frac %.4f steps %.2f" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"]), d["engine_plan"], d.get("breakdown_ms_per_iter"))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
}
BARGS="--config c4" run c4 A=1
BARGS="--config c4" run c4_notiny CUADMM_TINY_SIGN=0
BARGS="--config c4 --batch 0" run c4_b0 A=1
BARGS="--config c2 --mode sgs" run c2_sgs A=1
