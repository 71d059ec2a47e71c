#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -5 gpurun_out/r3b_tests.log
run() { # name, env..., args
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown $BARGS > gpurun_out/r3b_$name.json 2> gpurun_out/r3b_$name.err
  python - gpurun_out/r3b_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f steps %.2f" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["newton_schulz_steps"]["mean"]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
P
}
BARGS="--batch 0" run b0 A=1
BARGS="--batch 100" run b100_occ4 A=1
BARGS="--batch 100" run b100_occ3 CUADMM_CU_OCC=3
BARGS="--batch 100" run b100_nofence CUADMM_CU_X=1
BARGS="--batch 0" run b0_again A=1
