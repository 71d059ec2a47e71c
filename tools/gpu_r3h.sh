#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fused.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r3h_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3h_tests.log
tail -4 gpurun_out/r3h_tests.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown $BARGS > gpurun_out/r3h_$name.json 2> gpurun_out/r3h_$name.err
  python - gpurun_out/r3h_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f frac %.4f steps %.2f" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"]))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
}
BARGS="--batch 100" run b100 A=1
BARGS="--batch 100" run b100_nohint CUADMM_PSD_HINT=0
BARGS="--batch 100" run b100_occ3 CUADMM_CU_OCC=3
BARGS="--batch 32" run b32 A=1
BARGS="--batch 0" run b0 A=1
BARGS="--batch 32 --steps 20 --warmup 5" run b32_s20 A=1
