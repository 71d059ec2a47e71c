#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown $BARGS > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err
  python - gpurun_out/r3e_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f frac %.4f steps %.2f" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"]))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
}
BARGS="--batch 0" run b0_prio0 CUADMM_CLOSED_PRIO=0
BARGS="--batch 0" run b0_prio1 CUADMM_CLOSED_PRIO=1
BARGS="--batch 100" run b100_prio0 CUADMM_CLOSED_PRIO=0
BARGS="--batch 100" run b100_prio1 CUADMM_CLOSED_PRIO=1
CUADMM_CU_DBG=1 timeout 300 python bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-breakdown --batch 40 2>&1 | grep "cu debug" | tail -1
