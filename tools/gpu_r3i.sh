#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-breakdown $BARGS > gpurun_out/r3i_$name.json 2> gpurun_out/r3i_$name.err
  python - gpurun_out/r3i_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "iters/s %.0f ms/step %.4f psd/iter %.4f frac %.4f steps %.2f launches %d" % (d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"], r["launches"]))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
}
BARGS="--batch 64 --steps 20 --warmup 5" run s20_w5 A=1
BARGS="--batch 64 --steps 20 --warmup 50" run s20_w50 A=1
BARGS="--batch 64 --steps 20 --warmup 200" run s20_w200 A=1
BARGS="--batch 64 --steps 40 --warmup 5" run s40_w5 A=1
BARGS="--batch 64 --steps 100 --warmup 5" run s100_w5 A=1
BARGS="--batch 0 --steps 20 --warmup 5" run b0_s20_w5 A=1
BARGS="--batch 0 --steps 20 --warmup 200" run b0_s20_w200 A=1
