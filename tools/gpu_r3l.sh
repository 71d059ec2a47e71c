#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2
run() { name=$1; shift
  timeout 900 python bench.py "$@" > gpurun_out/r3l_$name.json 2> gpurun_out/r3l_$name.err
  python - gpurun_out/r3l_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "value %.1f ms/step %.4f frac %.4f" % (d["value"], d["ms_per_step"], r["frac"]), {k:d.get(k) for k in ("time_to_tol","breakdown_ms_per_iter")}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
}
run c1 --config c1 --no-cpu-baseline
run c5 --config c5 --no-cpu-baseline
run c2_sparse --c-sparse --no-cpu-baseline
run c2_proj --projection-only
run c4_proj --config c4 --projection-only --steps 20 --warmup 3
run c2 --steps 20 --warmup 5
