#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/lines_t16.log
: > $O
line() { name=$1; shift
  timeout 900 python bench.py "$@" 2>/dev/null | grep '^{' > gpurun_out/t16_bench_$name.json
  python - gpurun_out/t16_bench_$name.json $name >> $O <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("%-14s %9.1f %s  ms/step %.4f frac %.4f" % (sys.argv[2], d["value"], d["unit"][:12], d["ms_per_step"], r["frac"]), d.get("breakdown_ms_per_iter"))
except Exception as e:
    print(sys.argv[2], "ERR", e)
P
}
line c2_driver --steps 20 --warmup 5 --no-cpu-baseline
line c2 --steps 200 --warmup 20 --no-cpu-baseline
line c1 --config c1 --no-cpu-baseline
line c5 --config c5 --no-cpu-baseline
line c4 --config c4 --no-cpu-baseline
bash tools/gpu_real_r4.sh > /dev/null 2>&1
cut -c1-200 gpurun_out/r04_real_data.log >> $O
python -m pytest tests/test_gpu_sharded_procs.py tests/test_gpu_longrun.py -q -m gpu -x 2>&1 | tail -3 >> $O
cat $O
