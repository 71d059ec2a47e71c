#!/bin/bash
# round 3, first GPU call: batching correctness + C2 A/B
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fused.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r3a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a_tests.log
for b in 0 32; do
  timeout 300 python bench.py --steps 200 --warmup 20 --batch $b --no-cpu-baseline > gpurun_out/r3a_bench_c2_b$b.json 2> gpurun_out/r3a_bench_c2_b$b.err
  timeout 300 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline > gpurun_out/r3a_bench_c2_b${b}_s20.json 2>> gpurun_out/r3a_bench_c2_b$b.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --batch 100 --no-cpu-baseline > gpurun_out/r3a_bench_c2_b100.json 2> gpurun_out/r3a_bench_c2_b100.err
tail -3 gpurun_out/r3a_tests.log
for f in gpurun_out/r3a_bench_c2_*.json; do echo $f; python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d["roofline"]
    print(d["value"], d["ms_per_step"], r["ms_per_iteration"], r["frac"], r["newton_schulz_steps"]["mean"], d["engine_plan"], d.get("breakdown_ms_per_iter"))
except Exception as e:
    print("ERR", e)
P
done
