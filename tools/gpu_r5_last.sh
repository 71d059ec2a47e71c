#!/bin/bash
# round 5, last check: c1 / c5 bench lines with the y_solve object, the bench tests
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in c1 c5; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2>gpurun_out/r05_bench_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/r05_bench_$c.json')); print('$c', round(d['value'],1), d.get('y_solve'))"
done
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_bench.py -q 2>&1 | tail -3
