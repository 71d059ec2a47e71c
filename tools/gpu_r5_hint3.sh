#!/bin/bash
# round 5: schedule warm start for the mid-size groups of the batched-GEMM path: the suite, c1 / c3 lines, taha1a, PlanarHand_N=10
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -10
for c in c1 c3; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2>gpurun_out/r05_bench_$c.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r05_bench_$c.json'))
print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), d['roofline'].get('newton_schulz_steps'))
"
done
timeout 600 python tools/run_large.py taha1a 11000 1500 2>&1 | grep RESULT | cut -c1-300
timeout 600 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 300 2>&1 | grep RESULT | cut -c1-300
