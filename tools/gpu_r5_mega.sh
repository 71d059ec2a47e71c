#!/bin/bash
# round 5: the mega-lift of the sign schedule -- projection tests, the solver tests that failed/were cut off, c1/c3/c5 bench lines
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_psd.py tests/test_gpu_solver.py tests/test_gpu_fused.py tests/test_gpu_batch.py -q -x > gpurun_out/r05_mega_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_mega_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_mega_tests.log | tail -8
for c in c1 c5 c3 c2 c4; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_mega_bench_$c.json 2>gpurun_out/r05_mega_bench_$c.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r05_mega_bench_$c.json'))
print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'steps', d['roofline'].get('newton_schulz_steps'), {k:round(v['ms'],3) for k,v in d.get('breakdown',{}).items() if isinstance(v,dict) and 'ms' in v})
"
done
