#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -12
for c in c1 c5 c3 c2 c4; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2>gpurun_out/r05_bench_$c.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r05_bench_$c.json'))
print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'steps', d['roofline'].get('newton_schulz_steps',{}).get('mean'), 'steady', d.get('steady_state',{}).get('value'))
"
done
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_c2_driver_$i.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r05_bench_c2_driver_$i.json')); print('driver', d['value'], d['roofline']['frac'], d['steady_state']['value'])"
done
bash tools/prof_round5.sh c2 c2_20 c1 c3 c4 c5 2>&1 | tail -12
