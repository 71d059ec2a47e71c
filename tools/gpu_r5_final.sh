#!/bin/bash
# round 5, closing run: the whole -m gpu suite, every bench line, profiling passes of c1 / c5 (their kernels changed last), smoke
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for c in c1 c5 c3 c2 c4; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2>gpurun_out/r05_bench_$c.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r05_bench_$c.json'))
print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'steady', d.get('steady_state',{}).get('value'))
"
done
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_c2_driver.json 2>gpurun_out/r05_bench_c2_driver.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_c2_driver.json')); print('driver', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
bash tools/prof_round5.sh c1 c5 2>&1 | tail -6
