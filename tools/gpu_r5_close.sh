#!/bin/bash
# round 5, closing: the whole -m gpu suite as the driver runs it (-x), the duo tests ten times over, smoke, the bench lines
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -6
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 600 python -m pytest tests/test_gpu_solver.py -q -x -k "duo" 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
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
CUADMM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --blocks-per-gpu 5000 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks one GPU (gloo):', d['n_gpus'], d['value'], d['allreduce_path']['value'])"
CUADMM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c1 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1 on 2 ranks, one GPU (gloo):', d['n_gpus'], round(d['value'],1), d.get('engine_plan'))"
bash tools/prof_round5.sh c1 c5 2>&1 | tail -2
