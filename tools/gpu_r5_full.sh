#!/bin/bash
# round 5: the whole -m gpu suite, every bench line, the GEMM ubench
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./tools/ubench/gemm_sym48.exe > gpurun_out/r05_gemm_sym48.log 2>&1
grep "N = 2016\|vs shipped" gpurun_out/r05_gemm_sym48.log | head -12
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -12
for c in c1 c5 c3 c2 c4; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2>gpurun_out/r05_bench_$c.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r05_bench_$c.json'))
print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'steps', d['roofline'].get('newton_schulz_steps',{}).get('mean'))
"
done
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_c2_driver.json 2>gpurun_out/r05_bench_c2_driver.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_c2_driver.json')); print('driver', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['eig_engine'])"
