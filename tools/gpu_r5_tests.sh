#!/bin/bash
# round 5: the whole -m gpu suite + the driver's bench line
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu -x > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -20
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_c2_driver.json 2>gpurun_out/r05_bench_c2_driver.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_c2_driver.json')); print(d['value'], d['roofline']['frac'], d['cpu_baseline'])"
