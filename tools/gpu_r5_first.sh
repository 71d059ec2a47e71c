#!/bin/bash
# round 5, first trip: fp64 MFMA shapes ubench, the self-launching bench test, the driver's bench line
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./tools/ubench/mfma_shapes.exe > gpurun_out/r05_mfma_shapes.log 2>&1
cat gpurun_out/r05_mfma_shapes.log
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q -x -k "launches_its_own" 2>&1 | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_c2_driver.json 2>gpurun_out/r05_bench_c2_driver.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_c2_driver.json')); print(d['value'], d['roofline']['frac'], d.get('steady_state'), d['cpu_baseline'])"
