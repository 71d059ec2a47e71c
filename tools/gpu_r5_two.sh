#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
export CUADMM_BENCH_BACKEND=gloo
timeout 600 python bench.py --gpus 2 --blocks-per-gpu 5000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/two.json 2> gpurun_out/two.err
echo "rc=$?"; tail -5 gpurun_out/two.err; head -c 600 gpurun_out/two.json
timeout 600 python bench.py --gpus 2 --config c1 --steps 40 --warmup 5 --no-cpu-baseline --time-to-tol 0 > gpurun_out/two_c1.json 2> gpurun_out/two_c1.err
echo "rc=$?"; tail -3 gpurun_out/two_c1.err; python -c "
import json; d=json.load(open('gpurun_out/two_c1.json')); print(d['n_gpus'], d['value'], d.get('breakdown_ms_per_iter'))"
