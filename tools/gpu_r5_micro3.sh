#!/bin/bash
# round 5: the wide-forest plan as the default: bqp to 1e-3, its tests, the tests that touch the sweeps
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/run_large.py bqp-r1-40-1 11000 20000 2>&1 | grep "RESULT" | cut -c1-500 | tee gpurun_out/r05_bqp.log
timeout 900 python tools/run_large.py bqp-r1-40-1 11000 20000 lead_tops=0 2>&1 | grep "RESULT" | cut -c1-500 | tee -a gpurun_out/r05_bqp.log
timeout 2400 python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_configs.py tests/test_gpu_longrun.py tests/test_gpu_sharded_procs.py -x -q 2>&1 | tail -4
