#!/bin/bash
# round 5: contention test, then the profiling passes (kernel trace + PMC) of every config
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_contention.py tests/test_gpu_ops.py -q -x -k "contention or second_process or tail_solve" > gpurun_out/r05_contention.log 2>&1
echo "rc=$?" >> gpurun_out/r05_contention.log
tail -5 gpurun_out/r05_contention.log
bash tools/prof_round5.sh c2 c2_20 c1 c3 c4 c5 2>&1 | tail -15
