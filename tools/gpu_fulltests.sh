#!/bin/bash
# the whole -m gpu suite, log under gpurun_out/
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -s > gpurun_out/gpu_tests_full.log 2>&1
echo "rc=$?" >> gpurun_out/gpu_tests_full.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gpu_tests_full.log | tail -30
