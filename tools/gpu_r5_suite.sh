#!/bin/bash
# round 5: the whole -m gpu suite, every failure listed
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gpu_tests.log | tail -30
