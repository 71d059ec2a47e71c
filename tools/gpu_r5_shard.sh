#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/dbg/triple_diff.py 2>&1 | tail -25
timeout 2400 python -m pytest tests/test_gpu_sharded_procs.py tests/test_gpu_moment_parity.py tests/test_gpu_sharded.py -q -x > gpurun_out/r05_shard_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_shard_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_shard_tests.log | tail -8
