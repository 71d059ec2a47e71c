#!/bin/bash
# after a change to shared plumbing: the whole -m gpu suite and the driver's bench line
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_full.log 2>&1
echo "rc=$?" >> gpurun_out/gpu_tests_full.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gpu_tests_full.log | tail -20
for c in c2 c1 c5; do timeout 300 python bench.py --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d.get('steady_state'))"; done
