#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
timeout 900 python -m pytest tests/test_gpu_solver.py -q -x -k "duo" 2>&1 | tail -2
done
