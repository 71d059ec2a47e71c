#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|resident trees\|dense tree" | cut -c1-700; }
run bqp-r1-40-1 11000 3000 lead_debug=1 lead_tops=32
run bqp-r1-40-1 11000 3000 lead_debug=1 lead_tops=16
run bqp-r1-40-1 11000 3000 lead_debug=1 lead_tops=32 tail_k=2048
