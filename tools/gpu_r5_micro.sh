#!/bin/bash
# round 5: micro trees (a thread per tree of one or two nodes) and the wide-forest plan (bqp-r1-40-1: a 1 024-column tail, the whole y-solve on the device)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_micro.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|resident trees" | cut -c1-700 | tee -a $O; }
run bqp-r1-40-1 11000 3000 lead_debug=1
run bqp-r1-40-1 11000 3000 lead_tops=0
run PlanarHand_N=10_MOMENT 11000 300 lead_debug=1
run PushBox_N=50_MOMENT 11000 1500 lead_debug=1
run PushBox_N=30_MOMENT 11000 1500 lead_debug=1
timeout 1500 python -m pytest tests/test_gpu_moment_parity.py -x -q -k "bit_for_bit or dense_tree_tops or falls_back or PlanarHand_N=10 or PushBox" 2>&1 | tail -3
