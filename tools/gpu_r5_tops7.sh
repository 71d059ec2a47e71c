#!/bin/bash
# round 5: after the long-column kernel and the tiny-tree launch: agreement tests, default plans timed
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops_default2.log
: > $O
timeout 1500 python -m pytest tests/test_gpu_moment_parity.py -x -q -k "dense_tree_tops or bit_for_bit or hybrid or round4" 2>&1 | tail -5 | tee -a $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|resident trees" | cut -c1-900 | tee -a $O; }
run PlanarHand_N=1_MOMENT 0 300 lead_debug=1
run pendulum_N=80 11000 300 lead_debug=1
run PushBox_N=30_MOMENT 11000 300 lead_debug=1
run PushBox_N=50_MOMENT 11000 300 lead_debug=1
run PushT_N=30_MOMENT 11000 300 lead_debug=1
run PlanarHand_N=10_MOMENT 11000 300 lead_debug=1
