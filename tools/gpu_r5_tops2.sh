#!/bin/bash
# round 5: dense tree tops -- the oracle trajectories of the moment relaxations, then A/B runs of tail size against tree tops
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops_ab.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|lead debug" | cut -c1-900 | tee -a $O; }
run PlanarHand_N=10_MOMENT 11000 300 tail_k=24576 lead_debug=1
run PushBox_N=30_MOMENT 11000 300
run PushBox_N=30_MOMENT 11000 300 tail_k=10240 lead_tops=32 lead_debug=1
run PushBox_N=30_MOMENT 11000 300 tail_k=10240 lead_tops=64 lead_debug=1
run PushBox_N=50_MOMENT 11000 300
run PushBox_N=50_MOMENT 11000 300 tail_k=12800 lead_tops=32 lead_debug=1
run PushT_N=30_MOMENT 11000 300 lead_debug=1
run PushT_N=30_MOMENT 11000 300 lead_tops=0
timeout 2400 python -m pytest tests/test_gpu_moment_parity.py -x -q -s 2>&1 | grep -v "^$" | cut -c1-600 | tail -40 | tee gpurun_out/r05_tops_parity.log
