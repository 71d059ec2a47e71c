#!/bin/bash
# round 5: after the planner's step of 1 024 columns: moment-relaxation trajectories (deviations printed), default plans timed
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_moment_parity.py -q -s 2>&1 | grep -v "^$" | cut -c1-560 | tail -40 | tee gpurun_out/r05_tops_parity.log
O=gpurun_out/r05_tops_default.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT" | cut -c1-900 | tee -a $O; }
run PlanarHand_N=1_MOMENT 0 1500
run PushBox_N=30_MOMENT 11000 1500
run PushBox_N=50_MOMENT 11000 1500
run PushT_N=30_MOMENT 11000 1500
run PlanarHand_N=10_MOMENT 11000 300
